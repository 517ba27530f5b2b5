for c in conv64 conv128 conv256 conv512; do
  for pb in 96 80; do
  echo "pixb $pb"
  UBR_PC_PIXB=$pb UBR_LIB=$PWD/ubresnet_amd/libubr_stamps.so python tools/microbench.py $c 30 noxf nostats 2>&1 | grep "ms \|stamps"
  UBR_PC_PIXB=$pb python tools/microbench.py $c 30 2>&1 | grep "ms \|stamps"
  done
done
