for c in conv64 conv128 conv256 conv512; do
  for pr in 1 0; do echo -n "pairs=$pr: "; UBR_CONV_PAIRS=$pr UBR_CONV_PC=0 python tools/microbench.py $c 30 2>&1 | grep "ms "; done
done
for pr in 1 0 1 0; do echo -n "pairs=$pr "; UBR_CONV_PAIRS=$pr python bench.py --steps 30 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown | python tools/benchline.py; done
