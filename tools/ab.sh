#!/bin/bash
# A/B two builds of the same ABI on ONE GPU box (box-to-box variation is ~2 %, larger than most kernel changes):
#   tools/ab.sh prev      (here, before a change) snapshot the last commit's sources into ubresnet_amd/libubr_prev.so
#   tools/ab.sh run [N]   (through gpurun) alternate bench.py between libubr_prev.so and the in-tree build, N rounds
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "prev" ]; then
  T=$(mktemp -d)
  mkdir -p $T/ubresnet_amd/csrc $T/include
  for f in $(git -C $R ls-tree --name-only HEAD ubresnet_amd/csrc/ include/); do git -C $R show HEAD:$f > $T/$f; done
  objs=""
  for f in $T/ubresnet_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -ffp-contract=off -c $f -o ${f%.hip}.o
    objs="$objs ${f%.hip}.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ubresnet_amd/libubr_prev.so $objs
  rm -rf $T
  echo "built ubresnet_amd/libubr_prev.so from HEAD"
else
  N=${2:-2}
  for i in $(seq $N); do
    echo -n "prev: "; UBR_LIB=$R/ubresnet_amd/libubr_prev.so python $R/bench.py --no-infer --no-cpu-baseline --no-breakdown 2>/dev/null | python $R/tools/benchline.py
    echo -n "new:  "; python $R/bench.py --no-infer --no-cpu-baseline --no-breakdown 2>/dev/null | python $R/tools/benchline.py
  done
fi
