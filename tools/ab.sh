#!/bin/bash
# A/B the last commit against the working tree on ONE GPU box (box-to-box variation is ~2 %, larger than most kernel changes):
#   tools/ab.sh prev      (here, before a change) export the last commit's whole tree into ab_prev/ and build its library
#   tools/ab.sh run [N] [bench args]   (through gpurun) alternate bench.py between ab_prev/ and the working tree, N rounds
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "prev" ]; then
  rm -rf $R/ab_prev && mkdir -p $R/ab_prev
  git -C $R archive HEAD -- ubresnet_amd include bench.py oracle tools/benchline.py | tar -x -C $R/ab_prev
  (cd $R/ab_prev && python -m ubresnet_amd.build > /dev/null 2>&1)
  ls -la $R/ab_prev/ubresnet_amd/libubresnet_hip.so
  echo "built ab_prev/ from $(git -C $R rev-parse --short HEAD)"
else
  N=${2:-2}
  shift; shift || true
  for i in $(seq $N); do
    echo -n "prev: "; (cd $R/ab_prev && python bench.py --no-infer --no-cpu-baseline --no-breakdown "$@" 2>/dev/null | python $R/tools/benchline.py)
    echo -n "new:  "; python $R/bench.py --no-infer --no-cpu-baseline --no-breakdown --no-extra-legs "$@" 2>/dev/null | python $R/tools/benchline.py
  done
fi
