#!/bin/bash
for s in conv64 conv128 conv256; do
  for fe in 0 1; do
    echo -n "fast_epi=$fe "; UBR_CONV_PC=0 UBR_CONV_FAST_EPI=$fe python tools/microbench.py $s 20 2>&1 | grep -v "amdgpu.ids"
    echo -n "fast_epi=$fe addend "; UBR_CONV_PC=0 UBR_CONV_FAST_EPI=$fe python tools/microbench.py $s 20 addend 2>&1 | grep -v "amdgpu.ids"
  done
done
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for fe in 1 0 1 0; do echo -n "step fast_epi=$fe "; UBR_CONV_FAST_EPI=$fe $B 2>/dev/null | python tools/benchline.py; done
