#!/bin/bash
# scratch: row-reuse A/B of the weight-gradient kernels
for s in wgrad16 wgrad32 wgrad64 wgrad128 wgrad256 wgrad512; do
  for rr in 0 1; do
    echo -n "rowreuse=$rr "; UBR_WGRAD_ROWREUSE=$rr python tools/microbench.py $s 20 kernelonly 2>&1 | grep -v "amdgpu.ids\|nsplit"
  done
done
B="python bench.py --steps 30 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for rr in 1 0 1 0; do echo -n "step rowreuse=$rr "; UBR_WGRAD_ROWREUSE=$rr $B | python tools/benchline.py; done
echo -n "target_n=512 "; UBR_WGRAD_TARGET_N=512 $B | python tools/benchline.py
echo -n "target_k=512 "; UBR_WGRAD_TARGET_K=512 $B | python tools/benchline.py
echo -n "both 512 "; UBR_WGRAD_TARGET_N=512 UBR_WGRAD_TARGET_K=512 $B | python tools/benchline.py
