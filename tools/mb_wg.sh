#!/bin/bash
for s in conv64 conv128 conv256 conv512; do
  for pc in 0 2; do
    echo -n "pc=$pc "; UBR_CONV_PC=$pc python tools/microbench.py $s 20 2>&1 | grep -v "amdgpu.ids"
  done
done
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for pc in 1 0 2 1 0 2; do echo -n "step conv_pc=$pc "; UBR_CONV_PC=$pc $B 2>/dev/null | python tools/benchline.py; done
