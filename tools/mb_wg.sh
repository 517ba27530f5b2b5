#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for pc in 1 0 1 0 1 0; do echo -n "step pc=$pc "; UBR_WGRAD_PC=$pc $B 2>/dev/null | python tools/benchline.py; done
python tools/fwdprobe.py 2>/dev/null
