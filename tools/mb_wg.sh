#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for v in 64 32 16 64 32 16; do echo -n "UBR_PHASE_MIN_C=$v "; UBR_PHASE_MIN_C=$v $B 2>/dev/null | python tools/benchline.py; done
for v in 64 32 16; do echo -n "infer UBR_PHASE_MIN_C=$v "; UBR_PHASE_MIN_C=$v python tools/inferprobe.py 2>/dev/null | cut -c1-200; done
