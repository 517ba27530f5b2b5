#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for v in 64 16 64 16; do echo -n "UBR_PHASE_MIN_C=$v "; UBR_PHASE_MIN_C=$v $B 2>/dev/null | python tools/benchline.py; done
for v in 64 16; do echo -n "infer UBR_PHASE_MIN_C=$v "; UBR_PHASE_MIN_C=$v python tools/inferprobe.py 2>/dev/null | cut -c100-220; done
UBR_PHASE_MIN_C=16 python tools/fwdprobe.py 2>/dev/null | grep "taps4\|taps16\|forward" | head
