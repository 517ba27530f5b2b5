#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for v in 1 0 1 0; do echo -n "step defer_reduce=$v "; UBR_DEFER_REDUCE=$v $B 2>/dev/null | python tools/benchline.py; done
