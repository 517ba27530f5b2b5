#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
echo -n "default(128) "; $B 2>/dev/null | python tools/benchline.py
echo -n "MA9=4 "; UBR_WGRAD_MA9=4 $B 2>/dev/null | python tools/benchline.py
echo -n "MA9=4 N=64 "; UBR_WGRAD_MA9=4 UBR_WGRAD_TARGET_N=64 $B 2>/dev/null | python tools/benchline.py
echo -n "side_priority=-1 "; UBR_SIDE_PRIORITY=-1 $B 2>/dev/null | python tools/benchline.py
echo -n "DEFER_REDUCE=0 "; UBR_DEFER_REDUCE=0 $B 2>/dev/null | python tools/benchline.py
echo -n "default(128) "; $B 2>/dev/null | python tools/benchline.py
python -m pytest tests -x -q -m gpu -k "wgrad or uresnet or plan or dp or aspp" 2>&1 | tail -1
