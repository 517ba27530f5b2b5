#!/bin/bash
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)"
B="python bench.py --steps 40 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs --no-breakdown"
for v in 0 1 -1 0 1 -1; do echo -n "side_priority=$v "; UBR_SIDE_PRIORITY=$v $B 2>&1 | python tools/benchline.py; done
