#!/bin/bash
# The three schedules of the weight gradients, same box, same build (profiles/r03_stream_options.txt):
#   side stream (default) | one stream, weight gradient before its block's data gradient | one stream, right after it
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { python bench.py --steps 30 --warmup 5 --no-infer --no-cpu-baseline --no-extra-legs | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.1f img/s  sum of kernel time %.2f ms' % (d['ms_per_step'], d['value'], d.get('kernel_time_ms_per_step', 0)))"; }
for i in 1 2; do
echo -n "two streams (side stream for weight gradients):        "; run
echo -n "one stream, weight gradient BEFORE the data gradient:  "; UBR_WGRAD_STREAM=0 run
echo -n "one stream, weight gradient AFTER the data gradient:   "; UBR_WGRAD_STREAM=0 UBR_WGRAD_ORDER=after run
done
