#!/bin/bash
# One GPU-box session that produces everything under profiles/ for a round tag (default r02):
#   bash tools/profile_round.sh r02        (run through gpurun; outputs land in gpurun_out/<tag>_*)
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --breakdown-file $O/${TAG}_bench_launch_breakdown.txt > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || echo "bench failed"
echo "bench done"; tail -c 300 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-infer --no-extra-legs > $O/${TAG}_rocprof.log 2>&1
cd $R
python tools/timeline.py $O/prof_$TAG > $O/${TAG}_two_stream_timeline.txt
cp $(ls $O/prof_$TAG/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
rm -rf $O/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_i_$TAG -- python3 $R/tools/inferprobe.py > $O/${TAG}_infer.log 2>&1
cp $(ls $O/prof_i_$TAG/*/*kernel_stats.csv | head -1) $O/${TAG}_infer_kernel_stats.csv
rm -rf $O/prof_i_$TAG
echo "trace done"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-infer --no-extra-legs > $O/${TAG}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-infer --no-extra-legs > $O/${TAG}_pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/${TAG}_pmc_traffic.json
rm -rf $O/pmc_f $O/pmc_w
echo "pmc done"
UBR_PLAN=0 python tools/hostprobe.py > $O/${TAG}_hostprobe.txt 2>&1
UBR_PLAN=1 python tools/hostprobe.py >> $O/${TAG}_hostprobe.txt 2>&1
python tools/fwdprobe.py > $O/${TAG}_fwdprobe.txt 2>/dev/null
python tools/check_store_hazard.py > $O/${TAG}_store_hazard_scan.txt 2>&1
cat $O/${TAG}_hostprobe.txt
