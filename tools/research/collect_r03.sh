#!/bin/bash
# Round-3 evidence (run through gpurun from the repo root): tools/research/collect_r03.sh [tag]
#   un-profiled default bench line (headline + configs), then per workload: the same command under
#   rocprofv3 --kernel-trace --stats (program directly after --) and the two HBM-traffic counter passes
#   (--pmc FETCH_SIZE / --pmc WRITE_SIZE, kernel-trace only, separate passes).  Everything under gpurun_out/<tag>/.
set -o pipefail
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
for wl in cfg4 cfg3 cfg3s; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 $R/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-traffic > $out/bench_${wl}_under_stats.json 2> $out/stats_$wl.err; echo "stats $wl rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_$wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 1 --sustain-seconds 0 --no-cpu-baseline --no-configs --no-traffic > /dev/null 2> $out/fetch_$wl.err; echo "fetch $wl rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write_$wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 1 --sustain-seconds 0 --no-cpu-baseline --no-configs --no-traffic > /dev/null 2> $out/write_$wl.err; echo "write $wl rc=$?"
done
find $out -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
