#!/bin/bash
# Collect the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01
# Writes everything under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards
# (tools/make_traffic.py builds profiles/traffic_<workload>.json from the two PMC passes).
set -o pipefail
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in cfg4 cfg3; do
  # 1. plain bench line (un-profiled numbers are the ones quoted)
  timeout -k 10 300 python3 $R/bench.py --workload $wl --check > $out/bench_$wl.json 2> $out/bench_$wl.err
  # 2. kernel trace + stats (program directly after --)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 $R/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_${wl}_under_stats.json 2>/dev/null
  # 3. HBM traffic counters, one pass each, kernel-trace only
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_$wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write_$wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
timeout -k 10 300 python3 $R/bench.py --workload cfg5 --steps 5 --warmup 1 > $out/bench_cfg5.json 2>/dev/null
timeout -k 10 300 python3 $R/bench.py --workload cfg2 --no-cpu-baseline > $out/bench_cfg2.json 2>/dev/null
timeout -k 10 300 python3 $R/tools/bench_elementwise.py > $out/elementwise.jsonl 2>/dev/null
ls $out
