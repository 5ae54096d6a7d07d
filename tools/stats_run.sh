#!/bin/bash
# tools/stats_run.sh <tag> <script relative to the repo root> [args...]: one un-profiled run + one rocprofv3 --kernel-trace --stats run
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
prog=$R/$1; shift
timeout -k 10 400 python3 $prog "$@" > $out/plain.jsonl 2> $out/plain.err || { tail -5 $out/plain.err; exit 1; }
cat $out/plain.jsonl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $prog "$@" > $out/under_stats.jsonl 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
python3 $R/tools/summarize_pmc.py $out k_ | tee $out/summary.txt
