#!/bin/bash
# Counter evidence for one program (run through gpurun from the repo root):
#   tools/collect_pmc.sh <tag> <python script + args...>     (env vars pass through)
# One rocprofv3 pass per counter group (PMC passes carry --kernel-trace only, as gpurun requires), program directly
# after `--`.  Everything lands under gpurun_out/<tag>/; tools/summarize_pmc.py turns it into one table for profiles/.
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
prog=$R/$1; shift          # script path relative to the repo root
echo "== plain run" ; timeout -k 10 600 python3 $prog "$@" > $out/plain.jsonl 2> $out/plain.err || { tail -5 $out/plain.err; exit 1; }
cat $out/plain.jsonl
echo "== kernel stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $prog "$@" > $out/under_stats.jsonl 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $grp"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $prog "$@" > /dev/null 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_ > $out/summary.txt; tail -5 $out/summary.txt
