#!/bin/bash
# round 4, first GPU call: the traffic-mix ceiling next to the product kernel, then SQ counters of cfg3 on placed maps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_first
mkdir -p $out
cd $R
echo "== mix ceiling, plain" && timeout -k 10 300 python3 tools/research/mix_ceiling.py --rounds 7 > $out/mix_plain.jsonl 2> $out/mix_plain.err && cat $out/mix_plain.jsonl &&
echo "== mix ceiling, placed" && timeout -k 10 300 python3 tools/research/mix_ceiling.py --place --rounds 7 > $out/mix_placed.jsonl 2> $out/mix_placed.err && cat $out/mix_placed.jsonl &&
cd /tmp && export TMPDIR=/tmp &&
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $grp"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $R/tools/tune_reproject.py --workload cfg3 --place --rounds 5 "" "flags=2" "flags=64" > $out/pmc$i.out 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_reproject > $out/summary.txt; cat $out/summary.txt
