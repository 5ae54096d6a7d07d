#!/bin/bash
# final-tree evidence for the parts that changed late in round 4: default line (plain-pair headline), one-pass unwind stats + counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_late2
mkdir -p $out
cd $R
S=$SECONDS; python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo bench rc=$? seconds=$((SECONDS-S))
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/prof_unwind.py > $out/stats.out 2>&1 || exit 1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $R/tools/prof_unwind.py > /dev/null 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_unwind k_scan_wsums > $out/unwind_summary.txt
cp $out/stats/*/*kernel_stats.csv $out/unwind_kernel_stats.csv
grep -A22 "k_unwind_onepass<UwSrcPix2>" $out/unwind_summary.txt | head -50
cd $R; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_late2/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("headline_allocation_policy"), d["roofline"].get("frac_other_policies"))
print({k:(v.get("kernel_ms_avg"),v.get("frac")) for k,v in d["configs"].items()})
print({k:(v.get("ms"),v.get("frac")) for k,v in d["evaluators"].items()})
PY
