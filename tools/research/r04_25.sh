#!/bin/bash
# counters of the one-pass unwind kernel (LDS-parked form): is it VALU-issue bound?
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_uw_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/prof_unwind.py > $out/stats.out 2>&1 || exit 1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $R/tools/prof_unwind.py > /dev/null 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_unwind > $out/summary.txt; cat $out/summary.txt
