#!/bin/bash
# counters of the Gnomonic point evaluators (k_tan_points) and posmap kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_tan_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/bench_tan_evaluators.py > $out/stats.out 2>&1 || exit 1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $R/tools/bench_tan_evaluators.py > /dev/null 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_tan k_posmap_tan > $out/summary.txt; cat $out/summary.txt
