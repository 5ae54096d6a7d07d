#!/bin/bash
# final tree: one-pass unwind kernel stats + counters (single-word links), long fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_final6
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/prof_unwind.py > $out/stats.out 2>&1 || exit 1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $R/tools/prof_unwind.py > /dev/null 2> $out/pmc$i.err || { echo "pass $i failed"; tail -3 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_unwind k_scan_wsums > $out/unwind_summary.txt
cp $out/stats/*/*kernel_stats.csv $out/unwind_kernel_stats.csv
grep -A22 "k_unwind_onepass<UwSrcPix2" $out/unwind_summary.txt | head -30
head -8 $out/unwind_summary.txt | cut -c1-180
cd $R
timeout -k 10 700 python3 tools/fuzz_parity.py --seconds 600 > $out/fuzz.txt 2>&1; echo fuzz rc=$?; tail -1 $out/fuzz.txt
