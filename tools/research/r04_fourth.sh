#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_fourth
mkdir -p $out
cd $R
echo "== mosaic V1"; PXL_GENERIC_V1=1 timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum' | cut -c1-330 | tee $out/mosaic_v1.txt
echo "== mosaic V1 + pair taps"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip_pair.so PXL_GENERIC_V1=1 timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum' | cut -c1-330 | tee $out/mosaic_v1_pair.txt
echo "== mosaic V2"; timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum' | cut -c1-330 | tee $out/mosaic_v2.txt
for wl in "cfg3s" "cfg3s --place" "cfg4 --strip 3/8" "cfg3 --strip 3/8" "cfg3 --place" "down2 --place"; do
  echo "== tune $wl"
  timeout -k 10 400 python3 tools/tune_reproject.py --workload $wl --rounds 7 "" "nt=1" "nt=0" 2>&1 | grep -v amdgpu.ids | tee -a $out/tune_nt.txt
done
cd /tmp && export TMPDIR=/tmp
echo "== unwind kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/unwind_stats -- python3 $R/tools/prof_unwind.py > $out/unwind_stats.out 2>&1
cat $out/unwind_stats/*/*kernel_stats.csv | cut -c1-160 | head -12
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/unwind_pmc$i -- python3 $R/tools/prof_unwind.py > /dev/null 2> $out/unwind_pmc$i.err || { echo "pass $i failed"; tail -3 $out/unwind_pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_unwind k_scan_wsums > $out/unwind_summary.txt; cat $out/unwind_summary.txt
