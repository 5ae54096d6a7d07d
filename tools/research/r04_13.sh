#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_13
mkdir -p $out
cd $R
for wl in "cfg4" "cfg4 --place" "cfg3s" "cfg4 --strip 3/8"; do
  echo "== tune (bursts) $wl"
  timeout -k 10 400 python3 tools/tune_reproject.py --workload $wl --rounds 5 --burst 3 "" "rh=32" "rh=64" "rh=8" "rh=32,ns=16,pf=4" "ns=4" 2>&1 | grep -v "amdgpu.ids\|place_pair" | tee -a $out/tune_rh_same_res.txt
done
echo "== fuzz"; timeout -k 10 420 python3 tools/fuzz_parity.py --seconds 360 --seed 404 --log $out/fuzz.log > $out/fuzz.out 2>&1; echo fuzz rc=$?; tail -12 $out/fuzz.out | cut -c1-300
echo "== unwind two-pass"; PXL_UNWIND_ONEPASS=0 timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2
for v in "" _win1 _win2 _win8; do echo "== unwind one-pass (default WIN=4) $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2; done
