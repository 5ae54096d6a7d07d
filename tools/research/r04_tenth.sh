#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_tenth
mkdir -p $out
cd $R
echo "== tan evaluators, per-pixel posmap"; PXL_TAN_GRID=0 timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep posmap
echo "== tan evaluators, grid posmap, 8 fronts"; timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep posmap
echo "== tan evaluators, grid posmap, 1 front"; PXL_POSMAP_FRONTS=1 timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep posmap
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "gnomonic or tan or placement or alloc" > $out/pytest_sel.log 2>&1; echo rc=$? >> $out/pytest_sel.log; tail -5 $out/pytest_sel.log | cut -c1-300
echo "== unwind two-pass"; PXL_UNWIND_ONEPASS=0 timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2
for v in "" _u4w4 _u4w8 _u4w16 _u8w16; do echo "== unwind one-pass $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2; done
