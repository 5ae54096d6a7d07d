#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_eighth
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "unwind or pix2sky or soa or safe or gnomonic or generic or tan" > $out/pytest_sel.log 2>&1; echo rc=$? >> $out/pytest_sel.log; tail -15 $out/pytest_sel.log | cut -c1-300
echo "== unwind two-pass"; PXL_UNWIND_ONEPASS=0 timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2
for v in "" _uw4 _uw16; do echo "== unwind one-pass $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2; done
echo "== mosaic v1"; PXL_GENERIC_V=1 timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum' | cut -c1-200
for v in "" _g2 _g8; do echo "== mosaic v3 $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum\|max_abs' | cut -c1-200; done
echo "== tan evaluators, per-pixel posmap"; PXL_TAN_GRID=0 timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep posmap
echo "== tan evaluators, grid posmap"; timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep -v amdgpu
