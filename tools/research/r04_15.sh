#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_15
mkdir -p $out
cd $R
timeout -k 10 240 python -m pytest tests -m gpu -q -x -k "generic or mosaic" > $out/pytest_sel.log 2>&1; rc=$?; echo rc=$rc >> $out/pytest_sel.log; tail -12 $out/pytest_sel.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
echo "== fuzz generic"; timeout -k 10 150 python3 tools/fuzz_parity.py --seconds 90 --seed 5 --only generic > $out/fuzz_generic.out 2>&1; rc=$?; echo rc=$rc; tail -3 $out/fuzz_generic.out | cut -c1-400
[ $rc -eq 0 ] || exit 1
echo "== mosaic v3"; PXL_GENERIC_V=3 timeout -k 10 200 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum' | cut -c1-200
for v in "" _a2 _ns32 _ns32a8; do echo "== mosaic ring $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 200 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum\|max_abs' | cut -c1-200; done
