#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_14
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "generic or gnomonic or tan or mosaic or unwind" > $out/pytest_sel.log 2>&1; echo rc=$? >> $out/pytest_sel.log; tail -4 $out/pytest_sel.log | cut -c1-300
for v in _exlaunch _nopair ""; do echo "== mosaic $v (exlaunch = third launch + 8-byte stores; nopair = exact in the lattice launch, 8-byte stores; default = + pair stores)"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum\|max_abs' | cut -c1-200; done
echo "== fuzz generic"; timeout -k 10 200 python3 tools/fuzz_parity.py --seconds 150 --seed 77 --only generic > $out/fuzz_generic.out 2>&1; echo rc=$?; tail -2 $out/fuzz_generic.out | cut -c1-300
