#!/bin/bash
# generic pixel kernel: XCD-contiguous tile order against the plain 2-D grid (mosaic of 16 patches, plans reused)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
echo "== 2-D grid   $(timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep 'plans reused\|"tiled"' | grep -o '"ms": [0-9.]*\|bit_identical_to_one_shot": [a-z]*' | tr '\n' ' ')"
echo "== XCD order  $(PXL_GENERIC_XCD=1 timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep 'plans reused\|"tiled"' | grep -o '"ms": [0-9.]*\|bit_identical_to_one_shot": [a-z]*' | tr '\n' ' ')"
done
PXL_GENERIC_XCD=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "generic or tan or mosaic" 2>&1 | tail -2
