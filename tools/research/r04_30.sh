#!/bin/bash
# generic pixel kernel: non-temporal stores, rows per group, rolled row loop (mosaic of 16 patches, plans reused)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
echo "== tree   $(timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep 'plans reused' | grep -o '"ms": [0-9.]*\|bit_identical_to_one_shot": [a-z]*' | tr '\n' ' ')"
for v in nt ntg2 ntr2 ntr4 r4 ntr8; do
echo "== $v   $(PXL_LIB_PATH=$R/variants/lib_$v.so timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep 'plans reused' | grep -o '"ms": [0-9.]*\|bit_identical_to_one_shot": [a-z]*' | tr '\n' ' ')"
done; done
