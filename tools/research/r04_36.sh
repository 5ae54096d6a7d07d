#!/bin/bash
# one-pass unwind: very large chunks, one workgroup per CU (LDS-parked groups), against the shipping 3 + 4 split
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
echo "== tree (3+4)  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
for v in u2l5 u4l8 u3l9 u4l9; do
echo "== $v  $(PXL_LIB_PATH=$R/variants/lib_$v.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done; done
PXL_LIB_PATH=$R/variants/lib_u4l8.so timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind" 2>&1 | tail -1
