#!/bin/bash
# one-pass unwind: chunk id from the ticket (shipping) against chunk id = blockIdx.x
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
echo "== ticket   $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
echo "== blockIdx $(PXL_LIB_PATH=$R/variants/lib_noticket.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
done
PXL_LIB_PATH=$R/variants/lib_noticket.so timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "unwind" 2>&1 | tail -2
