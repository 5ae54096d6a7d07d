#!/bin/bash
# one-pass unwind: what the look-back costs (a timing-only build that skips it: WRONG answers, never shipped)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
echo "== shipping      $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
echo "== no look-back  $(PXL_LIB_PATH=$R/variants/lib_nolookback.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
done
