#!/bin/bash
# one-pass unwind with single-word links: chunk split and look-back windows revisited
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
echo "== tree 3+4 $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
for v in u2l4 u3l3 u2l3 u4l4 w2; do
echo "== $v  $(PXL_LIB_PATH=$R/variants/lib_$v.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done; done
