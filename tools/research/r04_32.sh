#!/bin/bash
# streaming evaluators: non-temporal stores (tree) against plain stores (variants/lib_plain.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
echo "== plain  $(PXL_LIB_PATH=$R/variants/lib_plain.so timeout -k 10 300 python3 tools/research/evaluators_ab.py 2>/dev/null | tail -1)"
echo "== nt     $(timeout -k 10 300 python3 tools/research/evaluators_ab.py 2>/dev/null | tail -1)"
done
for rep in 1 2; do
echo "== unwind plain  $(PXL_LIB_PATH=$R/variants/lib_plain.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== unwind nt     $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
