#!/bin/bash
# single-word links after the grid fix: previous commit / tree (64-byte stride) / 32-byte / 8-byte stride; streaming form with one window per round
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for v in prev tree l4 l1; do
L=""; [ $v != tree ] && L="PXL_LIB_PATH=$R/variants/lib_$v.so"
echo "== $v big    $(env $L timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== $v 7168   $(env $L PXL_UNWIND_BIG_FROM=99999999999 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
for v in sw1 sw1l4 sw1u6; do
echo "== stream $v $(PXL_LIB_PATH=$R/variants/lib_$v.so PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
done
