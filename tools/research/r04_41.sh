#!/bin/bash
# link stride (8-byte words per link): 8 = one link per 64-byte line (tree), 4, 2, 16
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for v in tree l4 l2 l16; do
L=""; [ $v != tree ] && L="PXL_LIB_PATH=$R/variants/lib_$v.so"
echo "== $v big    $(env $L timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== $v 7168   $(env $L PXL_UNWIND_BIG_FROM=99999999999 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== $v stream $(env $L PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
echo "== l8 stream WIN1 $(PXL_LIB_PATH=$R/variants/lib_l8sw1.so PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
