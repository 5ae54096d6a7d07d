#!/bin/bash
# streaming unwind: chunk size x persistent workgroups per CU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== shipping  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== stream U4 x1 $(PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
for v in u3 u2 u1; do for per in 1 2; do
echo "== stream $v x$per  $(PXL_LIB_PATH=$R/variants/lib_$v.so PXL_UNWIND_STREAM=1 PXL_UNWIND_STREAM_PER_CU=$per timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done; done
echo "== shipping  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
