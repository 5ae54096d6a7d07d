#!/bin/bash
# one-pass unwind: chunk size beyond what two workgroups per CU allow (15-bit aggregate fields)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_uw5
mkdir -p $out
cd $R
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" > $out/tests.txt 2>&1; rc=$?; tail -3 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
echo "== U3+L4 (tree)  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
for v in u2l4 u3l3 u4l4 u8l0 u4l0; do
echo "== $v  $(PXL_LIB_PATH=$R/variants/lib_$v.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
done; done
for v in u2l4 u4l4; do PXL_LIB_PATH=$R/variants/lib_$v.so timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind" 2>&1 | tail -1; done
