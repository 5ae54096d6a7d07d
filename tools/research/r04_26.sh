#!/bin/bash
# one-pass unwind after the instruction diet (compare-based guess, y' by DPP, NaN-free fast path): whole GPU suite, timing against the previous commit, fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_uw6
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; rc=$?; tail -3 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
echo "== previous commit  $(PXL_LIB_PATH=$R/variants/lib_prev.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of|in place' | tail -3 | tr '\n' ' ')"
echo "== tree             $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of|in place' | tail -3 | tr '\n' ' ')"
done
timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 90 --only unwind > $out/fuzz.txt 2>&1; echo fuzz rc=$?; tail -1 $out/fuzz.txt
