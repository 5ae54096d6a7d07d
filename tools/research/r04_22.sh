#!/bin/bash
# two-link one-pass unwind (PXL_UNWIND_ONEPASS=3) against the shipping form; HEAD's library beside both
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_uw2
mkdir -p $out
cd $R
PXL_UNWIND_ONEPASS=3 timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" > $out/tests.txt 2>&1; rc=$?; tail -3 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
echo "== HEAD lib        $(PXL_LIB_PATH=$R/variants/lib_head.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
echo "== one link (now)  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
echo "== two links U=2   $(PXL_UNWIND_ONEPASS=3 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
echo "== two links U=3   $(PXL_LIB_PATH=$R/variants/lib_u3.so PXL_UNWIND_ONEPASS=3 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
echo "== two links U=1   $(PXL_LIB_PATH=$R/variants/lib_u1.so PXL_UNWIND_ONEPASS=3 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
done
PXL_UNWIND_ONEPASS=3 timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 40 --only unwind > $out/fuzz.txt 2>&1; echo fuzz rc=$?; tail -1 $out/fuzz.txt
