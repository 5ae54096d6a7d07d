#!/bin/bash
# one-pass unwind: HEAD against A (plain-wave fast path in the apply step) and B = tree (A + 32-bit validity tests in the compute step)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" 2>&1 | tail -2 || exit 1
for rep in 1 2 3; do
echo "== HEAD  $(PXL_LIB_PATH=$R/variants/lib_prev.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== A     $(PXL_LIB_PATH=$R/variants/lib_A.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== B     $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 40 --only unwind 2>&1 | tail -1
