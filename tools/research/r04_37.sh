#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" 2>&1 | tail -2 || exit 1
for rep in 1 2; do
echo "== 13 312-point chunks (default for 1e8)  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== 7 168-point chunks                     $(PXL_UNWIND_BIG_FROM=99999999999 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
for n in 3e6 8e6 2e7; do
echo "== n=$n big   $(PXL_N=$n PXL_UNWIND_BIG_FROM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== n=$n small $(PXL_N=$n PXL_UNWIND_BIG_FROM=99999999999 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
PXL_UNWIND_BIG_FROM=1 timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 60 --only unwind 2>&1 | tail -1
