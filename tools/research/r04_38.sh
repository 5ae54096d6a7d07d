#!/bin/bash
# pipelined streaming one-pass unwind (PXL_UNWIND_STREAM=1: carrier wave + 15 compute waves, chunks parked in LDS, 4-window look-back)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
PXL_UNWIND_STREAM=1 timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" 2>&1 | tail -2 || exit 1
for rep in 1 2 3; do
echo "== shipping  $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== stream    $(PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
for n in 3e6 2e7; do
echo "== n=$n shipping $(PXL_N=$n timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== n=$n stream   $(PXL_N=$n PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
PXL_UNWIND_STREAM=1 timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 60 --only unwind 2>&1 | tail -1
