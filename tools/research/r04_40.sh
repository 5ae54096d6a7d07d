#!/bin/bash
# single-word links: the shipping one-pass kernel and the streaming one, against the previous commit (three-word links)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" 2>&1 | tail -2 || exit 1
PXL_UNWIND_STREAM=1 timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" 2>&1 | tail -1 || exit 1
for rep in 1 2; do
echo "== previous commit        $(PXL_LIB_PATH=$R/variants/lib_prev.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== tree                   $(timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== tree, 2 windows/round  $(PXL_LIB_PATH=$R/variants/lib_w2.so timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== tree, 7168 chunks      $(PXL_UNWIND_BIG_FROM=99999999999 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== stream WIN4            $(PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== stream WIN2            $(PXL_LIB_PATH=$R/variants/lib_sw2.so PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
echo "== stream WIN1            $(PXL_LIB_PATH=$R/variants/lib_sw1.so PXL_UNWIND_STREAM=1 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep -E 'out-of' | tail -2 | tr '\n' ' ')"
done
timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 40 --only unwind 2>&1 | tail -1
