#!/bin/bash
# generic pixel kernel: right taps from the neighbouring lane (PXL_T3_SHARE builds) against the 16-byte pair loads
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
PXL_LIB_PATH=$R/variants/lib_share.so timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "generic or tan or mosaic" 2>&1 | tail -2 || exit 1
for rep in 1 2; do
echo "== tree      $(timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep 'plans reused\|"tiled"\|checksum' | grep -o '"ms": [0-9.]*\|bit_identical_to_one_shot": [a-z]*\|checksum": [0-9]*' | tr '\n' ' ')"
for v in share shareg2 shareg8; do
echo "== $v   $(PXL_LIB_PATH=$R/variants/lib_$v.so timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep 'plans reused\|"tiled"\|checksum' | grep -o '"ms": [0-9.]*\|bit_identical_to_one_shot": [a-z]*\|checksum": [0-9]*' | tr '\n' ' ')"
done; done
PXL_LIB_PATH=$R/variants/lib_share.so timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 60 --only generic 2>&1 | tail -1
