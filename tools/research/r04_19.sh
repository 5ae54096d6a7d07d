#!/bin/bash
# in-range vote of the unwind sources: parity, timing, fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_vote
mkdir -p $out
cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" > $out/tests.txt 2>&1; rc=$?; tail -3 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | tail -6 || exit 1
timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 60 --only unwind > $out/fuzz.txt 2>&1; echo fuzz rc=$?; tail -1 $out/fuzz.txt
