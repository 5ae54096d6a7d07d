#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_seventh
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "unwind or pix2sky or soa or safe" > $out/pytest_unwind.log 2>&1; echo rc=$? >> $out/pytest_unwind.log; tail -4 $out/pytest_unwind.log
grep -q "rc=0" $out/pytest_unwind.log || exit 1
echo "== unwind two-pass"; PXL_UNWIND_ONEPASS=0 timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -3
echo "== unwind one-pass"; timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -3
bash tools/research/r04_sixth.sh
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo rc=$? >> $out/pytest_all.log; tail -4 $out/pytest_all.log
