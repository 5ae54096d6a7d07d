#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_second
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo rc=$? >> $out/pytest.log; tail -4 $out/pytest.log
timeout -k 10 300 python3 tools/tune_reproject.py --workload cfg3 --place --rounds 9 "" "nt=1" "rh=16" "rh=16,nt=1" "rh=16,ns=16,pf=12" "ns=16,pf=12" "ns=16,pf=12,nt=1" "rh=64,ns=16,pf=12" > $out/tune_cfg3.txt 2>&1; cat $out/tune_cfg3.txt
timeout -k 10 300 python3 tools/bench_tan_mosaic.py > $out/tan_mosaic.txt 2>&1; cat $out/tan_mosaic.txt
