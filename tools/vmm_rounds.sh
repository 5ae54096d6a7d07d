#!/bin/bash
# scratch runner: ring depth 4 vs 8 with the new prefetch rule
set -o pipefail
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3 --place --rounds 9 "" "pf=3" "ns=4" "ns=4,pf=4" "rh=64" "rh=16" > gpurun_out/r03_tune_pf2_cfg3.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload cfg4 --place --rounds 7 "" "pf=3" "ns=4" "ns=4,pf=2" > gpurun_out/r03_tune_pf2_cfg4.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3s --place --rounds 9 "" "pf=3" "ns=4" "ns=4,pf=2" > gpurun_out/r03_tune_pf2_cfg3s.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload cfg4 --strip 3/8 --place --rounds 9 "" "pf=3" "ns=4" "ns=4,pf=2" > gpurun_out/r03_tune_pf2_strip.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload cfg2 --rounds 9 "" "pf=3" "ns=4" "ns=4,pf=2" > gpurun_out/r03_tune_pf2_cfg2.txt 2>&1 || exit 1
for f in cfg3 cfg4 cfg3s strip cfg2; do grep -A8 "^workload" gpurun_out/r03_tune_pf2_$f.txt; done
