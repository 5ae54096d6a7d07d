#!/bin/bash
set -o pipefail
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3 --strip 3/8 --place --rounds 11 "mintiles=16384" "mintiles=8192" "mintiles=4096" "mintiles=2048" "mintiles=2048,rh=64" > gpurun_out/r03_tune_mintiles2_cfg3strip.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3 --strip 1/4 --place --rounds 11 "" "mintiles=8192" "mintiles=4096" "mintiles=2048" > gpurun_out/r03_tune_mintiles2_cfg3strip4.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3s --strip 3/8 --rounds 11 "" "mintiles=8192" "mintiles=4096" "mintiles=2048" > gpurun_out/r03_tune_mintiles2_cfg3sstrip.txt 2>&1 || exit 1
for f in cfg3strip cfg3strip4 cfg3sstrip; do grep -A6 "^workload" gpurun_out/r03_tune_mintiles2_$f.txt; done
