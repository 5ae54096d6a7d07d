#!/bin/bash
set -o pipefail
timeout -k 10 300 python tools/tune_reproject.py --workload up4 --place --rounds 7 "" "rh=32" > gpurun_out/r03_tune_other4.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload down2 --place --rounds 7 "" "rh=16" >> gpurun_out/r03_tune_other4.txt 2>&1 || exit 1
timeout -k 10 300 python tools/tune_reproject.py --workload down4 --place --rounds 7 "" "rh=16" >> gpurun_out/r03_tune_other4.txt 2>&1 || exit 1
grep -A3 "^workload" gpurun_out/r03_tune_other4.txt
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 200 python tools/fuzz_parity.py --seconds 90 --seed 555 --only reproject 2>&1 | tail -1
