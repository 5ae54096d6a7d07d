#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_placement.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3 --place-native --rounds 9 "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_place_native.txt
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3 --place --rounds 9 "" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_place_native.txt
timeout -k 10 300 python tools/tune_reproject.py --workload cfg3 --rounds 9 "" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_place_native.txt
timeout -k 10 300 python tools/tune_reproject.py --workload cfg4 --place-native --rounds 7 "" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_place_native.txt
timeout -k 10 300 python tools/tune_reproject.py --workload cfg4 --rounds 7 "" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_place_native.txt
