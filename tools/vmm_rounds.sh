#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gnomonic or generic or tan" 2>&1 | tail -2
for rnd in 1 2 3; do
  PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_prev.so timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | grep sky2pix | sed 's/^{/{"build": "library sincos", /' >> gpurun_out/r03_fm_sincos_ab.jsonl || exit 1
  timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | grep sky2pix | sed 's/^{/{"build": "pxl_fm_sincos", /' >> gpurun_out/r03_fm_sincos_ab.jsonl || exit 1
done
PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_prev.so timeout -k 10 200 python tools/bench_tan_mosaic.py 2>/dev/null | grep variant | cut -c1-200 | sed 's/^{/{"build": "library sincos", /' >> gpurun_out/r03_fm_sincos_ab.jsonl
timeout -k 10 200 python tools/bench_tan_mosaic.py 2>/dev/null | grep variant | cut -c1-200 | sed 's/^{/{"build": "pxl_fm_sincos", /' >> gpurun_out/r03_fm_sincos_ab.jsonl
cat gpurun_out/r03_fm_sincos_ab.jsonl
