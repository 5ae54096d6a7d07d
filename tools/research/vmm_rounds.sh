#!/bin/bash
set -o pipefail
for rnd in 1 2 3; do
  PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_prev.so timeout -k 10 200 python tools/ab_tan_patch.py 2>/dev/null | sed 's/^{/{"build": "general asin for points", /' >> gpurun_out/r03_fm_asin_points_ab.jsonl || exit 1
  timeout -k 10 200 python tools/ab_tan_patch.py 2>/dev/null | sed 's/^{/{"build": "small-half asin when the wave allows", /' >> gpurun_out/r03_fm_asin_points_ab.jsonl || exit 1
done
cat gpurun_out/r03_fm_asin_points_ab.jsonl
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gnomonic or generic or tan" 2>&1 | tail -2
