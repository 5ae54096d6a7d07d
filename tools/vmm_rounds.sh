#!/bin/bash
# scratch runner: Gnomonic evaluators before / after pxl_fastmath.h on one box, alternating processes
set -o pipefail
for rnd in 1 2 3; do
  PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_before_fastmath.so timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | sed 's/^{/{"build": "before (device libm)", /' >> gpurun_out/r03_fm_before_after.jsonl || exit 1
  timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | sed 's/^{/{"build": "after (pxl_fastmath.h)", /' >> gpurun_out/r03_fm_before_after.jsonl || exit 1
done
PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_before_fastmath.so timeout -k 10 200 python tools/bench_tan_mosaic.py 2>/dev/null | sed 's/^{/{"build": "before (device libm)", /' >> gpurun_out/r03_fm_before_after.jsonl
timeout -k 10 200 python tools/bench_tan_mosaic.py 2>/dev/null | sed 's/^{/{"build": "after (pxl_fastmath.h)", /' >> gpurun_out/r03_fm_before_after.jsonl
cat gpurun_out/r03_fm_before_after.jsonl | cut -c1-260
