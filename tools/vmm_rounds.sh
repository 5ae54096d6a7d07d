#!/bin/bash
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gnomonic or generic or tan" 2>&1 | tail -2
for rnd in 1 2 3; do
  PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_prev.so timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | grep -v sky2pix | sed 's/^{/{"build": "general atan2", /' >> gpurun_out/r03_fm_tame_ab.jsonl || exit 1
  timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | grep -v sky2pix | sed 's/^{/{"build": "tame atan2 when the wave allows", /' >> gpurun_out/r03_fm_tame_ab.jsonl || exit 1
done
cat gpurun_out/r03_fm_tame_ab.jsonl
timeout -k 10 200 python tools/fuzz_parity.py --seconds 60 --seed 4242 --only maps 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_parity.py --seconds 60 --seed 4243 --only generic 2>&1 | tail -1
