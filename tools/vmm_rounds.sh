#!/bin/bash
set -o pipefail
timeout -k 10 500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for rnd in 1 2 3; do timeout -k 10 200 python tools/bench_tan_evaluators.py 2>/dev/null | tee -a gpurun_out/r03_tan_evaluators_final.jsonl; done
timeout -k 10 200 python tools/fuzz_parity.py --seconds 60 --seed 5151 --only maps 2>&1 | tail -1
