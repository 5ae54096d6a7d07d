#!/bin/bash
set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_entrypoints.py -x -q -m gpu -k native 2>&1 | tail -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include tools/native/native_bench.cpp -L pixell.jl_amd -lpixell_hip -Wl,-rpath,$PWD/pixell.jl_amd -o /tmp/native_bench || exit 1
for rnd in 1 2; do
  /tmp/native_bench 21600 1 refine 20 | tee -a gpurun_out/r03_native_host_placed.jsonl
  /tmp/native_bench 21600 1 refine 20 placed | tee -a gpurun_out/r03_native_host_placed.jsonl
  /tmp/native_bench 43200 3 same 10 | tee -a gpurun_out/r03_native_host_placed.jsonl
  /tmp/native_bench 43200 3 same 10 placed | tee -a gpurun_out/r03_native_host_placed.jsonl
done
