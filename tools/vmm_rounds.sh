#!/bin/bash
python -m pytest tests/test_gpu_placement.py -x -q 2>&1 | tail -25
for k in 1 2; do
    python tools/tune_reproject.py --workload cfg3 --rounds 5 --place-compact "" "flags=64" 2>&1 | grep -v amdgpu.ids | grep -v "^workload" | sed "s/^/compact cfg3: /"
done
