#!/bin/bash
python -m pytest tests/test_gpu_placement.py -x -q 2>&1 | tail -2
for k in 1 2 3 4; do
    python tools/tune_reproject.py --workload cfg3 --rounds 5 --place "" 2>&1 | grep -v amdgpu.ids | grep -v "^workload" | cut -c1-420 | sed "s/^/placed cfg3: /"
done
python tools/tune_reproject.py --workload cfg4 --rounds 5 --place "" 2>&1 | grep -v amdgpu.ids | grep -v "^workload" | cut -c1-420 | sed "s/^/placed cfg4: /"
