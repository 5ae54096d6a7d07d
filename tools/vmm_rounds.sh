#!/bin/bash
# scratch runner: previous build against the current one, alternating processes on one box
set -o pipefail
for rnd in 1 2 3; do
  for wl in cfg4 cfg3; do
    PXL_LIB_PATH=$PWD/tools/native/libpixell_hip_prev.so timeout -k 10 200 python tools/tune_reproject.py --workload $wl --place --rounds 9 "" 2>/dev/null | grep median | sed "s/^/prev $wl /" >> gpurun_out/r03_ab_waittree.txt || exit 1
    timeout -k 10 200 python tools/tune_reproject.py --workload $wl --place --rounds 9 "" 2>/dev/null | grep median | sed "s/^/new  $wl /" >> gpurun_out/r03_ab_waittree.txt || exit 1
  done
done
cat gpurun_out/r03_ab_waittree.txt
