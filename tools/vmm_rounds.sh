#!/bin/bash
# plain torch allocations against pj.place_pair, alternating fresh processes, cfg3 and cfg4 (full launch, stores only)
for k in 1 2 3; do
  for wl in cfg3 cfg4; do
    python tools/tune_reproject.py --workload $wl --rounds 5 "" "flags=64" 2>&1 | grep -v amdgpu.ids | grep -v "^workload" | sed "s/^/plain  $wl: /"
    python tools/tune_reproject.py --workload $wl --rounds 5 --place "" "flags=64" 2>&1 | grep -v amdgpu.ids | grep -v "^workload" | sed "s/^/placed $wl: /"
  done
done
