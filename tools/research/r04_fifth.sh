#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_fifth
mkdir -p $out
cd $R
for wl in "cfg3 --strip 3/8" "cfg3 --place" "cfg3" "cfg3s" "cfg4 --strip 3/8" "up4 --place" "cfg2" "cfg4 --place"; do
  echo "== tune (bursts) $wl"
  timeout -k 10 400 python3 tools/tune_reproject.py --workload $wl --rounds 5 --burst 3 "nt=0" "nt=1" "rh=16,nt=0" "rh=16,nt=1" "rh=8,nt=1" 2>&1 | grep -v amdgpu.ids | tee -a $out/tune_nt_bursts.txt
done
