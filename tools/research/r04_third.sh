#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_third
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -k "generic or gnomonic or tan or mosaic" > $out/pytest_generic.log 2>&1; echo rc=$? >> $out/pytest_generic.log; tail -3 $out/pytest_generic.log
echo "== mosaic, round-3 pixel kernel"; PXL_GENERIC_V1=1 timeout -k 10 300 python3 tools/bench_tan_mosaic.py > $out/tan_mosaic_v1.txt 2>&1; grep tiled $out/tan_mosaic_v1.txt | cut -c1-330
echo "== mosaic, 2 pixels per lane"; timeout -k 10 300 python3 tools/bench_tan_mosaic.py > $out/tan_mosaic_v2.txt 2>&1; grep -v amdgpu.ids $out/tan_mosaic_v2.txt | cut -c1-330
for wl in "cfg3" "cfg3 --place" "cfg3 --strip 3/8" "cfg3 --strip 0/8 --place" "cfg2" "up4 --place" "cfg4x2 --place"; do
  echo "== tune $wl"
  timeout -k 10 400 python3 tools/tune_reproject.py --workload $wl --rounds 7 "" "rh=16" "rh=16,nt=1" "rh=8" "rh=8,nt=1" "nt=1" 2>&1 | grep -v amdgpu.ids | tee -a $out/tune_rh_nt.txt
done
