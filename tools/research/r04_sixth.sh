#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_sixth
mkdir -p $out
cd $R
for v in pair pair_th64 pair_th64_5; do
  echo "== mosaic V1 $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip_$v.so PXL_GENERIC_V1=1 timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep '"tiled"\|checksum\|max_abs' | cut -c1-330 | tee $out/mosaic_$v.txt
done
cd /tmp && export TMPDIR=/tmp
for v in pair pair_th64; do
PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip_$v.so PXL_GENERIC_V1=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$v -- python3 $R/tools/bench_tan_mosaic.py > /dev/null 2>&1
echo "== kernel stats $v"; cat $out/stats_$v/*/*kernel_stats.csv | grep "generic\|lattice" | cut -c1-140
done
