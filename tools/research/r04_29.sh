#!/bin/bash
# GenericReprojectPlan: parity, the mosaic with plans reused, a default line
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_plan
mkdir -p $out
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "generic or abi or tan or mosaic" > $out/tests.txt 2>&1; rc=$?; tail -3 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep -v amdgpu.ids > $out/mosaic.txt; cut -c1-420 $out/mosaic.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mosaic -- python3 $R/tools/bench_tan_mosaic.py > /dev/null 2>&1
cat $out/stats_mosaic/*/*kernel_stats.csv | grep "generic\|lattice" | cut -c1-170
