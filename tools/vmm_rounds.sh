#!/bin/bash
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  rm -rf $R/gpurun_out/uw_$mode; mkdir -p $R/gpurun_out/uw_$mode
  PXL_UNWIND_MSPACE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/uw_$mode -- python3 $R/tools/prof_unwind.py > /dev/null 2>&1
  echo "PXL_UNWIND_MSPACE=$mode"; grep -h "k_unwind\|k_scan_wsums" $R/gpurun_out/uw_$mode/*/*kernel_stats.csv | cut -c1-160
done
