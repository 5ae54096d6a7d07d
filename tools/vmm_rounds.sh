#!/bin/bash
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fillfetch; mkdir -p $R/gpurun_out/fillfetch
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fillfetch -- python3 $R/tools/exp_fill_fetch.py > /dev/null 2>&1
python3 - $R <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/gpurun_out/fillfetch/*/*counter_collection.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "fill" in n.lower() or "pixarea" in n or "posmap" in n or "Fill" in n:
        print("%-70s FETCH_SIZE x2 = %8.4f GB" % (n[:70], 2 * float(r["Counter_Value"]) * 1024 / 1e9))
PY
