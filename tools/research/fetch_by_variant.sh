#!/bin/bash
# FETCH_SIZE per launch of k_reproject_dma for tile variants of one workload (rocprofv3 --pmc FETCH_SIZE, one run per variant)
wl=${1:-cfg3}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  d=$R/gpurun_out/fetchvar/$(echo "$v" | tr ',=' '__')_
  rm -rf $d; mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -- python3 $R/tools/tune_reproject.py --workload $wl --rounds 3 "$v" > $d/out.txt 2> $d/err.txt
  python3 - "$v" $d <<'PY'
import csv, glob, sys
v, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*counter_collection.csv")
vals = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k_reproject_dma" in r["Kernel_Name"])
ms = [l for l in open(d + "/out.txt") if "median" in l]
print("%-24s FETCH_SIZE x2 = %.3f GB per launch (%d launches)   %s" % (v or "(default)", 2 * vals[len(vals) // 2] * 1024 / 1e9, len(vals), ms[0].split("median")[1].split("min")[0].strip() if ms else ""))
PY
done
