#!/bin/bash
# Counters of the reprojection kernel's STORES into the two halves of one 45 GB allocation (tools/research/exp_placement_vmm pmc):
# which part of the memory system tells a fast-writing region from a slow one?  One rocprofv3 pass per counter group.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_store_regions
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
exe=$R/tools/research/exp_placement_vmm
$exe 1 pmc > $out/plain.jsonl 2>&1; cat $out/plain.jsonl
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- $exe 1 pmc > $out/pmc$i.jsonl 2> $out/pmc$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, json, collections
out = "$out"
for d in sorted(glob.glob(out + "/pmc*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: continue
    times = [json.loads(l) for l in open(d.rstrip("/") + ".jsonl") if l.startswith("{")]
    rows = [r for r in csv.DictReader(open(f[0])) if "k_reproject_dma" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by.items():
        v.sort()
        vals = [x for _, x in v]
        # per call of time_kernel: 3 warm-ups + 10 timed launches = 13 dispatches; first 13 = high half, next 13 = low half
        hi, lo = vals[3:13], vals[16:26]
        print("%-42s high half (%.3f ms) median %16.1f | low half (%.3f ms) median %16.1f | ratio %.3f" % (
            c, times[0]["kernel_ms"], sorted(hi)[len(hi)//2], times[1]["kernel_ms"], sorted(lo)[len(lo)//2],
            (sorted(lo)[len(lo)//2] / sorted(hi)[len(hi)//2]) if sorted(hi)[len(hi)//2] else float("nan")))
PY
