#!/bin/bash
# Per-channel view of the same two store-only cases as tools/research/pmc_store_regions.sh (un-summed TCC counters).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_store_channels
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
exe=$R/tools/research/exp_placement_vmm
i=0
for grp in "TCC_EA0_WRREQ" "TCC_BUSY" "TCC_EA0_WRREQ_STALL"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv json -d $out/pmc$i -- $exe 1 pmc > $out/pmc$i.jsonl 2> $out/pmc$i.err || echo "pass $i failed"
  ls -la $out/pmc$i/*/ | head; head -3 $out/pmc$i/*/*counter_collection.csv
done
