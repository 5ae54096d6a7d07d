#!/bin/bash
# streaming unwind variants (chunk size, waves per workgroup); then a default line with the plain-pair headline
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_stream2
mkdir -p $out
cd $R
for v in u2 u8 w8 w8u8 w4u8; do for per in 1 2 4; do
  echo "== $v per_cu=$per $(PXL_LIB_PATH=$R/variants/lib_$v.so PXL_UNWIND_ONEPASS=2 PXL_UNWIND_STREAM_PER_CU=$per timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 | tr '\n' ' ')"
done; done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_stream2/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("headline_allocation_policy"), d["roofline"].get("frac_other_policies"))
print({k:(v.get("kernel_ms_avg"),v.get("frac")) for k,v in d["configs"].items()})
PY
