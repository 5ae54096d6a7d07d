#!/bin/bash
# streaming one-pass unwind (PXL_UNWIND_ONEPASS=2): parity tests, timing beside the block form; then one default line with the >= 16 GiB rule
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_stream
mkdir -p $out
cd $R
PXL_UNWIND_ONEPASS=2 timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "unwind or pix2sky or soa or safe" > $out/tests_stream.txt 2>&1; rc=$?; tail -3 $out/tests_stream.txt; [ $rc -eq 0 ] || exit $rc
for v in 1 2; do echo "== PXL_UNWIND_ONEPASS=$v"; PXL_UNWIND_ONEPASS=$v timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -3 || exit 1; done
echo "== per CU 2"; PXL_UNWIND_ONEPASS=2 PXL_UNWIND_STREAM_PER_CU=2 timeout -k 10 120 python3 tools/prof_unwind.py 2>&1 | grep out-of | tail -2 || exit 1
PXL_UNWIND_ONEPASS=2 timeout -k 10 300 python3 tools/fuzz_parity.py --seconds 40 --only unwind > $out/fuzz_stream.txt 2>&1; echo fuzz rc=$?; tail -2 $out/fuzz_stream.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_stream/bench_default.json").read().strip().splitlines()[-1])
al=d["config"]["buffer_placement"]["allocation"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], al.get("placement"), al.get("policy"), d["roofline"].get("frac_other_policies"))
print({k:(v.get("kernel_ms_avg"),v.get("frac")) for k,v in d["configs"].items()})
PY
