#!/bin/bash
# final tree: whole GPU suite, default line, mosaic (one-shot / plans) with kernel stats, fuzz of every kind
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_final5
mkdir -p $out
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; rc=$?; tail -2 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
S=$SECONDS; python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo bench rc=$? seconds=$((SECONDS-S))
timeout -k 10 300 python3 tools/bench_tan_mosaic.py 2>&1 | grep -v amdgpu.ids > $out/mosaic.txt; cut -c1-330 $out/mosaic.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mosaic -- python3 $R/tools/bench_tan_mosaic.py > /dev/null 2>&1
cat $out/stats_mosaic/*/*kernel_stats.csv | grep "generic\|lattice" | cut -c1-170 | tee $out/mosaic_kernel_stats.txt
cp $out/stats_mosaic/*/*kernel_stats.csv $out/mosaic_kernel_stats.csv
cd $R
timeout -k 10 400 python3 tools/fuzz_parity.py --seconds 240 > $out/fuzz.txt 2>&1; echo fuzz rc=$?; tail -1 $out/fuzz.txt
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_final5/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("headline_allocation_policy"), d["roofline"].get("frac_other_policies"))
print({k:(v.get("kernel_ms_avg"),v.get("frac")) for k,v in d["configs"].items()})
print({k:(v.get("ms"),v.get("frac")) for k,v in d["evaluators"].items()})
print(d["evaluators"]["CAR->TAN mosaic"].get("one_shot"), d["evaluators"]["CAR->TAN mosaic"]["check"])
PY
