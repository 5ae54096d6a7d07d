#!/bin/bash
# two default lines and one --placed headline on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_final2
mkdir -p $out
cd $R
for k in 1 2; do S=$SECONDS; python3 bench.py > $out/bench_$k.json 2> $out/bench_$k.err; echo run $k rc=$? seconds=$((SECONDS-S)); done
python3 bench.py --placed --no-configs --no-cpu-baseline > $out/bench_placed.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("bench_1","bench_2","bench_placed"):
    d=json.loads(open("gpurun_out/r04_final2/%s.json"%f).read().strip().splitlines()[-1])
    al=d["config"]["buffer_placement"]["allocation"]
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], al.get("placement"), al.get("tries"), al.get("seconds"), d["roofline"].get("frac_other_policies"))
    if "configs" in d:
        print({k:(v.get("kernel_ms_avg"),v.get("frac")) for k,v in d["configs"].items()})
        print({k:(v.get("ms"),v.get("frac")) for k,v in d["evaluators"].items()})
PY
