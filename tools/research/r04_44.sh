#!/bin/bash
# alloc_maps as the default headline: placement tests, cfg3 with the fallback, two default lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_maps2
mkdir -p $out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_placement.py tests/test_gpu_entrypoints.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for wl in cfg3 cfg3s; do
python3 bench.py --workload $wl --no-configs --no-cpu-baseline --no-traffic --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); al=d['config']['buffer_placement']['allocation']
print('$wl maps', d['ms_per_step'], d['roofline']['frac'], al.get('placement'), al.get('seconds'))"
done
for k in 1 2; do S=$SECONDS; python3 bench.py > $out/bench_$k.json 2> $out/bench_$k.err; echo run $k rc=$? seconds=$((SECONDS-S)); done
python3 - <<'PY'
import json
for f in ("bench_1","bench_2"):
    d=json.loads(open("gpurun_out/r04_maps2/%s.json"%f).read().strip().splitlines()[-1])
    al=d["config"]["buffer_placement"]["allocation"]
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], al.get("placement"), al.get("seconds"), d["roofline"].get("headline_allocation_policy"), d["roofline"].get("frac_other_policies"), d["roofline"].get("traffic"))
    print({k:(v.get("kernel_ms_avg"),v.get("frac")) for k,v in d["configs"].items()})
    print({k:(v.get("ms"),v.get("frac")) for k,v in d["evaluators"].items()})
PY
