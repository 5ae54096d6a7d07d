#!/bin/bash
# the exact-size class-aware pair (alloc_maps -> place_pair_shifted): fresh processes, headline only, against the plain pair and --placed
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_maps
mkdir -p $out
cd $R
for k in 1 2 3; do
for pol in maps arena placed; do
python3 bench.py --$pol --no-configs --no-cpu-baseline --no-traffic --sustain-seconds 0 > $out/${pol}_$k.json 2> $out/${pol}_$k.err
python3 - "$out/${pol}_$k.json" $pol <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
al=d["config"]["buffer_placement"]["allocation"]
print(sys.argv[2], d["ms_per_step"], d["roofline"]["frac"], (al if isinstance(al,str) else {k:al.get(k) for k in ("placement","ballast_GiB","layout","destination_minor_class_share","boundaries_GiB_from_scout_start","seconds","probes")}))
PY
done
for wl in cfg3 cfg3s; do
python3 bench.py --workload $wl --maps --no-configs --no-cpu-baseline --no-traffic --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); al=d['config']['buffer_placement']['allocation']
print('$wl maps', d['ms_per_step'], d['roofline']['frac'], al.get('placement'), al.get('seconds'))"
done
done
