#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_twelfth
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "entrypoints or placement or bench or generic or mosaic" > $out/pytest_sel.log 2>&1; echo rc=$? >> $out/pytest_sel.log; tail -6 $out/pytest_sel.log | cut -c1-300
echo "== default bench"; timeout -k 10 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_twelfth/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['buffer_placement']['allocation'])
for k,v in d['configs'].items():
    if 'class_aware_placement' in v:
        a=v['api_default']
        print(k,'placed',v['class_aware_placement']['kernel_ms_avg'],v['class_aware_placement']['frac'],'plain',v['plain_first_placement']['kernel_ms_avg'],v['plain_first_placement']['frac'],'api',a['ms_per_call'],a['frac'],a['first_call_s'],a['held_over_pair'],a['allocation'])
m=d['evaluators']['CAR->TAN mosaic']; print('mosaic',m.get('ms'),m.get('frac'),m.get('check'),m.get('error'))
PY
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PXL_BENCH_POINTS=2e8 PXL_BENCH_SAMPLER=pairs timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 > /dev/null 2> $out/pmc$i.err || { echo "cfg5 pass $i failed"; tail -2 $out/pmc$i.err; }
done
python3 $R/tools/summarize_pmc.py $out k_sample k_build_rowpairs > $out/cfg5_counters.txt; tail -45 $out/cfg5_counters.txt
