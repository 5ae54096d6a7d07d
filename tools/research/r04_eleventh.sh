#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_eleventh
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "unwind or pix2sky or soa or safe" > $out/pytest_unwind.log 2>&1; echo rc=$? >> $out/pytest_unwind.log; tail -4 $out/pytest_unwind.log | cut -c1-300
grep -q "rc=0" $out/pytest_unwind.log || exit 1
echo "== unwind two-pass"; PXL_UNWIND_ONEPASS=0 timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2
for v in "" _u4w8 _u8w16 _u2w16; do echo "== unwind one-pass, parallel look-back $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2; done
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo rc=$? >> $out/pytest_all.log; tail -4 $out/pytest_all.log | cut -c1-300
timeout -k 10 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_eleventh/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_ms_avg'])
for k,v in d['configs'].items():
    if 'class_aware_placement' in v:
        print(k,'placed',v['class_aware_placement']['kernel_ms_avg'],v['class_aware_placement']['frac'],'plain',v['plain_first_placement']['kernel_ms_avg'],v['plain_first_placement']['frac'],'api',v['api_default'])
    else:
        print(k,{q:v.get(q) for q in ('ms_per_step','kernel_ms_avg','frac')})
for k,v in d['evaluators'].items(): print(k, v['ms'], v['frac'])
PY
