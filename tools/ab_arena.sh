# plain (two allocations) vs --arena (one allocation, destination above the source), alternating, fresh process each time
wl=${1:-cfg4}; n=${2:-3}
for i in $(seq $n); do
  python bench.py --workload $wl --steps 10 --warmup 2 --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl plain', d['ms_per_step'], d['roofline']['frac'])"
  python bench.py --workload $wl --steps 10 --warmup 2 --sustain-seconds 0 --arena 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl arena', d['ms_per_step'], d['roofline']['frac'])"
done
