# two separate allocations vs one allocation with the destination above the source (bench.py's default since late round 2),
# alternating, a fresh process each time
wl=${1:-cfg4}; n=${2:-3}
for i in $(seq $n); do
  python bench.py --workload $wl --steps 10 --warmup 2 --sustain-seconds 0 --no-cpu-baseline --no-configs --no-traffic --two-allocations 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl two-allocations', d['ms_per_step'], d['roofline']['frac'])"
  python bench.py --workload $wl --steps 10 --warmup 2 --sustain-seconds 0 --no-cpu-baseline --no-configs --no-traffic --arena 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl one-allocation', d['ms_per_step'], d['roofline']['frac'])"
done
