# A/B of the streaming evaluators with 2 / 4 / 8 points per lane per trip (same box, alternating processes)
set -e
cd $GRAFT_REPO_ROOT
for u in 2 8; do make -s -C pixell.jl_amd/csrc OUT=/tmp/libpxl_unr$u.so EXTRA=-DPXL_UNR=$u; done
for rep in 1 2 3; do
  for u in 4 2 8; do
    lib=/tmp/libpxl_unr$u.so; [ $u = 4 ] && lib=$GRAFT_REPO_ROOT/pixell.jl_amd/libpixell_hip.so
    echo -n "UNR=$u: "; PXL_LIB_PATH=$lib python tools/bench_elementwise.py 2>/dev/null | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin]
print(' '.join('%.3f' % x['ms'] for x in r[:3]))"
  done
done
