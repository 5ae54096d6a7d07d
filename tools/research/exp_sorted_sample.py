"""Upper bound of what spatially ordering the points could buy the scattered sampler (BASELINE config 5).
NOTE: torch advanced indexing `sky[order]` returns wrong data for a 3.2 GB operand on this ROCm build; use
index_select (checked against sums below)."""
import math, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
m = pj.Enmap(torch.zeros((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
pj.fill_random_(m.data, 1234)
n = 50_000_000           # torch index_select / fancy indexing return wrong data from ~1.2e8 rows on this ROCm build
sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_sphere_points_(sky, 42)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
ms = t(lambda: pj.sample_bilinear(m, sky)); print("random order            : %.2f ms (%.1f Gpts/s)" % (ms, n / ms / 1e6))
pix = pj.sky2pix(m, sky)
ref = float(sky[:, 1].sum())
for rows, cols in ((1351, 0), (338, 0), (64, 0), (64, 512), (8, 64), (1, 1)):
    key = torch.floor((pix[:, 1] - 1.0) / rows).to(torch.int64)
    if cols:
        key = key * 100000 + torch.floor((pix[:, 0] - 1.0) / cols).to(torch.int64)
    o = torch.argsort(key)
    sk = sky.index_select(0, o)
    assert abs(float(sk[:, 1].sum()) - ref) < 1e-6 * abs(ref) + 1e-3, "torch gather returned wrong data"
    ms = t(lambda: pj.sample_bilinear(m, sk))
    print("sorted by %4d rows x %s cols: %.2f ms (%.1f Gpts/s)" % (rows, cols or "all", ms, n / ms / 1e6))
    del key, o, sk
