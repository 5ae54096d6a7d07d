#!/usr/bin/env python3
"""Per-stage timing of the tile-binned sampler over a list of knob settings, one process, same inputs.
    python tools/tune_sampler.py "TILE_KB=900,NF=4" "TILE_KB=450,NF=2" ...
Stage times come from running the plan with PXL_SAMPLE_STOP = 1..5 (HIP-event medians) and differencing."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj

dev = torch.device("cuda:0")
n = int(float(os.environ.get("PXL_N", "1e9")))
reps = int(os.environ.get("PXL_REPS", "3"))
shape, wcs = pj.fullsky_geometry(2 * math.pi / int(os.environ.get("PXL_NX", "43200")))
data = torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev)
pj.fill_random_(data, 1234, 0, "normal")
m = pj.Enmap(data, wcs)
sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_sphere_points_(sky, 42)
out = torch.empty((1, n), dtype=torch.float64, device=dev)


def t(fn):
    fn(); torch.cuda.synchronize(dev); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(dev); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


ref = None
for cfg in sys.argv[1:] or [""]:
    for k in [k for k in os.environ if k.startswith("PXL_SAMPLE_")]:
        del os.environ[k]
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("=")
        os.environ["PXL_SAMPLE_" + k] = v
    plan = pj.SampleBinned(m, n)
    cum = []
    for stop in (1, 2, 3, 4, 5):
        os.environ["PXL_SAMPLE_STOP"] = str(stop)
        cum.append(t(lambda: plan.sample(sky, out=out)))
    del os.environ["PXL_SAMPLE_STOP"]
    plan.bind(sky)
    bound_ms = t(lambda: plan.sample_bound(out=out))
    if ref is None:
        ref = pj.sample_bilinear(m, sky[:2_000_000])
    ok = bool(torch.equal(out[:, :2_000_000].view(torch.int64), ref.view(torch.int64)))
    st = [cum[0]] + [cum[i] - cum[i - 1] for i in range(1, 5)]
    print(json.dumps({"cfg": cfg, "tiles": plan.tiles, "total_ms": round(cum[4], 2), "Gpts/s": round(n / cum[4] / 1e6, 2),
                      "count": round(st[0], 2), "tables": round(st[1], 2), "scatter": round(st[2], 2), "gather": round(st[3], 2),
                      "unpermute": round(st[4], 2), "sample_bound_ms": round(bound_ms, 2), "bits_equal_direct_first_2e6": ok, "workspace_GB": round(plan.workspace_bytes / 1e9, 1)}), flush=True)
    plan.close()
    torch.cuda.empty_cache()
