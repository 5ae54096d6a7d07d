#!/usr/bin/env python3
"""cfg5 under the profiler: N uniform-on-sphere points sampled from the 43200x21601 Float64 map, a few launches of
each sampler variant (PXL_MODES = comma list of direct,pairs).  Prints one JSON line per variant with the
HIP-event median; run it directly after `rocprofv3 ... --` (tools/collect_cfg5.sh)."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj

dev = torch.device("cuda:0")
n = int(float(os.environ.get("PXL_N", "1e9")))
reps = int(os.environ.get("PXL_REPS", "3"))
modes = os.environ.get("PXL_MODES", "direct,pairs").split(",")
shape, wcs = pj.fullsky_geometry(2 * math.pi / int(os.environ.get("PXL_NX", "43200")))
data = torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev)
pj.fill_random_(data, 1234, 0, "normal")
if os.environ.get("PXL_F32", "0") == "1":          # a Float32 map (PXL_F32=1)
    data = data.to(torch.float32)
m = pj.Enmap(data, wcs)
sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_sphere_points_(sky, 42)


def t(fn):
    fn(); torch.cuda.synchronize(dev); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(dev); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for mode in modes:
    if mode == "direct":
        ms = t(lambda: pj.sample_bilinear(m, sky))
    elif mode == "pairs":
        pairs = pj.SamplePairs(m)
        ms = t(lambda: pj.sample_bilinear(None, sky, pairs=pairs))
        del pairs
    else:
        raise SystemExit("unknown mode " + mode)
    print(json.dumps({"mode": mode, "n": n, "ms": round(ms, 3), "Gpts/s": round(n / ms / 1e6, 2),
                      "alg_GB/s": round(56.0 * n / ms / 1e6, 1), "frac_of_8TBs": round(56.0 * n / ms / 1e6 / 8000, 4)}), flush=True)
    torch.cuda.empty_cache()
