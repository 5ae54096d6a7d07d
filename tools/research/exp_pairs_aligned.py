#!/usr/bin/env python3
"""What would a sector-aligned cell layout buy the row-pair sampler?  Same kernel, same 1e9 random points, but with the
cell's left column forced odd (1-based), so that its two 16-byte entries share one 32-byte-aligned half sector: the
1.0-sector-per-point floor, against the 1.25 of the plain row-pair layout.  Uniform in pixel space both times."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj

dev = torch.device("cuda:0")
n = int(float(os.environ.get("PXL_N", "1e9")))
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
nx, ny = shape[:2]
data = torch.empty((ny, nx), dtype=torch.float64, device=dev)
pj.fill_random_(data, 1234, 0, "normal")
m = pj.Enmap(data, wcs)
pairs = pj.SamplePairs(m)


def t(fn, reps=3):
    fn(); torch.cuda.synchronize(dev); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(dev); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device=dev); g.manual_seed(7)
for label in ("any column", "odd left column (aligned cell)"):
    pix = torch.empty((n, 2), dtype=torch.float64, device=dev)
    pix[:, 0].uniform_(1.0, nx - 1.0, generator=g)
    pix[:, 1].uniform_(1.0, ny - 1.0, generator=g)
    if label.startswith("odd"):
        i0 = torch.floor(pix[:, 0]); fr = pix[:, 0] - i0
        i0 = i0 - ((i0.to(torch.int64) + 1) % 2).to(torch.float64)          # make it odd: 1, 3, 5, ...
        pix[:, 0] = torch.clamp(i0, min=1.0) + fr * 0.999
        del i0, fr
    sky = pj.pix2sky(m, pix, safe=False)
    del pix
    ms = t(lambda: pj.sample_bilinear(None, sky, pairs=pairs))
    print(json.dumps({"points": label, "n": n, "ms": round(ms, 3), "Gpts/s": round(n / ms / 1e6, 2)}), flush=True)
    del sky
    torch.cuda.empty_cache()
