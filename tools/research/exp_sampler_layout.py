#!/usr/bin/env python3
"""Scattered sampler: direct 4-tap gather vs the row-pair layout, by map footprint (Float64/Float32, two
resolutions).  Prints Gpts/s per case; 2e8 uniform-on-sphere points."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
n = int(float(os.environ.get("PXL_N", "2e8")))
sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_sphere_points_(sky, 42)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for nx in (43200, 21600, 10800):
    shape, wcs = pj.fullsky_geometry(2 * math.pi / nx)
    for dt in (torch.float64, torch.float32):
        data = torch.randn((shape[1], shape[0]), dtype=dt, device=dev)
        m = pj.Enmap(data, wcs)
        gb = data.numel() * data.element_size() / 1e9
        td = t(lambda: pj.sample_bilinear(m, sky))
        pairs = pj.SamplePairs(m)
        tb = t(lambda: pairs.rebuild(data))
        tp = t(lambda: pj.sample_bilinear(None, sky, pairs=pairs))
        print("map %5dx%5d %s %5.2f GB: direct %6.2f Gpts/s | pairs %6.2f Gpts/s (x%.2f), build %.2f ms" % (
            shape[0], shape[1], str(dt)[6:], gb, n / td / 1e6, n / tp / 1e6, td / tp, tb), flush=True)
        del pairs, data, m
        torch.cuda.empty_cache()
