#!/usr/bin/env python3
"""Scattered sampler vs map footprint, from L2-sized to HBM-sized maps: where does the gather rate change?
Points are uniform on the sphere; the map is a full-sky CC map of nx columns.  Prints Gpts/s for the direct 4-tap
kernel and the row-pair kernel.  (Evidence for DESIGN 9.3: which level of the hierarchy bounds config 5.)"""
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
for nx in (128, 256, 512, 1024, 2048, 4096, 8192, 16384, 43200):
    shape, wcs = pj.fullsky_geometry(2 * math.pi / nx)
    data = torch.randn((shape[1], shape[0]), dtype=torch.float64, device=dev)
    m = pj.Enmap(data, wcs)
    mb = data.numel() * 8 / 2**20
    td = t(lambda: pj.sample_bilinear(m, sky))
    pairs = pj.SamplePairs(m)
    tp = t(lambda: pj.sample_bilinear(None, sky, pairs=pairs))
    print("map %5dx%5d f64 %9.2f MiB: direct %6.2f Gpts/s (%6.1f G lane-loads/s) | pairs %6.2f Gpts/s (%6.1f G lane-loads/s)" % (
        shape[0], shape[1], mb, n / td / 1e6, 4 * n / td / 1e6, n / tp / 1e6, 2 * n / tp / 1e6), flush=True)
    del pairs, data, m
    torch.cuda.empty_cache()
