#!/usr/bin/env python3
"""Batched Gnomonic evaluators (A16, tan_proj.jl:44-75) on 1e8 points and the Gnomonic posmap of a 8192^2 patch:
HIP-event medians, points per second and the fraction of the 32 B/point (16 B/pixel for posmap) byte roofline."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
n = int(float(os.environ.get("PXL_N", "1e8")))
N = 8192
wcs = pj.Gnomonic((-0.5 / 60, 0.5 / 60), (N / 2 + 0.5, N / 2 + 0.5), (40.0, -25.0))
geom = ((N, N), wcs)
pi_ = torch.empty(n, dtype=torch.float64, device=dev).uniform_(1.0, float(N))          # two N-vectors (the reference has no
pj_ = torch.empty(n, dtype=torch.float64, device=dev).uniform_(1.0, float(N))          # 2xN method for Gnomonic, tan_proj.jl:44-75)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
ra, dec = pj.pix2sky(geom, pi_, pj_, safe=False)
for name, fn, bytes_ in (("pix2sky(i, j) Gnomonic, two N-vectors", lambda: pj.pix2sky(geom, pi_, pj_, safe=False), 32.0 * n),
                         ("sky2pix(ra, dec) Gnomonic, two N-vectors", lambda: pj.sky2pix(geom, ra, dec, safe=False), 32.0 * n),
                         ("posmap Gnomonic 8192^2", lambda: pj.posmap((N, N), wcs, device=dev), 16.0 * N * N)):
    ms = t(fn)
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "GB/s": round(bytes_ / ms / 1e6, 1), "frac_of_8TBs": round(bytes_ / ms / 1e6 / 8000, 4)}), flush=True)
