#!/usr/bin/env python3
"""Gnomonic pix2sky on 1e8 scattered points of a wide (68 degree) and of a small (4 degree) patch; one JSON line each."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
n = int(1e8)
for N, res, label in ((8192, 0.5, "68-degree patch at dec -25"), (480, 0.5, "4-degree patch at dec -25"), (480, 0.5, "4-degree patch at dec +60")):
    crval = (40.0, 60.0) if "+60" in label else (40.0, -25.0)
    wcs = pj.Gnomonic((-res / 60, res / 60), (N / 2 + 0.5, N / 2 + 0.5), crval)
    geom = ((N, N), wcs)
    pi_ = torch.empty(n, dtype=torch.float64, device=dev).uniform_(1.0, float(N))
    pj_ = torch.empty(n, dtype=torch.float64, device=dev).uniform_(1.0, float(N))
    ts = []
    pj.pix2sky(geom, pi_, pj_, safe=False); torch.cuda.synchronize()
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pj.pix2sky(geom, pi_, pj_, safe=False); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(json.dumps({"patch": label, "pix2sky_ms_median": round(ts[3], 4), "frac_of_8TBs": round(32.0 * n / ts[3] / 1e6 / 8000, 4)}), flush=True)
