#!/usr/bin/env python3
"""The row-pair re-layout of the 0.5-arcmin map (7.5 GB read, 20 GB written) with 1, 2, 4, 8 write fronts; interleaved rounds."""
import json, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
m = torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev)
pj.fill_random_(m, 3)
pairs = pj.SamplePairs(pj.Enmap(m, wcs))
ref = pairs.data.clone()
res = {}
for rnd in range(7):
    for f in ("1", "2", "4", "8"):
        os.environ["PXL_PAIRS_FRONTS"] = f
        pairs.data.zero_()
        pairs.rebuild(m); torch.cuda.synchronize()
        assert torch.equal(pairs.data, ref), f
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pairs.rebuild(m); b.record(); torch.cuda.synchronize()
        res.setdefault(f, []).append(a.elapsed_time(b))
nbytes = m.numel() * 8 + pairs.data.numel() * 8
for f, ts in res.items():
    ts.sort()
    print(json.dumps({"kernel": "k_build_rowpairs", "write_fronts": int(f), "ms_median": round(ts[3], 4), "GBs": round(nbytes / ts[3] / 1e6, 1)}), flush=True)
