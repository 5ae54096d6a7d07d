#!/usr/bin/env python3
"""pixareamap! (7.5 GB) and posmap (2 x 7.5 GB) of the 0.5-arcmin map, write-only, with 1, 2, 4, 8 write fronts per map; interleaved
rounds in one process."""
import json, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
m = pj.Enmap(torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
res = {}
for rnd in range(9):
    for f in ("1", "2", "4", "8"):
        os.environ["PXL_AREA_FRONTS"] = f
        pj.pixareamap_(m); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pj.pixareamap_(m); b.record(); torch.cuda.synchronize()
        res.setdefault(f, []).append(a.elapsed_time(b))
pres = {}
for rnd in range(9):
    for f in ("1", "2", "4", "8"):
        os.environ["PXL_POSMAP_FRONTS"] = f
        pj.posmap(shape, wcs, device=dev); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pm = pj.posmap(shape, wcs, device=dev); b.record(); torch.cuda.synchronize()
        pres.setdefault(f, []).append(a.elapsed_time(b))
        del pm
for f, ts in pres.items():
    ts.sort()
    print(json.dumps({"kernel": "posmap (two maps)", "write_fronts_per_map": int(f), "ms_median": round(ts[4], 4), "ms_min": round(ts[0], 4), "frac_of_8TBs": round(16.0 * shape[0] * shape[1] / ts[4] / 1e6 / 8000, 4)}), flush=True)
for f, ts in res.items():
    ts.sort()
    print(json.dumps({"write_fronts": int(f), "ms_median": round(ts[4], 4), "ms_min": round(ts[0], 4), "frac_of_8TBs": round(8.0 * shape[0] * shape[1] / ts[4] / 1e6 / 8000, 4)}), flush=True)
