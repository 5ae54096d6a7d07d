#!/usr/bin/env python3
"""Write-only kernels on the 43200 x 21601 Float64 map (7.47 GB): pixareamap in its two forms, posmap (two maps), and
the vendor fills of the same bytes (torch fill_ = a plain store kernel, hipMemsetAsync through zero_()) as ceilings."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
m = pj.Enmap(torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
npx = shape[0] * shape[1]
def t(fn, reps=15):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
def row(name, ms, nbytes):
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 4)}), flush=True)
for rnd in range(2):
    row("pixareamap! contiguous chunks (k_pixareamap_chunks)", t(lambda: pj.pixareamap_(m)), 8.0 * npx)
    os.environ["PXL_AREA_ROWS"] = "1"
    row("pixareamap! one row per block (k_pixareamap_car)", t(lambda: pj.pixareamap_(m)), 8.0 * npx)
    del os.environ["PXL_AREA_ROWS"]
    row("torch fill_(1.5) of the same map", t(lambda: m.data.fill_(1.5)), 8.0 * npx)
    row("torch zero_() (memset) of the same map", t(lambda: m.data.zero_()), 8.0 * npx)
    row("posmap safe=true, two maps (k_posmap_car)", t(lambda: pj.posmap(shape, wcs, device=dev)), 16.0 * npx)
