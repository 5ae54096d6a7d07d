#!/usr/bin/env python3
"""pix2sky!(...; safe=true) on a long 2xN batch, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_unwind.py`."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
n = int(float(os.environ.get("PXL_N", "1e8")))
pix = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_random_(pix, 1, kind="uniform")
pix.mul_(float(shape[1]))
out = torch.empty_like(pix)
for it in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pj.pix2sky_((shape, wcs), pix, out, safe=True); b.record(); torch.cuda.synchronize()
    print("out-of-place %.4f ms" % a.elapsed_time(b))
ang = out.clone()
for it in range(4):
    ang.copy_(out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pj.unwind_(ang); b.record(); torch.cuda.synchronize()
    print("unwind_ in place %.4f ms" % a.elapsed_time(b))
