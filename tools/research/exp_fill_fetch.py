#!/usr/bin/env python3
"""Does a plain write stream fetch?  zero_() / fill of 1.9, 7.5 and 22.4 GB buffers, and pixareamap (write-only, 16-byte stores, 64 KB
chunks) on the 7.5 GB map -- run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and read the counter per kernel."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
for gb in (1.866, 7.465, 22.4):
    t = torch.empty(int(gb * 1e9 / 8), dtype=torch.float64, device=dev)
    for _ in range(3):
        t.zero_()
    torch.cuda.synchronize()
    print("zero_", gb, "GB done", flush=True)
    del t
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
m = pj.Enmap(torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
for _ in range(3):
    pj.pixareamap_(m)
ra, dec = pj.posmap(shape, wcs, device=dev)
for _ in range(2):
    ra, dec = pj.posmap(shape, wcs, device=dev)
torch.cuda.synchronize()
