#!/usr/bin/env python3
"""Latency of the 2xN evaluators on small and medium batches (launch-bound regime): median wall time per call with
the stream drained after every call, and GPU time per call when calls are queued back to back."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
g = pj.fullsky_geometry(2 * math.pi / 43200)
print("%9s | %-28s | %-28s | %-28s" % ("n", "pix2sky! safe=true", "pix2sky! safe=false", "sky2pix! safe=true"))
for n in (10, 100, 1000, 4096, 4097, 10000, 30000, 100000, 300000, 1000000):
    pix = torch.rand((n, 2), dtype=torch.float64, device=dev) * 20000
    out = torch.empty_like(pix)
    cells = []
    for fn in (lambda: pj.pix2sky_(g, pix, out, safe=True), lambda: pj.pix2sky_(g, pix, out, safe=False),
               lambda: pj.sky2pix_(g, pix, out, safe=True)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        wall = sorted(ts)[len(ts) // 2] * 1e6
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record(); torch.cuda.synchronize()
        cells.append("%7.1f us sync, %6.1f us queued" % (wall, a.elapsed_time(b) * 1e3 / 50))
    print("%9d | %s | %s | %s" % (n, *cells), flush=True)
