#!/usr/bin/env python3
"""Reference point for the roofline fraction: what do the vendor's own device-to-device copies reach on buffers
the size of the cfg4 maps?  (torch copy_ = hipMemcpyAsync D2D; clone of an expression = an elementwise kernel.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = 43200 * 21601 * 3
src = torch.empty(n, dtype=torch.float64, device=dev).normal_()
dst = torch.empty_like(src)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
gb = 2 * n * 8 / 1e9
for name, fn in (("hipMemcpy D2D (dst.copy_(src))", lambda: dst.copy_(src)),
                 ("elementwise kernel (torch.mul(src, 2, out=dst))", lambda: torch.mul(src, 2.0, out=dst))):
    ms = t(fn)
    print("%-50s %7.3f ms  %7.1f GB/s  (%.1f%% of 8 TB/s)" % (name, ms, gb / ms * 1e3, gb / ms * 1e3 / 80))
