#!/usr/bin/env python3
"""Where do the ~100 us of fixed cost per strip launch go?  Time one 1/8-strip interior launch (a) alone, with the
device idle before it, and (b) as the average of 20 launches queued back to back (what a running job does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry("cfg4")
nx, ny, nc = shape_in
for world in (8, 16):
    lay = pj.DecStripLayout(shape_in, wcs_in, shape_out, wcs_out, world // 2, world)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, src_rows=lay.src_window, dst_rows=lay.dst_window, device=dev)
    src = torch.empty((nc, lay.src_window[1], nx), dtype=torch.float64, device=dev); pj.fill_random_(src, 1)
    dst = torch.empty((nc, lay.dst_window[1], shape_out[0]), dtype=torch.float64, device=dev)
    r0, nr = lay.interior[0], lay.interior[1] - lay.interior[0]
    plan.build_tables(); plan.execute_rows(src, dst, r0, nr); torch.cuda.synchronize()
    alone = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.execute_rows(src, dst, r0, nr); b.record(); torch.cuda.synchronize(); alone.append(a.elapsed_time(b))
    queued = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            plan.execute_rows(src, dst, r0, nr)
        b.record(); torch.cuda.synchronize(); queued.append(a.elapsed_time(b) / 20)
    steps = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            plan.build_tables(); plan.execute_rows(src, dst, r0, nr)
        b.record(); torch.cuda.synchronize(); steps.append(a.elapsed_time(b) / 20)
    print("1/%d strip: alone %.4f ms | 20 queued back to back %.4f ms each | with build_tables %.4f ms each" % (
        world, sorted(alone)[4], sorted(queued)[2], sorted(steps)[2]), flush=True)
