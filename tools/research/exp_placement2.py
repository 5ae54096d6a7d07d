"""Which stream carries the placement effect?  For several re-allocations: full kernel, loads only (flags=2),
stores only (flags=64)."""
import math, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry("cfg4")
nx, ny, nc = shape_in
plans = {}
for name, flags in (("full", "0"), ("loads", "2"), ("stores", "64")):
    os.environ["PXL_REPROJECT_FLAGS"] = flags
    plans[name] = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    plans[name].build_tables()
def t(plan, src, dst, reps=5):
    plan.execute_rows(src, dst, 0, ny); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.execute_rows(src, dst, 0, ny); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
random.seed(int.from_bytes(os.urandom(4), "little"))
keep = []
for k in range(8):
    src = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
    dst = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
    pj.fill_random_(src, 1234)
    print("placement %d: full %.3f  loads-only %.3f  stores-only %.3f ms" % (
        k, t(plans["full"], src, dst), t(plans["loads"], src, dst), t(plans["stores"], src, dst)))
    del src, dst
    keep.append(torch.empty(int(random.uniform(0.2, 6.0) * 2**30), dtype=torch.uint8, device=dev))
    if len(keep) > 3:
        keep.pop(0)
    torch.cuda.empty_cache()
