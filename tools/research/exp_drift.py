"""Does the kernel time drift inside a long-running process (clock / power-state ramps)?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry("cfg4")
nx, ny, nc = shape_in
src = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
dst = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
pj.fill_random_(src, 1234)
plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
plan.build_tables()
n = 600
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for a, b in ev:
    a.record(); plan.execute_rows(src, dst, 0, ny); b.record()
torch.cuda.synchronize()
ts = [a.elapsed_time(b) for a, b in ev]
for lo in range(0, n, 60):
    seg = sorted(ts[lo:lo + 60])
    print("launches %3d-%3d: min %.3f  median %.3f  max %.3f ms" % (lo, lo + 59, seg[0], seg[30], seg[-1]))
