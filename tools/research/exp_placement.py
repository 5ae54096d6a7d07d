"""Is the process-to-process spread a property of where the buffers land physically?  Re-allocate the maps
several times inside ONE process (with differently sized dummies in between) and time each placement."""
import math, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry("cfg4")
nx, ny, nc = shape_in
plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
plan.build_tables()
def t(src, dst, reps=7):
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
    print("placement %d: src %x dst %x  -> %.3f ms" % (k, src.data_ptr(), dst.data_ptr(), t(src, dst)))
    del src, dst
    keep.append(torch.empty(int(random.uniform(0.2, 6.0) * 2**30), dtype=torch.uint8, device=dev))   # shuffle the heap
    if len(keep) > 3:
        keep.pop(0)
    torch.cuda.empty_cache()
