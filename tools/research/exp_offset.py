"""Does the relative placement of the source and destination buffers matter?  Same plan, same data, the
destination view shifted by a few byte offsets inside one larger allocation."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry(sys.argv[1] if len(sys.argv) > 1 else "cfg4")
nx, ny, nc = shape_in
nxo, nyo = shape_out
src = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
pj.fill_random_(src, 1234)
n_dst = nc * nyo * nxo
pad = 64 * 1024 * 1024 // 8
big = torch.empty(n_dst + pad, dtype=torch.float64, device=dev)
plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
plan.build_tables()
print("src ptr %x  big ptr %x  delta mod 2^30 = %d" % (src.data_ptr(), big.data_ptr(), (big.data_ptr() - src.data_ptr()) % (1 << 30)))
def t(dst, reps=7):
    plan.execute_rows(src, dst, 0, nyo); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.execute_rows(src, dst, 0, nyo); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for off_bytes in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 16 << 20, (32 << 20) + 8192):
    dst = big[off_bytes // 8: off_bytes // 8 + n_dst].view(nc, nyo, nxo)
    print("dst offset %9d B: %.4f ms" % (off_bytes, t(dst)))
