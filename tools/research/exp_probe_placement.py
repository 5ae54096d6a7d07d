#!/usr/bin/env python3
"""Does the WRITE speed of a region of device memory, measured with a plain fill, predict how fast the cfg4 reprojection
runs with its destination there?  One big allocation; per 4 GiB window: the rate of a 1 GiB fill; then the kernel with
the destination at several offsets (source at the far end or at the start).  Prints one JSON line."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
shape_in, wcs_in, shape_out, wcs_out = bench.workload_geometry("cfg4")[:4]
sh = pj.DecStripReprojector(shape_in, wcs_in, shape_out, wcs_out, rank=0, world=1, device=dev)
ns = math.prod(sh.src_tensor_shape()); nd = math.prod(sh.dst_tensor_shape())
GiB = 1 << 30
total_gib = int(os.environ.get("PXL_ARENA_GIB", "96"))
arena = torch.empty(total_gib * GiB // 8, dtype=torch.float64, device=dev)
def ev_time(fn, reps=3):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(dev); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
fill = {}
for off in range(0, total_gib, 4):
    w = arena[off * GiB // 8:(off + 1) * GiB // 8]
    w.zero_(); torch.cuda.synchronize(dev)
    fill[off] = round(1.0737 / ev_time(lambda: w.zero_()) * 1e3, 1)       # GB/s... 1 GiB in ms -> GB/s
need = (nd * 8 + GiB - 1) // GiB
rows = []
src_hi = arena[(total_gib * GiB - ns * 8) // 8:].view(sh.src_tensor_shape())
src_lo = arena[:ns].view(sh.src_tensor_shape())
sh.plan.build_tables()
n = sh.dst_window[1]
for off in range(0, total_gib - need + 1, 8):
    dst = arena[off * GiB // 8: off * GiB // 8 + nd].view(sh.dst_tensor_shape())
    # source away from the destination: at the far end unless the destination reaches into it
    far = (off + need) * GiB <= total_gib * GiB - ns * 8
    src = src_hi if far else src_lo
    if not far and off * GiB < ns * 8:
        continue
    src.zero_(); torch.cuda.synchronize(dev)
    ms = ev_time(lambda: sh.plan.execute_rows(src, dst, 0, n), reps=5)
    wins = [fill[o] for o in range(off - off % 4, off + need, 4) if o in fill]
    rows.append({"dst_offset_GiB": off, "src": "high end" if far else "low end", "kernel_ms": round(ms, 3),
                 "fill_GBs_of_its_windows_min_mean": [min(wins), round(sum(wins) / len(wins), 1)]})
print(json.dumps({"arena_GiB": total_gib, "fill_GBs_per_4GiB_window": fill, "kernel_by_destination": rows}))
