#!/usr/bin/env python3
"""Where in one big allocation is the reprojection's DESTINATION fast?  One arena (PXL_ARENA_GIB, default 200), the source
at the far end (or at the start when the destination reaches into it), the destination slid through the arena in steps of
PXL_STEP_GIB; per position the full launch and the same launch with stores only (PXL_REPROJECT_FLAGS=64), HIP-event
medians.  One JSON line per position; a last line with the extremes.  Workload: PXL_WL (cfg3 default)."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = os.environ.get("PXL_WL", "cfg3")
shape_in, wcs_in, shape_out, wcs_out = bench.workload_geometry(wl)[:4]
sh = pj.DecStripReprojector(shape_in, wcs_in, shape_out, wcs_out, rank=0, world=1, device=dev)
os.environ["PXL_REPROJECT_FLAGS"] = "64"
st = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)       # stores only
del os.environ["PXL_REPROJECT_FLAGS"]
ns = math.prod(sh.src_tensor_shape()); nd = math.prod(sh.dst_tensor_shape())
GiB = 1 << 30
total = int(os.environ.get("PXL_ARENA_GIB", "200"))
step = float(os.environ.get("PXL_STEP_GIB", "2"))
arena = torch.empty(total * GiB // 8, dtype=torch.float64, device=dev)
print(json.dumps({"workload": wl, "arena_GiB": total, "arena_va": hex(arena.data_ptr()), "dst_GiB": round(nd * 8 / GiB, 2), "src_GiB": round(ns * 8 / GiB, 2)}), flush=True)


def ev_time(fn, reps=5):
    fn(); torch.cuda.synchronize(dev)
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(dev); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


src_hi = arena[arena.numel() - ns:].view(sh.src_tensor_shape())
src_lo = arena[:ns].view(sh.src_tensor_shape())
pj.fill_random_(src_hi, 1234); pj.fill_random_(src_lo, 1234)
sh.plan.build_tables(); st.build_tables()
n = sh.dst_window[1]
rows = []
off = 0.0
while (off * GiB + nd * 8) <= total * GiB:
    o = int(off * GiB) // 8 // 32 * 32                 # 256-byte aligned element offset
    far = (o + nd) <= arena.numel() - ns
    if not far and o < ns:
        off += step
        continue
    src = src_hi if far else src_lo
    dst = arena[o:o + nd].view(sh.dst_tensor_shape())
    full = ev_time(lambda: sh.plan.execute_rows(src, dst, 0, n))
    stores = ev_time(lambda: st.execute_rows(src, dst, 0, n))
    rows.append((off, full, stores))
    print(json.dumps({"dst_offset_GiB": off, "full_ms": round(full, 4), "stores_only_ms": round(stores, 4), "src": "high" if far else "low"}), flush=True)
    if not far:
        pj.fill_random_(src_lo, 1234)                  # the destination may have overwritten it
    off += step
fs = sorted(r[1] for r in rows)
print(json.dumps({"positions": len(rows), "full_ms_min": round(fs[0], 4), "full_ms_median": round(fs[len(fs) // 2], 4), "full_ms_max": round(fs[-1], 4)}))
