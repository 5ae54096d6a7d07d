#!/usr/bin/env python3
"""Strip efficiency of the dec-strip sharded step on ONE GPU (no peers here: the halo rows are already resident, so
this is the compute side of a rank's step): for world = 1, 2, 4, 8 the middle rank's strip is timed as
  (a) one launch of all its rows (tables already built),
  (b) the step as issued eagerly: build_tables + interior rows + boundary rows,
  (c) the same launches replayed from one HIP graph (captured through torch.cuda.CUDAGraph).
Prints ms, % of 8 TB/s on the strip's algorithmic bytes, and the ideal (full map / world)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pixell_jl_amd as pj

dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry(os.environ.get("PXL_WORKLOAD", "cfg4"))
nx, ny, nc = shape_in
nxo = shape_out[0]


def med(fn, reps=15, inner=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / inner)
    return sorted(ts)[len(ts) // 2]


full_ms = None
for world in (1, 2, 4, 8):
    sh = pj.DecStripReprojector(shape_in, wcs_in, shape_out, wcs_out, world // 2, world, dev)
    src, dst = sh.alloc_src(), sh.alloc_dst()
    pj.fill_random_(src, 1)
    n = sh.dst_window[1]
    i_lo, i_hi = sh.interior if world > 1 else (0, n)
    sh.plan.build_tables()

    def one():
        sh.plan.execute_rows(src, dst, 0, n)

    def step():
        sh.plan.build_tables()
        if i_hi > i_lo:
            sh.plan.execute_rows(src, dst, i_lo, i_hi - i_lo)
            if i_lo > 0:
                sh.plan.execute_rows(src, dst, 0, i_lo)
            if i_hi < n:
                sh.plan.execute_rows(src, dst, i_hi, n - i_hi)
        else:
            sh.plan.execute_rows(src, dst, 0, n)

    t_one, t_step = med(one), med(step)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        step(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    t_graph = med(g.replay)
    alg = 8.0 * nc * (sh.src_window[1] * nx + n * nxo)
    if world == 1:
        full_ms = t_one
    f = lambda ms: round(alg / (ms * 1e-3) / 1e9 / 8000, 4)
    print(json.dumps({"world": world, "rows": n, "interior": [i_lo, i_hi], "one_launch_ms": round(t_one, 4), "eager_step_ms": round(t_step, 4),
                      "graph_step_ms": round(t_graph, 4), "ideal_ms": round(full_ms / world, 4), "frac_one_launch": f(t_one),
                      "frac_eager_step": f(t_step), "frac_graph_step": f(t_graph)}), flush=True)
    del src, dst, sh, g
    torch.cuda.empty_cache()
