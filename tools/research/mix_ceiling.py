#!/usr/bin/env python3
"""Structure-free ceiling of the 2x-refinement traffic mix (tools/research/mix_ceiling.hip), beside k_reproject_dma on the SAME maps.

    python tools/research/mix_ceiling.py [--place] [--rounds 9]

Prints one JSON line per variant: median / min ms and the algorithmic GB/s of cfg3 (9.332 GB).  Build the library first:
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/research/mix_ceiling.hip -o tools/research/libmix_ceiling.so
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import pixell_jl_amd as pj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--place", action="store_true")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--workload", default="cfg3")
    args = ap.parse_args()
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmix_ceiling.so"))
    lib.mix_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.mix_launch.restype = C.c_int
    dev = torch.device("cuda:0")
    shape_in, wcs_in, shape_out, wcs_out, desc = bench.workload_geometry(args.workload)
    nx, ny, nc = shape_in
    nxo, nyo = shape_out
    assert nc == 1
    keep = None
    if args.place:
        src, dst, info = pj.place_pair((1, ny, nx), (1, nyo, nxo), dtype=torch.float64, device=dev)
        keep = info.pop("arena")
        print(json.dumps({"placement": info["placement"], "source": info["source"], "classes": info["classes"]}))
    else:
        src = torch.empty((1, ny, nx), dtype=torch.float64, device=dev)
        dst = torch.empty((1, nyo, nxo), dtype=torch.float64, device=dev)
    pj.fill_random_(src, 1234)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    stream = torch.cuda.current_stream(dev)
    alg = 8.0 * (nx * ny + nxo * nyo)

    def real():
        plan.execute(src, dst)

    def mix(rh, depth, mode, width=4):
        def f():
            rc = lib.mix_launch(src.data_ptr(), dst.data_ptr(), nx, ny, nxo, nyo, rh, depth, mode, width, C.c_void_p(stream.cuda_stream))
            assert rc == 0, rc
        return f

    variants = [("k_reproject_dma (the product kernel, bit-exact output)", real)]
    for rh, depth in ((32, 4), (32, 8), (64, 8), (16, 4), (64, 4), (32, 2), (8, 4), (8, 2), (16, 2), (16, 8)):
        variants.append(("mix loads+stores rh=%d depth=%d" % (rh, depth), mix(rh, depth, 0)))
    for rh, depth, width in ((16, 4, 2), (32, 4, 2), (16, 4, 8), (32, 4, 8), (8, 2, 8)):
        variants.append(("mix loads+stores rh=%d depth=%d, %d-column tiles" % (rh, depth, 128 * width), mix(rh, depth, 0, width)))
    variants.append(("mix loads + NON-TEMPORAL stores rh=16 depth=4", mix(16, 4, 3)))
    variants.append(("mix stores only rh=32, 256-column tiles", mix(32, 4, 1, 2)))
    variants.append(("mix stores only rh=32, 1024-column tiles", mix(32, 4, 1, 8)))
    variants.append(("mix stores only rh=32", mix(32, 4, 1)))
    variants.append(("mix loads only rh=32 depth=4", mix(32, 4, 2)))
    variants.append(("mix loads only rh=32 depth=8", mix(32, 8, 2)))
    times = {name: [] for name, _ in variants}
    for name, f in variants:          # warm-up
        f()
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for name, f in variants:      # interleaved A/B
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f()
            e0.record()
            for _k in range(5):
                f()
            e1.record()
            e1.synchronize()
            times[name].append(e0.elapsed_time(e1) / 5)
    for name, _ in variants:
        t = times[name]
        med = statistics.median(t)
        print(json.dumps({"variant": name, "median_ms": round(med, 4), "min_ms": round(min(t), 4),
                          "cfg3_algorithmic_GBs": round(alg / med / 1e6, 1), "frac_of_8TBs": round(alg / med / 1e6 / 8000, 4)}))


if __name__ == "__main__":
    main()
