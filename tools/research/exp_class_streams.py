#!/usr/bin/env python3
"""Do the streaming evaluators care which memory class their buffers are in?  One 200 GiB allocation, classes mapped with the store
probe; pix2sky! (16 B in + 16 B out per point) with input and output in the SAME class and in DIFFERENT classes, posmap (two write-only
maps) likewise, a plain copy for scale.  One JSON line per case."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import pixell_jl_amd as pj

dev = torch.device("cuda:0")
GiB = 1 << 30
arena = torch.empty(int(os.environ.get("PXL_ARENA_GIB", "200")) * GiB, dtype=torch.uint8, device=dev)
offs, labels, info = pj.map_classes(arena, step_gib=2)
runs = []
for o, l in zip(offs, labels):
    if not runs or runs[-1][0] != l:
        runs.append([l, o, o])
    runs[-1][2] = o + 2 * GiB
print(json.dumps({"classes": info["classes"], "runs_label_from_to_GiB": [[l, a // GiB, b // GiB] for l, a, b in runs]}), flush=True)
big = sorted([r for r in runs if r[2] - r[1] >= 16 * GiB], key=lambda r: -(r[2] - r[1]))
A = big[0]
B = next(r for r in big[1:] if r[0] != A[0])
f64 = arena.view(torch.float64)


def buf(off, n):
    return f64[off // 8: off // 8 + n]


def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
g = (shape, wcs)
n = 300_000_000                       # 4.8 GB in + 4.8 GB out
nb = 2 * n
half = 8 * GiB
cases = {"same class": (A[1], A[1] + half), "different classes": (A[1], B[1])}
for name, (o_in, o_out) in cases.items():
    pix = buf(o_in, nb).view(n, 2)
    out = buf(o_out, nb).view(n, 2)
    pj.fill_random_(pix, 1, kind="uniform"); pix.mul_(float(shape[1]))
    ms = t(lambda: pj.pix2sky_(g, pix, out, safe=False))
    print(json.dumps({"kernel": "pix2sky!(safe=false) 3e8 points", "buffers": name, "ms": round(ms, 4), "frac_of_8TBs": round(32.0 * n / ms / 1e6 / 8000, 4)}), flush=True)
    ms = t(lambda: pj.sky2pix_(g, pix, out, safe=True))
    print(json.dumps({"kernel": "sky2pix!(safe=true) 3e8 points", "buffers": name, "ms": round(ms, 4), "frac_of_8TBs": round(32.0 * n / ms / 1e6 / 8000, 4)}), flush=True)
    ms = t(lambda: out.copy_(pix))
    print(json.dumps({"kernel": "torch copy 4.8 GB", "buffers": name, "ms": round(ms, 4), "frac_of_8TBs": round(32.0 * n / ms / 1e6 / 8000, 4)}), flush=True)
lib = pj.load_library()
wref = wcs.to_struct()
sh2 = (C.c_int64 * 2)(shape[0], shape[1])
npx = shape[0] * shape[1]
for name, (o_ra, o_dec) in cases.items():
    ra, dec = buf(o_ra, npx), buf(o_dec, npx)

    def posmap():
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        pj._lib.check(lib.pxl_posmap_car_f64(C.byref(wref), sh2, 0, shape[1], C.c_void_p(ra.data_ptr()), C.c_void_p(dec.data_ptr()), 1, s))
    ms = t(posmap)
    print(json.dumps({"kernel": "posmap 43200x21601 (two write-only maps)", "buffers": name, "ms": round(ms, 4), "frac_of_8TBs": round(16.0 * npx / ms / 1e6 / 8000, 4)}), flush=True)
