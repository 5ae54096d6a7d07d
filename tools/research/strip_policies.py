#!/usr/bin/env python3
"""One rank's strip of config 4 on one GPU, for world sizes 2 / 4 / 8 (middle rank): kernel time of the strip's launch with its
maps from alloc_pair (plain), alloc_maps (the default policy) and pj.place_pair (144 GiB of head-room).  Bursts of 6 launches."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 43200, dims=(3,))
shape_out = shape_in[:2]
wcs_out = pj.CarClenshawCurtis(wcs_in.cdelt, (wcs_in.crpix[0] + 0.5, wcs_in.crpix[1] + 0.5), wcs_in.crval, wcs_in.unit)


def burst(sh, src, dst, reps=6):
    n = sh.dst_window[1]
    ts = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sh.plan.execute_rows(src, dst, 0, n); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts[1:])[len(ts[1:]) // 2]


for world in (2, 4, 8):
    rank = world // 2
    sh = pj.DecStripReprojector(shape_in, wcs_in, shape_out, wcs_out, rank, world, dev)
    sh.plan.build_tables()
    nbytes = 8.0 * (math.prod(sh.src_tensor_shape()) + math.prod(sh.dst_tensor_shape()))
    row = {"world": world, "rank": rank, "pair_GiB": round(nbytes / 2**30, 2)}
    for name in ("plain", "maps", "placed"):
        torch.cuda.empty_cache()
        if name == "plain":
            src, dst, hold = sh.alloc_pair()
            how = "alloc_pair"
        elif name == "maps":
            src, dst, info = sh.alloc_maps()
            how = info.get("placement")
        else:
            src, dst, info = pj.place_pair(sh.src_tensor_shape(), sh.dst_tensor_shape(), device=dev)
            how = info.get("placement")
        src.normal_()
        ms = burst(sh, src, dst)
        row[name] = {"ms": round(ms, 4), "frac": round(nbytes / ms / 1e6 / 8000, 4), "how": how}
        del src, dst
        hold = info = None
    print(json.dumps(row), flush=True)
