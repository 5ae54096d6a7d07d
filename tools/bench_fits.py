#!/usr/bin/env python3
"""PCIe-inclusive rates of the FITS path either side of the hot path: write_map (HBM -> big-endian file) and
read_map / read_map_rows (file -> HBM), streamed through pinned buffers (pixell.jl_amd/fits_io.py).

    python tools/bench_fits.py [--res-arcmin 1.0] [--ncomp 3] [--dir /tmp]

The file is written to --dir (page-cache warm on re-read, so the read figure is host-memcpy + PCIe, not disk)."""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pixell_jl_amd as pj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res-arcmin", type=float, default=1.0)
    ap.add_argument("--ncomp", type=int, default=3)
    ap.add_argument("--dir", default="/tmp")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    nx = int(round(360 * 60 / args.res_arcmin))
    shape, wcs = pj.fullsky_geometry(2 * math.pi / nx, dims=(args.ncomp,))
    m = pj.Enmap(torch.empty((args.ncomp, shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
    pj.fill_random_(m.data, 7)
    gb = m.data.numel() * 8 / 1e9
    path = os.path.join(args.dir, "pxl_bench_%d.fits" % os.getpid())
    out = {"map": list(shape), "GB": round(gb, 3), "chunk_MB": float(os.environ.get("PXL_FITS_CHUNK_MB", "128"))}
    try:
        t0 = time.perf_counter(); pj.write_map(path, m); torch.cuda.synchronize(); t1 = time.perf_counter()
        out["write_map_GBps"] = round(gb / (t1 - t0), 2)
        for rep in range(2):
            t0 = time.perf_counter(); back = pj.read_map(path, device=dev); torch.cuda.synchronize(); t1 = time.perf_counter()
            out["read_map_GBps_pass%d" % (rep + 1)] = round(gb / (t1 - t0), 2)
        out["roundtrip_identical"] = bool(torch.equal(back.data, m.data))
        del back
        ny = shape[1]
        t0 = time.perf_counter(); strip, _, _ = pj.read_map_rows(path, ny // 8, ny // 8, device=dev); torch.cuda.synchronize(); t1 = time.perf_counter()
        out["read_map_rows_one_eighth_GBps"] = round(strip.numel() * 8 / 1e9 / (t1 - t0), 2)
        out["strip_identical"] = bool(torch.equal(strip, m.data[:, ny // 8:ny // 8 + ny // 8, :]))
    finally:
        if os.path.exists(path):
            os.remove(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
