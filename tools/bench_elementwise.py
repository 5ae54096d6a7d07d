#!/usr/bin/env python3
"""HBM-roofline numbers for the streaming evaluators: HIP-event timing of
pix2sky / sky2pix on 2xN batches (32 B/point), posmap (16 B/pixel, write-only), pixareamap (8 B/pixel) and
the scattered sampler.  Prints one JSON line per kernel."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pixell_jl_amd as pj  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
    g = (shape, wcs)
    n = 400_000_000
    pix = torch.empty((n, 2), dtype=torch.float64, device=dev)
    pj.fill_random_(pix, 1, kind="uniform")
    pix.mul_(float(shape[1]))
    out = torch.empty_like(pix)
    rows = []

    def rec(name, ms, bytes_, units, unit_name):
        rows.append({"kernel": name, "ms": round(ms, 4), "GB/s": round(bytes_ / ms / 1e6, 1),
                     "frac_of_8TBs": round(bytes_ / ms / 1e6 / 8000, 4), unit_name + "/s": round(units / ms * 1e3, 1)})

    rec("pix2sky! 2xN safe=false (k_pix2sky_pairs)", timeit(lambda: pj.pix2sky_(g, pix, out, safe=False)), 32.0 * n, n / 1e6, "Mpts")
    sky = out.clone()
    rec("sky2pix! 2xN safe=false (k_sky2pix_pairs)", timeit(lambda: pj.sky2pix_(g, sky, out, safe=False)), 32.0 * n, n / 1e6, "Mpts")
    rec("sky2pix! 2xN safe=true  (k_sky2pix_pairs)", timeit(lambda: pj.sky2pix_(g, sky, out, safe=True)), 32.0 * n, n / 1e6, "Mpts")
    # SoA forms (two N-vectors in, two out): pix2sky(m, ivec, jvec) and the three sky2pix roundings
    ip, jp = pix[:, 0].contiguous(), pix[:, 1].contiguous()
    rec("pix2sky(i, j) SoA safe=true (k_pix2sky_soa)", timeit(lambda: pj.pix2sky(g, ip, jp, safe=True)), 32.0 * n, n / 1e6, "Mpts")
    rec("sky2pix(ra, dec) SoA safe=true (k_sky2pix_soa)", timeit(lambda: pj.sky2pix(g, ip, jp, safe=True)), 32.0 * n, n / 1e6, "Mpts")
    del ip, jp
    nb = 100_000_000
    small = pix[:nb].clone()
    small_out = torch.empty_like(small)
    rec("pix2sky! 2xN safe=true (fused rewind + verified-scan unwrap), 1e8 pts",
        timeit(lambda: pj.pix2sky_(g, small, small_out, safe=True), reps=5), 32.0 * nb, nb / 1e6, "Mpts")
    del small, small_out
    del pix, out, sky
    npx = shape[0] * shape[1]
    rec("posmap safe=true (k_posmap_car)", timeit(lambda: pj.posmap(shape, wcs, device=dev)), 16.0 * npx, npx / 1e6, "Mpix")
    m = pj.Enmap(torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
    rec("pixareamap! (k_pixareamap_car)", timeit(lambda: pj.pixareamap_(m)), 8.0 * npx, npx / 1e6, "Mpix")
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
