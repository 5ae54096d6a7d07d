#!/usr/bin/env python3
"""Differential fuzzing of libpixell_hip.so against the CPU oracle (test infrastructure, like tests/).

    python tools/fuzz_parity.py --seconds 240 --seed 1 [--log gpurun_out/fuzz.log]

Random geometries well beyond the fixed cases of tests/test_gpu_parity.py: tiny (1..40 pixel) and wide
(up to 20000 columns) maps, scale ratios 1/8..8, flips, Fejer1 offsets, nearly-periodic maps either side of
the 1e-8 test, dec-strip windows, partial execute_rows launches, every kernel variant, random tile-shape knobs,
Float64 and Float32 storage, NaN/Inf/huge coordinates.  Every output buffer sits between NaN canaries, so an
out-of-window write is caught as well as a wrong value.  Every comparison is bit for bit.  A failing case
prints its full parameters (re-run with --only KIND --seed S to reproduce) and the exit code is 1.
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

import pixell_jl_amd as pj       # noqa: E402
from oracle import oracle as O   # noqa: E402

DEV = None
KNOBS = ("PXL_REPROJECT_RH", "PXL_REPROJECT_PAIRS", "PXL_REPROJECT_NS", "PXL_REPROJECT_PF", "PXL_REPROJECT_NT")
PAD = 4099      # canary elements either side of an output buffer


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    it = np.int64 if a.dtype == np.float64 else np.int32
    same = a.view(it) == b.view(it)
    return bool((same | (np.isnan(a) & np.isnan(b))).all())


def to_dev(a, dtype=np.float64):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(DEV)


class Guarded:
    """An output tensor carved out of a larger NaN-filled buffer; check() verifies the canaries."""

    def __init__(self, shape, dtype):
        n = int(np.prod(shape))
        self.buf = torch.full((n + 2 * PAD,), float("nan"), dtype=dtype, device=DEV)
        self.t = self.buf[PAD:PAD + n].view(shape)
        self.n = n

    def canaries_intact(self):
        return bool(torch.isnan(self.buf[:PAD]).all()) and bool(torch.isnan(self.buf[PAD + self.n:]).all())


def rand_geometry(rng, cls=None):
    """(shape, wcs): periodic full-circle maps or partial-sky patches in several size classes."""
    cls = cls or rng.choice(["tiny", "small", "small", "medium", "wide", "tall"])
    lo, hi = {"tiny": ((1, 40), (1, 40)), "small": ((30, 700), (20, 300)), "medium": ((700, 5000), (200, 1500)),
              "wide": ((5000, 20000), (8, 120)), "tall": ((8, 200), (1500, 6000))}[cls]
    nx = int(rng.integers(lo[0], lo[1] + 1))
    ny = int(rng.integers(hi[0], hi[1] + 1))
    sx = -1.0 if rng.random() < 0.7 else 1.0
    sy = 1.0 if rng.random() < 0.7 else -1.0
    mode = rng.choice(["periodic", "patch", "nearly"])
    if mode == "periodic":
        cd1 = sx * 360.0 / nx
        cd2 = sy * rng.uniform(20, 180) / max(ny, 2)
        crpix = (rng.uniform(1, nx + 1), rng.uniform(1, ny + 1))
    elif mode == "nearly":       # nx * |cdelt| within ~1e-7 rad of 2*pi: either side of the 1e-8 periodicity test
        eps = rng.choice([0.0, 1e-10, 3e-9, 3e-8, 1e-6, -1e-10, -3e-9, -3e-8, -1e-6])
        cd1 = sx * (360.0 + math.degrees(eps)) / nx
        cd2 = sy * rng.uniform(20, 180) / max(ny, 2)
        crpix = (rng.uniform(1, nx + 1), rng.uniform(1, ny + 1))
    else:
        res = 10 ** rng.uniform(-2.5, 0.3)
        cd1 = sx * res * rng.uniform(0.8, 1.25)
        cd2 = sy * res
        crpix = (rng.uniform(-30, nx + 30), rng.uniform(-30, ny + 30))
    if rng.random() < 0.15:
        crpix = (round(crpix[0]) + 0.5, round(crpix[1]))       # the reference's own half-integer conventions
    crval = (rng.uniform(-180, 180), 0.0 if rng.random() < 0.8 else rng.uniform(-30, 30))
    ctor = pj.CarClenshawCurtis if rng.random() < 0.8 else pj.CarFejer1
    return (nx, ny), ctor((cd1, cd2), crpix, crval)


def related_geometry(rng, shape, wcs):
    """A destination geometry derived from the source: scale 1/8..8, fractional shifts, same sky region."""
    nx, ny = shape
    s = 2.0 ** rng.uniform(-3, 3)
    if rng.random() < 0.3:
        s = float(rng.choice([0.25, 0.5, 1.0, 2.0, 3.0, 4.0]))
    nxo = max(1, min(int(round(nx / s)) + int(rng.integers(-2, 3)), 24000))
    nyo = max(1, min(int(round(ny / s)) + int(rng.integers(-2, 3)), 8000))
    if abs(nx * abs(wcs.cdelt[0]) - 360.0) < 1e-3 and rng.random() < 0.7:
        cd1 = math.copysign(360.0 / nxo, wcs.cdelt[0] if rng.random() < 0.8 else -wcs.cdelt[0])
    else:
        cd1 = wcs.cdelt[0] * s * (1 if rng.random() < 0.8 else -1)
    cd2 = wcs.cdelt[1] * ny / nyo * (1 if rng.random() < 0.8 else -1) * rng.uniform(0.9, 1.1)
    crpix = (wcs.crpix[0] / s + rng.uniform(-3, 3), (wcs.crpix[1] / s if cd2 * wcs.cdelt[1] > 0 else nyo - wcs.crpix[1] / s)
             + rng.uniform(-3, 3))
    crval = (wcs.crval[0] + rng.uniform(-2, 2) * abs(wcs.cdelt[0]), wcs.crval[1])
    return (nxo, nyo), pj.CarClenshawCurtis((cd1, cd2), crpix, crval)


def fuzz_reproject(rng):
    shape_in, wcs_in = rand_geometry(rng)
    if rng.random() < 0.7:
        shape_out, wcs_out = related_geometry(rng, shape_in, wcs_in)
    else:
        shape_out, wcs_out = rand_geometry(rng, cls=rng.choice(["tiny", "small", "medium"]))
    if shape_in[0] * shape_in[1] > 12e6 or shape_out[0] * shape_out[1] > 12e6:
        return "skipped-size"
    nc = int(rng.choice([1, 1, 2, 3]))
    f32 = rng.random() < 0.35
    nx, ny = shape_in
    nxo, nyo = shape_out
    # dec-strip windows: destination rows [d0, d0+dn), source rows = what they need (+ optional slack), or the full map
    if rng.random() < 0.5 and nyo > 2:
        d0 = int(rng.integers(0, nyo - 1))
        dn = int(rng.integers(1, nyo - d0 + 1))
    else:
        d0, dn = 0, nyo
    s_lo, s_hi = O.reproject_src_rows(wcs_in, shape_in, wcs_out, shape_out, d0, dn)
    mode = rng.choice(["full", "exact", "slack", "short"])
    if mode == "full" or s_hi <= s_lo:
        s0, sn = 0, ny
    elif mode == "exact":
        s0, sn = s_lo, s_hi - s_lo
    elif mode == "slack":
        s0 = max(0, s_lo - int(rng.integers(0, 5)))
        sn = min(ny, s_hi + int(rng.integers(0, 5))) - s0
    else:                         # a window that misses needed rows: those taps read as zero (oracle semantics)
        s0 = min(ny - 1, s_lo + int(rng.integers(0, 3)))
        sn = max(1, min(ny, s_hi - int(rng.integers(0, 3))) - s0)
    src = rng.normal(size=(nc, sn, nx))
    if rng.random() < 0.1:
        src[rng.random(src.shape) < 0.01] = np.nan
    knobs = {}
    if rng.random() < 0.6:
        knobs["PXL_REPROJECT_RH"] = str(rng.choice([1, 2, 3, 8, 16, 32, 64]))
    if rng.random() < 0.6:
        knobs["PXL_REPROJECT_PAIRS"] = str(rng.choice([1, 2, 4]))
    if rng.random() < 0.4:
        knobs["PXL_REPROJECT_NT"] = str(rng.choice([0, 1]))           # non-temporal (the default since round 4) / plain stores
    if rng.random() < 0.4:
        knobs["PXL_REPROJECT_NS"] = str(rng.choice([2, 3, 4, 8]))
    if rng.random() < 0.4:
        knobs["PXL_REPROJECT_PF"] = str(rng.choice([1, 2, 3]))
    variant = int(rng.choice([0, 0, 0, 1, 2]))
    params = dict(shape_in=shape_in, wcs_in=wcs_in, shape_out=shape_out, wcs_out=wcs_out, nc=nc, f32=f32,
                  src_rows=(s0, sn), dst_rows=(d0, dn), knobs=knobs, variant=variant)
    if f32:
        src = src.astype(np.float32)
        expect = O.reproject_f32(wcs_in, (nx, ny, nc), src, wcs_out, shape_out, src_row0=s0, src_nrows=sn,
                                 dst_row0=d0, dst_nrows=dn)
    else:
        expect = O.reproject(wcs_in, (nx, ny, nc), src, wcs_out, shape_out, src_row0=s0, src_nrows=sn,
                             dst_row0=d0, dst_nrows=dn)
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(knobs)
    try:
        plan = pj.ReprojectPlan((nx, ny, nc), wcs_in, shape_out, wcs_out, src_rows=(s0, sn), dst_rows=(d0, dn), device=DEV)
    finally:
        for k in KNOBS:
            os.environ.pop(k, None)
    plan.set_variant(variant)
    assert plan.src_rows_needed() == (s_lo, s_hi), ("src_rows_needed", params, plan.src_rows_needed(), (s_lo, s_hi))
    tdt = torch.float32 if f32 else torch.float64
    d_src = to_dev(src, np.float32 if f32 else np.float64)
    g = Guarded((nc, dn, nxo), tdt)
    if rng.random() < 0.5 or dn < 2:
        plan.execute(d_src, g.t)
        got = g.t.cpu().numpy()
        assert bits_equal(got, expect), ("reproject", params, float(np.nanmax(np.abs(got.astype(float) - expect))))
    else:                                                     # partial launches: untouched rows must stay NaN
        r0 = int(rng.integers(0, dn - 1))
        nr = int(rng.integers(1, dn - r0 + 1))
        plan.build_tables()
        plan.execute_rows(d_src, g.t, r0, nr)
        got = g.t.cpu().numpy()
        assert bits_equal(got[:, r0:r0 + nr], expect[:, r0:r0 + nr]), ("execute_rows", params, r0, nr)
        rest = np.concatenate([got[:, :r0].ravel(), got[:, r0 + nr:].ravel()])
        assert np.isnan(rest).all(), ("execute_rows wrote outside its rows", params, r0, nr)
    assert g.canaries_intact(), ("reproject wrote outside dst", params)
    plan.close()
    return "f32" if f32 else "f64"


def special_values(rng, a, scale):
    """Sprinkle NaN/Inf/huge/zero/denormal values into a coordinate array."""
    n = a.size
    if n == 0:
        return a
    flat = a.reshape(-1)
    k = int(rng.integers(0, max(1, n // 50) + 1))
    idx = rng.integers(0, n, k)
    pool = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e300, -1e300, 5e-324, 1e15 * scale, -1e9 * scale])
    flat[idx] = pool[rng.integers(0, pool.size, k)]
    return a


def rand_n(rng):
    return int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 1023, 1024, 1025, 4096, int(rng.integers(1, 3000)),
                           int(rng.integers(3000, 400000))]))


def fuzz_elementwise(rng):
    shape, wcs = rand_geometry(rng)
    n = rand_n(rng)
    g = (shape, wcs)
    span = 3.0 * max(shape)
    pix = special_values(rng, rng.uniform(-span, span, (n, 2)), span)
    if rng.random() < 0.5:
        pix = np.stack([rng.uniform(0, shape[0] + 1, n), rng.uniform(0, shape[1] + 1, n)], axis=1)
    d = to_dev(pix)
    # NaN/Inf poison the sequential unwrap from that element on; parity there is covered by the dedicated tests,
    # here only finite inputs go through WRAP_UNWIND
    finite = bool(np.isfinite(pix).all())
    which = rng.choice(["p2s", "p2s_soa", "s2p", "s2p_soa"])
    params = dict(shape=shape, wcs=wcs, n=n, which=which)
    if which == "p2s":
        assert bits_equal(pj.pix2sky(g, d, safe=False).cpu().numpy(), O.pix2sky(wcs, pix, O.WRAP_NONE)), params
        assert bits_equal(pj.pix2sky_rewind(g, d).cpu().numpy(), O.pix2sky(wcs, pix, O.WRAP_REWIND)), params
        if finite:
            assert bits_equal(pj.pix2sky(g, d, safe=True).cpu().numpy(), O.pix2sky(wcs, pix, O.WRAP_UNWIND)), params
    elif which == "p2s_soa":
        for safe in (True, False):
            ra, dec = pj.pix2sky(g, to_dev(pix[:, 0]), to_dev(pix[:, 1]), safe=safe)
            era, edec = O.pix2sky_soa(wcs, pix[:, 0].copy(), pix[:, 1].copy(), safe=safe)
            assert bits_equal(ra.cpu().numpy(), era) and bits_equal(dec.cpu().numpy(), edec), (params, safe)
    else:
        sky = np.stack([rng.uniform(-math.pi, math.pi, n) + 2 * math.pi * rng.integers(-9, 10, n),
                        rng.uniform(-math.pi / 2, math.pi / 2, n) + math.pi * rng.integers(-3, 4, n)], axis=1)
        sky = special_values(rng, sky, 1.0)
        safe = bool(rng.random() < 0.6)
        if which == "s2p":
            got = pj.sky2pix(g, to_dev(sky), safe=safe).cpu().numpy()
            assert bits_equal(got, O.sky2pix(wcs, shape, sky, safe=safe, form=O.FORM_RECIP)), (params, safe)
        else:
            ra, dec = sky[:, 0].copy(), sky[:, 1].copy()
            x, y = pj.sky2pix(g, to_dev(ra), to_dev(dec), safe=safe)
            ex, ey = O.sky2pix_soa(wcs, shape, ra, dec, safe=safe, form=O.FORM_RECIP_AV)
            assert bits_equal(x.cpu().numpy(), ex) and bits_equal(y.cpu().numpy(), ey), (params, safe, "av")
            x, y = pj.sky2pix_broadcast(g, to_dev(ra), to_dev(dec), safe=safe)
            ex, ey = O.sky2pix_soa(wcs, shape, ra, dec, safe=safe, form=O.FORM_DIV)
            assert bits_equal(x.cpu().numpy(), ex) and bits_equal(y.cpu().numpy(), ey), (params, safe, "div")
    return which


def fuzz_unwind(rng):
    n = rand_n(rng) + int(rng.choice([0, 0, 700000]))
    period = float(rng.choice([2 * math.pi, 360.0, 1.0, 43200.000000000007, rng.uniform(0.1, 100)]))
    ref = float(rng.choice([0.0, period / 2, rng.uniform(-3 * period, 3 * period)]))
    kind = rng.choice(["walk", "jumps", "ties", "const"])
    if kind == "walk":
        a = np.cumsum(rng.normal(0, period * rng.uniform(0.01, 0.6), n)) + rng.uniform(-5, 5) * period
    elif kind == "jumps":
        a = rng.uniform(-60 * period, 60 * period, n)
    elif kind == "ties":
        a = np.arange(n, dtype=np.float64) * (period / 2) * float(rng.choice([1, -1, 3])) + rng.choice([0.0, 0.25])
    else:
        a = np.full(n, rng.uniform(-4, 4) * period)
    params = dict(n=n, period=period, ref=ref, kind=kind)
    got = pj.unwind_(to_dev(a), period, ref).cpu().numpy()
    assert bits_equal(got, O.unwind_row(a.copy(), period, ref)), params
    got = pj.rewind_(to_dev(a), period, ref).cpu().numpy()
    pick = np.arange(n) if n <= 20000 else rng.integers(0, n, 20000)
    exp = np.array([O.rewind(v, period, ref) for v in a[pick]], dtype=np.float64)
    assert bits_equal(got[pick], exp), ("rewind", params)
    if n:
        two = np.stack([a, -0.37 * a[::-1]], axis=1)
        got = pj.unwind_(to_dev(two), period, ref).cpu().numpy()
        exp = np.stack([O.unwind_row(two[:, 0].copy(), period, ref), O.unwind_row(two[:, 1].copy(), period, ref)], axis=1)
        assert bits_equal(got, exp), ("2xN", params)
        # pix2sky!(safe=true) OUT OF PLACE through an identity WCS: the one-pass kernel (decoupled look-back) beyond 8192 points,
        # the single-block form below; in place: the two-pass form.  Period 2 pi, ref 0 are the evaluator's own.
        ident = ((10, 10), pj.CarClenshawCurtis((1.0, 1.0), (0.0, 0.0), (0.0, 0.0), 1.0))
        d = to_dev(two)
        exp = O.pix2sky(ident[1], two, O.WRAP_UNWIND)
        got = pj.pix2sky_(ident, d, torch.empty_like(d), safe=True).cpu().numpy()
        assert bits_equal(got, exp), ("pix2sky! out of place", params)
        got = pj.pix2sky_(ident, d, d, safe=True).cpu().numpy()
        assert bits_equal(got, exp), ("pix2sky! in place", params)
    return kind


def fuzz_posmap(rng):
    shape, wcs = rand_geometry(rng, cls=rng.choice(["tiny", "small", "medium", "wide"]))
    if shape[0] * shape[1] > 6e6:
        return "skipped-size"
    ny = shape[1]
    r0 = int(rng.integers(0, ny))
    nr = int(rng.integers(1, ny - r0 + 1))
    safe = bool(rng.random() < 0.6)
    ra, dec = pj.posmap(shape, wcs, device=DEV, row0=r0, nrows=nr, safe=safe)
    era, edec = O.posmap(wcs, shape, row0=r0, nrows=nr, safe=safe)
    params = dict(shape=shape, wcs=wcs, rows=(r0, nr), safe=safe)
    assert bits_equal(ra.data.cpu().numpy(), era) and bits_equal(dec.data.cpu().numpy(), edec), params
    return "posmap"


def fuzz_sample(rng):
    shape, wcs = rand_geometry(rng, cls=rng.choice(["tiny", "small", "medium", "wide", "tall"]))
    nx, ny = shape
    if nx * ny > 8e6:
        return "skipped-size"
    nc = int(rng.choice([1, 2, 3]))
    f32 = rng.random() < 0.4
    if rng.random() < 0.4 and ny > 2:
        s0 = int(rng.integers(0, ny - 1))
        sn = int(rng.integers(1, ny - s0 + 1))
    else:
        s0, sn = 0, ny
    n = rand_n(rng)
    src = rng.normal(size=(nc, sn, nx))
    # points concentrated on the map's footprint, plus uniform-on-sphere and far-away periods
    ci, cj = rng.uniform(-2, nx + 3, n), rng.uniform(-2, ny + 3, n)
    fra, fdec = O.pix2sky_soa(wcs, ci, cj, safe=False)
    u = rng.random(n) < 0.3
    fra = np.where(u, rng.uniform(-math.pi, math.pi, n), fra) + 2 * math.pi * (rng.random(n) < 0.1) * rng.integers(-4, 5, n)
    fdec = np.where(u, np.arcsin(rng.uniform(-1, 1, n)), fdec)
    sky = special_values(rng, np.stack([fra, fdec], axis=1), 1.0)
    params = dict(shape=shape, wcs=wcs, nc=nc, f32=f32, src_rows=(s0, sn), n=n)
    if f32:
        src = src.astype(np.float32)
        expect = O.sample_bilinear_f32(wcs, (nx, ny, nc), src, sky, src_row0=s0, src_nrows=sn)
    else:
        expect = O.sample_bilinear(wcs, (nx, ny, nc), src, sky, src_row0=s0, src_nrows=sn)
    m = pj.Enmap(to_dev(src, np.float32 if f32 else np.float64), wcs)
    d_sky = to_dev(sky)
    got = pj.sample_bilinear(m, d_sky, src_rows=(s0, sn), full_shape=(nx, ny, nc)).cpu().numpy()
    assert bits_equal(got, expect), ("sample", params)
    pairs = pj.SamplePairs(m, src_rows=(s0, sn), full_shape=(nx, ny, nc))
    got = pj.sample_bilinear(None, d_sky, pairs=pairs).cpu().numpy()
    assert bits_equal(got, expect), ("sample via row pairs", params)
    return "sample_f32" if f32 else "sample_f64"


def fuzz_maps(rng):
    """pixareamap! on random CAR geometries (device sin vs the host's: a few ulp of sin(dec) ~ 1, scaled by the RA pixel
    width; constant along RA, every row) and the Gnomonic posmap against the oracle's evaluators (libm-level tolerance)."""
    if rng.random() < 0.6:
        shape, wcs = rand_geometry(rng, cls=rng.choice(["tiny", "small", "medium", "wide", "tall"]))
        if shape[0] * shape[1] > 6e6:
            return "skipped-size"
        pm = pj.pixareamap(shape, wcs, device=DEV)
        col = pm.data[:, 0].cpu().numpy()
        ref = O.pixarea_rows(wcs, shape[1])
        tol = 8 * np.finfo(float).eps * abs(wcs.cdelt[0] * wcs.unit)
        assert np.abs(col - ref).max() <= tol, ("pixareamap", shape, wcs, float(np.abs(col - ref).max()), tol)
        assert bool((pm.data == pm.data[:, :1]).all()), ("pixareamap not constant along RA", shape, wcs)
        return "pixareamap"
    res = 10 ** rng.uniform(-2.5, 0.0)
    nx, ny = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    if res * max(nx, ny) > 40.0:          # patches beyond ~40 degrees approach the horizon, where asin / atan2 amplify an ulp
        res = 40.0 / max(nx, ny)          # of difference between the device's and the host's libm beyond any fixed bound
    wcs = pj.Gnomonic((-res if rng.random() < 0.7 else res, res * rng.uniform(0.9, 1.1)),
                      (nx / 2 + rng.uniform(-40, 40), ny / 2 + rng.uniform(-40, 40)), (rng.uniform(-180, 180), rng.uniform(-89, 89)))
    ra, dec = pj.posmap((nx, ny), wcs, device=DEV)
    jj, ii = np.meshgrid(np.arange(1, ny + 1, dtype=float), np.arange(1, nx + 1, dtype=float), indexing="ij")
    era, edec = O.pix2sky_tan(wcs, ii.ravel(), jj.ravel())
    gra, gdec = ra.data.cpu().numpy().ravel(), dec.data.cpu().numpy().ravel()
    dra = np.abs(np.angle(np.exp(1j * (gra - era))))            # RA compared on the circle
    # both sides take asin / atan2 of direction cosines: an ulp there is 1e-16 / cos(dec) in the angle, so a pixel within arc seconds
    # of a celestial pole (found by seed 5151: a patch centred at dec 88.4 degrees, 1.3e-11) needs the bound scaled with it
    tol = 1e-11 + 4e-16 / np.maximum(np.cos(edec), 1e-12)
    assert np.nanmax(dra - tol) < 0 and np.nanmax(np.abs(gdec - edec) - tol) < 0, ("tan posmap", (nx, ny), wcs, float(np.nanmax(dra)), float(np.nanmax(np.abs(gdec - edec))))
    assert np.array_equal(np.isnan(gra), np.isnan(era))
    return "tan_posmap"


def fuzz_generic(rng):
    """CAR <-> Gnomonic reprojection (tolerance-checked: FP64 transcendentals): the tiled kernel against the oracle and
    against the per-pixel kernel, random resolutions, patch sizes and centres (seam, high declinations, patches partly
    off the source, the Gnomonic horizon), one or two components."""
    res = float(rng.choice([60.0, 30.0, 20.0, 10.0, 4.0])) / 60.0                  # degrees per pixel
    ra0 = float(rng.choice([rng.uniform(-180, 180), 179.95, -179.9, 0.0]))
    dec0 = float(rng.choice([rng.uniform(-60, 60), rng.uniform(-88, 88), 0.0]))

    def tan_geometry(scale=1.0):
        r = res * scale * rng.uniform(0.7, 1.4)
        nx, ny = int(rng.integers(20, 420)), int(rng.integers(20, 330))
        crpix = (nx / 2 + rng.uniform(-0.3, 0.3) * nx, ny / 2 + rng.uniform(-0.3, 0.3) * ny)
        sx = -1.0 if rng.random() < 0.7 else 1.0
        return (nx, ny), pj.Gnomonic((sx * r, r * rng.uniform(0.9, 1.1)), crpix, (ra0 + rng.uniform(-3, 3) * res, dec0 + rng.uniform(-3, 3) * res))

    def car_geometry():
        if res >= 10.0 / 60.0 and rng.random() < 0.6:                               # periodic full sky
            return pj.fullsky_geometry(math.radians(res))
        half = res * rng.uniform(60, 250)
        d0 = max(-89.0, min(89.0, dec0))
        box = [[math.radians(ra0 + half), math.radians(ra0 - half)], [math.radians(max(-89.5, d0 - half * 0.7)), math.radians(min(89.5, d0 + half * 0.7))]]
        return pj.geometry(box, math.radians(res))

    direction = rng.choice(["car->tan", "car->tan", "car->tan", "tan->car", "tan->tan"])
    if direction == "car->tan":
        (s_in, w_in), p_in = car_geometry(), 0
        (s_out, w_out), p_out = tan_geometry(), 1
    elif direction == "tan->car":
        (s_in, w_in), p_in = tan_geometry(), 1
        (s_out, w_out), p_out = car_geometry(), 0
        if s_out[0] * s_out[1] > 400000:
            return "skipped-size"
    else:
        (s_in, w_in), p_in = tan_geometry(), 1
        (s_out, w_out), p_out = tan_geometry(rng.uniform(0.5, 2.0)), 1
    s_in = tuple(int(v) for v in s_in[:2]); s_out = tuple(int(v) for v in s_out[:2])
    if s_in[0] * s_in[1] > 3000000 or s_out[0] * s_out[1] > 400000 or min(s_in + s_out) < 2:
        return "skipped-size"
    nc = int(rng.choice([1, 1, 2]))
    yy, xx = np.meshgrid(np.arange(s_in[1]), np.arange(s_in[0]), indexing="ij")
    src = np.stack([np.sin(0.011 * (c + 1) * xx + 0.3) * np.cos(0.013 * yy) + 0.001 * c * xx for c in range(nc)])
    params = dict(direction=direction, s_in=s_in, w_in=w_in, s_out=s_out, w_out=w_out, nc=nc)
    exp = O.reproject_generic(w_in, p_in, (s_in[0], s_in[1], nc), src, w_out, p_out, s_out)
    m = pj.Enmap(to_dev(src if nc > 1 else src[0]), w_in)
    tiled = pj.reproject(m, s_out, w_out).data.cpu().numpy().reshape(nc, s_out[1], s_out[0])
    os.environ["PXL_GENERIC_EXACT"] = "1"
    try:
        exact = pj.reproject(m, s_out, w_out).data.cpu().numpy().reshape(nc, s_out[1], s_out[0])
    finally:
        del os.environ["PXL_GENERIC_EXACT"]
    assert np.array_equal(np.isnan(exact), np.isnan(exp)) and np.array_equal(np.isnan(tiled), np.isnan(exp)), ("generic: NaN pattern", params)
    e1 = float(np.nanmax(np.abs(exact - exp))) if exp.size else 0.0
    e2 = float(np.nanmax(np.abs(tiled - exp))) if exp.size else 0.0
    e3 = float(np.nanmax(np.abs(tiled - exact))) if exp.size else 0.0
    assert e1 < 1e-9 and e2 < 1e-9 and e3 < 2e-10, ("generic: tolerance", (e1, e2, e3), params)
    return direction


KINDS = {"reproject": (fuzz_reproject, 0.5), "elementwise": (fuzz_elementwise, 0.15), "unwind": (fuzz_unwind, 0.1),
         "posmap": (fuzz_posmap, 0.05), "sample": (fuzz_sample, 0.15), "generic": (fuzz_generic, 0.05), "maps": (fuzz_maps, 0.03)}


def main():
    global DEV
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default=None, choices=list(KINDS))
    ap.add_argument("--max-cases", type=int, default=10**9)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "the fuzzer drives the HIP library: it needs the GPU"
    DEV = torch.device("cuda:0")
    pj.load_library()
    O.set_threads(min(O.max_threads(), 16))
    names = list(KINDS) if args.only is None else [args.only]
    w = np.array([KINDS[k][1] for k in names])
    w /= w.sum()
    master = np.random.default_rng(args.seed)
    counts, fails = {}, 0
    t0 = last = time.time()
    case = 0
    log = open(args.log, "a") if args.log else None
    while time.time() - t0 < args.seconds and case < args.max_cases:
        kind = names[int(master.choice(len(names), p=w))]
        cseed = int(master.integers(0, 2**62))
        rng = np.random.default_rng(cseed)
        try:
            tag = KINDS[kind][0](rng)
            counts[kind + ":" + str(tag)] = counts.get(kind + ":" + str(tag), 0) + 1
        except AssertionError as e:
            fails += 1
            msg = "FAIL kind=%s case_seed=%d: %s" % (kind, cseed, e)
            print(msg, flush=True)
            if log:
                log.write(msg + "\n")
                log.flush()
            if fails >= 10:
                break
        case += 1
        if time.time() - last > 30:
            last = time.time()
            line = "[%5.0fs] %d cases, %d failures" % (last - t0, case, fails)
            print(line, flush=True)
            if log:
                log.write(line + "\n")
                log.flush()
    summary = "fuzz_parity seed=%d: %d cases in %.0f s, %d failures; %s" % (
        args.seed, case, time.time() - t0, fails, ", ".join("%s=%d" % kv for kv in sorted(counts.items())))
    print(summary, flush=True)
    if log:
        log.write(summary + "\n")
        log.close()
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
