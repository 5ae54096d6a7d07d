"""Pin the CPU oracle against every known-answer literal the reference's own tests hold for the hot path
(transcribed in tests/golden/reference_literals.json with file:line provenance), the reference's
pixel-area data files, and wcslib vectors mirroring test_geometry.jl:66-80.  CPU only."""
import math
import os

import numpy as np
import pytest

from conftest import ARCMIN, DEG, GOLDEN, bits_equal, isapprox, unhex


def _box(lit):
    b = lit["box_deg"]
    return (b[0][0] * DEG, b[1][0] * DEG), (b[0][1] * DEG, b[1][1] * DEG)


def test_fullsky_geometry_literals(O, literals):
    for lit in literals["fullsky_geometry"]:
        res = eval(lit["res_expr"]) * DEG                      # deg2rad(1/60) etc.
        shape, w = O.fullsky_geometry(res)
        assert list(shape) == lit["shape"], lit["src"]
        for k in ("cdelt", "crpix", "crval"):
            if k in lit:
                assert isapprox(list(getattr(w, k)), lit[k]), (lit["src"], k)


def test_fullsky_geometry_asserts(O):
    with pytest.raises(AssertionError):
        O.fullsky_geometry(0.0123)                              # does not divide the sky


def test_geometry_literals(O, literals):
    for lit in literals["geometry"]:
        p1, p2 = _box(lit)
        shape, w = O.geometry(p1, p2, lit["res_arcmin"] * ARCMIN)
        assert list(shape) == lit["shape"], lit["src"]
        if "cdelt" in lit:
            assert isapprox(list(w.cdelt), lit["cdelt"])
        if lit.get("crpix_exact"):
            assert list(w.crpix) == lit["crpix"]
        else:
            assert isapprox(list(w.crpix), lit["crpix"])
        if lit.get("crval_exact"):
            assert list(w.crval) == lit["crval"]
        else:
            assert isapprox(list(w.crval), lit["crval"])


def test_pix2sky_literals(O, literals):
    shape, w = O.fullsky_geometry(1 * DEG)
    for lit in literals["pix2sky_1deg"]:
        # the 2-vector method: scalar path (rewind) then a no-op unwind (car_proj.jl:155-162)
        ra, dec = O.pix2sky_soa(w, [lit["pix"][0]], [lit["pix"][1]], safe=True)
        assert isapprox([ra[0], dec[0]], lit["sky"]), lit["src"]
        # the 2xN array method with safe=true on one point gives the same answer
        sky = O.pix2sky(w, [lit["pix"]], O.WRAP_UNWIND)
        assert isapprox(sky[0], lit["sky"]), lit["src"]
    lit = literals["pix2sky_scalar_1deg_docstring"]
    ra, dec = O.pix2sky_soa(w, [lit["pix"][0]], [lit["pix"][1]], safe=True)
    assert isapprox([ra[0] / DEG, dec[0] / DEG], lit["sky_deg"])


def test_roundtrips_and_2pik_invariance(O, literals):
    shape, w = O.fullsky_geometry(1 * DEG)
    for lit in literals["roundtrip_1deg"]:
        ra, dec = O.pix2sky_soa(w, [lit["pix"][0]], [lit["pix"][1]], safe=True)
        add = [k * math.pi for k in lit.get("add_pi", [0, 0])]
        x, y = O.sky2pix_soa(w, shape, [ra[0] + add[0]], [dec[0] + add[1]], safe=True, form=O.FORM_DIV)
        assert isapprox([x[0], y[0]], lit["pix"]), lit["src"]
        # vector form (test_geometry.jl:59-60) and 2xN form
        x, y = O.sky2pix_soa(w, shape, [ra[0] + add[0]], [dec[0] + add[1]], safe=True, form=O.FORM_RECIP_AV)
        assert isapprox([x[0], y[0]], lit["pix"]), lit["src"]
        p = O.sky2pix(w, shape, [[ra[0] + add[0], dec[0] + add[1]]], safe=True)
        assert isapprox(p[0], lit["pix"]), lit["src"]


def test_sky2pix_literals(O, literals):
    shape, w = O.fullsky_geometry(1 * DEG)
    for lit in literals["sky2pix_1deg"]:
        x, y = O.sky2pix_soa(w, shape, [lit["sky_deg"][0] * DEG], [lit["sky_deg"][1] * DEG], form=O.FORM_DIV)
        assert isapprox([x[0], y[0]], lit["pix"]), lit["src"]


def test_wrap_vector(O, literals):
    lit = literals["wrap_box_1deg"]
    p1, p2 = _box(lit)
    shape, w = O.geometry(p1, p2, lit["res_deg"] * DEG)
    got = [O.sky2pix_soa(w, shape, [ra * DEG], [0.0], safe=True, form=O.FORM_DIV)[0][0] for ra in lit["ra_deg"]]
    assert isapprox(got, lit["pix_ra"])


def _norm_last(first, step, stop):
    return first + step * ((stop - first) // step)


def test_slice_geometry_literals(O, literals):
    shape0, w0 = O.fullsky_geometry(1 * DEG)
    for lit in literals["slice_geometry_1deg"]:
        (fx, sx, ex), (fy, sy, ey) = lit["sel"]
        shape, w = O.slice_geometry(w0, (fx, fy), (sx, sy), (_norm_last(fx, sx, ex), _norm_last(fy, sy, ey)))
        assert list(shape) == lit["shape"], lit["src"]
        assert isapprox(list(w.cdelt), lit["cdelt"]), lit["src"]
        assert isapprox(list(w.crpix), lit["crpix"]), lit["src"]
        assert isapprox(list(w.crval), lit["crval"]), lit["src"]


def test_skyarea_literals(O, literals):
    shape0, w0 = O.fullsky_geometry(1 * DEG)
    for lit in literals["skyarea"]:
        if lit["kind"] == "fullsky_1deg":
            assert math.isclose(O.skyarea_cyl(w0, shape0), lit["area_over_pi"] * math.pi, rel_tol=1.5e-8)
        elif lit["kind"] == "fullsky_1deg_slice":
            (fx, sx, ex), (fy, sy, ey) = lit["sel"]
            shape, w = O.slice_geometry(w0, (fx, fy), (sx, sy), (_norm_last(fx, sx, ex), _norm_last(fy, sy, ey)))
            assert math.isclose(O.skyarea_cyl(w, shape), lit["area"], rel_tol=1.5e-8)
        else:
            p1, p2 = _box(lit)
            shape, w = O.geometry(p1, p2, lit["res_arcmin"] * ARCMIN)
            assert math.isclose(O.skyarea_cyl(w, shape), lit["area"], rel_tol=1.5e-8)


def test_pixareamap_reference_data(O, literals):
    """test_geometry.jl:287-316: sum(abs(pm[1,:] - python-pixell column)) < 100eps()."""
    for lit in literals["pixareamap"]:
        if lit["kind"] == "fullsky_1deg":
            shape, w = O.fullsky_geometry(1 * DEG)
        else:
            p1, p2 = _box(lit)
            shape, w = O.geometry(p1, p2, lit["res_arcmin"] * ARCMIN)
        ref = np.loadtxt(os.path.join(GOLDEN, lit["file"]))
        assert ref.size == shape[1]
        got = O.pixarea_rows(w, shape[1])
        assert np.abs(got - ref).sum() < lit["tol_sum_abs"], lit["src"]


def test_gnomonic_literals(O, literals):
    g = literals["gnomonic"]

    class W:
        cdelt, crpix, crval, unit = g["cdelt"], g["crpix"], g["crval"], DEG
    for lit in g["pix2sky"]:
        ra, dec = O.pix2sky_tan(W, [lit["pix"][0]], [lit["pix"][1]])
        assert isapprox([ra[0], dec[0]], lit["sky"])
    a, d = g["roundtrip_sky"]
    x, y = O.sky2pix_tan(W, [a], [d])
    ra, dec = O.pix2sky_tan(W, x, y)
    assert abs(ra[0] - a) < 1e-12 and abs(dec[0] - d) < 1e-12


def test_wcslib_vectors(O, wcslib_vectors):
    """The reference cross-checks its CAR fast path against wcslib with `≈` (test_geometry.jl:66-80);
    so do we, on vectors produced by wcslib 7.6 (tests/golden/gen_wcslib_vectors.py)."""
    for case in wcslib_vectors["cases"]:
        g = case["geom"]

        class W:
            cdelt, crpix, crval, unit = g["cdelt"], g["crpix"], g["crval"], DEG
        # (a) the reference's own draw, pi .* rand(2, 1024): direct `≈`, safe=false
        sky = O.pix2sky(W, unhex(case["pix_small"]), O.WRAP_NONE)
        assert isapprox(sky, unhex(case["pix2world_small_deg"]) * DEG), g["name"]
        got = O.sky2pix(W, g["shape"], unhex(case["world_small_deg"]) * DEG, safe=False)
        assert isapprox(got, unhex(case["world2pix_small"])), g["name"]
        # (b) points all over the map: wcslib reports RA in [0, 360), so compare RA modulo 2pi and
        #     pixels after the reference's own safe=true rewind
        sky = O.pix2sky(W, unhex(case["pix"]), O.WRAP_NONE)
        ref = unhex(case["pix2world_deg"]) * DEG
        dra = np.abs((sky[:, 0] - ref[:, 0] + math.pi) % (2 * math.pi) - math.pi)
        assert np.max(dra) < 1e-12 and np.max(np.abs(sky[:, 1] - ref[:, 1])) < 1e-12
        got = O.sky2pix(W, g["shape"], unhex(case["world_deg"]) * DEG, safe=True)
        assert np.max(np.abs(got - unhex(case["world2pix"]))) < 1e-7      # pixel units; wcslib works in degrees


def test_every_pixel_range_invariants(O):
    """test_geometry.jl:207-223 on the 1-degree full-sky map, and config 1 of BASELINE.json
    (1024x513 full-sky round trip, max error <= 1e-9 pix)."""
    for res_deg, tol in ((1.0, 1e-9), (360.0 / 1024, 1e-9)):
        shape, w = O.fullsky_geometry(res_deg * DEG)
        nx, ny = shape
        ii, jj = np.meshgrid(np.arange(1, nx + 1, dtype=float), np.arange(1, ny + 1, dtype=float))
        ii, jj = ii.ravel(), jj.ravel()
        ra, dec = O.pix2sky_soa(w, ii, jj, safe=True)
        assert np.all((-math.pi <= ra) & (ra <= math.pi))
        assert np.all((-math.pi / 2 <= dec) & (dec <= math.pi / 2))
        x, y = O.sky2pix_soa(w, shape, ra, dec, safe=True, form=O.FORM_DIV)
        assert np.all((1 <= x) & (x <= nx)) and np.all((1 <= y) & (y <= ny))
        ra_u, dec_u = O.pix2sky_soa(w, ii, jj, safe=False)
        x, y = O.sky2pix_soa(w, shape, ra_u, dec_u, safe=True, form=O.FORM_DIV)
        assert np.all((1 <= x) & (x <= nx)) and np.all((1 <= y) & (y <= ny))
        assert np.max(np.abs(x - ii)) <= tol and np.max(np.abs(y - jj)) <= tol


def test_safe_equals_unsafe_on_small_box(O):
    """test_geometry.jl:164-181"""
    shape, w = O.geometry((10 * DEG, -5 * DEG), (-10 * DEG, 5 * DEG), 1 * DEG)
    ii, jj = np.meshgrid(np.arange(1, shape[0] + 1, dtype=float), np.arange(1, shape[1] + 1, dtype=float))
    ii, jj = ii.ravel(), jj.ravel()
    ra, dec = O.pix2sky_soa(w, ii, jj, safe=True)
    ra_u, dec_u = O.pix2sky_soa(w, ii, jj, safe=False)
    assert isapprox(ra, ra_u) and isapprox(dec, dec_u)
    x, y = O.sky2pix_soa(w, shape, ra, dec, safe=True, form=O.FORM_DIV)
    xu, yu = O.sky2pix_soa(w, shape, ra, dec, safe=False, form=O.FORM_DIV)
    assert isapprox(x, xu) and isapprox(y, yu)


def test_jl_mod_semantics(O):
    """Julia mod: result has the sign of the divisor; exact zero keeps the divisor's sign."""
    assert O.jl_mod(5.0, 3.0) == 2.0
    assert O.jl_mod(-1.0, 3.0) == 2.0
    assert O.jl_mod(1.0, -3.0) == -2.0
    assert math.copysign(1.0, O.jl_mod(6.0, -3.0)) == -1.0 and O.jl_mod(6.0, -3.0) == 0.0
    assert O.jl_mod(-1e-20, 3.0) == 3.0            # tiny negative remainder rounds up to the period itself
    assert O.rewind(3 * math.pi) == pytest.approx(-math.pi)
    assert O.rewind(-math.pi) == -math.pi


def test_unwind_definition(O):
    """unwind! = rewind + DSP.unwrap (PARITY UNPINNED: DSP.jl is not in the reference tree).  Checks the
    defining properties: differences between neighbours end up <= period/2, values stay congruent."""
    rng = np.random.default_rng(3)
    a = np.cumsum(rng.normal(0, 1.0, 500)) + 40.0
    u = O.unwind_row(a)
    assert np.all(np.abs(np.diff(u)) <= math.pi + 1e-12)
    k = (u - a) / (2 * math.pi)
    assert np.max(np.abs(k - np.round(k))) < 1e-9
    assert abs(u[0]) <= math.pi                     # first element is just rewound


def test_bilinear_reproduces_affine_field(O):
    """Bilinear interpolation is exact for m = a + b*i + c*j away from the seam (SURVEY 8(c) fixtures)."""
    shape, w = O.fullsky_geometry(2 * math.pi / 64)
    nx, ny = shape
    jj, ii = np.meshgrid(np.arange(1, ny + 1, dtype=float), np.arange(1, nx + 1, dtype=float), indexing="ij")
    src = (0.25 + 3.0 * ii - 2.0 * jj)[None]
    rng = np.random.default_rng(5)
    x = 2 + (nx - 3) * rng.random(2000)
    y = 1 + (ny - 1) * rng.random(2000)
    ra, dec = O.pix2sky_soa(w, x, y, safe=False)
    out = O.sample_bilinear(w, (nx, ny, 1), src, np.stack([ra, dec], 1))
    px = O.sky2pix(w, shape, np.stack([ra, dec], 1), safe=True)
    expect = 0.25 + 3.0 * px[:, 0] - 2.0 * px[:, 1]
    assert np.max(np.abs(out[0] - expect)) < 1e-11


def test_reproject_identity_and_seam(O):
    """Reprojecting onto the same geometry returns the map; a half-pixel RA shift averages neighbours
    across the RA seam (pixel nx <-> pixel 1)."""
    shape, w = O.fullsky_geometry(2 * math.pi / 32)
    nx, ny = shape
    rng = np.random.default_rng(7)
    src = rng.normal(size=(2, ny, nx))
    same = O.reproject(w, (nx, ny, 2), src, w, shape)
    assert np.max(np.abs(same - src)) < 1e-12

    class Shift:
        cdelt, crval, unit = tuple(w.cdelt), tuple(w.crval), w.unit
        crpix = (w.crpix[0] - 0.5, w.crpix[1])      # output pixel i sits at source x = i + 0.5
    out = O.reproject(w, (nx, ny, 2), src, Shift, shape)
    expect = 0.5 * (src + np.roll(src, -1, axis=2))
    assert np.max(np.abs(out - expect)) < 1e-12


def test_reproject_windows_match_full(O):
    """A dec-strip window (+ the rows it needs) reproduces the same bits as the full-map call."""
    shape, w = O.fullsky_geometry(2 * math.pi / 48)
    nx, ny = shape
    shape_o, w_o = O.fullsky_geometry(2 * math.pi / 96)
    rng = np.random.default_rng(11)
    src = rng.normal(size=(1, ny, nx))
    full = O.reproject(w, (nx, ny, 1), src, w_o, shape_o)
    for lo, hi in ((0, 13), (13, 30), (30, shape_o[1])):
        s_lo, s_hi = O.reproject_src_rows(w, shape, w_o, shape_o, lo, hi - lo)
        part = O.reproject(w, (nx, ny, 1), src[:, s_lo:s_hi], w_o, shape_o, src_row0=s_lo, src_nrows=s_hi - s_lo,
                           dst_row0=lo, dst_nrows=hi - lo)
        assert bits_equal(part, full[:, lo:hi])


def test_bilinear_agrees_with_scipy_map_coordinates(O):
    """R1 is absent from the reference (parity unpinned), so its definition is cross-checked against an
    independent implementation of the convention Pixell.jl says it mirrors (python-pixell order-1
    interpolation == scipy.ndimage.map_coordinates(order=1), RA wrapping on a full-sky map)."""
    from scipy.ndimage import map_coordinates
    shape, w = O.fullsky_geometry(2 * math.pi / 96)
    nx, ny = shape
    rng = np.random.default_rng(21)
    src = rng.normal(size=(ny, nx))
    n = 5000
    ra = rng.uniform(-math.pi, math.pi, n)
    dec = rng.uniform(-math.pi / 2, math.pi / 2, n) * 0.999
    sky = np.stack([ra, dec], 1)
    got = O.sample_bilinear(w, (nx, ny, 1), src[None], sky)[0]
    pix = O.sky2pix(w, shape, sky, safe=True)                   # 1-based (x, y)
    ref = map_coordinates(src, [pix[:, 1] - 1, pix[:, 0] - 1], order=1, mode="grid-wrap")
    assert np.max(np.abs(got - ref)) < 1e-12
    # and the regular-grid reprojection is the same sampler evaluated at the output pixel centres
    shape_o, w_o = O.fullsky_geometry(2 * math.pi / 160)
    out = O.reproject(w, (nx, ny, 1), src[None], w_o, shape_o)[0]
    xs, ys = O.reproject_tables(w, shape, w_o, shape_o)
    yy, xx = np.meshgrid(ys - 1, xs - 1, indexing="ij")
    inner = slice(1, shape_o[1] - 1)                            # keep away from the poles' zero rows
    ref = map_coordinates(src, [yy[inner].ravel(), xx[inner].ravel()], order=1, mode="grid-wrap")
    assert np.max(np.abs(out[inner].ravel() - ref)) < 1e-12


def test_unwind_agrees_with_numpy_unwrap(O):
    """unwind!'s arithmetic lives in DSP.jl (not in the reference tree; parity unpinned).  numpy.unwrap is an
    independent implementation of the same published algorithm: same wrap counts, values equal to rounding."""
    rng = np.random.default_rng(22)
    a = np.cumsum(rng.normal(0, 1.2, 20000)) + 7.0
    ours = O.unwind_row(a)
    wound = np.array([O.rewind(v) for v in a])
    ref = np.unwrap(wound)
    assert np.max(np.abs(ours - ref)) < 1e-9
    assert np.array_equal(np.round((ours - wound) / (2 * math.pi)), np.round((ref - wound) / (2 * math.pi)))


def _golden():
    import json
    with open(os.path.join(GOLDEN, "oracle_golden.json")) as f:
        return json.load(f)


def test_oracle_reproduces_its_frozen_outputs(O):
    """Guards the (unpinned) definition of R1 and the evaluators against silent drift of the oracle itself."""
    g = _golden()
    shape, w = O.fullsky_geometry(2 * math.pi / 64)
    nx, ny = shape
    assert list(shape) == g["geometry"]["shape"] and list(w.crpix) == g["geometry"]["crpix"]
    jj, ii = np.meshgrid(np.arange(1, ny + 1, dtype=float), np.arange(1, nx + 1, dtype=float), indexing="ij")
    src = (((jj - 1) * nx + ii) ** 2)[None]
    shape2, w2 = O.fullsky_geometry(2 * math.pi / 128)

    class Shift:
        cdelt, crval, unit = tuple(w.cdelt), tuple(w.crval), w.unit
        crpix = (w.crpix[0] + 0.5, w.crpix[1] + 0.5)
    assert bits_equal(O.reproject(w, (nx, ny, 1), src, w2, shape2).ravel(), unhex(g["refined"], 1).ravel())
    assert bits_equal(O.reproject(w, (nx, ny, 1), src, Shift, shape).ravel(), unhex(g["shifted"], 1).ravel())
    pix = unhex(g["pix"])
    assert bits_equal(O.pix2sky(w, pix, O.WRAP_NONE), unhex(g["pix2sky_unsafe"]))
    assert bits_equal(O.pix2sky(w, pix, O.WRAP_REWIND), unhex(g["pix2sky_rewind"]))
    assert bits_equal(O.pix2sky(w, pix, O.WRAP_UNWIND), unhex(g["pix2sky_unwind"]))
    sky = unhex(g["pix2sky_unsafe"])
    assert bits_equal(O.sky2pix(w, shape, sky, safe=True, form=O.FORM_RECIP), unhex(g["sky2pix_safe_recip"]))


def test_gnomonic_against_wcslib_tan(O, wcslib_vectors):
    """test_geometry.jl:92-119 pins the Gnomonic evaluators against wcslib's TAN; so do we (wcslib 7.6 vectors)."""
    t = wcslib_vectors["tan"]
    g = t["geom"]

    class W:
        cdelt, crpix, crval, unit = g["cdelt"], g["crpix"], g["crval"], DEG
    pix = unhex(t["pix"])
    ref = unhex(t["pix2world_deg"]) * DEG
    ra, dec = O.pix2sky_tan(W, pix[:, 0], pix[:, 1])
    dra = np.abs((ra - ref[:, 0] + math.pi) % (2 * math.pi) - math.pi)          # wcslib reports RA in [0, 360)
    assert np.max(dra) < 1e-11 and np.max(np.abs(dec - ref[:, 1])) < 1e-11
    x, y = O.sky2pix_tan(W, ref[:, 0], ref[:, 1])
    assert np.max(np.abs(x - pix[:, 0])) < 1e-7 and np.max(np.abs(y - pix[:, 1])) < 1e-7


def test_reproject_against_wcslib_plus_scipy(O, wcslib_vectors):
    """End-to-end independent check of R1: wcslib 7.6 for both coordinate steps and scipy's order-1
    map_coordinates for the gather (python-pixell's pipeline), frozen in the golden file, against the oracle's
    reprojection of the same map.  Differences are rounding of the two coordinate chains (~1e-13 pixel)."""
    t = wcslib_vectors["reproject_wcslib_scipy"]
    gi, go = t["geom_in"], t["geom_out"]

    class Win:
        cdelt, crpix, crval, unit = gi["cdelt"], gi["crpix"], gi["crval"], DEG

    class Wout:
        cdelt, crpix, crval, unit = go["cdelt"], go["crpix"], go["crval"], DEG
    nx, ny = gi["shape"]
    src = unhex(t["src"], 1).reshape(ny, nx)
    out = O.reproject(Win, (nx, ny, 1), src[None], Wout, go["shape"])[0]
    r0, r1 = t["rows"]                                     # 1-based rows covered by the reference (poles skipped)
    ref = unhex(t["expected"], 1).reshape(r1 - r0 + 1, go["shape"][0])
    assert np.max(np.abs(out[r0 - 1:r1] - ref)) < 1e-11


def test_rad2deg_follows_julia_base_multiply_form(O, pj):
    """Base.rad2deg(z::AbstractFloat) = z * (180 / oftype(z, pi)) (julia base/math.jl) -- ONE multiply by the Float64
    constant 180/pi.  The round-1 restatement z / (pi/180) differs by 1 ulp on ~11 % of inputs; geometry()
    (enmap_geom.jl:100-102) must use the published form.  Inputs below are resolutions 2pi/n where the two forms
    differ, so a regression to the division form fails here.  The constant is checked against its hex literal."""
    k = 180 / math.pi
    assert k.hex() == "0x1.ca5dc1a63c1f8p+5"
    seen = 0
    for n in (142, 198, 324, 597, 765, 1031, 1500):
        res = 2 * math.pi / n
        assert res * k != res / (math.pi / 180), "input no longer discriminates the two forms"
        # a box whose RA midpoint is also a discriminating value
        p1, p2 = (res * 7.0, -0.25), (res * 7.0 - res * 40, 0.25)
        for geom in (lambda: O.geometry(p1, p2, res), lambda: pj.geometry([[p1[0], p2[0]], [p1[1], p2[1]]], res)):
            shape, w = geom()
            assert w.cdelt[0] == -res * k and w.cdelt[1] == res * k
            mid = (p1[0] + p2[0]) / 2
            assert w.crval[0] == mid * k and w.crval[1] == 0.0
            seen += 1
    assert seen == 14
