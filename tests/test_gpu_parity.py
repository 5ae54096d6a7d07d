"""GPU parity: every device entry of libpixell_hip.so (called through the C ABI via the host package)
against the CPU oracle on the same seeded inputs.  Bar: bit-exact for pix<->sky coordinate arithmetic;
interpolated values within 1e-10 absolute (BASELINE.json north_star) -- and, because the kernels restate
the oracle operation for operation, we additionally require them to be bit-identical."""
import math

import numpy as np
import pytest

from conftest import ARCMIN, DEG, bits_equal

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL_INTERP = 1e-10       # Float64 agreement bound on interpolated values (north_star)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import pixell_jl_amd as pj
    pj.load_library()                      # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)


def geoms(pj):
    g = {}
    g["fullsky_1deg"] = pj.fullsky_geometry(1 * DEG)
    g["fullsky_odd"] = ((45, 23), pj.CarClenshawCurtis((-8.0, 8.181818181818182), (22.5, 12.0), (4.0, 0.0)))
    g["box"] = pj.geometry([[10 * DEG, -10 * DEG], [-5 * DEG, 5 * DEG]], 0.5 * ARCMIN)
    g["box_flipped"] = pj.geometry([[-10 * DEG, 10 * DEG], [5 * DEG, -5 * DEG]], 1 * ARCMIN)
    g["fejer1"] = ((360, 180), pj.CarFejer1((-1.0, 1.0), (180.5, 90.5), (0.5, 0.0)))
    return g


# ---- elementwise evaluators -----------------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 63, 1000, 70001])
def test_pix2sky_2xN_bit_exact(pj, O, dev, n):
    rng = np.random.default_rng(100 + n)
    for name, (shape, wcs) in geoms(pj).items():
        pix = np.stack([rng.uniform(-50, shape[0] + 50, n), rng.uniform(-50, shape[1] + 50, n)], axis=1)
        pix_d = to_dev(pix.reshape(n, 2), dev)
        got = pj.pix2sky((shape, wcs), pix_d, safe=False).cpu().numpy()
        assert bits_equal(got, O.pix2sky(wcs, pix.reshape(n, 2), O.WRAP_NONE)), name
        got = pj.pix2sky_rewind((shape, wcs), pix_d).cpu().numpy()
        assert bits_equal(got, O.pix2sky(wcs, pix.reshape(n, 2), O.WRAP_REWIND)), name
        # safe=true on arrays = rewind + unwrap along the point axis
        got = pj.pix2sky((shape, wcs), pix_d, safe=True).cpu().numpy()
        assert bits_equal(got, O.pix2sky(wcs, pix.reshape(n, 2), O.WRAP_UNWIND)), name


def test_pix2sky_inplace(pj, O, dev):
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    rng = np.random.default_rng(1)
    pix = rng.uniform(0, 400, (5000, 2))
    buf = to_dev(pix, dev)
    out = pj.pix2sky_((shape, wcs), buf, buf, safe=False)
    assert out.data_ptr() == buf.data_ptr()
    assert bits_equal(buf.cpu().numpy(), O.pix2sky(wcs, pix, O.WRAP_NONE))


def test_pix2sky_soa_bit_exact(pj, O, dev):
    rng = np.random.default_rng(2)
    for name, (shape, wcs) in geoms(pj).items():
        ip = rng.uniform(-3 * shape[0], 3 * shape[0], 33333)
        jp = rng.uniform(-3 * shape[1], 3 * shape[1], 33333)
        for safe in (True, False):
            ra, dec = pj.pix2sky((shape, wcs), to_dev(ip, dev), to_dev(jp, dev), safe=safe)
            era, edec = O.pix2sky_soa(wcs, ip, jp, safe=safe)
            assert bits_equal(ra.cpu().numpy(), era) and bits_equal(dec.cpu().numpy(), edec), (name, safe)


@pytest.mark.parametrize("safe", [True, False])
def test_sky2pix_all_forms_bit_exact(pj, O, dev, safe):
    rng = np.random.default_rng(3)
    n = 50001
    for name, (shape, wcs) in geoms(pj).items():
        # on-sky, off-sky and "crazy" angles many periods away (test_geometry.jl:63-64)
        ra = np.concatenate([rng.uniform(-math.pi, math.pi, n), rng.uniform(-40 * math.pi, 40 * math.pi, n)])
        dec = np.concatenate([rng.uniform(-math.pi / 2, math.pi / 2, n), rng.uniform(-17 * math.pi, 17 * math.pi, n)])
        sky = np.stack([ra, dec], axis=1)
        got = pj.sky2pix((shape, wcs), to_dev(sky, dev), safe=safe).cpu().numpy()            # A11
        assert bits_equal(got, O.sky2pix(wcs, shape, sky, safe=safe, form=O.FORM_RECIP)), name
        x, y = pj.sky2pix((shape, wcs), to_dev(ra, dev), to_dev(dec, dev), safe=safe)         # A13
        ex, ey = O.sky2pix_soa(wcs, shape, ra, dec, safe=safe, form=O.FORM_RECIP_AV)
        assert bits_equal(x.cpu().numpy(), ex) and bits_equal(y.cpu().numpy(), ey), name
        x, y = pj.sky2pix_broadcast((shape, wcs), to_dev(ra, dev), to_dev(dec, dev), safe=safe)   # A12
        ex, ey = O.sky2pix_soa(wcs, shape, ra, dec, safe=safe, form=O.FORM_DIV)
        assert bits_equal(x.cpu().numpy(), ex) and bits_equal(y.cpu().numpy(), ey), name


def test_device_matches_host_scalar_methods(pj, dev):
    """The host scalar methods (which stay on the CPU, as in the reference) and the device broadcast form
    are the same arithmetic."""
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    rng = np.random.default_rng(4)
    ra = rng.uniform(-10, 10, 257)
    dec = rng.uniform(-3, 3, 257)
    x, y = pj.sky2pix_broadcast((shape, wcs), to_dev(ra, dev), to_dev(dec, dev), safe=True)
    host = np.array([pj.sky2pix((shape, wcs), float(a), float(d)) for a, d in zip(ra, dec)])
    assert bits_equal(x.cpu().numpy(), host[:, 0]) and bits_equal(y.cpu().numpy(), host[:, 1])
    a, d = pj.pix2sky((shape, wcs), to_dev(ra * 20, dev), to_dev(dec * 20, dev), safe=True)
    host = np.array([pj.pix2sky((shape, wcs), float(i), float(j)) for i, j in zip(ra * 20, dec * 20)])
    assert bits_equal(a.cpu().numpy(), host[:, 0]) and bits_equal(d.cpu().numpy(), host[:, 1])


def test_reference_literals_on_device(pj, dev, literals):
    """The reference's known answers through the device path (test_geometry.jl:52-64,82-87)."""
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    pix = np.array([lit["pix"] for lit in literals["pix2sky_1deg"]])
    sky = pj.pix2sky_rewind((shape, wcs), to_dev(pix, dev)).cpu().numpy()
    for row, lit in zip(sky, literals["pix2sky_1deg"]):
        assert np.allclose(row, lit["sky"], rtol=1.5e-8, atol=0), lit["src"]
    lit = literals["wrap_box_1deg"]
    b = lit["box_deg"]
    shape, wcs = pj.geometry([[b[0][0] * DEG, b[0][1] * DEG], [b[1][0] * DEG, b[1][1] * DEG]], lit["res_deg"] * DEG)
    ra = np.array(lit["ra_deg"], dtype=float) * DEG
    x, _ = pj.sky2pix_broadcast((shape, wcs), to_dev(ra, dev), to_dev(np.zeros_like(ra), dev), safe=True)
    assert np.allclose(x.cpu().numpy(), lit["pix_ra"], rtol=1.5e-8)


def test_wcslib_vectors_on_device(pj, dev, wcslib_vectors):
    from conftest import unhex, isapprox
    for case in wcslib_vectors["cases"]:
        g = case["geom"]
        wcs = pj.CarClenshawCurtis(g["cdelt"], g["crpix"], g["crval"])
        sky = pj.pix2sky((g["shape"], wcs), to_dev(unhex(case["pix_small"]), dev), safe=False).cpu().numpy()
        assert isapprox(sky, unhex(case["pix2world_small_deg"]) * DEG)
        pix = pj.sky2pix((g["shape"], wcs), to_dev(unhex(case["world_small_deg"]) * DEG, dev), safe=False).cpu().numpy()
        assert isapprox(pix, unhex(case["world2pix_small"]))


# ---- whole-map writers ----------------------------------------------------------------------------

def test_posmap_bit_exact(pj, O, dev):
    for name, (shape, wcs) in geoms(pj).items():
        if shape[0] * shape[1] > 4_000_000:
            continue
        ra, dec = pj.posmap(shape, wcs, device=dev)
        era, edec = O.posmap(wcs, shape)
        assert ra.shape == shape and bits_equal(ra.data.cpu().numpy(), era), name
        assert bits_equal(dec.data.cpu().numpy(), edec), name
        # safe=false and a strip
        ra, dec = pj.posmap(shape, wcs, device=dev, row0=3, nrows=7, safe=False)
        era, edec = O.posmap(wcs, shape, row0=3, nrows=7, safe=False)
        assert bits_equal(ra.data.cpu().numpy(), era) and bits_equal(dec.data.cpu().numpy(), edec), name


def test_posmap_range_invariants_fullsky(pj, dev):
    """test_geometry.jl:207-223 on device."""
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    ra, dec = pj.posmap(shape, wcs, device=dev)
    assert float(ra.data.min()) >= -math.pi and float(ra.data.max()) <= math.pi
    assert float(dec.data.min()) >= -math.pi / 2 and float(dec.data.max()) <= math.pi / 2
    sky = torch.stack([ra.data.reshape(-1), dec.data.reshape(-1)], dim=1).contiguous()
    pix = pj.sky2pix((shape, wcs), sky, safe=True)
    assert float(pix[:, 0].min()) >= 1 and float(pix[:, 0].max()) <= shape[0]
    assert float(pix[:, 1].min()) >= 1 and float(pix[:, 1].max()) <= shape[1]


def test_pixareamap_vs_reference_data(pj, O, dev, literals):
    """test_geometry.jl:287-316: first column vs the python-pixell data files, sum|diff| < 100 eps."""
    import os
    from conftest import GOLDEN
    for lit in literals["pixareamap"]:
        if lit["kind"] == "fullsky_1deg":
            shape, wcs = pj.fullsky_geometry(1 * DEG)
        else:
            b = lit["box_deg"]
            shape, wcs = pj.geometry([[b[0][0] * DEG, b[0][1] * DEG], [b[1][0] * DEG, b[1][1] * DEG]],
                                     lit["res_arcmin"] * ARCMIN)
        pm = pj.pixareamap(shape, wcs, device=dev)
        col = pm.data[:, 0].cpu().numpy()
        ref = np.loadtxt(os.path.join(GOLDEN, lit["file"]))
        assert np.abs(col - ref).sum() < lit["tol_sum_abs"], lit["src"]
        # device sin vs glibc sin: a few ulp of sin(dec) ~ 1, scaled by the RA pixel width
        assert np.abs(col - O.pixarea_rows(wcs, shape[1])).max() < 8 * np.finfo(float).eps * abs(wcs.cdelt[0] * wcs.unit)
        assert bool((pm.data == pm.data[:, :1]).all())          # constant along RA
        m = pj.Enmap(torch.zeros((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
        assert np.abs(pj.pixareamap(m).data[:, 0].cpu().numpy() - ref).sum() < lit["tol_sum_abs"]


def test_gnomonic_on_device(pj, O, dev, literals):
    """test_geometry.jl:92-119: device TAN evaluators vs the oracle (libm-level tolerance), literals, and
    the full posmap of the 1827x1825 patch with the reference's own L1 bound (sum|diff| < 1e-9)."""
    g = literals["gnomonic"]
    wcs = pj.Gnomonic(g["cdelt"], g["crpix"], g["crval"])
    shape = tuple(g["shape"])
    for lit in g["pix2sky"]:
        a, d = pj.pix2sky((shape, wcs), float(lit["pix"][0]), float(lit["pix"][1]))
        assert np.allclose([a, d], lit["sky"], rtol=1.5e-8)
    ra, dec = pj.posmap(shape, wcs, device=dev)
    jj, ii = np.meshgrid(np.arange(1, shape[1] + 1, dtype=float), np.arange(1, shape[0] + 1, dtype=float), indexing="ij")
    era, edec = O.pix2sky_tan(wcs, ii.ravel(), jj.ravel())
    assert np.abs(ra.data.cpu().numpy().ravel() - era).sum() < 1e-9
    assert np.abs(dec.data.cpu().numpy().ravel() - edec).sum() < 1e-9
    x, y = pj.sky2pix((shape, wcs), to_dev(era, dev), to_dev(edec, dev))
    assert np.abs(x.cpu().numpy() - ii.ravel()).max() < 1e-6 and np.abs(y.cpu().numpy() - jj.ravel()).max() < 1e-6
    ex, ey = O.sky2pix_tan(wcs, era, edec)
    assert np.abs(x.cpu().numpy() - ex).max() < 1e-7 and np.abs(y.cpu().numpy() - ey).max() < 1e-7


# ---- reprojection ---------------------------------------------------------------------------------

def _shifted(pj, wcs, dx, dy):
    return type(wcs)(wcs.cdelt, (wcs.crpix[0] + dx, wcs.crpix[1] + dy), wcs.crval, wcs.unit)


def reproject_cases(pj):
    fs64 = pj.fullsky_geometry(2 * math.pi / 64)
    fs128 = pj.fullsky_geometry(2 * math.pi / 128)
    fs256 = pj.fullsky_geometry(2 * math.pi / 256)
    fs500 = pj.fullsky_geometry(2 * math.pi / 500)
    fs1000 = pj.fullsky_geometry(2 * math.pi / 1000)
    odd = ((45, 23), pj.CarClenshawCurtis((-8.0, 8.181818181818182), (22.5, 12.0), (4.0, 0.0)))
    box = pj.geometry([[10 * DEG, -10 * DEG], [-5 * DEG, 5 * DEG]], 2 * ARCMIN)
    box_fine = pj.geometry([[6 * DEG, -7 * DEG], [-3 * DEG, 4 * DEG]], 1 * ARCMIN)
    box_flip = pj.geometry([[-8 * DEG, 9 * DEG], [4 * DEG, -4 * DEG]], 1.5 * ARCMIN)
    wide = pj.geometry([[179 * DEG, -179 * DEG], [-60 * DEG, 60 * DEG]], 30 * ARCMIN)
    cases = {
        "refine2x_fullsky": (fs128, fs256),                     # the headline pattern (config 2/3)
        "refine2x_to_1000": (fs500, fs1000),
        "same_res_half_pixel_shift": (fs256, (fs256[0], _shifted(pj, fs256[1], 0.5, 0.5))),   # config 4 pattern
        "same_res_quarter_shift": (fs500, (fs500[0], _shifted(pj, fs500[1], -0.25, 0.3))),
        "identity": (fs128, fs128),
        "coarsen2x": (fs256, fs128),
        "coarsen4x": (fs256, fs64),
        "refine8x": (fs64, (fs500[0], fs500[1])),
        "odd_nx_source": (odd, fs128),
        "odd_nx_dest": (fs128, odd),
        "fullsky_to_box": (fs1000, box),
        "box_to_box_fine": (box, box_fine),
        "box_to_flipped_box": (box, box_flip),
        "flipped_box_to_box": (box_flip, box),
        "box_to_fullsky_mostly_zero": (box, fs256),
        "wide_box_to_fullsky": (wide, fs500),                  # non-periodic source, rewind jump inside tiles
        "fullsky_to_wide": (fs500, wide),
    }
    return cases


@pytest.mark.parametrize("ncomp", [1, 3])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_reproject_vs_oracle(pj, O, dev, ncomp, variant):
    rng = np.random.default_rng(42)
    for name, ((shape_in, wcs_in), (shape_out, wcs_out)) in reproject_cases(pj).items():
        nx, ny = shape_in[:2]
        src = rng.normal(size=(ncomp, ny, nx))
        expect = O.reproject(wcs_in, (nx, ny, ncomp), src, wcs_out, shape_out)
        plan = pj.ReprojectPlan((nx, ny, ncomp), wcs_in, shape_out, wcs_out, device=dev)
        plan.set_variant(variant)
        dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
        plan.execute(to_dev(src, dev), dst)
        got = dst.cpu().numpy()
        assert np.isfinite(got).all(), name
        err = np.abs(got - expect).max()
        assert err <= TOL_INTERP, (name, variant, err)
        assert bits_equal(got, expect), (name, variant, "not bit-identical", err)
        plan.close()


def test_reproject_enmap_api(pj, O, dev):
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["refine2x_fullsky"]
    rng = np.random.default_rng(5)
    src = rng.normal(size=(shape_in[1], shape_in[0]))
    m = pj.Enmap(to_dev(src, dev), wcs_in)
    out = pj.reproject(m, shape_out, wcs_out)
    assert out.shape == shape_out and out.wcs == wcs_out
    expect = O.reproject(wcs_in, shape_in, src[None], wcs_out, shape_out)[0]
    assert bits_equal(out.data.cpu().numpy(), expect)


@pytest.mark.parametrize("rh", [1, 5, 64])
def test_reproject_tile_heights(pj, O, dev, rh, monkeypatch):
    """Different tile heights / lane widths walk the LDS ring differently; results must not change."""
    for pairs in (1, 2, 4):
        monkeypatch.setenv("PXL_REPROJECT_RH", str(rh))
        monkeypatch.setenv("PXL_REPROJECT_PAIRS", str(pairs))
        rng = np.random.default_rng(6)
        for name in ("refine2x_to_1000", "same_res_quarter_shift", "coarsen2x", "box_to_flipped_box"):
            (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)[name]
            src = rng.normal(size=(1, shape_in[1], shape_in[0]))
            plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
            dst = torch.empty(plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
            plan.execute(to_dev(src, dev), dst)
            assert bits_equal(dst.cpu().numpy(), O.reproject(wcs_in, shape_in, src, wcs_out, shape_out)), (name, rh, pairs)


def test_reproject_windows(pj, O, dev):
    """Dec-strip windows: a shard that holds only the rows it needs gives the same bits as the full map,
    and execute_rows() can split a strip into interior + boundary launches."""
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["same_res_quarter_shift"]
    nx, ny = shape_in
    rng = np.random.default_rng(8)
    src = rng.normal(size=(3, ny, nx))
    full = O.reproject(wcs_in, (nx, ny, 3), src, wcs_out, shape_out)
    nyo = shape_out[1]
    for lo, hi in ((0, 60), (60, 200), (200, nyo)):
        probe = pj.ReprojectPlan((nx, ny, 3), wcs_in, shape_out, wcs_out, dst_rows=(lo, hi - lo), device=dev)
        s_lo, s_hi = probe.src_rows_needed()
        assert (s_lo, s_hi) == O.reproject_src_rows(wcs_in, shape_in, wcs_out, shape_out, lo, hi - lo)
        plan = pj.ReprojectPlan((nx, ny, 3), wcs_in, shape_out, wcs_out, src_rows=(s_lo, s_hi - s_lo),
                                dst_rows=(lo, hi - lo), device=dev)
        s = to_dev(src[:, s_lo:s_hi], dev)
        dst = torch.empty(plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
        plan.execute(s, dst)
        assert bits_equal(dst.cpu().numpy(), full[:, lo:hi])
        dst2 = torch.full_like(dst, float("nan"))
        plan.build_tables()
        mid = (hi - lo) // 3
        plan.execute_rows(s, dst2, mid, (hi - lo) - mid)
        plan.execute_rows(s, dst2, 0, mid)
        assert bits_equal(dst2.cpu().numpy(), full[:, lo:hi])


def test_reproject_properties_linearity(pj, dev):
    """Size-independent property: reproject is linear in the map (exactly, for power-of-two scalings)."""
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["refine2x_to_1000"]
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn((shape_in[1], shape_in[0]), dtype=torch.float64, generator=g).to(dev)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    ra = pj.reproject(pj.Enmap(a, wcs_in), shape_out, wcs_out, plan=plan).data
    r4 = pj.reproject(pj.Enmap(4.0 * a, wcs_in), shape_out, wcs_out, plan=plan).data
    assert torch.equal(r4, 4.0 * ra)
    ones = pj.reproject(pj.Enmap(torch.ones_like(a), wcs_in), shape_out, wcs_out, plan=plan).data
    assert float((ones - 1.0).abs().max()) < 1e-13       # weights sum to one on a full-sky map


# ---- scattered sampling ---------------------------------------------------------------------------

@pytest.mark.parametrize("ncomp", [1, 3])
def test_sample_bilinear_vs_oracle(pj, O, dev, ncomp):
    rng = np.random.default_rng(9)
    n = 20011
    for name, (shape, wcs) in geoms(pj).items():
        if shape[0] * shape[1] > 4_000_000:
            continue
        nx, ny = shape
        src = rng.normal(size=(ncomp, ny, nx))
        u1, u2 = rng.random(n), rng.random(n)
        sky = np.stack([2 * math.pi * u1 - math.pi, np.arcsin(2 * u2 - 1)], axis=1)     # uniform on the sphere
        sky[:50, 0] += 6 * math.pi                                                        # far-away periods
        m = pj.Enmap(to_dev(src if ncomp > 1 else src[0], dev), wcs)
        got = pj.sample_bilinear(m, to_dev(sky, dev)).cpu().numpy()
        expect = O.sample_bilinear(wcs, (nx, ny, ncomp), src, sky)
        assert np.abs(got - expect).max() <= TOL_INTERP, name
        assert bits_equal(got, expect), name


def test_sample_nonfinite_and_empty(pj, O, dev):
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 64)
    src = np.random.default_rng(1).normal(size=(shape[1], shape[0]))
    m = pj.Enmap(to_dev(src, dev), wcs)
    sky = np.array([[0.1, 0.2], [float("nan"), 0.0], [0.3, float("inf")], [1.0, -1.0]])
    got = pj.sample_bilinear(m, to_dev(sky, dev)).cpu().numpy()[0]
    expect = O.sample_bilinear(wcs, (shape[0], shape[1], 1), src[None], sky)[0]
    assert np.isnan(got[1]) and np.isnan(got[2]) and np.isnan(expect[1]) and np.isnan(expect[2])
    assert got[0] == expect[0] and got[3] == expect[3]
    empty = pj.sample_bilinear(m, torch.empty((0, 2), dtype=torch.float64, device=dev))
    assert empty.shape == (1, 0)


def test_sample_degenerate_windows(pj, O, dev):
    """The direct sampler on windows that have fewer than two elements to read: an EMPTY resident window (a rank whose
    strip holds none of the map's rows; torch hands out a null data pointer for it) and 1 x 1 maps, periodic and not.
    Every point issues its gathers unconditionally, so the kernel must aim the unused ones at memory that exists
    (they used to read address 0 / one element past a 1-element plane)."""
    rng = np.random.default_rng(5)
    n = 3000
    sky = np.stack([2 * math.pi * rng.random(n) - math.pi, np.arcsin(2 * rng.random(n) - 1)], axis=1)
    d_sky = to_dev(sky, dev)
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 64)
    nx, ny = shape
    for f32 in (False, True):
        dt = torch.float32 if f32 else torch.float64
        for r0 in (0, 7, ny):
            empty = pj.Enmap(torch.empty((0, nx), dtype=dt, device=dev), wcs)
            assert empty.data.data_ptr() == 0 or empty.data.numel() == 0
            got = pj.sample_bilinear(empty, d_sky, src_rows=(r0, 0), full_shape=(nx, ny, 1)).cpu().numpy()
            assert got.shape == (1, n) and not got.any()             # nothing resident: every tap reads as zero
    for periodic in (True, False):
        w1 = pj.CarClenshawCurtis((-360.0, 180.0), (1.0, 1.0), (0.0, 0.0)) if periodic else \
            pj.CarClenshawCurtis((-1.0, 1.0), (1.0, 1.0), (0.0, 0.0))
        for f32 in (False, True):
            src = np.array([[[1.75]]])
            m = pj.Enmap(torch.from_numpy(src.astype(np.float32) if f32 else src).to(dev)[0], w1)
            near = np.stack([0.02 * (rng.random(n) - 0.5), 0.02 * (rng.random(n) - 0.5)], axis=1)
            both = np.concatenate([near, sky])
            got = pj.sample_bilinear(m, to_dev(both, dev)).cpu().numpy()
            if f32:
                expect = O.sample_bilinear_f32(w1, (1, 1, 1), src.astype(np.float32), both)
                assert np.array_equal(got.view(np.int32), expect.view(np.int32)), periodic
            else:
                expect = O.sample_bilinear(w1, (1, 1, 1), src, both)
                assert bits_equal(got, expect), periodic
            assert np.abs(got).max() > 0


def test_sample_matches_reproject(pj, dev):
    """Consistency of the two samplers: sampling at the output pixel centres of a reprojection gives the
    reprojected map to rounding (they use the reference's two different sky2pix roundings)."""
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["same_res_quarter_shift"]
    g = torch.Generator(device="cpu").manual_seed(2)
    a = torch.randn((shape_in[1], shape_in[0]), dtype=torch.float64, generator=g).to(dev)
    m = pj.Enmap(a, wcs_in)
    rep = pj.reproject(m, shape_out, wcs_out).data
    ra, dec = pj.posmap(shape_out, wcs_out, device=dev, safe=False)
    sky = torch.stack([ra.data.reshape(-1), dec.data.reshape(-1)], dim=1).contiguous()
    smp = pj.sample_bilinear(m, sky).reshape(rep.shape)
    assert float((smp - rep).abs().max()) < 1e-9


# ---- error behaviour ------------------------------------------------------------------------------

def test_errors_are_reported_not_swallowed(pj, dev):
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    with pytest.raises(RuntimeError):                      # CPU tensors are refused: no CPU fallback
        pj.pix2sky((shape, wcs), torch.zeros((4, 2), dtype=torch.float64))
    with pytest.raises(TypeError):
        pj.pix2sky((shape, wcs), torch.zeros((4, 2), dtype=torch.float32, device=dev))
    bad = pj.CarClenshawCurtis((0.0, 1.0), (1.0, 1.0), (0.0, 0.0))
    with pytest.raises(pj.PixellHipError) as ei:
        pj.pix2sky((shape, bad), torch.zeros((4, 2), dtype=torch.float64, device=dev))
    assert ei.value.code == -22 and "WCS" in str(ei.value)
    with pytest.raises(pj.PixellHipError):
        pj.ReprojectPlan(shape, wcs, shape, wcs, dst_rows=(100, 500), device=dev)
    with pytest.raises(AssertionError):                    # car_proj.jl:156
        pj.pix2sky((shape, wcs), [1.0, 2.0, 3.0])


# ---- dec-strip sharding on the device (single GPU stands in for every rank; no collective needed) ---

@pytest.mark.parametrize("world", [2, 4, 8])
def test_dec_strip_plans_on_device(pj, dev, world):
    """Each rank's window plan (own rows + halo, interior/boundary split) reproduces the full-map launch
    bit for bit.  Halo rows are copied from the full map here; the exchange itself is covered by the gloo
    tests (tests/test_sharding_gloo.py) and uses the same DecStripLayout."""
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["same_res_quarter_shift"]
    nx, ny = shape_in
    nc = 3
    g = torch.Generator(device="cpu").manual_seed(3)
    full_src = torch.randn((nc, ny, nx), dtype=torch.float64, generator=g).to(dev)
    full_plan = pj.ReprojectPlan((nx, ny, nc), wcs_in, shape_out, wcs_out, device=dev)
    full_dst = torch.empty(full_plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
    full_plan.execute(full_src, full_dst)
    for rank in range(world):
        L = pj.sharding.DecStripLayout((nx, ny, nc), wcs_in, shape_out, wcs_out, rank, world)
        plan = pj.ReprojectPlan((nx, ny, nc), wcs_in, shape_out, wcs_out, src_rows=L.src_window,
                                dst_rows=L.dst_window, device=dev)
        # the C library's host-side row logic agrees with the Python planner
        assert plan.src_rows_needed() == L.need[rank]
        assert plan.rows_covered(*L.own[rank]) == L.interior
        src = torch.full(L.src_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
        src[:, L.own_slice(), :] = full_src[:, L.own[rank][0]:L.own[rank][1], :]
        dst = torch.full(L.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
        plan.build_tables()
        i_lo, i_hi = L.interior
        plan.execute_rows(src, dst, i_lo, i_hi - i_lo)                 # before the halo "arrives"
        for _, lo, hi in L.recvs:
            src[:, lo - L.buf_lo:hi - L.buf_lo, :] = full_src[:, lo:hi, :]
        if i_lo > 0:
            plan.execute_rows(src, dst, 0, i_lo)
        if i_hi < L.dst_window[1]:
            plan.execute_rows(src, dst, i_hi, L.dst_window[1] - i_hi)
        lo, n = L.dst_window
        assert torch.equal(dst, full_dst[:, lo:lo + n, :]), (world, rank)


def test_dec_strip_reprojector_world1(pj, dev):
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["refine2x_to_1000"]
    sh = pj.DecStripReprojector(shape_in, wcs_in, shape_out, wcs_out, 0, 1, dev)
    src, dst = sh.alloc_src(), sh.alloc_dst()
    pj.fill_random_(src, 7)
    e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    sh.step(src, dst, events=e)
    torch.cuda.synchronize()
    assert e[0].elapsed_time(e[1]) > 0
    ref = pj.reproject(pj.Enmap(src[0], wcs_in), shape_out, wcs_out).data
    assert torch.equal(dst[0], ref)


def test_fill_random_is_sharding_invariant(pj, dev):
    """The benchmark's synthetic map is keyed by absolute pixel index, so a strip generated on its own
    equals the same rows of the full map."""
    full = torch.empty((40, 64), dtype=torch.float64, device=dev)
    pj.fill_random_(full, 1234, 0)
    part = torch.empty((10, 64), dtype=torch.float64, device=dev)
    pj.fill_random_(part, 1234, 15 * 64)
    assert torch.equal(part, full[15:25])
    assert abs(float(full.mean())) < 0.1 and abs(float(full.std()) - 1.0) < 0.1
    sky = torch.empty((10000, 2), dtype=torch.float64, device=dev)
    pj.fill_sphere_points_(sky, 42)
    assert float(sky[:, 0].min()) >= -math.pi and float(sky[:, 0].max()) < math.pi
    assert float(sky[:, 1].abs().max()) <= math.pi / 2 and abs(float(torch.sin(sky[:, 1]).mean())) < 0.05


# ---- unwind! on long batches: the parallel scan must reproduce the sequential recurrence bit for bit ----

def _identity_wcs(pj):
    # unit = 1, cdelt = 1, crval = 0, crpix = 0: pix2sky is the identity, so tests can feed angles directly
    return (10, 10), pj.CarClenshawCurtis((1.0, 1.0), (0.0, 0.0), (0.0, 0.0), 1.0)


@pytest.mark.parametrize("n", [4097, 100003, 2_000_000])
def test_unwind_long_batches_bit_exact(pj, O, dev, n):
    g = _identity_wcs(pj)
    rng = np.random.default_rng(n)
    walk = np.cumsum(rng.normal(0, 1.5, (n, 2)), axis=0) + rng.uniform(-50, 50, 2)   # wanders over many periods
    got = pj.pix2sky(g, to_dev(walk, dev), safe=True).cpu().numpy()
    assert bits_equal(got, O.pix2sky(g[1], walk, O.WRAP_UNWIND))
    jumps = rng.uniform(-400, 400, (n, 2))                                           # a wrap at almost every step
    got = pj.pix2sky(g, to_dev(jumps, dev), safe=True).cpu().numpy()
    assert bits_equal(got, O.pix2sky(g[1], jumps, O.WRAP_UNWIND))


def test_unwind_one_pass_equals_two_pass(pj, O, dev):
    """Out-of-place pix2sky!(safe=true) on a long batch takes the one-pass kernel (decoupled look-back between 512-point wave
    chunks, k_unwind_onepass); the in-place call keeps the sums -> scan -> apply form.  Same arithmetic per element: the two
    must agree bit for bit with each other and with the oracle's serial recurrence -- on a walk that wraps all the time (carries
    of thousands of periods cross every chunk boundary), with the batch length not a multiple of anything, with a NaN late in one
    row (everything after it is NaN, nothing before it changes) and with a half-period tie in the middle (the verification fails
    and the fallback produces the answer)."""
    g = _identity_wcs(pj)
    n = 3_000_017
    rng = np.random.default_rng(4)
    walk = np.cumsum(rng.normal(0.3, 2.5, (n, 2)), axis=0)
    cases = {"walk": walk}
    w2 = walk.copy()
    w2[2_500_003, 1] = float("nan")
    cases["late NaN"] = w2
    w3 = walk.copy()
    w3[1_500_000:, 0] = w3[1_499_999, 0] + math.pi * np.arange(1, n - 1_500_000 + 1)      # exact half-period steps from there on
    cases["ties"] = w3
    for name, a in cases.items():
        exp = O.pix2sky(g[1], a, O.WRAP_UNWIND)
        d = to_dev(a, dev)
        one = pj.pix2sky_(g, d, torch.empty_like(d), safe=True).cpu().numpy()        # out of place: one pass, 7 168-point chunks
        two = pj.pix2sky_(g, d, d, safe=True).cpu().numpy()                          # in place: two passes
        assert _same_bits_or_nan(one, exp), name
        assert _same_bits_or_nan(two, exp), name
    # a longer batch (1 256 chunks), on coordinates inside the interval (the wave vote) and wandering ones mixed
    n = 9_000_011
    rng = np.random.default_rng(9)
    a = rng.uniform(-3.0, 3.0, (n, 2))
    a[n // 3: n // 2] += np.cumsum(rng.normal(0.0, 2.0, (n // 2 - n // 3, 2)), axis=0)
    d = to_dev(a, dev)
    assert bits_equal(pj.pix2sky_(g, d, torch.empty_like(d), safe=True).cpu().numpy(), O.pix2sky(g[1], a, O.WRAP_UNWIND))


def test_unwind_in_range_vote_bit_exact(pj, O, dev):
    """The unwind sources skip the exact-fmod machinery when a whole wave's coordinates already lie in [ref - P/2, ref + P/2)
    (a wave vote; `UwSrcPix2::to_m` / `UwSrcAng2::to_m`).  Same bits as the general path: batches where every wave votes yes,
    batches where waves alternate, the interval's end points and their neighbours, a real full-sky geometry, long and short
    batches (one-pass, two-pass and single-block kernels), and `unwind!` with a reference that is not zero."""
    rng = np.random.default_rng(77)
    ident = _identity_wcs(pj)
    pi = math.pi
    edge = np.array([-pi, np.nextafter(-pi, 0), np.nextafter(-pi, -4), pi, np.nextafter(pi, 0), np.nextafter(pi, 4), 0.0, -0.0, 1e-300, -1e-300])
    for n in (700, 50_000, 1_200_003):
        inside = rng.uniform(-pi, pi, (n, 2))
        inside[:, 1] = rng.uniform(-pi / 2, pi / 2, n)
        mixed = inside.copy()
        blocks = (np.arange(n) // 64) % 3 == 1                       # every third wave leaves the interval
        mixed[blocks, 0] += rng.choice([-4, -2, 2, 6], blocks.sum()) * pi
        edges = inside.copy()
        at = rng.integers(0, n, 4 * len(edge))
        edges[at, 0] = np.tile(edge, 4)
        edges[at[::2], 1] = np.tile(edge, 4)[::2]
        for name, a in (("inside", inside), ("mixed", mixed), ("edges", edges)):
            exp = O.pix2sky(ident[1], a, O.WRAP_UNWIND)
            d = to_dev(a, dev)
            assert bits_equal(pj.pix2sky_(ident, d, torch.empty_like(d), safe=True).cpu().numpy(), exp), (name, n, "out of place")
            assert bits_equal(pj.pix2sky_(ident, d, d, safe=True).cpu().numpy(), exp), (name, n, "in place")
    # a real geometry: every pixel of the map is inside the interval, half a pixel beyond the seam is not
    shape, wcs = pj.fullsky_geometry(2 * pi / 21600)
    n = 400_001
    pix = np.stack([rng.uniform(0.5, shape[0] + 0.5, n), rng.uniform(1, shape[1], n)], axis=1)
    pix[::4097, 0] = rng.choice([-3.0, 0.25, shape[0] + 0.75, 3.0 * shape[0]], len(pix[::4097]))
    exp = O.pix2sky(wcs, pix, O.WRAP_UNWIND)
    d = to_dev(pix, dev)
    assert bits_equal(pj.pix2sky_((shape, wcs), d, torch.empty_like(d), safe=True).cpu().numpy(), exp)
    # unwind! with its own period and reference
    for period, ref in ((2 * pi, 0.0), (360.0, 180.0), (1.0, -0.3)):
        for n in (900, 300_001):
            a = ref + rng.uniform(-period / 2, period / 2, (n, 2))
            a[at[at < n], 0] = ref + rng.choice([-period / 2, period / 2, np.nextafter(period / 2, 0)], (at < n).sum())
            b = a.copy()
            b[(np.arange(n) // 64) % 2 == 1, 1] += 3 * period
            for name, x in (("inside", a), ("mixed", b)):
                exp = np.stack([O.unwind_row(x[:, 0].copy(), period, ref), O.unwind_row(x[:, 1].copy(), period, ref)], axis=1)
                assert bits_equal(pj.unwind_(to_dev(x, dev), period, ref).cpu().numpy(), exp), (name, n, period, ref)


def test_unwind_ties_and_nonfinite(pj, O, dev):
    """Adversarial inputs: steps of exactly half a period (rint ties-to-even decides), and NaN/Inf, which in
    the sequential recurrence poison every later element -- the verified scan must hand those to the serial
    kernel and still match the oracle bit for bit."""
    g = _identity_wcs(pj)
    n = 20000
    k = np.arange(n, dtype=np.float64)
    half = math.pi                                         # half of the 2*pi period
    ties = np.stack([k * half, -k * half + 0.25], axis=1)
    got = pj.pix2sky(g, to_dev(ties, dev), safe=True).cpu().numpy()
    assert bits_equal(got, O.pix2sky(g[1], ties, O.WRAP_UNWIND))
    tern = np.stack([(k % 3) * half, (k % 5) * half * 0.5], axis=1)
    got = pj.pix2sky(g, to_dev(tern, dev), safe=True).cpu().numpy()
    assert bits_equal(got, O.pix2sky(g[1], tern, O.WRAP_UNWIND))
    bad = np.random.default_rng(0).uniform(-20, 20, (n, 2))
    bad[7000, 0] = float("nan")
    bad[12345, 1] = float("inf")
    got = pj.pix2sky(g, to_dev(bad, dev), safe=True).cpu().numpy()
    exp = O.pix2sky(g[1], bad, O.WRAP_UNWIND)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert bits_equal(got[:7000], exp[:7000]) and bits_equal(got[:12345, 1], exp[:12345, 1])
    assert np.isnan(got[7000:, 0]).all() and np.isnan(got[12345:, 1]).all()


# ---- CAR <-> Gnomonic (non-separable) reprojection: tolerance-checked (FP64 transcendentals) ---------

def test_generic_reproject_car_tan(pj, O, dev, literals):
    g = literals["gnomonic"]
    tan_wcs = pj.Gnomonic(g["cdelt"], g["crpix"], g["crval"])
    tan_shape = (600, 500)
    tan_wcs = pj.Gnomonic(g["cdelt"], (300.5, 250.5), g["crval"])
    # a CAR box around the same field (RA 97.5 deg, DEC -7.5 deg), 0.5 arcmin pixels like the TAN patch
    car_shape, car_wcs = pj.geometry([[101 * DEG, 94 * DEG], [-11 * DEG, -4 * DEG]], 0.5 * ARCMIN)
    rng = np.random.default_rng(12)
    # smooth maps (so a last-bit difference in x, y moves values by ~1e-13, not by a pixel-to-pixel jump)
    def smooth(shape, nc):
        yy, xx = np.meshgrid(np.arange(shape[1]), np.arange(shape[0]), indexing="ij")
        return np.stack([np.sin(0.01 * (c + 1) * xx) * np.cos(0.013 * yy) + 0.001 * c * xx for c in range(nc)])
    for (s_in, w_in, p_in), (s_out, w_out, p_out), nc in (
            ((car_shape, car_wcs, 0), (tan_shape, tan_wcs, 1), 2),
            ((tan_shape, tan_wcs, 1), (car_shape, car_wcs, 0), 1),
            ((tan_shape, tan_wcs, 1), ((300, 260), pj.Gnomonic(g["cdelt"], (120.0, 100.0), (98.0, -7.0)), 1), 1)):
        src = smooth(s_in, nc)
        m = pj.Enmap(to_dev(src if nc > 1 else src[0], dev), w_in)
        out = pj.reproject(m, s_out, w_out)
        assert out.shape[:2] == tuple(s_out)
        got = out.data.cpu().numpy().reshape(nc, s_out[1], s_out[0])
        exp = O.reproject_generic(w_in, p_in, (s_in[0], s_in[1], nc), src, w_out, p_out, s_out)
        assert np.isfinite(got).all()
        assert np.abs(got - exp).max() < 1e-9, (p_in, p_out, np.abs(got - exp).max())
        assert np.abs(exp).max() > 0.1                      # the maps overlap: this is not a comparison of zeros
    # the tiled kernel (coordinates interpolated per 128 x 32 tile, checked to 1e-10 pixel) against the per-pixel one
    # (PXL_GENERIC_EXACT=1) and the oracle: a periodic full-sky CAR source seen from a TAN patch that straddles the
    # RA = 180 deg seam (tiles across the rewind jump must fall back to exact evaluation), at 0.5 and at 20 arcmin
    # (coarse pixels: the interpolant fails its check everywhere and every tile takes the exact path)
    import os
    for res_arcmin, n_car in ((0.5, 43200), (20.0, 1080)):
        fshape, fwcs = pj.fullsky_geometry(2 * math.pi / n_car)
        if n_car > 2000:           # a declination strip of the full-sky map is enough for the oracle's patience
            fshape, fwcs = pj.slice_geometry(fshape, fwcs, None, (10801 - 700, 10801 + 700))
        patch_wcs = pj.Gnomonic((res_arcmin / 60, res_arcmin / 60), (300.5, 200.5), (179.9, 0.3))
        patch_shape = (600, 400)
        src = smooth(fshape, 1)
        m = pj.Enmap(to_dev(src[0], dev), fwcs)
        tiled = pj.reproject(m, patch_shape, patch_wcs).data.cpu().numpy()
        os.environ["PXL_GENERIC_EXACT"] = "1"
        try:
            exact = pj.reproject(m, patch_shape, patch_wcs).data.cpu().numpy()
        finally:
            del os.environ["PXL_GENERIC_EXACT"]
        exp = O.reproject_generic(fwcs, 0, (fshape[0], fshape[1], 1), src, patch_wcs, 1, patch_shape)[0]
        assert np.abs(exact - exp).max() < 1e-9 and np.abs(tiled - exp).max() < 1e-9, (res_arcmin, np.abs(tiled - exp).max())
        assert np.abs(tiled - exact).max() < 1e-10, (res_arcmin, np.abs(tiled - exact).max())
        assert np.abs(exp).max() > 0.1
    # consistency with the separable kernel when both maps are CAR
    fs = pj.fullsky_geometry(2 * math.pi / 200)
    fs2 = pj.fullsky_geometry(2 * math.pi / 300)
    src = smooth(fs[0], 1)
    exp = O.reproject(fs[1], fs[0], src, fs2[1], fs2[0])
    gen = O.reproject_generic(fs[1], 0, (fs[0][0], fs[0][1], 1), src, fs2[1], 0, fs2[0])
    assert bits_equal(gen, exp)


def test_generic_reproject_plan_equals_one_shot(pj, O, dev):
    """GenericReprojectPlan (pxl_generic_plan_*: the coordinate lattice and the per-pixel tile list kept between calls) gives the
    bits of the one-shot entry -- same kernels, same lattice -- for one and several components, for CAR -> TAN, TAN -> CAR and
    TAN -> TAN, on a patch that straddles the RA seam of a periodic source (tiles across the rewind jump are evaluated per pixel:
    the plan must launch that kernel too) and on coarse pixels (every tile per pixel); it is reusable on other maps of the same
    geometry, checks what it is given, and matches the oracle."""
    rng = np.random.default_rng(5)
    # a declination strip of the 0.5' full-sky map (full rings: periodic), as in the test above; the interpolant passes its check at
    # 0.5' pixels and fails it everywhere at 40'
    fshape, fwcs = pj.fullsky_geometry(2 * math.pi / 43200)
    fshape, fwcs = pj.slice_geometry(fshape, fwcs, None, (10801 - 500, 10801 + 500))
    seam = ((700, 500), pj.Gnomonic((0.5 / 60, 0.5 / 60), (350.5, 250.5), (179.9, 0.3)))
    inner = ((520, 300), pj.Gnomonic((0.5 / 60, 0.5 / 60), (260.5, 150.5), (40.0, -1.0)))
    coarse = ((300, 200), pj.Gnomonic((40.0 / 60, 40.0 / 60), (150.5, 100.5), (10.0, 3.0)))
    xx = np.arange(fshape[0])[None, :]
    yy = np.arange(fshape[1])[:, None]
    smooth = np.stack([np.sin(0.01 * (c + 1) * xx) * np.cos(0.013 * yy) + 0.001 * c * xx for c in range(2)])
    for (oshape, owcs), nc, want_exact in ((seam, 1, "some"), (inner, 2, "none"), (coarse, 2, "all")):
        src = smooth[:nc]
        m = pj.Enmap(to_dev(src if nc > 1 else src[0], dev), fwcs)
        one = pj.reproject(m, oshape, owcs)
        plan = pj.GenericReprojectPlan(fshape, fwcs, oshape, owcs, device=dev)
        ex, tot = plan.tiles()
        assert tot == -(-oshape[0] // 128) * -(-oshape[1] // 32)
        assert {"some": 0 < ex < tot, "none": ex == 0, "all": ex == tot}[want_exact], (ex, tot)
        out = pj.Enmap(torch.full_like(one.data, float("nan")), owcs)
        got = pj.reproject(m, oshape, owcs, out=out, plan=plan)
        assert got is out
        assert bits_equal(out.data.cpu().numpy(), one.data.cpu().numpy()), want_exact
        exp = O.reproject_generic(fwcs, 0, (fshape[0], fshape[1], nc), src, owcs, 1, oshape)
        assert np.abs(out.data.cpu().numpy().reshape(nc, oshape[1], oshape[0]) - exp).max() < 1e-9
        # the same plan on another map of the geometry, another component count
        m2 = pj.Enmap(torch.randn((3, fshape[1], fshape[0]), dtype=torch.float64, device=dev), fwcs)
        assert bits_equal(pj.reproject(m2, oshape, owcs, plan=plan).data.cpu().numpy(), pj.reproject(m2, oshape, owcs).data.cpu().numpy())
        with pytest.raises(ValueError):
            pj.reproject(m2, (oshape[0] + 1, oshape[1]), owcs, plan=plan)
        with pytest.raises(ValueError):
            plan.execute(m2.data[:, :-1, :].contiguous(), out.data)
        plan.close()
    # TAN -> CAR and TAN -> TAN
    tshape, twcs = inner
    tmap = pj.Enmap(to_dev(smooth[0][:tshape[1], :tshape[0]].copy(), dev), twcs)
    cshape, cwcs = pj.geometry([[41 * DEG, 39 * DEG], [-2 * DEG, 0 * DEG]], 0.5 * ARCMIN)
    for oshape, owcs in ((cshape, cwcs), ((200, 180), pj.Gnomonic((0.6 / 60, 0.6 / 60), (100.0, 90.0), (40.2, -1.1)))):
        plan = pj.GenericReprojectPlan(tshape, twcs, oshape, owcs, device=dev)
        assert bits_equal(pj.reproject(tmap, oshape, owcs, plan=plan).data.cpu().numpy(), pj.reproject(tmap, oshape, owcs).data.cpu().numpy())
    with pytest.raises(TypeError):
        pj.reproject(tmap, cshape, cwcs, plan=pj.ReprojectPlan(fshape, fwcs, fshape, fwcs, device=dev))


def test_rewind_and_unwind_entries(pj, O, dev):
    """Standalone rewind! / unwind! (enmap_ops.jl:15-32) with the periods and reference angles the reference
    itself uses: 2*pi about 0 for angles, the pixel period about the map centre for sky2pix."""
    rng = np.random.default_rng(31)
    a = rng.uniform(-80, 80, 100001)
    for period, ref in ((2 * math.pi, 0.0), (360.0, 181.0), (43200.000000000007, 21601.0)):
        got = pj.rewind_(to_dev(a * (period / 6.0), dev), period, ref).cpu().numpy()
        exp = np.array([O.rewind(v, period, ref) for v in a * (period / 6.0)])
        assert bits_equal(got, exp), (period, ref)
    walk = np.cumsum(rng.normal(0, 2.0, 300000)) + 11.0
    for period, ref in ((2 * math.pi, 0.0), (5.0, 1.25)):
        got = pj.unwind_(to_dev(walk, dev), period, ref).cpu().numpy()          # 1-D vector
        assert bits_equal(got, O.unwind_row(walk, period, ref)), (period, ref)
        two = np.stack([walk, -0.5 * walk[::-1]], axis=1)
        got = pj.unwind_(to_dev(two, dev), period, ref).cpu().numpy()           # 2xN batch, dims=2
        exp = np.stack([O.unwind_row(two[:, 0], period, ref), O.unwind_row(two[:, 1], period, ref)], axis=1)
        assert bits_equal(got, exp), (period, ref)
    short = rng.uniform(-20, 20, 100)
    assert bits_equal(pj.unwind_(to_dev(short, dev)).cpu().numpy(), O.unwind_row(short))


def test_device_matches_frozen_golden(pj, dev):
    """The committed golden vectors (tests/golden/oracle_golden.json) through the device path, bit for bit."""
    import json
    import os
    from conftest import GOLDEN, unhex
    with open(os.path.join(GOLDEN, "oracle_golden.json")) as f:
        g = json.load(f)
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 64)
    nx, ny = shape
    jj, ii = np.meshgrid(np.arange(1, ny + 1, dtype=float), np.arange(1, nx + 1, dtype=float), indexing="ij")
    m = pj.Enmap(to_dev(((jj - 1) * nx + ii) ** 2, dev), wcs)
    shape2, wcs2 = pj.fullsky_geometry(2 * math.pi / 128)
    assert bits_equal(pj.reproject(m, shape2, wcs2).data.cpu().numpy().ravel(), unhex(g["refined"], 1).ravel())
    shifted = pj.CarClenshawCurtis(wcs.cdelt, (wcs.crpix[0] + 0.5, wcs.crpix[1] + 0.5), wcs.crval)
    assert bits_equal(pj.reproject(m, shape, shifted).data.cpu().numpy().ravel(), unhex(g["shifted"], 1).ravel())
    pix = to_dev(unhex(g["pix"]), dev)
    assert bits_equal(pj.pix2sky(m, pix, safe=False).cpu().numpy(), unhex(g["pix2sky_unsafe"]))
    assert bits_equal(pj.pix2sky_rewind(m, pix).cpu().numpy(), unhex(g["pix2sky_rewind"]))
    assert bits_equal(pj.pix2sky(m, pix, safe=True).cpu().numpy(), unhex(g["pix2sky_unwind"]))
    sky = to_dev(unhex(g["pix2sky_unsafe"]), dev)
    assert bits_equal(pj.sky2pix(m, sky, safe=True).cpu().numpy(), unhex(g["sky2pix_safe_recip"]))
    x, y = pj.sky2pix_broadcast(m, sky[:, 0].contiguous(), sky[:, 1].contiguous(), safe=True)
    assert bits_equal(torch.stack([x, y], dim=1).cpu().numpy(), unhex(g["sky2pix_safe_div"]))



# ---- randomized geometry sweep --------------------------------------------------------------------

def _random_geometry(pj, rng, periodic_ok=True):
    """A random CAR geometry: either a full-circle map (periodic in RA) or a partial-sky patch, with random
    resolution, sign of cdelt (flips) and fractional crpix."""
    if periodic_ok and rng.random() < 0.5:
        nx = int(rng.integers(40, 700))
        ny = int(rng.integers(20, 300))
        sgn = -1.0 if rng.random() < 0.7 else 1.0
        cd1 = sgn * 360.0 / nx
        span = rng.uniform(20, 180)
        cd2 = (1.0 if rng.random() < 0.7 else -1.0) * span / ny
        crpix = (rng.uniform(1, nx), rng.uniform(1, ny))
        crval = (rng.uniform(-180, 180), 0.0)
    else:
        nx = int(rng.integers(30, 600))
        ny = int(rng.integers(20, 300))
        res = rng.uniform(0.02, 0.6)
        cd1 = (-1.0 if rng.random() < 0.7 else 1.0) * res * rng.uniform(0.8, 1.25)
        cd2 = (1.0 if rng.random() < 0.7 else -1.0) * res
        crpix = (rng.uniform(-20, nx + 20), rng.uniform(-20, ny + 20))
        crval = (rng.uniform(-180, 180), 0.0)
    return (nx, ny), pj.CarClenshawCurtis((cd1, cd2), crpix, crval)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_reproject_random_geometries(pj, O, dev, seed):
    """60 random source/destination geometry pairs per seed (periodic and partial-sky sources, up- and
    down-sampling, flips, odd sizes, fractional reference pixels, overlapping and disjoint footprints) through
    every kernel variant, bit for bit against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    for case in range(60):
        shape_in, wcs_in = _random_geometry(pj, rng)
        shape_out, wcs_out = _random_geometry(pj, rng)
        if rng.random() < 0.5:          # make the footprints overlap more often than chance
            wcs_out = pj.CarClenshawCurtis(wcs_out.cdelt, wcs_out.crpix, (wcs_in.crval[0] + rng.uniform(-5, 5), 0.0))
        nc = int(rng.integers(1, 3))
        src = rng.normal(size=(nc, shape_in[1], shape_in[0]))
        expect = O.reproject(wcs_in, (shape_in[0], shape_in[1], nc), src, wcs_out, shape_out)
        d_src = to_dev(src, dev)
        for variant in (0, 2, 1):
            plan = pj.ReprojectPlan((shape_in[0], shape_in[1], nc), wcs_in, shape_out, wcs_out, device=dev)
            plan.set_variant(variant)
            dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
            plan.execute(d_src, dst)
            got = dst.cpu().numpy()
            assert bits_equal(got, expect), (seed, case, variant, shape_in, wcs_in, shape_out, wcs_out,
                                             float(np.nanmax(np.abs(got - expect))))
            plan.close()


# ---- Float32 maps (Enmap{Float32}): Float64 coordinates and weights, one rounding at the store --------

def _f32_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.int32), b.view(np.int32))


@pytest.mark.parametrize("variant", [0, 1])
def test_reproject_float32_vs_oracle(pj, O, dev, variant):
    rng = np.random.default_rng(4242)
    for name, ((shape_in, wcs_in), (shape_out, wcs_out)) in reproject_cases(pj).items():
        nx, ny = shape_in[:2]
        nc = 2
        src = rng.normal(size=(nc, ny, nx)).astype(np.float32)
        expect = O.reproject_f32(wcs_in, (nx, ny, nc), src, wcs_out, shape_out)
        plan = pj.ReprojectPlan((nx, ny, nc), wcs_in, shape_out, wcs_out, device=dev)
        plan.set_variant(variant)
        dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float32, device=dev)
        plan.execute(torch.from_numpy(src).to(dev), dst)
        got = dst.cpu().numpy()
        assert np.isfinite(got).all(), name
        assert _f32_equal(got, expect), (name, variant, float(np.abs(got - expect).max()))
        plan.close()


def test_reproject_float32_random_geometries(pj, O, dev):
    rng = np.random.default_rng(5151)
    for case in range(60):
        shape_in, wcs_in = _random_geometry(pj, rng)
        shape_out, wcs_out = _random_geometry(pj, rng)
        if rng.random() < 0.5:
            wcs_out = pj.CarClenshawCurtis(wcs_out.cdelt, wcs_out.crpix, (wcs_in.crval[0] + rng.uniform(-5, 5), 0.0))
        src = rng.normal(size=(1, shape_in[1], shape_in[0])).astype(np.float32)
        expect = O.reproject_f32(wcs_in, (shape_in[0], shape_in[1], 1), src, wcs_out, shape_out)
        m = pj.Enmap(torch.from_numpy(src[0]).to(dev), wcs_in)
        out = pj.reproject(m, shape_out, wcs_out)
        assert out.data.dtype == torch.float32
        assert _f32_equal(out.data.cpu().numpy()[None], expect), (case, shape_in, wcs_in, shape_out, wcs_out)


def test_sample_float32_vs_oracle(pj, O, dev):
    rng = np.random.default_rng(6161)
    for name, (shape, wcs) in geoms(pj).items():
        if shape[0] * shape[1] > 4_000_000:
            continue
        nx, ny = shape
        src = rng.normal(size=(2, ny, nx)).astype(np.float32)
        n = 12345
        sky = np.stack([2 * math.pi * rng.random(n) - math.pi, np.arcsin(2 * rng.random(n) - 1)], axis=1)
        m = pj.Enmap(torch.from_numpy(src).to(dev), wcs)
        got = pj.sample_bilinear(m, to_dev(sky, dev))
        assert got.dtype == torch.float32
        assert _f32_equal(got.cpu().numpy(), O.sample_bilinear_f32(wcs, (nx, ny, 2), src, sky)), name


def test_float32_and_float64_agree(pj, dev):
    """The Float32 path is the Float64 path with storage rounding: results agree to Float32 precision."""
    (shape_in, wcs_in), (shape_out, wcs_out) = reproject_cases(pj)["refine2x_to_1000"]
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn((shape_in[1], shape_in[0]), dtype=torch.float32, generator=g).to(dev)
    r32 = pj.reproject(pj.Enmap(a, wcs_in), shape_out, wcs_out).data
    r64 = pj.reproject(pj.Enmap(a.double(), wcs_in), shape_out, wcs_out).data
    assert torch.equal(r32, r64.float())
    with pytest.raises(TypeError):
        plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
        plan.execute(a, torch.empty((shape_out[1], shape_out[0]), dtype=torch.float64, device=dev))


def test_unwind_inplace_and_overlap(pj, O, dev):
    """pix2sky!(m, buf, buf; safe=true) on a long batch: the in-place form verifies before it stores, so both the
    clean case and the half-period-tie case (which falls back to the multi-pass form) match the oracle; partially
    overlapping buffers are refused."""
    g = _identity_wcs(pj)
    n = 300_001
    rng = np.random.default_rng(77)
    walk = np.cumsum(rng.normal(0, 1.5, (n, 2)), axis=0)
    buf = to_dev(walk, dev)
    pj.pix2sky_(g, buf, buf, safe=True)
    assert bits_equal(buf.cpu().numpy(), O.pix2sky(g[1], walk, O.WRAP_UNWIND))
    k = np.arange(n, dtype=np.float64)
    ties = np.stack([k * math.pi, 0.5 * walk[:, 1]], axis=1)
    buf = to_dev(ties, dev)
    pj.pix2sky_(g, buf, buf, safe=True)
    assert bits_equal(buf.cpu().numpy(), O.pix2sky(g[1], ties, O.WRAP_UNWIND))
    big = torch.zeros((n + 8, 2), dtype=torch.float64, device=dev)
    with pytest.raises(RuntimeError):
        pj.pix2sky_(g, big[:n], big[8:], safe=True)
    # the same refusal for small batches (the single-block form, n <= 8192), and exact aliasing still works there
    for ns in (2, 100, 4096, 8192):
        with pytest.raises(RuntimeError):
            pj.pix2sky_(g, big[:ns], big[1:ns + 1], safe=True)
        small = to_dev(walk[:ns], dev)
        pj.pix2sky_(g, small, small, safe=True)
        assert bits_equal(small.cpu().numpy(), O.pix2sky(g[1], walk[:ns], O.WRAP_UNWIND))


def test_sample_row_pair_layout(pj, O, dev):
    """The row-pair copy of a map (two adjacent entries hold a point's whole 2x2 neighbourhood) gives the same bits
    as the direct gather and as the oracle: Float64 and Float32, full maps and declination strips, periodic and
    partial-sky maps, points far outside and non-finite ones."""
    rng = np.random.default_rng(99)
    for name, (shape, wcs) in geoms(pj).items():
        if shape[0] * shape[1] > 4_000_000:
            continue
        nx, ny = shape
        n = 30011
        sky = np.stack([2 * math.pi * rng.random(n) - math.pi, np.arcsin(2 * rng.random(n) - 1)], axis=1)
        sky[:40, 0] += 4 * math.pi
        sky[40] = (float("nan"), 0.1)
        sky[41] = (0.2, float("inf"))
        d_sky = to_dev(sky, dev)
        for nc, f32, (r0, nr) in ((1, False, (0, ny)), (3, False, (ny // 4, ny // 2)), (2, True, (0, ny)), (1, True, (3, 5))):
            src = rng.normal(size=(nc, nr, nx))
            if f32:
                src = src.astype(np.float32)
                expect = O.sample_bilinear_f32(wcs, (nx, ny, nc), src, sky, src_row0=r0, src_nrows=nr)
            else:
                expect = O.sample_bilinear(wcs, (nx, ny, nc), src, sky, src_row0=r0, src_nrows=nr)
            t = torch.from_numpy(src).to(dev)
            m = pj.Enmap(t if nc > 1 else t[0], wcs)
            pairs = pj.SamplePairs(m, src_rows=(r0, nr), full_shape=(nx, ny, nc))
            got = pj.sample_bilinear(None, d_sky, pairs=pairs).cpu().numpy()
            direct = pj.sample_bilinear(m, d_sky, src_rows=(r0, nr), full_shape=(nx, ny, nc)).cpu().numpy()
            if f32:
                assert _f32_equal(np.nan_to_num(got), np.nan_to_num(expect)) and _f32_equal(np.nan_to_num(got), np.nan_to_num(direct)), (name, nc, r0, nr)
            else:
                assert bits_equal(got, expect) and bits_equal(got, direct), (name, nc, r0, nr)
            assert np.array_equal(np.isnan(got), np.isnan(expect))
    # the pair buffer is caller-owned: wrong sizes are refused
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    m = pj.Enmap(torch.zeros((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
    with pytest.raises(ValueError):
        pj.SamplePairs(m, out=torch.empty(10, dtype=torch.float64, device=dev))
    # ... and so is one that does not start on a 64-byte sector (the layout keeps every cell inside one sector)
    lib = pj.load_library()
    need = lib.pxl_sample_pairs_elems(pj._lib.shape_arr((shape[0], shape[1], 1)), shape[1])
    assert need == 2 * 4 * ((shape[0] + 2) // 3) * (shape[1] + 1)              # groups of 4 entries, 3 new columns each
    with pytest.raises(pj.PixellHipError, match="64-byte"):
        pj.SamplePairs(m, out=torch.empty(need + 8, dtype=torch.float64, device=dev)[1:need + 1])
    # tiny maps: a Float32 row (groups of 8 entries, 7 new columns) can be the longer one
    assert lib.pxl_sample_pairs_elems(pj._lib.shape_arr((8, 8, 2)), 4) == 2 * 16 * 5 * 2


def _same_bits_or_nan(a, b):
    """Bit equality where both are numbers, NaN where either is (the payload of a NaN is not part of the contract)."""
    a, b = np.asarray(a), np.asarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and bits_equal(np.where(na, 0.0, a), np.where(nb, 0.0, b))


@pytest.mark.parametrize("n", [2, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 32769, 70000])
def test_unwind_small_batches_single_block(pj, O, dev, n):
    """Batches up to 8192 points run in one launch of one block (rounds of 4096 points with LDS carries), longer
    ones in the multi-kernel form: same bits either side of every boundary, out of place and in place, for smooth
    walks, a wrap at every step, exact half-period ties (serial recurrence inside the block) and NaN/Inf."""
    g = _identity_wcs(pj)
    rng = np.random.default_rng(4000 + n)
    k = np.arange(n, dtype=np.float64)
    cases = {
        "walk": np.cumsum(rng.normal(0, 1.5, (n, 2)), axis=0) + rng.uniform(-50, 50, 2),
        "jumps": rng.uniform(-400, 400, (n, 2)),
        "ties": np.stack([k * math.pi, -k * math.pi + 0.25], axis=1),
    }
    bad = rng.uniform(-20, 20, (n, 2))
    bad[n // 2, 0] = float("nan")
    bad[(2 * n) // 3, 1] = float("inf")
    cases["nonfinite"] = bad
    for name, a in cases.items():
        exp = O.pix2sky(g[1], a, O.WRAP_UNWIND)
        got = pj.pix2sky(g, to_dev(a, dev), safe=True).cpu().numpy()
        assert _same_bits_or_nan(got, exp), (name, n, "out of place")
        buf = to_dev(a, dev)
        pj.pix2sky_(g, buf, buf, safe=True)
        assert _same_bits_or_nan(buf.cpu().numpy(), exp), (name, n, "in place")
        for period, ref in ((2 * math.pi, 0.0), (5.0, 1.25)):
            col = a[:, 0].copy()
            assert _same_bits_or_nan(pj.unwind_(to_dev(col, dev), period, ref).cpu().numpy(), O.unwind_row(col, period, ref)), (name, n, period)
            two = pj.unwind_(to_dev(a, dev), period, ref).cpu().numpy()
            exp2 = np.stack([O.unwind_row(a[:, 0].copy(), period, ref), O.unwind_row(a[:, 1].copy(), period, ref)], axis=1)
            assert _same_bits_or_nan(two, exp2), (name, n, period, "2xN")


def test_gnomonic_pix2sky_arrays_on_device(pj, O, dev, literals):
    """pix2sky on device arrays of a Gnomonic map (pxl_pix2sky_tan_f64): against the oracle and as the inverse of
    the device sky2pix."""
    g = literals["gnomonic"]
    wcs = pj.Gnomonic(g["cdelt"], g["crpix"], g["crval"])
    shape = tuple(g["shape"])
    rng = np.random.default_rng(17)
    ip, jp = rng.uniform(1, shape[0], 40001), rng.uniform(1, shape[1], 40001)
    ra, dec = pj.pix2sky((shape, wcs), to_dev(ip, dev), to_dev(jp, dev))
    era, edec = O.pix2sky_tan(wcs, ip, jp)
    assert np.abs(ra.cpu().numpy() - era).max() < 1e-12 and np.abs(dec.cpu().numpy() - edec).max() < 1e-12
    x, y = pj.sky2pix((shape, wcs), ra, dec)
    assert np.abs(x.cpu().numpy() - ip).max() < 1e-6 and np.abs(y.cpu().numpy() - jp).max() < 1e-6


def test_gnomonic_wide_fields_on_device(pj, O, dev):
    """The evaluators' own atan2 / asin / sincos / rsqrt (pxl_fastmath.h) across their whole domain on the device: patches centred
    from pole to pole, points out to ~85 degrees from the centre (every octant of atan2, both halves of asin, several
    periods of the sincos reduction), against the oracle's glibc at a few ulp of the angle / 1e-9 of a pixel."""
    rng = np.random.default_rng(23)
    n = 200001
    for crval in ((0.0, 0.0), (97.5, -7.5), (-170.0, 45.0), (10.0, 89.9), (200.0, -89.0), (359.0, 60.0), (-720.5, -30.0)):
        wcs = pj.Gnomonic((-1.0 / 60, 1.0 / 60), (1000.5, 900.5), crval)
        shape = (2000, 1800)
        # plane radius r = tan(distance from the centre): up to tan(85 deg) = 11.4 rad = 39 300 pixels of 1 arcmin
        rad = np.tan(np.radians(rng.uniform(0.0, 85.0, n))) / np.radians(1.0 / 60)
        phi = rng.uniform(0.0, 2 * np.pi, n)
        phi[:64] = np.arange(64) * (np.pi / 32)                       # the octant boundaries
        ip, jp = 1000.5 + rad * np.cos(phi), 900.5 + rad * np.sin(phi)
        ra, dec = pj.pix2sky((shape, wcs), to_dev(ip, dev), to_dev(jp, dev), safe=False)
        era, edec = O.pix2sky_tan(wcs, ip, jp)
        # both sides carry ~1e-16 / cos(dec) of conditioning (the reference's form takes asin / atan2 of direction cosines)
        tol = 4e-16 * np.abs(era) + 2e-15 / np.maximum(np.cos(edec), 1e-6)
        assert (np.abs(ra.cpu().numpy() - era) < tol).all()
        assert (np.abs(dec.cpu().numpy() - edec) < tol).all()
        x, y = pj.sky2pix((shape, wcs), to_dev(era, dev), to_dev(edec, dev), safe=False)
        ex, ey = O.sky2pix_tan(wcs, era, edec)
        r = rad * np.radians(1.0 / 60)             # plane radius; d(pixel)/d(angle) grows like 1 + r^2 towards the horizon
        tol = 1e-9 + 2e-14 * rad * (1.0 + r * r)
        assert (np.abs(x.cpu().numpy() - ex) < tol).all() and (np.abs(y.cpu().numpy() - ey) < tol).all()
    # non-finite and absurd pixel coordinates: NaN in, NaN out; where the plane radius overflows or is infinite the device gives
    # NaN in DEC (documented in pxl_tan.h) or the oracle's number, never another finite one
    wcs = pj.Gnomonic((-1.0 / 60, 1.0 / 60), (1000.5, 900.5), (30.0, 20.0))
    ip = np.array([np.nan, 5.0, np.inf, -np.inf, 7.0, 1e200, -1e300, 3.0, np.nan, 1e15])
    jp = np.array([4.0, np.nan, 6.0, 8.0, np.inf, 2.0, 9.0, -1e250, np.nan, -1e15])
    ra, dec = pj.pix2sky(((2000, 1800), wcs), to_dev(ip, dev), to_dev(jp, dev), safe=False)
    era, edec = O.pix2sky_tan(wcs, ip, jp)
    ra, dec = ra.cpu().numpy(), dec.cpu().numpy()
    wild = ~(np.abs(jp) < 1e150) | ~(np.abs(ip) < 1e150)
    for got, exp in ((ra, era), (dec, edec)):
        assert np.isnan(got[np.isnan(exp)]).all(), (got, exp)
        ok = ~np.isnan(exp) & ~wild
        assert np.abs(got[ok] - exp[ok]).max() < 1e-9, (got, exp)
        w = ~np.isnan(exp) & wild
        assert (np.isnan(got[w]) | (np.abs(got[w] - exp[w]) < 1e-9)).all(), (got, exp)
    # angles many periods away from the centre, and beyond the fast sincos range (the library path)
    wcs = pj.Gnomonic((-1.0 / 60, 1.0 / 60), (1000.5, 900.5), (30.0, 20.0))
    base_ra, base_dec = np.radians(30.0) + rng.uniform(-0.3, 0.3, 4096), np.radians(20.0) + rng.uniform(-0.3, 0.3, 4096)
    for turns in (1, -3, 1000, 200000):
        a = base_ra + turns * 2 * np.pi
        x, y = pj.sky2pix(((2000, 1800), wcs), to_dev(a, dev), to_dev(base_dec, dev), safe=False)
        ex, ey = O.sky2pix_tan(wcs, a, base_dec)
        assert np.abs(x.cpu().numpy() - ex).max() < 1e-6 and np.abs(y.cpu().numpy() - ey).max() < 1e-6


def test_gnomonic_celestial_pole_is_finite_on_device(pj, O, dev):
    """the sine handed to asin is a product of three rounded factors and overshoots 1 by up to 4.4e-16 for a
    pixel ON a celestial pole (one patch centre in nine); pxl_fm_asin saturates that band to +-pi/2 instead of returning NaN.
    (1) sky2pix(ra, +-pi/2) -> pix2sky for many patch centres must give +-pi/2 back; (2) a posmap whose grid contains the pole
    pixel is finite, with the pole's declination at that pixel."""
    rng = np.random.default_rng(88)
    d0s = np.concatenate([rng.uniform(5.0, 89.999, 160), rng.uniform(-89.999, -5.0, 160), [90.0, -90.0, 89.99999, 45.0, 30.0, 60.0]])
    ras = to_dev(rng.uniform(-np.pi, np.pi, 64), dev)
    worst = 0.0
    for d0 in d0s:
        wcs = pj.Gnomonic((-1.0 / 60, 1.0 / 60), (1000.5, 900.5), (float(rng.uniform(-180, 180)), float(d0)))
        pole = math.copysign(math.pi / 2, d0)                 # the pole in front of this tangent plane
        x, y = pj.sky2pix(((2000, 1800), wcs), ras, torch.full_like(ras, pole), safe=False)
        ra, dec = pj.pix2sky(((2000, 1800), wcs), x, y, safe=False)
        dec = dec.cpu().numpy()
        assert np.isfinite(dec).all(), (d0, dec)
        assert np.isfinite(ra.cpu().numpy()).all(), d0
        worst = max(worst, np.abs(dec - pole).max())
    # asin near 1 has square-root conditioning: |ddec| <= sqrt(2 eps k) for a sine that is k ulp off
    assert worst < 1e-7, worst
    for d0, res_arcmin in ((88.0, 1.0), (-89.5, 0.5), (75.0, 4.0)):
        wcs0 = pj.Gnomonic((-res_arcmin / 60, res_arcmin / 60), (256.5, 256.5), (33.0, d0))
        pole = math.copysign(math.pi / 2, d0)
        xp, yp = O.sky2pix_tan(wcs0, np.array([0.0]), np.array([pole]))
        # move the reference pixel so that the pole lands exactly on a pixel centre inside the grid
        wcs = pj.Gnomonic(wcs0.cdelt, (256.5 + (round(xp[0]) - xp[0]), 256.5 + (round(yp[0]) - yp[0])), (33.0, d0))
        ip, jp = int(round(xp[0])), int(round(yp[0]))
        shape = (max(512, ip + 8), max(512, jp + 8))
        assert 1 <= ip <= shape[0] and 1 <= jp <= shape[1], (ip, jp)
        ra, dec = pj.posmap(shape, wcs, device=dev)
        ra, dec = ra.data.cpu().numpy(), dec.data.cpu().numpy()
        assert np.isfinite(ra).all() and np.isfinite(dec).all(), d0
        assert abs(dec[jp - 1, ip - 1] - pole) < 1e-7, (d0, dec[jp - 1, ip - 1])
        jj, ii = np.meshgrid(np.arange(1, shape[1] + 1, dtype=float), np.arange(1, shape[0] + 1, dtype=float), indexing="ij")
        era, edec = O.pix2sky_tan(wcs, ii.ravel(), jj.ravel())
        ok = np.isfinite(edec)
        assert np.abs(dec.ravel()[ok] - edec[ok]).max() < 1e-7


def _tan_pix2sky_longdouble(wcs, ii, jj):
    """tan_proj.jl:59-75 operation for operation in numpy long double (x87 80-bit here): the yardstick for the polar-patch test."""
    L = np.longdouble
    pi = L("3.14159265358979323846264338327950288")
    unit, scale = L(wcs.unit), L(1.0) / L(wcs.cdelt[0])
    a0, d0 = L(wcs.crval[0]) * (pi / 180), L(wcs.crval[1]) * (pi / 180)
    X = (L(wcs.crpix[0]) - ii.astype(L)) * unit / scale
    Y = (L(wcs.crpix[1]) - jj.astype(L)) * unit / scale
    D = np.arctan(np.sqrt(X * X + Y * Y))
    B = np.arctan2(-X, Y)
    XX = np.sin(d0) * np.sin(D) * np.cos(B) + np.cos(d0) * np.cos(D)
    YY = np.sin(D) * np.sin(B)
    return a0 + np.arctan2(YY, XX), np.arcsin(np.sin(d0) * np.cos(D) - np.cos(d0) * np.sin(D) * np.cos(B))


def test_gnomonic_polar_patch_posmap_l1_on_device(pj, O, dev):
    """the reference's bar for its Gnomonic code is an L1 sum over a whole posmap, sum|diff| < 1e-9 for the
    3 334 275 pixels of its 1827 x 1825 patch (test_geometry.jl:116-119), i.e. 3.0e-16 per pixel.  Evaluated here on POLAR patches
    (centres at |dec| >= 88 degrees, the pole inside the grid), scaled by the pixel count.  On such a patch that bar cannot hold for
    ANY double-precision evaluation, the reference's own formula included: dec = asin(s) has d(dec)/ds = 1 / cos(dec) and RA is the
    longitude of a point at distance cos(dec) from the axis, so one ulp of the intermediate direction cosine is eps / cos(dec) of
    angle, and cos(dec) < 0.1 on the whole patch (measured on the MI355X, 1024^2 at 0.5' centred on dec 88.39: the ORACLE -- glibc, the
    reference's operation order in double -- is 1.43e-9 from the long-double value of the same formula in L1(dec) against a bar of
    3.1e-10; the device 1.85e-9; in RA as a distance on the sky, dRA cos(dec), 6.5e-11 both).  So the yardstick is the reference's
    formula in long double, and the device's L1 error against it may exceed the oracle's by half of the oracle's plus the
    reference's bar, for DEC and for RA cos(dec): the same conditioning, the same class of result, no NaN anywhere."""
    if np.finfo(np.longdouble).eps > 2e-19:
        pytest.skip("long double is not wider than double here")
    per_pixel = 1e-9 / (1827 * 1825)
    for crval, res_arcmin, n in (((40.0, 88.39), 0.5, 1024), ((-120.0, -89.2), 1.0, 768), ((0.0, 90.0), 0.5, 512)):
        wcs = pj.Gnomonic((-res_arcmin / 60, res_arcmin / 60), (n / 2 + 0.5, n / 2 + 0.5), crval)
        ra, dec = pj.posmap((n, n), wcs, device=dev)
        ra, dec = ra.data.cpu().numpy().ravel(), dec.data.cpu().numpy().ravel()
        assert np.isfinite(ra).all() and np.isfinite(dec).all()
        jj, ii = np.meshgrid(np.arange(1, n + 1, dtype=float), np.arange(1, n + 1, dtype=float), indexing="ij")
        ii, jj = ii.ravel(), jj.ravel()
        era, edec = O.pix2sky_tan(wcs, ii, jj)
        tra, tdec = _tan_pix2sky_longdouble(wcs, ii, jj)
        cosd = np.cos(tdec).astype(float)

        def l1(a, d):
            dra = np.abs((a - tra).astype(float))
            dra = np.minimum(dra, np.abs(dra - 2 * np.pi))    # atan2's branch cut passes through the pole
            return (dra * cosd).sum(), np.abs((d - tdec).astype(float)).sum()
        bound = per_pixel * n * n
        dev_ra, dev_dec = l1(ra, dec)
        ora_ra, ora_dec = l1(era, edec)
        print("polar patch %s: device L1 ra*cos(dec) %.3e dec %.3e | oracle %.3e %.3e | reference's bar %.3e" % (crval, dev_ra, dev_dec, ora_ra, ora_dec, bound))
        assert dev_ra < 1.5 * ora_ra + bound and dev_dec < 1.5 * ora_dec + bound, (crval, dev_ra, ora_ra, dev_dec, ora_dec, bound)


@pytest.mark.parametrize("f32", [False, True])
def test_reproject_every_tile_shape(pj, O, dev, f32, monkeypatch):
    """The LDS-DMA kernel is instantiated per (storage type, lane width, 16-byte chunks per source-row segment):
    sweep lane widths x scale factors so that every instantiation the plan can pick runs against the oracle
    (tools/fuzz_parity.py reaches them at random; this makes it deterministic)."""
    rng = np.random.default_rng(321)
    nx_in, ny_in = 3072, 24
    wcs_in = pj.CarClenshawCurtis((-360.0 / nx_in, 2.0), (nx_in / 2 + 0.5, 12.0), (0.3, 0.0))
    src = rng.normal(size=(2, ny_in, nx_in))
    src = src.astype(np.float32) if f32 else src
    d_src = torch.from_numpy(src).to(dev)
    for pairs in (1, 2, 4):
        monkeypatch.setenv("PXL_REPROJECT_PAIRS", str(pairs))
        for sx in (0.2, 0.45, 0.7, 0.95, 1.2, 1.45, 1.9, 2.4, 2.9, 3.4, 3.9, 4.4, 4.9):
            nxo = max(64, int(round(nx_in / sx)) & ~1)
            wcs_out = pj.CarClenshawCurtis((-360.0 / nxo, 2.0 * rng.uniform(0.9, 1.1)), (nxo / 2 + 0.25, 12.3), (0.3, 0.0))
            shape_out = (nxo, 20)
            plan = pj.ReprojectPlan((nx_in, ny_in, 2), wcs_in, shape_out, wcs_out, device=dev)
            dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=d_src.dtype, device=dev)
            plan.execute(d_src, dst)
            got = dst.cpu().numpy()
            if f32:
                assert _f32_equal(got, O.reproject_f32(wcs_in, (nx_in, ny_in, 2), src, wcs_out, shape_out)), (pairs, sx)
            else:
                assert bits_equal(got, O.reproject(wcs_in, (nx_in, ny_in, 2), src, wcs_out, shape_out)), (pairs, sx)
            plan.close()


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 100001])
def test_soa_forms_unaligned_and_odd(pj, O, dev, n):
    """The SoA kernels take two points per lane with 16-byte accesses when all four arrays allow it; vectors that
    start 8 bytes off (views into a larger tensor) and odd lengths take the scalar accesses -- same bits either way."""
    shape, wcs = pj.fullsky_geometry(1 * DEG)
    rng = np.random.default_rng(n)
    a, b = rng.uniform(-400, 400, n + 1), rng.uniform(-200, 200, n + 1)
    for off in (0, 1):
        da, db = to_dev(a, dev)[off:off + n], to_dev(b, dev)[off:off + n]
        assert da.is_contiguous() and (da.data_ptr() % 16 == 0) == (off == 0)
        for safe in (True, False):
            ra, dec = pj.pix2sky((shape, wcs), da, db, safe=safe)
            era, edec = O.pix2sky_soa(wcs, a[off:off + n], b[off:off + n], safe=safe)
            assert bits_equal(ra.cpu().numpy(), era) and bits_equal(dec.cpu().numpy(), edec), (n, off, safe)
            sa, sb = to_dev(a * 0.01, dev)[off:off + n], to_dev(b * 0.01, dev)[off:off + n]
            x, y = pj.sky2pix((shape, wcs), sa, sb, safe=safe)
            ex, ey = O.sky2pix_soa(wcs, shape, a[off:off + n] * 0.01, b[off:off + n] * 0.01, safe=safe, form=O.FORM_RECIP_AV)
            assert bits_equal(x.cpu().numpy(), ex) and bits_equal(y.cpu().numpy(), ey), (n, off, safe)
