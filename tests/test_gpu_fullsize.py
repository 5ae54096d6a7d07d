"""Parity at BASELINE.json's full sizes, where the CPU oracle cannot finish a whole map in seconds:
(1) sampled output rows against the oracle fed the same source rows (bit-identical), and
(2) size-independent properties -- identity, linearity, partition of unity, half-pixel-shift = neighbour
mean across the RA seam, pix->sky->pix round trips within 1e-9 pixel, range invariants."""
import math

import numpy as np
import pytest

from conftest import DEG, bits_equal

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import pixell_jl_amd as pj
    pj.load_library()
    return torch.device("cuda:0")


def _rows_vs_oracle(pj, O, src, dst, shape_in, wcs_in, shape_out, wcs_out, rows):
    nx, ny, nc = shape_in
    for r in rows:
        s_lo, s_hi = O.reproject_src_rows(wcs_in, shape_in, wcs_out, shape_out, r, 1)
        s = src[:, s_lo:s_hi, :].cpu().numpy()
        exp = O.reproject(wcs_in, shape_in, s, wcs_out, shape_out, src_row0=s_lo, src_nrows=s_hi - s_lo,
                          dst_row0=r, dst_nrows=1)
        got = dst[:, r:r + 1, :].cpu().numpy()
        assert np.abs(got - exp).max() <= 1e-10, r
        assert bits_equal(got, exp), r


def test_config3_refine2x_fullsize(pj, O, dev):
    """21600x10801 -> 43200x21601 (BASELINE config 3)."""
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 21600)
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 43200)
    assert shape_in == (21600, 10801) and shape_out == (43200, 21601)
    nx, ny = shape_in
    src = torch.empty((1, ny, nx), dtype=torch.float64, device=dev)
    pj.fill_random_(src, 1234)
    plan = pj.ReprojectPlan((nx, ny, 1), wcs_in, shape_out, wcs_out, device=dev)
    dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
    plan.execute(src, dst)
    assert bool(torch.isfinite(dst).all())
    _rows_vs_oracle(pj, O, src, dst, (nx, ny, 1), wcs_in, shape_out, wcs_out,
                    [0, 1, 2, 31, 32, 33, 10800, 10801, 21599, 21600])
    # output pixel (2i-1, 2j-1) coincides with source pixel (i, j): exact copy there up to rounding of x, y
    sub = dst[0, 0::2, 0::2]
    assert float((sub - src[0]).abs().max()) < 1e-9
    # linearity (exact for a power-of-two scale) and partition of unity
    dst2 = torch.empty_like(dst)
    plan.execute(src * 2.0, dst2)
    assert torch.equal(dst2, dst * 2.0)
    plan.execute(torch.ones_like(src), dst2)
    assert float((dst2 - 1.0).abs().max()) < 1e-13
    del dst2
    # cross-check variants on the full map: LDS-DMA kernel == register-staged kernel == direct gather
    for variant in (2, 1):
        plan.set_variant(variant)
        other = torch.empty_like(dst)
        plan.execute(src, other)
        assert torch.equal(other, dst), variant
        del other


def test_config4_shifted_iqu_fullsize(pj, O, dev):
    """43200x21601x3 -> same shape, half-pixel-shifted WCS (BASELINE config 4, the bench workload)."""
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 43200, dims=(3,))
    nx, ny, nc = shape_in
    wcs_out = pj.CarClenshawCurtis(wcs_in.cdelt, (wcs_in.crpix[0] + 0.5, wcs_in.crpix[1] + 0.5), wcs_in.crval)
    shape_out = (nx, ny)
    src = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
    for c in range(nc):
        pj.fill_random_(src[c], 1234 + c)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
    plan.execute(src, dst)
    assert bool(torch.isfinite(dst).all())
    _rows_vs_oracle(pj, O, src, dst, shape_in, wcs_in, shape_out, wcs_out, [0, 1, 31, 32, 10800, 21599, 21600])
    # output pixel (i, j) sits at source (i - 0.5, j - 0.5): the mean of the 2x2 block, RA wrapping at the
    # seam (pixel 0 <-> pixel nx), rows below the map reading as zero
    left = torch.roll(src, 1, dims=2)
    blk = 0.5 * (0.5 * left[:, 1:, :] + 0.5 * src[:, 1:, :]) + 0.5 * (0.5 * left[:, :-1, :] + 0.5 * src[:, :-1, :])
    assert float((dst[:, 1:, :] - blk).abs().max()) < 1e-12
    del left, blk


def test_config1_roundtrip_and_posmap_fullsize(pj, dev):
    """pix2sky -> sky2pix round trip over EVERY pixel of the 1024x513 map (config 1) and of the 0.5-arcmin
    map (config 4 geometry): <= 1e-9 pixel, plus the range invariants of test_geometry.jl:213-222."""
    for nx in (1024, 43200):
        shape, wcs = pj.fullsky_geometry(2 * math.pi / nx)
        ra, dec = pj.posmap(shape, wcs, device=dev)
        assert float(ra.data.min()) >= -math.pi and float(ra.data.max()) <= math.pi
        assert float(dec.data.min()) >= -math.pi / 2 and float(dec.data.max()) <= math.pi / 2
        # on-sky declinations are only touched by rewind's own rounding (one ulp of pi)
        ra_u, dec_u = pj.posmap(shape, wcs, device=dev, safe=False)
        assert float((dec.data - dec_u.data).abs().max()) <= 1e-15
        step = 1 if nx == 1024 else 16            # every pixel / every 16th row of the big map
        sky = torch.stack([ra.data[::step].reshape(-1), dec.data[::step].reshape(-1)], dim=1).contiguous()
        pix = pj.sky2pix((shape, wcs), sky, safe=True)
        jj, ii = torch.meshgrid(torch.arange(1, shape[1] + 1, step, dtype=torch.float64, device=dev),
                                torch.arange(1, shape[0] + 1, dtype=torch.float64, device=dev), indexing="ij")
        assert float((pix[:, 0] - ii.reshape(-1)).abs().max()) <= 1e-9
        assert float((pix[:, 1] - jj.reshape(-1)).abs().max()) <= 1e-9
        assert float(pix[:, 0].min()) >= 1 and float(pix[:, 0].max()) <= shape[0]
        del ra, dec, ra_u, dec_u, sky, pix, ii, jj


def test_config5_scattered_sample_properties(pj, O, dev):
    """1.25e8 uniform-on-sphere points (config 5's share of one of 8 GPUs: a 2 GB coordinate batch) from the
    0.5-arcmin map: seeded subsets from both ends of the batch against the oracle through BOTH samplers (direct
    gather and row-pair copy), all points equal between the two, sample of a constant map is that constant."""
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
    nx, ny = shape
    m = pj.Enmap(torch.empty((ny, nx), dtype=torch.float64, device=dev), wcs)
    pj.fill_random_(m.data, 1234)
    n = 125_000_000
    sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
    pj.fill_sphere_points_(sky, 42)
    out = pj.sample_bilinear(m, sky)
    assert bool(torch.isfinite(out).all())
    # the row-pair copy of the same map (20 GB: offsets beyond 2^31 bytes) gives the same bits for every point
    pairs = pj.SamplePairs(m)
    out_pairs = pj.sample_bilinear(None, sky, pairs=pairs)
    assert torch.equal(out_pairs, out)
    del pairs
    # subsets vs oracle: the oracle only needs the rows those points touch -> take points in a DEC band, from the first
    # and the last 200 000 points of the batch (the tail sits beyond 2^31 bytes of the coordinate buffer)
    rows = m.data[11999:12041].cpu().numpy()[None]
    for lo in (0, n - 200000):
        pix = pj.sky2pix(m, sky[lo:lo + 200000], safe=True)
        band = (pix[:, 1] >= 12000) & (pix[:, 1] < 12040)
        idx = torch.nonzero(band).reshape(-1) + lo
        assert idx.numel() > 100
        exp = O.sample_bilinear(wcs, (nx, ny, 1), rows, sky[idx].cpu().numpy(), src_row0=11999, src_nrows=42)
        assert bits_equal(out[:, idx].cpu().numpy(), exp)
        assert bits_equal(out_pairs[:, idx].cpu().numpy(), exp)
    del out_pairs
    m.data.fill_(3.25)
    assert float((pj.sample_bilinear(m, sky) - 3.25).abs().max()) < 1e-12


def test_getindex_on_a_map_beyond_2GiB(pj, dev):
    """Enmap.getindex (copy semantics, negative and non-unit steps) on a 7.5 GB map: checked element-wise
    against arithmetic on the known content m[j, i] = j * nx + i (exact in Float64)."""
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
    nx, ny = shape
    data = (torch.arange(ny, dtype=torch.float64, device=dev).mul_(nx).reshape(ny, 1)
            + torch.arange(nx, dtype=torch.float64, device=dev).reshape(1, nx))
    m = pj.Enmap(data, wcs)
    g = m.getindex((40000, -7, 3), (21000, -3, 12))            # both axes backwards with steps
    xs = torch.arange(40000, 2, -7, dtype=torch.float64, device=dev)          # 1-based columns
    ys = torch.arange(21000, 11, -3, dtype=torch.float64, device=dev)
    assert g.shape == (xs.numel(), ys.numel())
    expect = (ys.reshape(-1, 1) - 1) * nx + (xs.reshape(1, -1) - 1)
    assert torch.equal(g.data, expect)
    big = m.getindex((2, 43199), (21601, -1, 1))                               # 7.5 GB result, DEC reversed
    assert big.shape == (43198, 21601)
    assert float(big.data[0, 0]) == (ny - 1) * nx + 1 and float(big.data[-1, -1]) == nx - 2
    assert float(big.data[12345, 23456]) == (ny - 1 - 12345) * nx + 23457
    s_expect = float(data[:, 1:-1].sum())
    assert abs(float(big.data.sum()) - s_expect) <= 1e-9 * abs(s_expect)


def test_config4_float32_storage_fullsize(pj, dev):
    """The config-4 geometry with Float32 maps (11.2 GB per map: plane offsets beyond 2^31 elements in the Float32
    kernels too): the Float32 path must equal the Float64 path on the widened input, rounded once -- every value of
    the 2.8e9-element output is compared."""
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 43200, dims=(3,))
    nx, ny, nc = shape_in
    wcs_out = pj.CarClenshawCurtis(wcs_in.cdelt, (wcs_in.crpix[0] + 0.5, wcs_in.crpix[1] + 0.5), wcs_in.crval)
    src64 = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
    pj.fill_random_(src64, 99)
    src32 = src64.float()
    src64.copy_(src32)                                   # the widened Float32 values, exactly
    plan = pj.ReprojectPlan(shape_in, wcs_in, (nx, ny), wcs_out, device=dev)
    dst32 = torch.empty((nc, ny, nx), dtype=torch.float32, device=dev)
    plan.execute(src32, dst32)
    del src32
    dst64 = torch.empty((nc, ny, nx), dtype=torch.float64, device=dev)
    plan.execute(src64, dst64)
    del src64
    for c in range(nc):                                  # plane by plane keeps the temporaries small
        assert torch.equal(dst32[c], dst64[c].float()), c
    plan.close()


def test_config3_same_resolution_shift_fullsize(pj, O, dev):
    """21600x10801 -> same shape, half-pixel-shifted WCS (second workload; bench `cfg3s`)."""
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 21600)
    nx, ny = shape_in
    wcs_out = pj.CarClenshawCurtis(wcs_in.cdelt, (wcs_in.crpix[0] + 0.5, wcs_in.crpix[1] + 0.5), wcs_in.crval)
    src = torch.empty((1, ny, nx), dtype=torch.float64, device=dev)
    pj.fill_random_(src, 99)
    plan = pj.ReprojectPlan((nx, ny, 1), wcs_in, (nx, ny), wcs_out, device=dev)
    dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
    plan.execute(src, dst)
    assert bool(torch.isfinite(dst).all())
    _rows_vs_oracle(pj, O, src, dst, (nx, ny, 1), wcs_in, (nx, ny), wcs_out, [0, 1, 15, 16, 17, 5400, 10799, 10800])
    left = torch.roll(src, 1, dims=2)
    blk = 0.5 * (0.5 * left[:, 1:, :] + 0.5 * src[:, 1:, :]) + 0.5 * (0.5 * left[:, :-1, :] + 0.5 * src[:, :-1, :])
    assert float((dst[:, 1:, :] - blk).abs().max()) < 1e-12
    del left, blk
    for variant in (2, 1):
        plan.set_variant(variant)
        other = torch.empty_like(dst)
        plan.execute(src, other)
        assert torch.equal(other, dst), variant
        del other


@pytest.mark.parametrize("workload", ["cfg3", "cfg3s"])
def test_strips_of_config3_fullsize(pj, dev, workload):
    """What a rank of a sharded config-3 job launches: the 1/8 and 1/4 declination strips (window plans, interior rows first,
    boundary rows after the halo) of the full-size maps reproduce the unsharded launch bit for bit.  These launches are small
    enough for the plan's tile-height floor to matter (short tiles below 8 192 tiles per launch)."""
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 21600)
    nx, ny = shape_in
    if workload == "cfg3":
        shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 43200)
    else:
        shape_out, wcs_out = (nx, ny), pj.CarClenshawCurtis(wcs_in.cdelt, (wcs_in.crpix[0] + 0.5, wcs_in.crpix[1] + 0.5), wcs_in.crval)
    full_src = torch.empty((1, ny, nx), dtype=torch.float64, device=dev)
    pj.fill_random_(full_src, 4321)
    full_plan = pj.ReprojectPlan((nx, ny, 1), wcs_in, shape_out, wcs_out, device=dev)
    full_dst = torch.empty(full_plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
    full_plan.execute(full_src, full_dst)
    for rank, world in ((0, 8), (3, 8), (7, 8), (1, 4), (2, 3)):
        L = pj.sharding.DecStripLayout((nx, ny, 1), wcs_in, shape_out, wcs_out, rank, world)
        plan = pj.ReprojectPlan((nx, ny, 1), wcs_in, shape_out, wcs_out, src_rows=L.src_window, dst_rows=L.dst_window, device=dev)
        src = torch.full(L.src_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
        src[:, L.own_slice(), :] = full_src[:, L.own[rank][0]:L.own[rank][1], :]
        dst = torch.full(L.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
        plan.build_tables()
        i_lo, i_hi = L.interior
        plan.execute_rows(src, dst, i_lo, i_hi - i_lo)
        for _, lo, hi in L.recvs:
            src[:, lo - L.buf_lo:hi - L.buf_lo, :] = full_src[:, lo:hi, :]
        if i_lo > 0:
            plan.execute_rows(src, dst, 0, i_lo)
        if i_hi < L.dst_window[1]:
            plan.execute_rows(src, dst, i_hi, L.dst_window[1] - i_hi)
        lo, n = L.dst_window
        assert torch.equal(dst, full_dst[:, lo:lo + n, :]), (workload, rank, world)
        del src, dst, plan


@pytest.mark.parametrize("res_in,res_out", [(10800, 43200), (43200, 21600), (43200, 10800), (5400, 43200)])
def test_other_scale_factors_fullsize(pj, O, dev, res_in, res_out):
    """4x / 8x refinement and 2x / 4x coarsening onto or from the 0.5-arcmin grid: the plan picks other tile heights and prefetch
    distances for them (`k_reproject_dma` bullets); sampled rows against the oracle, every kernel variant identical."""
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / res_in)
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / res_out)
    nx, ny = shape_in
    src = torch.empty((1, ny, nx), dtype=torch.float64, device=dev)
    pj.fill_random_(src, res_in + res_out)
    plan = pj.ReprojectPlan((nx, ny, 1), wcs_in, shape_out, wcs_out, device=dev)
    dst = torch.full(plan.dst_tensor_shape(), float("nan"), dtype=torch.float64, device=dev)
    plan.execute(src, dst)
    assert bool(torch.isfinite(dst).all())
    nyo = shape_out[1]
    _rows_vs_oracle(pj, O, src, dst, (nx, ny, 1), wcs_in, shape_out, wcs_out, [0, 1, 3, 4, 7, 8, 9, nyo // 2, nyo - 2, nyo - 1])
    for variant in (2, 1):
        plan.set_variant(variant)
        other = torch.empty_like(dst)
        plan.execute(src, other)
        assert torch.equal(other, dst), variant
        del other
