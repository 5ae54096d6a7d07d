"""Property-based checks (hypothesis) of the host logic and the oracle: invariants the reference's tests only
probe at a few points (rewind range, slice/pixel consistency, Julia range normalisation, round trips)."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

import pixell_jl_amd as pj
from oracle import oracle as O

finite = st.floats(min_value=-1e6, max_value=1e6, allow_nan=False, allow_infinity=False)


@settings(max_examples=300, deadline=None)
@given(x=finite, period=st.floats(min_value=1e-3, max_value=1e4), ref=st.floats(min_value=-1e3, max_value=1e3))
def test_rewind_lands_in_its_window_and_is_congruent(x, period, ref):
    r = pj.rewind(x, period, ref)
    assert r == O.rewind(x, period, ref)                       # host mirror == oracle, bit for bit
    assert ref - period / 2 - 1e-9 * period <= r <= ref + period / 2 + 1e-9 * period
    k = (x - r) / period
    assert abs(k - round(k)) < 1e-6 * max(1.0, abs(k))


@settings(max_examples=300, deadline=None)
@given(first=st.integers(1, 400), stop=st.integers(1, 400), step=st.integers(-9, 9).filter(lambda s: s != 0))
def test_jlrange_matches_python_range(first, stop, step):
    r = pj.JlRange(first, stop, step)
    ref = list(range(first, stop + (1 if step > 0 else -1), step))
    assert r.length == len(ref)
    if ref:
        assert r.last == ref[-1]
        assert list(np.arange(1, 401)[r.to_slice()]) == ref


@settings(max_examples=200, deadline=None)
@given(fx=st.integers(1, 300), nx=st.integers(1, 50), sx=st.integers(-4, 4).filter(lambda s: s != 0),
       fy=st.integers(1, 150), ny=st.integers(1, 30), sy=st.integers(-3, 3).filter(lambda s: s != 0),
       k=st.integers(0, 49), l=st.integers(0, 29))
def test_slice_geometry_keeps_pixels_on_the_sky(fx, nx, sx, fy, ny, sy, k, l):
    """The reference's slicing convention (enmap_ops.jl:154-167): a strided slice is a coarser grid whose pixel
    (k+1, l+1) is centred on the BLOCK of |step| parent pixels starting at first + k*step, i.e. at parent
    coordinate first + k*step + (step - sign(step))/2; for unit steps that is the parent pixel itself."""
    shape0, w0 = pj.fullsky_geometry(math.radians(1))
    lx, ly = fx + (nx - 1) * sx, fy + (ny - 1) * sy
    if not (1 <= lx <= shape0[0] and 1 <= ly <= shape0[1] and fx <= shape0[0] and fy <= shape0[1]):
        return
    shape, w = pj.slice_geometry(shape0, w0, (fx, sx, lx), (fy, sy, ly))
    assert shape == (nx, ny)
    k, l = k % nx, l % ny
    a, d = pj.pix2sky((shape, w), float(k + 1), float(l + 1), safe=False)
    cx = fx + k * sx + (sx - (1 if sx > 0 else -1)) / 2
    cy = fy + l * sy + (sy - (1 if sy > 0 else -1)) / 2
    a0, d0 = pj.pix2sky((shape0, w0), float(cx), float(cy), safe=False)
    assert abs(a - a0) < 1e-12 and abs(d - d0) < 1e-12


@settings(max_examples=200, deadline=None)
@given(i=st.floats(-500, 900), j=st.floats(-300, 500), nx=st.sampled_from([360, 720, 1024, 43200]))
def test_pix_sky_pix_roundtrip_modulo_period(i, j, nx):
    shape, w = pj.fullsky_geometry(2 * math.pi / nx)
    a, d = pj.pix2sky((shape, w), i, j, safe=True)
    x, y = pj.sky2pix((shape, w), a, d, safe=True)
    per_x, per_y = nx, 2 * (shape[1] - 1)
    assert abs(((x - i + per_x / 2) % per_x) - per_x / 2) < 1e-6
    assert abs(((y - j + per_y / 2) % per_y) - per_y / 2) < 1e-6
    assert shape[0] / 2 + 1 - per_x / 2 - 1e-6 <= x <= shape[0] / 2 + 1 + per_x / 2 + 1e-6


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10**6), n=st.integers(1, 400), scale=st.floats(0.1, 50.0))
def test_unwind_matches_sequential_definition(seed, n, scale):
    rng = np.random.default_rng(seed)
    a = np.cumsum(rng.normal(0, scale, n))
    got = O.unwind_row(a)
    prev = None
    for k in range(n):
        m = O.rewind(a[k])
        y = m if prev is None else m - np.rint((m - prev) / (2 * math.pi)) * (2 * math.pi)
        assert got[k] == y
        prev = y
