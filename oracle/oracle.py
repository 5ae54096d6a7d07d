"""ctypes loader for the CPU oracle (oracle/pixell_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (pixell.jl_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

WRAP_NONE, WRAP_REWIND, WRAP_UNWIND = 0, 1, 2
FORM_RECIP, FORM_DIV, FORM_RECIP_AV = 0, 1, 2


class CarWCS(C.Structure):
    _fields_ = [("cdelt", C.c_double * 2), ("crpix", C.c_double * 2), ("crval", C.c_double * 2),
                ("unit", C.c_double)]


def build():
    """Compile liboracle.so with gcc (committed recipe: oracle/Makefile)."""
    subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or \
                os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pixell_oracle.c")):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.pxl_jl_mod.restype = C.c_double
        _lib.pxl_jl_mod.argtypes = [C.c_double, C.c_double]
        _lib.pxl_rewind_cpu.restype = C.c_double
        _lib.pxl_rewind_cpu.argtypes = [C.c_double] * 3
        _lib.pxl_skyarea_cyl_cpu.restype = C.c_double
    return _lib


def _w(wcs):
    """Accept anything with cdelt/crpix/crval/unit (tuple-likes) and return a CarWCS."""
    if isinstance(wcs, CarWCS):
        return wcs
    out = CarWCS()
    out.cdelt[:] = [float(v) for v in wcs.cdelt]
    out.crpix[:] = [float(v) for v in wcs.crpix]
    out.crval[:] = [float(v) for v in wcs.crval]
    out.unit = float(wcs.unit)
    return out


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i64(vals):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def set_threads(n):
    lib().pxl_oracle_set_threads(int(n))


def max_threads():
    return lib().pxl_oracle_max_threads()


def jl_mod(x, y):
    return lib().pxl_jl_mod(float(x), float(y))


def rewind(a, period=2 * np.pi, ref=0.0):
    return lib().pxl_rewind_cpu(float(a), float(period), float(ref))


def unwind_row(a, period=2 * np.pi, ref=0.0):
    a = _f64(a).copy()
    lib().pxl_unwind_row_cpu(C.c_int64(a.size), C.c_int64(1), _dp(a), C.c_double(period), C.c_double(ref))
    return a


def pix2sky(wcs, pix, wrap_mode=WRAP_UNWIND, out=None):
    """pix: (n, 2) array of (ra_pix, dec_pix) pairs == Julia 2xN column-major.  Returns (n, 2).
    out: a caller-allocated (n, 2) result array (bench.py's cpu_baseline touches it before timing)."""
    pix = _f64(pix)
    sky = np.empty_like(pix) if out is None else out
    assert sky.shape == pix.shape and sky.dtype == np.float64 and sky.flags.c_contiguous
    rc = lib().pxl_pix2sky_car_f64_cpu(C.byref(_w(wcs)), C.c_int64(pix.shape[0]), _dp(pix), _dp(sky),
                                       C.c_int(wrap_mode))
    assert rc == 0
    return sky


def pix2sky_soa(wcs, ipix, jpix, safe=True):
    ipix, jpix = _f64(ipix), _f64(jpix)
    ra, dec = np.empty_like(ipix), np.empty_like(jpix)
    rc = lib().pxl_pix2sky_car_soa_f64_cpu(C.byref(_w(wcs)), C.c_int64(ipix.size), _dp(ipix), _dp(jpix),
                                           _dp(ra), _dp(dec), C.c_int(bool(safe)))
    assert rc == 0
    return ra, dec


def sky2pix(wcs, shape, sky, safe=True, form=FORM_RECIP, out=None):
    sky = _f64(sky)
    pix = np.empty_like(sky) if out is None else out
    assert pix.shape == sky.shape and pix.dtype == np.float64 and pix.flags.c_contiguous
    rc = lib().pxl_sky2pix_car_f64_cpu(C.byref(_w(wcs)), _i64(shape[:2]), C.c_int64(sky.shape[0]), _dp(sky),
                                       _dp(pix), C.c_int(bool(safe)), C.c_int(form))
    assert rc == 0
    return pix


def sky2pix_soa(wcs, shape, ra, dec, safe=True, form=FORM_RECIP_AV):
    ra, dec = _f64(ra), _f64(dec)
    x, y = np.empty_like(ra), np.empty_like(dec)
    rc = lib().pxl_sky2pix_car_soa_f64_cpu(C.byref(_w(wcs)), _i64(shape[:2]), C.c_int64(ra.size), _dp(ra),
                                           _dp(dec), _dp(x), _dp(y), C.c_int(bool(safe)), C.c_int(form))
    assert rc == 0
    return x, y


def posmap(wcs, shape, row0=0, nrows=None, safe=True, out=None):
    """Returns (ra, dec) arrays of shape (nrows, nx) (C order == Julia (nx, nrows) column-major).  out: caller-allocated (ra, dec)."""
    nx, ny = int(shape[0]), int(shape[1])
    nrows = ny - row0 if nrows is None else nrows
    if out is None:
        ra = np.empty((nrows, nx)); dec = np.empty((nrows, nx))
    else:
        ra, dec = out
        assert ra.shape == (nrows, nx) and dec.shape == (nrows, nx) and ra.flags.c_contiguous and dec.flags.c_contiguous
    rc = lib().pxl_posmap_car_f64_cpu(C.byref(_w(wcs)), _i64((nx, ny)), C.c_int64(row0), C.c_int64(nrows),
                                      _dp(ra), _dp(dec), C.c_int(bool(safe)))
    assert rc == 0
    return ra, dec


def fullsky_geometry(res, shape=None):
    resx, resy = (res, res) if np.isscalar(res) else res
    sh = _i64(shape if shape is not None else (0, 0))
    w = CarWCS()
    rc = lib().pxl_fullsky_geometry_cpu(C.c_double(resx), C.c_double(resy), sh, C.byref(w))
    if rc != 0:
        raise AssertionError("resolution does not evenly divide the sky (rc=%d)" % rc)
    return (int(sh[0]), int(sh[1])), w


def geometry(pos1, pos2, res):
    resx, resy = (res, res) if np.isscalar(res) else res
    p1 = (C.c_double * 2)(*pos1); p2 = (C.c_double * 2)(*pos2)
    sh = _i64((0, 0)); w = CarWCS()
    rc = lib().pxl_geometry_cpu(p1, p2, C.c_double(resx), C.c_double(resy), sh, C.byref(w))
    if rc != 0:
        raise AssertionError("resolution does not evenly divide the sky (rc=%d)" % rc)
    return (int(sh[0]), int(sh[1])), w


def slice_geometry(wcs, first, step, last):
    sh = _i64((0, 0)); w = CarWCS()
    rc = lib().pxl_slice_geometry_cpu(C.byref(_w(wcs)), _i64(first), _i64(step), _i64(last), sh, C.byref(w))
    assert rc == 0
    return (int(sh[0]), int(sh[1])), w


def pixarea_rows(wcs, nrows):
    out = np.empty(nrows)
    lib().pxl_pixarea_rows_cpu(C.byref(_w(wcs)), C.c_int64(nrows), _dp(out))
    return out


def skyarea_cyl(wcs, shape):
    return lib().pxl_skyarea_cyl_cpu(C.byref(_w(wcs)), _i64(shape[:2]))


def sky2pix_tan(wcs, ra, dec):
    ra, dec = _f64(np.atleast_1d(ra)), _f64(np.atleast_1d(dec))
    x, y = np.empty_like(ra), np.empty_like(dec)
    lib().pxl_sky2pix_tan_f64_cpu(C.byref(_w(wcs)), C.c_int64(ra.size), _dp(ra), _dp(dec), _dp(x), _dp(y))
    return x, y


def pix2sky_tan(wcs, ipix, jpix):
    ipix, jpix = _f64(np.atleast_1d(ipix)), _f64(np.atleast_1d(jpix))
    ra, dec = np.empty_like(ipix), np.empty_like(jpix)
    lib().pxl_pix2sky_tan_f64_cpu(C.byref(_w(wcs)), C.c_int64(ipix.size), _dp(ipix), _dp(jpix), _dp(ra), _dp(dec))
    return ra, dec


def is_periodic(wcs, nx):
    return bool(lib().pxl_car_is_periodic_cpu(C.byref(_w(wcs)), C.c_int64(nx)))


def _shape3(shape):
    return (int(shape[0]), int(shape[1]), int(shape[2]) if len(shape) > 2 else 1)


def sample_bilinear(wcs_in, shape_in, src, sky, src_row0=0, src_nrows=None):
    """src: (nc, nrows, nx) C-order (== Julia (nx, nrows, nc)); sky: (n, 2).  Returns (nc, n)."""
    nx, ny, nc = _shape3(shape_in)
    src = _f64(src)
    src_nrows = ny - src_row0 if src_nrows is None else src_nrows
    assert src.size == nx * src_nrows * nc
    sky = _f64(sky)
    n = sky.shape[0]
    out = np.empty((nc, n))
    rc = lib().pxl_sample_car_bilinear_f64_cpu(C.byref(_w(wcs_in)), _i64((nx, ny, nc)), _dp(src),
                                               C.c_int64(src_row0), C.c_int64(src_nrows), C.c_int64(n),
                                               _dp(sky), _dp(out))
    assert rc == 0
    return out


def reproject_tables(wcs_in, shape_in, wcs_out, shape_out):
    xs = np.empty(int(shape_out[0])); ys = np.empty(int(shape_out[1]))
    lib().pxl_reproject_tables_cpu(C.byref(_w(wcs_in)), _i64(shape_in[:2]), C.byref(_w(wcs_out)),
                                   _i64(shape_out[:2]), _dp(xs), _dp(ys))
    return xs, ys


def reproject(wcs_in, shape_in, src, wcs_out, shape_out, src_row0=0, src_nrows=None, dst_row0=0,
              dst_nrows=None, out=None):
    """src: (nc, src_nrows, nx) C-order.  Returns dst (nc, dst_nrows, nx_out); out: a caller-allocated dst."""
    nx, ny, nc = _shape3(shape_in)
    nxo, nyo = int(shape_out[0]), int(shape_out[1])
    src = _f64(src)
    src_nrows = ny - src_row0 if src_nrows is None else src_nrows
    dst_nrows = nyo - dst_row0 if dst_nrows is None else dst_nrows
    assert src.size == nx * src_nrows * nc
    dst = np.empty((nc, dst_nrows, nxo)) if out is None else out
    assert dst.shape == (nc, dst_nrows, nxo) and dst.dtype == np.float64 and dst.flags.c_contiguous
    rc = lib().pxl_reproject_car_bilinear_f64_cpu(
        C.byref(_w(wcs_in)), _i64((nx, ny, nc)), _dp(src), C.c_int64(src_row0), C.c_int64(src_nrows),
        C.byref(_w(wcs_out)), _i64((nxo, nyo)), _dp(dst), C.c_int64(dst_row0), C.c_int64(dst_nrows))
    assert rc == 0, rc
    return dst


def reproject_src_rows(wcs_in, shape_in, wcs_out, shape_out, dst_row0, dst_nrows):
    lo, hi = C.c_int64(), C.c_int64()
    rc = lib().pxl_reproject_src_rows_cpu(C.byref(_w(wcs_in)), _i64(shape_in[:2]), C.byref(_w(wcs_out)),
                                          _i64(shape_out[:2]), C.c_int64(dst_row0), C.c_int64(dst_nrows),
                                          C.byref(lo), C.byref(hi))
    assert rc == 0
    return lo.value, hi.value


def reproject_generic(wcs_in, proj_in, shape_in, src, wcs_out, proj_out, shape_out):
    """CAR (proj 0) / Gnomonic (proj 1) non-separable reprojection.  src: (nc, ny, nx); returns (nc, nyo, nxo)."""
    nx, ny, nc = _shape3(shape_in)
    nxo, nyo = int(shape_out[0]), int(shape_out[1])
    src = _f64(src)
    assert src.size == nx * ny * nc
    dst = np.empty((nc, nyo, nxo))
    rc = lib().pxl_reproject_generic_bilinear_f64_cpu(
        C.byref(_w(wcs_in)), C.c_int(proj_in), _i64((nx, ny, nc)), _dp(src), C.byref(_w(wcs_out)), C.c_int(proj_out),
        _i64((nxo, nyo)), _dp(dst))
    assert rc == 0
    return dst


def reproject_f32(wcs_in, shape_in, src32, wcs_out, shape_out, **kw):
    """Float32 maps: taps widened to Float64 (exact), the R1 arithmetic in Float64, one rounding to Float32 --
    what Julia does when a Float64 expression is stored into a Float32 array."""
    src32 = np.ascontiguousarray(src32, dtype=np.float32)
    return reproject(wcs_in, shape_in, src32.astype(np.float64), wcs_out, shape_out, **kw).astype(np.float32)


def sample_bilinear_f32(wcs_in, shape_in, src32, sky, **kw):
    src32 = np.ascontiguousarray(src32, dtype=np.float32)
    return sample_bilinear(wcs_in, shape_in, src32.astype(np.float64), sky, **kw).astype(np.float32)
