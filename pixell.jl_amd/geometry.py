"""Geometry constructors and slicing (host-side, O(1)); mirrors /root/reference/src/enmap_geom.jl and
enmap_ops.jl:142-168.  These define the synthetic benchmark maps bit-for-bit."""
import math

from .wcs import CarClenshawCurtis, PI, TWOPI, sliced_wcs


def _jl_round(x: float) -> int:
    """Julia round(Int, x): ties to even (Python's round does the same)."""
    return int(round(x))


def _sign(x: float) -> float:
    return 1.0 if x > 0 else (-1.0 if x < 0 else x)


def _jl_div(a: int, b: int) -> int:
    """Julia `div` on Ints: truncation toward zero."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def _rad2deg(x: float) -> float:
    """Base.rad2deg(z::AbstractFloat) = z * (180 / oftype(z, pi)): one multiply by the Float64 constant 180/pi."""
    return x * (180 / PI)


def create_car_wcs(W, cdelt, crpix, crval):
    """enmap_geom.jl:13-19 (unit = pi/180)."""
    return W((cdelt[0], cdelt[1]), (crpix[0], crpix[1]), (crval[0], crval[1]), PI / 180)


def fullsky_geometry(res, W=CarClenshawCurtis, shape=None, dims=()):
    """enmap_geom.jl:47-73.  `res` in radians (a number, or a (res_ra, res_dec) tuple)."""
    resx, resy = (res, res) if not isinstance(res, (tuple, list)) else res
    if shape is None:
        shape = (_jl_round(TWOPI / resx + 0), _jl_round(PI / resy + 1))   # CAR has pixels on poles
    nx, ny = int(shape[0]), int(shape[1])
    if not abs(resx * nx - TWOPI) < 1e-8:
        raise AssertionError("Horizontal resolution does not evenly divide the sky; this is required for SHTs.")
    if not abs(resy * (ny - 1) - PI) < 1e-8:
        raise AssertionError("Vertical resolution does not evenly divide the sky; this is required for SHTs.")
    wcs = create_car_wcs(W,
                         (-360.0 / nx, 180.0 / (ny - 1)),
                         (math.floor(nx / 2) + 0.5, (ny + 1) / 2),
                         (resy * 90 / PI, 0.0))
    return (nx, ny) + tuple(dims), wcs


def geometry(bbox_coords, res, W=CarClenshawCurtis):
    """enmap_geom.jl:77-108.  bbox_coords = [[ra1, ra2], [dec1, dec2]] in radians (column k is corner k)."""
    resx, resy = (res, res) if not isinstance(res, (tuple, list)) else res
    if not abs(TWOPI / resx - round(TWOPI / resx)) < 1e-8:
        raise AssertionError("Horizontal resolution does not evenly divide the sky; this is required for SHTs.")
    if not abs(TWOPI / resy - round(TWOPI / resy)) < 1e-8:
        raise AssertionError("Vertical resolution does not evenly divide the sky; this is required for SHTs.")
    pos1 = (float(bbox_coords[0][0]), float(bbox_coords[1][0]))
    pos2 = (float(bbox_coords[0][1]), float(bbox_coords[1][1]))
    r = (resx, resy)
    shape, cdelt, crpix, crval = [], [], [], []
    for k in range(2):
        shape.append(_jl_round(abs(pos1[k] - pos2[k]) / r[k]))
        mid = (pos1[k] + pos2[k]) / 2
        cv = mid if k == 0 else 0.0
        cd = abs(r[k]) * _sign(pos2[k] - pos1[k])
        crpix.append(1 - (pos1[k] - cv) / cd)
        cdelt.append(_rad2deg(cd))
        crval.append(_rad2deg(cv))
    return tuple(shape), create_car_wcs(W, cdelt, crpix, crval)


class JlRange:
    """A Julia-style inclusive range first:step:stop with 1-based indices (last() is normalised)."""

    def __init__(self, first, stop=None, step=1):
        if stop is None:
            stop = first
        if step == 0:
            raise ValueError("step cannot be zero")
        self.first, self.step = int(first), int(step)
        n = (int(stop) - self.first) // self.step        # floor division handles both signs
        self.length = max(n + 1, 0)
        self.last = self.first + self.step * (self.length - 1) if self.length > 0 else self.first - self.step

    def to_slice(self):
        """0-based Python slice selecting the same elements."""
        lo = self.first - 1
        hi = self.last - 1 + (1 if self.step > 0 else -1)
        return slice(lo, hi if hi >= 0 else None, self.step)


def _as_range(sel, n):
    if isinstance(sel, JlRange):
        return sel
    if sel is None or sel is Ellipsis or (isinstance(sel, slice) and sel == slice(None)):
        return JlRange(1, n)
    if isinstance(sel, int):
        return JlRange(sel, sel)
    if isinstance(sel, (tuple, list)):
        if len(sel) == 2:
            return JlRange(sel[0], sel[1])
        return JlRange(sel[0], sel[2], sel[1])      # (first, step, stop) like first:step:stop
    raise TypeError("cannot interpret %r as a 1-based range" % (sel,))


def slice_geometry(shape_all, wcs, sel_x=None, sel_y=None):
    """enmap_ops.jl:154-167.  Selections are 1-based inclusive: JlRange, (first, stop),
    (first, step, stop), an int, or None for `:`."""
    sel = (_as_range(sel_x, shape_all[0]), _as_range(sel_y, shape_all[1]))
    starts = [s.first - 1 if s.step > 0 else s.first for s in sel]
    steps = [s.step for s in sel]
    sel_sizes = [s.last - s.first + s.step for s in sel]
    crpix = tuple((wcs.crpix[k] - (starts[k] + 0.5)) / steps[k] + 0.5 for k in range(2))
    cdelt = tuple(wcs.cdelt[k] * steps[k] for k in range(2))
    shape = tuple(_jl_div(sel_sizes[k], steps[k]) for k in range(2))
    return shape + tuple(shape_all[2:]), sliced_wcs(wcs, cdelt, crpix)


def pad_geometry(shape, wcs, npix_ra, npix_dec, mode="center"):
    """center_pad / corner_pad geometry, car_proj.jl:303-326."""
    if mode == "center":
        new_shape = (shape[0] + 2 * npix_ra, shape[1] + 2 * npix_dec) + tuple(shape[2:])
        return new_shape, type(wcs)(wcs.cdelt, (wcs.crpix[0] + npix_ra, wcs.crpix[1] + npix_dec), wcs.crval, wcs.unit)
    if mode == "corner":
        return (shape[0] + npix_ra, shape[1] + npix_dec) + tuple(shape[2:]), wcs
    raise ValueError("mode must be 'center' or 'corner'")


def skyarea(shape, wcs):
    """skyarea_cyl, arbitrary_wcs.jl:125-132 (host scalar)."""
    d0, dd = wcs.crval[1] * wcs.unit, wcs.cdelt[1] * wcs.unit
    da = wcs.cdelt[0] * wcs.unit
    e0 = d0 + (0.5 - wcs.crpix[1]) * dd
    e1 = d0 + ((shape[1] + 0.5) - wcs.crpix[1]) * dd
    d1, d2 = min(e0, e1), max(e0, e1)
    d1, d2 = max(-PI / 2, d1), min(PI / 2, d2)
    return (math.sin(d2) - math.sin(d1)) * abs(da) * shape[0]


def extent_cyl(shape, wcs, signed=False):
    """extent_cyl, arbitrary_wcs.jl:134-148: (RA extent at the mean cos(dec), DEC extent) in radians."""
    n_a, n_d = shape[0], shape[1]
    d0, dd = wcs.crval[1] * wcs.unit, wcs.cdelt[1] * wcs.unit
    da = wcs.cdelt[0] * wcs.unit
    e0 = d0 + (0.5 - wcs.crpix[1]) * dd
    e1 = d0 + ((n_d + 0.5) - wcs.crpix[1]) * dd
    d1, d2 = min(e0, e1), max(e0, e1)
    d1, d2 = max(-PI / 2, d1), min(PI / 2, d2)
    dsign = 1 if d1 <= d2 else -1
    mean_cos = (math.sin(d2) - math.sin(d1)) / (d2 - d1)
    ext = (n_a * da * mean_cos, (d2 - d1) * dsign)
    return ext if signed else (abs(ext[0]), abs(ext[1]))


def _fftfreq(n, fs):
    """AbstractFFTs.fftfreq(n, fs): [0, 1, ..., ceil(n/2)-1, -floor(n/2), ..., -1] * fs / n."""
    half = (n + 1) // 2
    return [k * fs / n for k in range(half)] + [(k - n) * fs / n for k in range(half, n)]


def laxes_cyl(shape, wcs):
    """laxes_cyl, arbitrary_wcs.jl:157-162: multipole axes (l_ra, l_dec) of a cylindrical patch."""
    ext = extent_cyl(shape[:2], wcs, signed=True)
    da_bar, dd_bar = ext[0] / shape[0], ext[1] / shape[1]
    return _fftfreq(shape[0], TWOPI / da_bar), _fftfreq(shape[1], TWOPI / dd_bar)
