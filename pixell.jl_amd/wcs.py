"""WCS types and host-side scalar evaluators (mirrors /root/reference/src/projections/car_proj.jl,
tan_proj.jl and the top of enmap_ops.jl).

Scalar calls stay on the host -- exactly as SURVEY 8(b) assigns them ("only the batched/whole-map callers
go to the GPU; scalar calls stay in Julia") -- and are written op-for-op like the reference so that they
agree bit-for-bit with the device kernels (Python floats are IEEE doubles evaluated one operation at a
time: no FMA contraction).
"""
import math
from dataclasses import dataclass
from typing import Tuple

from ._lib import CarWCSStruct

PI = math.pi                 # Float64(pi)
TWOPI = 2 * math.pi          # Julia `2pi`
DEG = math.pi / 180          # enmap_geom.jl:18


def jl_mod(x: float, y: float) -> float:
    """Julia Base.mod for Float64 (base/float.jl): result takes the sign of y."""
    r = math.fmod(x, y)
    if r == 0.0:
        return math.copysign(r, y)
    if (r > 0.0) != (y > 0.0):
        return r + y
    return r


def rewind(angle: float, period: float = TWOPI, ref_angle: float = 0.0) -> float:
    """enmap_ops.jl:10-13."""
    half = period / 2
    return (ref_angle + jl_mod((angle - ref_angle) + half, period)) - half


@dataclass(frozen=True)
class _WCSBase:
    cdelt: Tuple[float, float]
    crpix: Tuple[float, float]
    crval: Tuple[float, float]
    unit: float = DEG

    naxis = 2            # car_proj.jl:49-54

    def __post_init__(self):
        object.__setattr__(self, "cdelt", (float(self.cdelt[0]), float(self.cdelt[1])))
        object.__setattr__(self, "crpix", (float(self.crpix[0]), float(self.crpix[1])))
        object.__setattr__(self, "crval", (float(self.crval[0]), float(self.crval[1])))
        object.__setattr__(self, "unit", float(self.unit))

    def to_struct(self) -> CarWCSStruct:
        s = CarWCSStruct()
        s.cdelt[:] = self.cdelt
        s.crpix[:] = self.crpix
        s.crval[:] = self.crval
        s.unit = self.unit
        return s

    def __repr__(self):   # car_proj.jl:56-65
        return "%s{Float64}(naxis=2,cdelt=%s,crval=%s,crpix=%s)" % (
            type(self).__name__, self.cdelt, self.crval, self.crpix)


class AbstractCARWCS(_WCSBase):
    """car_proj.jl:2"""


class CarClenshawCurtis(AbstractCARWCS):
    """car_proj.jl:7-12 (pixels on the poles)."""


class CarFejer1(AbstractCARWCS):
    """car_proj.jl:14-19."""


class Gnomonic(_WCSBase):
    """tan_proj.jl:4-9."""


def getunit(wcs): return wcs.unit          # car_proj.jl:21
def getcdelt(wcs): return wcs.cdelt        # car_proj.jl:22
def getcrpix(wcs): return wcs.crpix        # car_proj.jl:23
def getcrval(wcs): return wcs.crval        # car_proj.jl:24


def sliced_wcs(wcs, cdelt, crpix):
    """car_proj.jl:275-278"""
    return type(wcs)(tuple(cdelt), tuple(crpix), wcs.crval, wcs.unit)


def is_periodic(wcs, nx: int) -> bool:
    """Full-circle test with the threshold of enmap_geom.jl:55; decides RA wrap of the bilinear taps."""
    return abs(nx * abs(wcs.cdelt[0] * wcs.unit) - TWOPI) < 1e-8


# ---- scalar evaluators ---------------------------------------------------------------------------

def pix2sky_scalar(shape, wcs: AbstractCARWCS, ra_pixel: float, dec_pixel: float, safe: bool = True):
    """car_proj.jl:141-152 (safe -> rewind)."""
    a0, d0 = wcs.crval[0] * wcs.unit, wcs.crval[1] * wcs.unit
    da, dd = wcs.cdelt[0] * wcs.unit, wcs.cdelt[1] * wcs.unit
    a = a0 + (float(ra_pixel) - wcs.crpix[0]) * da
    d = d0 + (float(dec_pixel) - wcs.crpix[1]) * dd
    if safe:
        return rewind(a), rewind(d)
    return a, d


def sky2pix_scalar(shape, wcs: AbstractCARWCS, ra: float, dec: float, safe: bool = True):
    """car_proj.jl:220-234 (division form)."""
    a0, d0 = wcs.crval[0] * wcs.unit, wcs.crval[1] * wcs.unit
    da, dd = wcs.cdelt[0] * wcs.unit, wcs.cdelt[1] * wcs.unit
    x = wcs.crpix[0] + (float(ra) - a0) / da
    y = wcs.crpix[1] + (float(dec) - d0) / dd
    if safe:
        cx, cy = shape[0] / 2 + 1, shape[1] / 2 + 1
        x = rewind(x, abs(TWOPI / da), cx)
        y = rewind(y, abs(TWOPI / dd), cy)
    return x, y


def sky2pix_tan_scalar(shape, wcs: Gnomonic, ra: float, dec: float):
    """tan_proj.jl:44-57"""
    scale = 1.0 / wcs.cdelt[0]
    a0, d0 = wcs.crval[0] * DEG, wcs.crval[1] * DEG
    A = math.cos(dec) * math.cos(ra - a0)
    F = scale / wcs.unit / (math.sin(d0) * math.sin(dec) + A * math.cos(d0))
    line = -F * (math.cos(d0) * math.sin(dec) - A * math.sin(d0))
    sample = -F * math.cos(dec) * math.sin(ra - a0)
    return wcs.crpix[0] - sample, wcs.crpix[1] - line


def pix2sky_tan_scalar(shape, wcs: Gnomonic, ra_pixel: float, dec_pixel: float):
    """tan_proj.jl:59-75"""
    scale = 1.0 / wcs.cdelt[0]
    a0, d0 = wcs.crval[0] * DEG, wcs.crval[1] * DEG
    X = (wcs.crpix[0] - ra_pixel) * wcs.unit / scale
    Y = (wcs.crpix[1] - dec_pixel) * wcs.unit / scale
    D = math.atan(math.sqrt(X * X + Y * Y))
    B = math.atan2(-X, Y)
    XX = math.sin(d0) * math.sin(D) * math.cos(B) + math.cos(d0) * math.cos(D)
    YY = math.sin(D) * math.sin(B)
    return a0 + math.atan2(YY, XX), math.asin(math.sin(d0) * math.cos(D) - math.cos(d0) * math.sin(D) * math.cos(B))


@dataclass(frozen=True)
class SkyBoundingBox:
    """enmap_ops.jl:171-188"""
    ra_min: float
    dec_min: float
    ra_max: float
    dec_max: float

    @classmethod
    def from_corners(cls, c1, c2):
        return cls(min(c1[0], c2[0]), min(c1[1], c2[1]), max(c1[0], c2[0]), max(c1[1], c2[1]))

    def __contains__(self, skycoords):
        a, d = skycoords
        return (self.ra_min <= a <= self.ra_max) and (self.dec_min <= d <= self.dec_max)
