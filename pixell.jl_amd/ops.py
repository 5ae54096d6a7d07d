"""Device operators: the reference's batched evaluators and the reprojection, through the C ABI.

Names, argument meaning and `safe` keyword follow /root/reference/src/projections/car_proj.jl and
src/enmap_ops.jl.  Dispatch mirrors the reference's methods:

    pix2sky(m, pix2xN)            car_proj.jl:118-122   device, 2xN batch == torch tensor of shape (N, 2)
    pix2sky(m, ivec, jvec)        car_proj.jl:141-152   device, two N-vectors (broadcast form)
    pix2sky(m, i, j)              car_proj.jl:141-152   host scalar
    pix2sky(m, [i, j])            car_proj.jl:155-162   host scalar, length-2 vector
    sky2pix(m, sky2xN)            car_proj.jl:196-200   device
    sky2pix(m, ravec, decvec)     car_proj.jl:235-252   device
    sky2pix(m, ra, dec)           car_proj.jl:220-234   host scalar
    sky2pix(m, [ra, dec])         car_proj.jl:255-259   host scalar

`m` is an Enmap or a (shape, wcs) pair (enmap_ops.jl:60-66).  Every device path calls libpixell_hip.so;
nothing here computes on the CPU in its place.
"""
import ctypes as C

import torch

from . import _lib, placement
from ._lib import FORM_DIV, FORM_RECIP, FORM_RECIP_AV, WRAP_NONE, WRAP_REWIND, WRAP_UNWIND
from .enmap import Enmap, _geom
from .wcs import (AbstractCARWCS, Gnomonic, pix2sky_scalar, pix2sky_tan_scalar, sky2pix_scalar,
                  sky2pix_tan_scalar)


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev_f64(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch tensor" % what)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the MI355X path has no CPU fallback "
                           "(CPU arrays stay with the reference implementation)" % what)
    if t.dtype != torch.float64:
        raise TypeError("%s must be float64" % what)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)
    return t


def _dev_map(t: torch.Tensor, what: str) -> torch.Tensor:
    """Map storage may be Float64 or Float32 (coordinates are always Float64)."""
    if isinstance(t, torch.Tensor) and t.dtype == torch.float32:
        if not t.is_cuda:
            raise RuntimeError("%s must live on the GPU: the MI355X path has no CPU fallback" % what)
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % what)
        return t
    return _dev_f64(t, what)


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _is_scalar(x):
    return isinstance(x, (int, float))


def _wcs_ref(wcs):
    return C.byref(wcs.to_struct())


def _shape2(shape):
    return _lib.shape_arr((shape[0], shape[1]))


# ---- pix2sky ------------------------------------------------------------------------------------

def pix2sky_(m, pixcoords, skycoords, safe=True):
    """pix2sky!(m, pixcoords, skycoords; safe) -- car_proj.jl:92-115.  Returns skycoords."""
    shape, wcs = _geom(m)
    _require_car(wcs)
    pix = _dev_f64(pixcoords, "pixcoords")
    sky = _dev_f64(skycoords, "skycoords")
    if pix.dim() != 2 or pix.shape[1] != 2 or sky.shape != pix.shape:
        raise ValueError("coordinate batches are (N, 2) tensors (Julia 2xN)")
    with torch.cuda.device(pix.device):
        _lib.check(_lib.load().pxl_pix2sky_car_f64(_wcs_ref(wcs), pix.shape[0], _ptr(pix), _ptr(sky),
                                                   WRAP_UNWIND if safe else WRAP_NONE, _stream(pix)))
    return sky


def pix2sky(m, p1, p2=None, safe=True):
    shape, wcs = _geom(m)
    if isinstance(wcs, Gnomonic):
        return _pix2sky_tan(shape, wcs, p1, p2)
    _require_car(wcs)
    if p2 is None:
        if isinstance(p1, torch.Tensor):
            return pix2sky_(m, p1, torch.empty_like(p1), safe=safe)
        if len(p1) != 2:                       # @assert length(pixcoords) == 2, car_proj.jl:156
            raise AssertionError("length(pixcoords) == 2")
        # car_proj.jl:157: the inner scalar call does not forward `safe` (always rewinds);
        # the unwind! that follows is a no-op on one point.
        return list(pix2sky_scalar(shape, wcs, p1[0], p1[1], safe=True))
    if _is_scalar(p1) and _is_scalar(p2):
        return pix2sky_scalar(shape, wcs, p1, p2, safe=safe)
    ip, jp = _dev_f64(p1, "ra_pixel"), _dev_f64(p2, "dec_pixel")
    if ip.shape != jp.shape:
        raise ValueError("ra_pixel and dec_pixel must have the same shape")
    ra, dec = torch.empty_like(ip), torch.empty_like(jp)
    with torch.cuda.device(ip.device):
        _lib.check(_lib.load().pxl_pix2sky_car_soa_f64(_wcs_ref(wcs), ip.numel(), _ptr(ip), _ptr(jp), _ptr(ra),
                                                       _ptr(dec), int(bool(safe)), _stream(ip)))
    return ra, dec


def pix2sky_rewind(m, pixcoords):
    """2xN pix2sky with the per-element rewind of the scalar method (PXL_WRAP_REWIND)."""
    shape, wcs = _geom(m)
    pix = _dev_f64(pixcoords, "pixcoords")
    sky = torch.empty_like(pix)
    with torch.cuda.device(pix.device):
        _lib.check(_lib.load().pxl_pix2sky_car_f64(_wcs_ref(wcs), pix.shape[0], _ptr(pix), _ptr(sky),
                                                   WRAP_REWIND, _stream(pix)))
    return sky


# ---- rewind / unwind on device arrays (enmap_ops.jl:10-32) ----------------------------------------

def rewind_(angles: torch.Tensor, period=2 * 3.141592653589793, ref_angle=0.0):
    """rewind!(angles; period, ref_angle), elementwise, in place."""
    a = _dev_f64(angles, "angles")
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().pxl_rewind_f64(_ptr(a), a.numel(), float(period), float(ref_angle), _stream(a)))
    return angles


def unwind_(angles: torch.Tensor, period=2 * 3.141592653589793, ref_angle=0.0):
    """unwind!(angles; dims=2, ...) for an (N, 2) coordinate batch, or along a 1-D vector; in place."""
    a = _dev_f64(angles, "angles")
    if a.dim() == 2 and a.shape[1] == 2:
        n, nrow = a.shape[0], 2
    elif a.dim() == 1:
        n, nrow = a.shape[0], 1
    else:
        raise ValueError("unwind_ takes an (N, 2) batch or a 1-D vector")
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().pxl_unwind_f64(_ptr(a), n, nrow, float(period), float(ref_angle), _stream(a)))
    return angles


# ---- sky2pix ------------------------------------------------------------------------------------

def sky2pix_(m, skycoords, pixcoords, safe=True):
    """sky2pix!(m, skycoords, pixcoords; safe) -- car_proj.jl:165-193.  Returns pixcoords."""
    shape, wcs = _geom(m)
    _require_car(wcs)
    sky = _dev_f64(skycoords, "skycoords")
    pix = _dev_f64(pixcoords, "pixcoords")
    if sky.dim() != 2 or sky.shape[1] != 2 or sky.shape != pix.shape:
        raise ValueError("coordinate batches are (N, 2) tensors (Julia 2xN)")
    with torch.cuda.device(sky.device):
        _lib.check(_lib.load().pxl_sky2pix_car_f64(_wcs_ref(wcs), _shape2(shape), sky.shape[0], _ptr(sky),
                                                   _ptr(pix), int(bool(safe)), FORM_RECIP, _stream(sky)))
    return pix


def sky2pix(m, p1, p2=None, safe=True):
    shape, wcs = _geom(m)
    if isinstance(wcs, Gnomonic):
        return _sky2pix_tan(shape, wcs, p1, p2)
    _require_car(wcs)
    if p2 is None:
        if isinstance(p1, torch.Tensor):
            return sky2pix_(m, p1, torch.empty_like(p1), safe=safe)
        if len(p1) != 2:                       # car_proj.jl:256
            raise AssertionError("length(skycoords) == 2")
        return list(sky2pix_scalar(shape, wcs, p1[0], p1[1], safe=safe))
    if _is_scalar(p1) and _is_scalar(p2):
        return sky2pix_scalar(shape, wcs, p1, p2, safe=safe)
    ra, dec = _dev_f64(p1, "ra"), _dev_f64(p2, "dec")
    if ra.shape != dec.shape:
        raise ValueError("ra and dec must have the same shape")
    ip, jp = torch.empty_like(ra), torch.empty_like(dec)
    with torch.cuda.device(ra.device):
        _lib.check(_lib.load().pxl_sky2pix_car_soa_f64(_wcs_ref(wcs), _shape2(shape), ra.numel(), _ptr(ra),
                                                       _ptr(dec), _ptr(ip), _ptr(jp), int(bool(safe)),
                                                       FORM_RECIP_AV, _stream(ra)))
    return ip, jp


def sky2pix_broadcast(m, ra, dec, safe=True):
    """sky2pix.(Ref(m), ra, dec): the scalar (division) form of car_proj.jl:220-234 over two vectors."""
    shape, wcs = _geom(m)
    ra, dec = _dev_f64(ra, "ra"), _dev_f64(dec, "dec")
    ip, jp = torch.empty_like(ra), torch.empty_like(dec)
    with torch.cuda.device(ra.device):
        _lib.check(_lib.load().pxl_sky2pix_car_soa_f64(_wcs_ref(wcs), _shape2(shape), ra.numel(), _ptr(ra),
                                                       _ptr(dec), _ptr(ip), _ptr(jp), int(bool(safe)),
                                                       FORM_DIV, _stream(ra)))
    return ip, jp


def _require_car(wcs):
    if not isinstance(wcs, AbstractCARWCS):
        raise TypeError("only CAR WCS (CarClenshawCurtis / CarFejer1) and Gnomonic are accelerated; "
                        "generic WCSTransform maps stay on the reference's wcslib path")


# ---- Gnomonic -----------------------------------------------------------------------------------

def _pix2sky_tan(shape, wcs, p1, p2):
    if _is_scalar(p1) and _is_scalar(p2):
        return pix2sky_tan_scalar(shape, wcs, p1, p2)
    ip, jp = _dev_f64(p1, "ra_pixel"), _dev_f64(p2, "dec_pixel")
    ra, dec = torch.empty_like(ip), torch.empty_like(jp)
    with torch.cuda.device(ip.device):
        _lib.check(_lib.load().pxl_pix2sky_tan_f64(_wcs_ref(wcs), ip.numel(), _ptr(ip), _ptr(jp), _ptr(ra),
                                                   _ptr(dec), _stream(ip)))
    return ra, dec


def _sky2pix_tan(shape, wcs, p1, p2):
    if _is_scalar(p1) and _is_scalar(p2):
        return sky2pix_tan_scalar(shape, wcs, p1, p2)
    ra, dec = _dev_f64(p1, "ra"), _dev_f64(p2, "dec")
    ip, jp = torch.empty_like(ra), torch.empty_like(dec)
    with torch.cuda.device(ra.device):
        _lib.check(_lib.load().pxl_sky2pix_tan_f64(_wcs_ref(wcs), ra.numel(), _ptr(ra), _ptr(dec), _ptr(ip),
                                                   _ptr(jp), _stream(ra)))
    return ip, jp


# ---- whole-map writers --------------------------------------------------------------------------

def posmap(shape, wcs, device="cuda", row0=0, nrows=None, safe=True):
    """posmap(shape, wcs) -- enmap_ops.jl:190-203: (ra_map, dec_map) Enmaps of the pixel centres.
    row0/nrows select a declination strip (0-based rows) for sharded use."""
    nx, ny = int(shape[0]), int(shape[1])
    nrows = ny - row0 if nrows is None else nrows
    dev = torch.device(device)
    ra = torch.empty((nrows, nx), dtype=torch.float64, device=dev)
    dec = torch.empty((nrows, nx), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        lib = _lib.load()
        if isinstance(wcs, Gnomonic):
            rc = lib.pxl_posmap_tan_f64(_wcs_ref(wcs), _shape2(shape), row0, nrows, _ptr(ra), _ptr(dec), _stream(ra))
        else:
            _require_car(wcs)
            rc = lib.pxl_posmap_car_f64(_wcs_ref(wcs), _shape2(shape), row0, nrows, _ptr(ra), _ptr(dec),
                                        int(bool(safe)), _stream(ra))
        _lib.check(rc)
    strip_wcs = wcs
    if (row0, nrows) != (0, ny):
        from .geometry import slice_geometry
        _, strip_wcs = slice_geometry((nx, ny), wcs, None, (row0 + 1, row0 + nrows))
    return Enmap(ra, strip_wcs), Enmap(dec, strip_wcs)


def pixareamap_(pixareas: Enmap):
    """pixareamap!(pixareas) -- enmap_ops.jl:124-138."""
    shape, wcs = pixareas.shape, pixareas.wcs
    _require_car(wcs)
    data = _dev_f64(pixareas.data, "pixareas")
    nplanes = data.shape[0] if data.dim() == 3 else 1
    with torch.cuda.device(data.device):
        for c in range(nplanes):
            plane = data[c] if data.dim() == 3 else data
            _lib.check(_lib.load().pxl_pixareamap_car_f64(_wcs_ref(wcs), _shape2(shape), 0, shape[1], _ptr(plane),
                                                          _stream(data)))
    return pixareas


def pixareamap(m, wcs=None, device="cuda"):
    """pixareamap(m::Enmap) / pixareamap(shape, wcs) -- car_proj.jl:264-272."""
    if isinstance(m, Enmap):
        return pixareamap_(m.similar())
    shape = tuple(m)
    data = torch.empty(tuple(reversed(shape)), dtype=torch.float64, device=device)
    return pixareamap_(Enmap(data, wcs))


# ---- reprojection -------------------------------------------------------------------------------

class ReprojectPlan:
    """Owns a pxl_reproject_plan (device coordinate tables).  src/dst windows describe declination
    strips of sharded maps (0-based row0, nrows); full maps by default."""

    def __init__(self, shape_in, wcs_in, shape_out, wcs_out, src_rows=None, dst_rows=None, device="cuda"):
        _require_car(wcs_in)
        _require_car(wcs_out)
        self.shape_in = (int(shape_in[0]), int(shape_in[1]), int(shape_in[2]) if len(shape_in) > 2 else 1)
        self.shape_out = (int(shape_out[0]), int(shape_out[1]))
        self.wcs_in, self.wcs_out = wcs_in, wcs_out
        self.src_rows = (0, self.shape_in[1]) if src_rows is None else (int(src_rows[0]), int(src_rows[1]))
        self.dst_rows = (0, self.shape_out[1]) if dst_rows is None else (int(dst_rows[0]), int(dst_rows[1]))
        self.device = torch.device(device)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pxl_reproject_plan_create(
                _wcs_ref(wcs_in), _lib.shape_arr(self.shape_in), self.src_rows[0], self.src_rows[1],
                _wcs_ref(wcs_out), _lib.shape_arr(self.shape_out), self.dst_rows[0], self.dst_rows[1],
                C.byref(self._h)))

    @property
    def ncomp(self):
        return self.shape_in[2]

    def src_tensor_shape(self):
        return (self.ncomp, self.src_rows[1], self.shape_in[0])

    def dst_tensor_shape(self):
        return (self.ncomp, self.dst_rows[1], self.shape_out[0])

    def set_variant(self, variant: int):
        _lib.check(_lib.load().pxl_reproject_plan_set_variant(self._h, int(variant)))

    def src_rows_needed(self):
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().pxl_reproject_plan_src_rows(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def rows_covered(self, have_lo, have_hi):
        lo, hi = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().pxl_reproject_plan_rows_covered(self._h, have_lo, have_hi, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def _check(self, src, dst):
        src, dst = _dev_map(src, "src"), _dev_map(dst, "dst")
        if src.dtype != dst.dtype:
            raise TypeError("src and dst must have the same dtype (Float64 or Float32)")
        if src.numel() != self.ncomp * self.src_rows[1] * self.shape_in[0]:
            raise ValueError("src has %d elements, plan expects %s" % (src.numel(), (self.src_tensor_shape(),)))
        if dst.numel() != self.ncomp * self.dst_rows[1] * self.shape_out[0]:
            raise ValueError("dst has %d elements, plan expects %s" % (dst.numel(), (self.dst_tensor_shape(),)))
        if src.data_ptr() == dst.data_ptr():
            raise ValueError("src and dst must not alias")
        return src, dst

    def execute(self, src, dst):
        src, dst = self._check(src, dst)
        lib = _lib.load()
        fn = lib.pxl_reproject_execute_f32 if src.dtype == torch.float32 else lib.pxl_reproject_execute
        with torch.cuda.device(self.device):
            _lib.check(fn(self._h, _ptr(src), _ptr(dst), _stream(dst)))
        return dst

    def build_tables(self):
        with torch.cuda.device(self.device):
            s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.load().pxl_reproject_build_tables(self._h, s))

    def execute_rows(self, src, dst, r0, nr):
        src, dst = self._check(src, dst)
        lib = _lib.load()
        fn = lib.pxl_reproject_execute_rows_f32 if src.dtype == torch.float32 else lib.pxl_reproject_execute_rows
        with torch.cuda.device(self.device):
            _lib.check(fn(self._h, _ptr(src), _ptr(dst), r0, nr, _stream(dst)))
        return dst

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().pxl_reproject_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reproject(m: Enmap, shape_out, wcs_out, out: Enmap = None, plan: ReprojectPlan = None) -> Enmap:
    """Bilinear CAR -> CAR reprojection of every component of `m` onto (shape_out, wcs_out).
    Not in the reference; composes posmap(out) o sky2pix(in) o 2x2 gather + lerp."""
    nxo, nyo = int(shape_out[0]), int(shape_out[1])
    if isinstance(m.wcs, Gnomonic) or isinstance(wcs_out, Gnomonic):
        return _reproject_generic(m, (nxo, nyo), wcs_out, out, plan)
    if plan is None:
        plan = ReprojectPlan(m.shape, m.wcs, shape_out, wcs_out, device=m.device)
    if out is None:
        # the library's allocation policy for map-sized outputs (placement.empty_map): by default a destination of 3 GiB or more
        # is placed across a boundary between two memory classes of the HBM -- no head-room is kept -- because the kernel's
        # eight write fronts store 15 % faster there; PXL_ALLOC_POLICY=plain or pj.set_allocation_policy("plain")
        # turn that into torch.empty.  Allocate once and pass `out=` (and `plan=`) when reprojecting repeatedly.
        oshape = (nyo, nxo) if m.data.dim() == 2 else (m.data.shape[0], nyo, nxo)
        data, _info = placement.empty_map(oshape, dtype=m.data.dtype, device=m.device)
        out = Enmap(data, wcs_out)
    plan.execute(m.data, out.data)
    return out


def _proj_code(w):
    if not isinstance(w, (AbstractCARWCS, Gnomonic)):
        raise TypeError("generic reprojection handles CAR and Gnomonic WCS only")
    return 1 if isinstance(w, Gnomonic) else 0   # PXL_PROJ_TAN / PXL_PROJ_CAR


class GenericReprojectPlan:
    """Owns a pxl_generic_plan: the coordinate lattice of a CAR <-> Gnomonic reprojection (42 exact evaluations and 12 check
    points per 128 x 32 output tile) and the list of tiles that are evaluated per pixel.  What ReprojectPlan's tables are to
    the separable CAR -> CAR path: make it once per pair of geometries, execute it on as many maps as there are --
    `pj.reproject(m, shape_out, wcs_out, out=out, plan=plan)`.  Same results as the one-shot call, bit for bit."""

    def __init__(self, shape_in, wcs_in, shape_out, wcs_out, device="cuda"):
        self.shape_in = (int(shape_in[0]), int(shape_in[1]))
        self.shape_out = (int(shape_out[0]), int(shape_out[1]))
        self.wcs_in, self.wcs_out = wcs_in, wcs_out
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.load().pxl_generic_plan_create(
                _wcs_ref(wcs_in), _proj_code(wcs_in), _shape2(self.shape_in),
                _wcs_ref(wcs_out), _proj_code(wcs_out), _shape2(self.shape_out), s, C.byref(self._h)))

    def tiles(self):
        """(tiles evaluated per pixel, all tiles)"""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().pxl_generic_plan_tiles(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def execute(self, src, dst):
        src, dst = _dev_f64(src, "src"), _dev_f64(dst, "dst")
        nx, ny = self.shape_in
        if src.dim() not in (2, 3) or tuple(src.shape[-2:]) != (ny, nx):
            raise ValueError("src has shape %s, plan expects (..., %d, %d)" % (tuple(src.shape), ny, nx))
        nc = src.shape[0] if src.dim() == 3 else 1
        if dst.numel() != nc * self.shape_out[0] * self.shape_out[1]:
            raise ValueError("dst has %d elements, plan expects %d x %d x %d" % (dst.numel(), nc, self.shape_out[1], self.shape_out[0]))
        if src.device != self.device or dst.device != self.device:
            raise ValueError("plan lives on %s" % (self.device,))
        if src.data_ptr() == dst.data_ptr():
            raise ValueError("src and dst must not alias")
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pxl_generic_plan_execute(self._h, nc, _ptr(src), _ptr(dst), _stream(dst)))
        return dst

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().pxl_generic_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _reproject_generic(m: Enmap, shape_out, wcs_out, out=None, plan=None) -> Enmap:
    """CAR <-> Gnomonic (non-separable) bilinear reprojection: pxl_reproject_generic_bilinear_f64, or a GenericReprojectPlan."""
    data = _dev_f64(m.data, "map data")
    code = _proj_code
    code(m.wcs), code(wcs_out)
    nc = data.shape[0] if data.dim() == 3 else 1
    if out is None:
        oshape = (shape_out[1], shape_out[0]) if data.dim() == 2 else (nc, shape_out[1], shape_out[0])
        out = Enmap(torch.empty(oshape, dtype=torch.float64, device=data.device), wcs_out)
    if plan is not None:
        if not isinstance(plan, GenericReprojectPlan):
            raise TypeError("a CAR <-> Gnomonic reprojection takes a GenericReprojectPlan")
        if plan.shape_in != (int(m.shape[0]), int(m.shape[1])) or plan.shape_out != (int(shape_out[0]), int(shape_out[1])):
            raise ValueError("plan was made for %s -> %s" % (plan.shape_in, plan.shape_out))
        plan.execute(data, out.data)
        return out
    with torch.cuda.device(data.device):
        _lib.check(_lib.load().pxl_reproject_generic_bilinear_f64(
            _wcs_ref(m.wcs), code(m.wcs), _lib.shape_arr((m.shape[0], m.shape[1], nc)), _ptr(data),
            _wcs_ref(wcs_out), code(wcs_out), _shape2(shape_out), _ptr(out.data), _stream(data)))
    return out


class SamplePairs:
    """Row-pair copy of a map for scattered sampling (pxl_sample_build_pairs_*): 8/3 of the footprint, ONE random
    64-byte sector per point instead of 2.25.  Build once per map, pass to sample_bilinear(..., pairs=...)."""

    def __init__(self, m: Enmap, src_rows=None, full_shape=None, out: torch.Tensor = None):
        data = _dev_map(m.data, "map data")
        _require_car(m.wcs)
        self.wcs = m.wcs
        self.shape = tuple(m.shape if full_shape is None else full_shape)
        self.nc = data.shape[0] if data.dim() == 3 else 1
        self.src_rows = (0, self.shape[1]) if src_rows is None else (int(src_rows[0]), int(src_rows[1]))
        self.dtype = data.dtype
        lib = _lib.load()
        shp = _lib.shape_arr((self.shape[0], self.shape[1], self.nc))
        n = lib.pxl_sample_pairs_elems(shp, self.src_rows[1])
        if n < 0:
            raise _lib.PixellHipError(-22, _lib.last_error())
        if data.numel() != self.nc * self.src_rows[1] * self.shape[0]:
            raise ValueError("map data does not match the resident window")
        self.data = out if out is not None else torch.empty(n, dtype=data.dtype, device=data.device)
        if self.data.numel() != n or self.data.dtype != data.dtype or not self.data.is_contiguous():
            raise ValueError("pair buffer must hold %d contiguous %s elements" % (n, data.dtype))
        self.rebuild(data)

    def rebuild(self, data: torch.Tensor):
        """Re-derive the pair copy after the map changed (one streaming pass)."""
        data = _dev_map(data, "map data")
        lib = _lib.load()
        fn = lib.pxl_sample_build_pairs_f32 if data.dtype == torch.float32 else lib.pxl_sample_build_pairs_f64
        with torch.cuda.device(data.device):
            _lib.check(fn(_lib.shape_arr((self.shape[0], self.shape[1], self.nc)), _ptr(data), self.src_rows[1],
                          _ptr(self.data), _stream(data)))
        return self


def sample_bilinear(m: Enmap, skycoords: torch.Tensor, src_rows=None, full_shape=None, pairs: SamplePairs = None) -> torch.Tensor:
    """Bilinear sample of every component of `m` at a 2xN batch of (ra, dec): fused
    sky2pix!(safe=true) [car_proj.jl:165-193] + 2x2 gather.  Returns a (nc, N) tensor.
    src_rows/full_shape describe `m.data` as a declination strip of a larger map.  pairs: a SamplePairs copy of
    the same map (then `m` may be None): same result, fewer random memory sectors per point."""
    sky = _dev_f64(skycoords, "skycoords")
    lib = _lib.load()
    if pairs is not None:
        out = torch.empty((pairs.nc, sky.shape[0]), dtype=pairs.dtype, device=sky.device)
        fn = lib.pxl_sample_car_bilinear_pairs_f32 if pairs.dtype == torch.float32 else lib.pxl_sample_car_bilinear_pairs_f64
        with torch.cuda.device(sky.device):
            _lib.check(fn(_wcs_ref(pairs.wcs), _lib.shape_arr((pairs.shape[0], pairs.shape[1], pairs.nc)), _ptr(pairs.data),
                          pairs.src_rows[0], pairs.src_rows[1], sky.shape[0], _ptr(sky), _ptr(out), _stream(sky)))
        return out
    data = _dev_map(m.data, "map data")
    _require_car(m.wcs)
    shape = m.shape if full_shape is None else tuple(full_shape)
    nc = data.shape[0] if data.dim() == 3 else 1
    row0, nrows = (0, shape[1]) if src_rows is None else src_rows
    out = torch.empty((nc, sky.shape[0]), dtype=data.dtype, device=sky.device)
    fn = lib.pxl_sample_car_bilinear_f32 if data.dtype == torch.float32 else lib.pxl_sample_car_bilinear_f64
    with torch.cuda.device(sky.device):
        _lib.check(fn(
            _wcs_ref(m.wcs), _lib.shape_arr((shape[0], shape[1], nc)), _ptr(data), row0, nrows, sky.shape[0],
            _ptr(sky), _ptr(out), _stream(sky)))
    return out


# ---- synthetic inputs (benchmark plumbing) --------------------------------------------------------

def fill_random_(t: torch.Tensor, seed: int, offset: int = 0, kind: str = "normal"):
    t = _dev_f64(t, "tensor")
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().pxl_fill_random_f64(_ptr(t), t.numel(), seed, offset, 0 if kind == "normal" else 1,
                                                   _stream(t)))
    return t


def fill_sphere_points_(sky: torch.Tensor, seed: int, offset: int = 0):
    sky = _dev_f64(sky, "skycoords")
    with torch.cuda.device(sky.device):
        _lib.check(_lib.load().pxl_fill_sphere_points_f64(_ptr(sky), sky.shape[0], seed, offset, _stream(sky)))
    return sky
