"""Enmap container over a device tensor (mirrors /root/reference/src/enmap.jl:10-81,176).

Memory layout is the reference's: Julia column-major (nx, ny[, nc]) with RA contiguous.  In torch's
row-major terms that is a C-contiguous tensor of shape ([nc,] ny, nx); `Enmap.shape` reports the Julia
shape so geometry code reads like the reference.
"""
import torch

from .geometry import JlRange, _as_range, slice_geometry
from .wcs import AbstractCARWCS


class NoWCS:
    """enmap.jl:20"""


class Enmap:
    def __init__(self, data: torch.Tensor, wcs):
        if data.dim() not in (2, 3):
            raise ValueError("Enmap data must be ([nc,] ny, nx)")
        self.data = data
        self.wcs = wcs

    # ---- array traits forwarded to the parent (enmap.jl:22-36)
    @property
    def shape(self):
        """Julia-order shape (nx, ny[, nc])."""
        return tuple(reversed(self.data.shape))

    def size(self):
        return self.shape

    @property
    def ncomp(self):
        return self.data.shape[0] if self.data.dim() == 3 else 1

    @property
    def device(self):
        return self.data.device

    def parent(self):
        return self.data

    def getwcs(self):
        return self.wcs

    def __repr__(self):   # enmap.jl:27-30
        return "Enmap(shape=%s,wcs=%r)" % (self.shape, self.wcs)

    # ---- similar / copy never share a mutable WCS (test_enmap.jl:68-90); CAR WCS are immutable values
    def similar(self):
        return Enmap(torch.empty_like(self.data), self.wcs)

    def copy(self):
        return Enmap(self.data.clone(), self.wcs)

    deepcopy = copy

    # ---- slicing re-derives the WCS through slice_geometry (enmap.jl:40-43,65-74)
    def view(self, sel_x=None, sel_y=None, sel_c=None):
        """1-based inclusive Julia-style selections (see geometry.slice_geometry); returns a view."""
        if isinstance(sel_x, int) or isinstance(sel_y, int):
            # a dropped RA/DEC axis gives up the WCS and returns the parent view (enmap.jl:47-52)
            return self._raw_view(sel_x, sel_y, sel_c)
        _, new_wcs = slice_geometry(self.shape, self.wcs, sel_x, sel_y)
        return Enmap(self._raw_view(sel_x, sel_y, sel_c), new_wcs)

    def _raw_view(self, sel_x, sel_y, sel_c):
        nx, ny = self.shape[0], self.shape[1]

        def sl(sel, n):
            if isinstance(sel, int):
                return sel - 1
            r = _as_range(sel, n)
            if r.step < 0:
                raise NotImplementedError("negative-step views need a copy: use getindex()")
            return r.to_slice()
        idx = (sl(sel_y, ny), sl(sel_x, nx))
        if self.data.dim() == 3:
            idx = ((slice(None) if sel_c is None else sl(sel_c, self.data.shape[0])),) + idx
        return self.data[idx]

    def getindex(self, sel_x=None, sel_y=None, sel_c=None):
        """m[sel_x, sel_y] with copy semantics; supports negative steps (test_enmap.jl:17-31).
        Built from basic strided slices + flip + one contiguous copy (TensorIterator paths, which split
        operands beyond 32-bit indexing); torch's gather-style kernels (index_select, advanced indexing)
        were observed to return wrong data for operands above ~1.9 GB on this ROCm build."""
        rx, ry = _as_range(sel_x, self.shape[0]), _as_range(sel_y, self.shape[1])

        def cut(t, r, dim):
            if r.length == 0:
                return t.narrow(dim, 0, 0)
            lo, hi = (r.first, r.last) if r.step > 0 else (r.last, r.first)       # 1-based inclusive, ascending
            idx = [slice(None)] * t.dim()
            idx[dim] = slice(lo - 1, hi, abs(r.step))
            t = t[tuple(idx)]
            return torch.flip(t, dims=[dim]) if r.step < 0 else t
        d = cut(cut(self.data, rx, -1), ry, -2)
        if sel_c is not None and self.data.dim() == 3:
            d = cut(d, _as_range(sel_c, self.data.shape[0]), 0)
        _, new_wcs = slice_geometry(self.shape, self.wcs, rx, ry)
        return Enmap(d.contiguous(), new_wcs)


    # ---- broadcasting (enmap.jl:86-174): elementwise arithmetic keeps the first operand's WCS
    #      (combine(x::AbstractWCSTransform, y) = copy(x), enmap.jl:107-110); the arithmetic itself is the
    #      parent array's (torch), exactly as the reference defers to the parent's BroadcastStyle
    def _binary(self, other, op, reflected=False):
        o = other.data if isinstance(other, Enmap) else other
        res = op(o, self.data) if reflected else op(self.data, o)
        return Enmap(res, self.wcs)

    def __add__(self, o): return self._binary(o, torch.add)
    def __radd__(self, o): return self._binary(o, torch.add, True)
    def __sub__(self, o): return self._binary(o, torch.sub)
    def __rsub__(self, o): return self._binary(o, torch.sub, True)
    def __mul__(self, o): return self._binary(o, torch.mul)
    def __rmul__(self, o): return self._binary(o, torch.mul, True)
    def __truediv__(self, o): return self._binary(o, torch.div)
    def __rtruediv__(self, o): return self._binary(o, torch.div, True)
    def __pow__(self, o): return self._binary(o, torch.pow)
    def __neg__(self): return Enmap(-self.data, self.wcs)

    def assign(self, other):
        """m .= other (broadcasted assignment keeps m's WCS; test_enmap.jl:83-88)."""
        self.data.copy_(other.data if isinstance(other, Enmap) else other)
        return self

    def sum(self):
        return self.data.sum()

    # ---- pad (car_proj.jl:280-327): zero padding with the WCS shifted (center) or kept (corner)
    def pad(self, npix_ra, npix_dec=None, mode="center"):
        from .geometry import pad_geometry
        npix_dec = npix_ra if npix_dec is None else npix_dec
        new_shape, new_wcs = pad_geometry(self.shape, self.wcs, npix_ra, npix_dec, mode)
        lead = tuple(self.data.shape[:-2])
        arr = torch.zeros(lead + (new_shape[1], new_shape[0]), dtype=self.data.dtype, device=self.data.device)
        ny, nx = self.data.shape[-2:]
        if mode == "center":
            arr[..., npix_dec:npix_dec + ny, npix_ra:npix_ra + nx] = self.data
        else:
            arr[..., :ny, :nx] = self.data
        return Enmap(arr, new_wcs)


def getwcs(x):
    """enmap.jl:24-25"""
    return x.wcs if isinstance(x, Enmap) else NoWCS()


def _geom(m_or_geom):
    """Accept an Enmap or a (shape, wcs) pair -- the two call styles of enmap_ops.jl:60-66."""
    if isinstance(m_or_geom, Enmap):
        return m_or_geom.shape, m_or_geom.wcs
    shape, wcs = m_or_geom
    return tuple(shape), wcs


__all__ = ["Enmap", "NoWCS", "getwcs", "JlRange", "AbstractCARWCS"]
