"""FITS image interop for CAR maps (mirrors read_map / write_map, /root/reference/src/enmap.jl:198-237).

The on-disk format either side of the hot path: a primary image HDU, BITPIX -64 (or -32 on read), big-endian,
NAXIS1 = RA fastest, WCS cards CTYPE `RA---CAR` / `DEC--CAR` in degrees (header of the reference's
test/data/test.fits).  The header is parsed on the host; the data block is copied raw to HBM and byte-swapped
there (pxl_fits_decode_f64 / pxl_fits_encode_f64).  Only what Pixell.jl's own read_map/write_map need is
implemented -- no extensions, no scaling keywords, no compressed images.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .enmap import Enmap
from .wcs import CarClenshawCurtis

BLOCK = 2880
CARD = 80

# getunit(), arbitrary_wcs.jl:16-34
_UNITS = {"deg": math.pi / 180, "rad": 1.0, "arcmin": math.pi / 180 / 60, "arcsec": math.pi / 180 / 60 / 60,
          "mas": math.pi / 180 / 60 / 60 / 1000}


def _parse_value(raw: str):
    v = raw.split("/")[0].strip() if not raw.strip().startswith("'") else raw
    v = v.strip()
    if v.startswith("'"):
        end = v.find("'", 1)
        while end != -1 and v[end:end + 2] == "''":
            end = v.find("'", end + 2)
        return v[1:end].rstrip()
    if v in ("T", "F"):
        return v == "T"
    try:
        return int(v)
    except ValueError:
        return float(v.replace("D", "E"))


def read_header(path):
    """Primary-HDU header as an ordered dict, plus the byte offset of the data block."""
    cards = {}
    offset = 0
    with open(path, "rb") as f:
        done = False
        while not done:
            block = f.read(BLOCK)
            if len(block) < BLOCK:
                raise ValueError("truncated FITS header in %s" % path)
            offset += BLOCK
            for i in range(0, BLOCK, CARD):
                card = block[i:i + CARD].decode("ascii", "replace")
                key = card[:8].strip()
                if key == "END":
                    done = True
                    break
                if card[8:10] == "= " and key:
                    cards[key] = _parse_value(card[10:])
    if not cards.get("SIMPLE", False):
        raise ValueError("%s is not a simple FITS file" % path)
    return cards, offset


def wcs_from_header(h):
    """convert(CarClenshawCurtis{Float64}, WCS.from_header(...)) -- enmap.jl:213-219 (trim=true branch)."""
    if h.get("CTYPE1") != "RA---CAR" or h.get("CTYPE2") != "DEC--CAR":      # the reference @asserts the same
        raise AssertionError("only RA---CAR / DEC--CAR maps are supported (got %r, %r)" % (h.get("CTYPE1"), h.get("CTYPE2")))
    cu1, cu2 = h.get("CUNIT1", "deg"), h.get("CUNIT2", "deg")
    if cu1 != cu2:
        raise AssertionError("Units of RA and DEC must be the same.")
    unit = _UNITS.get(cu1, math.pi / 180)                                     # unknown unit: assume degrees (:32-33)
    return CarClenshawCurtis((float(h.get("CDELT1", 1.0)), float(h.get("CDELT2", 1.0))),
                             (float(h.get("CRPIX1", 0.0)), float(h.get("CRPIX2", 0.0))),
                             (float(h.get("CRVAL1", 0.0)), float(h.get("CRVAL2", 0.0))), unit)


def read_map(path, device="cuda", sel=None, verbose=False, dtype=None):
    """read_map(path; sel) -> Enmap on the device.  sel = (sel_x, sel_y[, sel_c]) with the 1-based inclusive
    selections of geometry.slice_geometry (e.g. ((11, 20), (21, 40), (1, 2)) for 11:20, 21:40, 1:2).
    The element type follows the file (BITPIX -64 -> Float64, -32 -> Float32, like the reference's read);
    dtype=torch.float64 widens a Float32 file on the device."""
    h, offset = read_header(path)
    bitpix = h["BITPIX"]
    if bitpix not in (-64, -32):
        raise ValueError("BITPIX %d not supported (only -64 and -32 image HDUs)" % bitpix)
    naxis = h["NAXIS"]
    if naxis not in (2, 3):
        raise ValueError("NAXIS %d not supported" % naxis)
    dims = [h["NAXIS%d" % (k + 1)] for k in range(naxis)]                      # (nx, ny[, nc]) = Julia shape
    n = int(np.prod(dims))
    raw = np.memmap(path, dtype=np.uint8, mode="r", offset=offset, shape=(n * (-bitpix // 8),))
    dev = torch.device(device)
    d_raw = torch.from_numpy(np.array(raw)).to(dev)            # one host copy of the data block, then H2D
    want = dtype if dtype is not None else (torch.float64 if bitpix == -64 else torch.float32)
    out = torch.empty(tuple(reversed(dims)), dtype=want, device=dev)
    with torch.cuda.device(dev):
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if want == torch.float32:
            if bitpix != -32:
                raise ValueError("a BITPIX -64 file cannot be read as Float32 without loss; read it as Float64")
            _lib.check(_lib.load().pxl_fits_swap_f32(C.c_void_p(d_raw.data_ptr()), C.c_void_p(out.data_ptr()), n, s))
        else:
            _lib.check(_lib.load().pxl_fits_decode_f64(C.c_void_p(d_raw.data_ptr()), C.c_void_p(out.data_ptr()), n, bitpix, s))
    wcs = wcs_from_header(h)
    # IAU <-> COSMO (enmap.jl:178-195,206-211): flip U (third Stokes plane) when the file says POLCCONV = IAU
    if "STOKES" in [v for v in h.values() if isinstance(v, str)] and h.get("POLCCONV", "COSMO") == "IAU" and naxis == 3:
        if verbose:
            print("convert to IAU: flip U")
        if dims[2] >= 3:
            out[2].neg_()
    m = Enmap(out, wcs)
    if sel is not None:
        m = m.getindex(*sel)
    return m


def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, int):
        v = "%20d" % value
    elif isinstance(value, float):
        v = "%20s" % repr(float(value)).upper().replace("E+", "E")
    else:
        v = "'%-8s'" % value
        v = "%-20s" % v
    card = "%-8s= %s" % (key, v)
    if comment:
        card += " / " + comment
    return ("%-80s" % card)[:80]


def write_map(path, m: Enmap):
    """write_map(fname, emap) -- enmap.jl:225-237: BITPIX -64 primary HDU + the CAR WCS cards, degrees."""
    data = m.data
    if data.dtype not in (torch.float64, torch.float32) or not data.is_contiguous():
        raise TypeError("write_map needs a contiguous float64 or float32 map")
    f32 = data.dtype == torch.float32
    shape = m.shape
    wcs = m.wcs
    cards = [_card("SIMPLE", True, "file does conform to FITS standard"),
             _card("BITPIX", -32 if f32 else -64, "number of bits per data pixel"),
             _card("NAXIS", len(shape), "number of data axes")]
    for k, nk in enumerate(shape):
        cards.append(_card("NAXIS%d" % (k + 1), int(nk), "length of data axis %d" % (k + 1)))
    cards.append(_card("EXTEND", True))
    scale = wcs.unit / (math.pi / 180)                                         # header is written in degrees
    cards += [_card("WCSAXES", 2), _card("CRPIX1", wcs.crpix[0]), _card("CRPIX2", wcs.crpix[1]),
              _card("CDELT1", wcs.cdelt[0] * scale), _card("CDELT2", wcs.cdelt[1] * scale),
              _card("CUNIT1", "deg"), _card("CUNIT2", "deg"), _card("CTYPE1", "RA---CAR"), _card("CTYPE2", "DEC--CAR"),
              _card("CRVAL1", wcs.crval[0] * scale), _card("CRVAL2", wcs.crval[1] * scale), "%-80s" % "END"]
    header = "".join(cards)
    header += " " * (-len(header) % BLOCK)
    n = data.numel()
    raw = torch.empty(n, dtype=torch.int32 if f32 else torch.int64, device=data.device)
    with torch.cuda.device(data.device):
        s = C.c_void_p(torch.cuda.current_stream(data.device).cuda_stream)
        if f32:
            _lib.check(_lib.load().pxl_fits_swap_f32(C.c_void_p(data.data_ptr()), C.c_void_p(raw.data_ptr()), n, s))
        else:
            _lib.check(_lib.load().pxl_fits_encode_f64(C.c_void_p(data.data_ptr()), C.c_void_p(raw.data_ptr()), n, s))
    payload = raw.cpu().numpy().tobytes()
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(payload)
        f.write(b"\0" * (-len(payload) % BLOCK))
