"""FITS image interop for CAR maps (mirrors read_map / write_map, /root/reference/src/enmap.jl:198-237).

The on-disk format either side of the hot path: a primary image HDU, BITPIX -64 (or -32 on read), big-endian,
NAXIS1 = RA fastest, WCS cards CTYPE `RA---CAR` / `DEC--CAR` in degrees (header of the reference's
test/data/test.fits).  The header is parsed on the host; the data block is copied raw to HBM and byte-swapped
there (pxl_fits_decode_f64 / pxl_fits_encode_f64).  Only what Pixell.jl's own read_map/write_map need is
implemented -- no extensions, no scaling keywords, no compressed images.
"""
import ctypes as C
import math
import os

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


# ---- streaming between the file and HBM ------------------------------------------------------------
# A 43200 x 21601 x 3 Float64 map is 22.4 GB on disk: the data block moves in chunks through two pinned host
# buffers, so the host never holds more than 2 x CHUNK bytes of it, the H2D (or D2H) copy of one chunk runs on a
# copy stream while the file system serves the next, and the byte swap happens on the device.

def _chunk_bytes():
    return max(4096, int(float(os.environ.get("PXL_FITS_CHUNK_MB", "128")) * (1 << 20)) & ~4095)


def _io_threads():
    """Host threads that move one chunk between the file and the pinned buffer (os.preadv / os.pwritev release the
    GIL).  Measured on a 5.6 GB map, page-cache-warm file (profiles/r02_fits_io.json): read_map 21.6 GB/s with one
    thread, 45-48 with four (PCIe Gen5 x16 territory), 32-41 with eight; write_map stays at ~10.5 GB/s whatever the
    thread count (fresh page-cache pages).  PXL_FITS_THREADS overrides."""
    v = os.environ.get("PXL_FITS_THREADS")
    return max(1, int(v)) if v else max(1, min(4, (os.cpu_count() or 2) // 2))


def _parallel_io(pool, fn, fd, view, offset, nbytes, nthreads):
    """fn = os.preadv or os.pwritev on [offset, offset + nbytes) of fd against view[:nbytes], split over the pool in
    4 KiB-aligned pieces.  Returns the number of bytes moved."""
    if nthreads <= 1 or nbytes < (8 << 20):
        piece = [(0, nbytes)]
    else:
        step = ((nbytes + nthreads - 1) // nthreads + 4095) & ~4095
        piece = [(a, min(step, nbytes - a)) for a in range(0, nbytes, step)]

    def run(a, m):
        done = 0
        while done < m:                      # short reads / writes are legal
            k = fn(fd, [view[a + done:a + m]], offset + a + done)
            if k <= 0:
                break
            done += k
        return done

    if len(piece) == 1:
        return run(*piece[0])
    return sum(f.result() for f in [pool.submit(run, a, m) for a, m in piece])


class _Stager:
    """Two pinned host buffers + two device staging buffers + a copy stream (double buffering)."""

    def __init__(self, dev, nbytes):
        self.dev = dev
        self.nbytes = nbytes
        self.host = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self.view = [memoryview(h.numpy()) for h in self.host]
        self.raw = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.busy = [None, None]                  # event after which buffer b may be reused


def _spans_to_chunks(spans, esz, chunk_elems):
    """Split contiguous runs (file_offset, n_elements, dst_element_offset) into chunks of <= chunk_elems."""
    for off, n, dst in spans:
        done = 0
        while done < n:
            m = min(chunk_elems, n - done)
            yield off + done * esz, m, dst + done
            done += m


def _decode(raw_ptr, out_ptr, n, bitpix, want, stream):
    lib = _lib.load()
    if want == torch.float32:
        _lib.check(lib.pxl_fits_swap_f32(C.c_void_p(raw_ptr), C.c_void_p(out_ptr), n, stream))
    else:
        _lib.check(lib.pxl_fits_decode_f64(C.c_void_p(raw_ptr), C.c_void_p(out_ptr), n, bitpix, stream))


def _read_spans(path, spans, out, bitpix, dev):
    """File -> pinned host -> HBM -> big-endian decode into the flat tensor `out` (element offsets in spans)."""
    esz = -bitpix // 8
    total = sum(n for _, n, _ in spans)
    if total == 0:
        return
    chunk = min(_chunk_bytes() // esz, max(n for _, n, _ in spans))
    st = _Stager(dev, chunk * esz)
    cur = torch.cuda.current_stream(dev)
    s = C.c_void_p(cur.cuda_stream)
    osz = out.element_size()
    from concurrent.futures import ThreadPoolExecutor
    nthr = _io_threads()
    with open(path, "rb", buffering=0) as f, torch.cuda.device(dev), ThreadPoolExecutor(max_workers=nthr) as pool:
        for i, (off, n, dst) in enumerate(_spans_to_chunks(spans, esz, chunk)):
            b = i & 1
            if st.busy[b] is not None:
                st.busy[b].synchronize()          # the copy and the decode that used this pair are done
            got = _parallel_io(pool, os.preadv, f.fileno(), st.view[b], off, n * esz, nthr)
            if got != n * esz:
                raise ValueError("truncated FITS data block in %s" % path)
            with torch.cuda.stream(st.copy_stream):
                st.raw[b][:n * esz].copy_(st.host[b][:n * esz], non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(st.copy_stream)
            cur.wait_event(copied)
            _decode(st.raw[b].data_ptr(), out.data_ptr() + dst * osz, n, bitpix, out.dtype, s)
            st.busy[b] = torch.cuda.Event()
            st.busy[b].record(cur)
    for ev in st.busy:
        if ev is not None:
            ev.synchronize()                      # the staging buffers die with this call


def _file_layout(path):
    h, offset = read_header(path)
    bitpix = h["BITPIX"]
    if bitpix not in (-64, -32):
        raise ValueError("BITPIX %d not supported (only -64 and -32 image HDUs)" % bitpix)
    naxis = h["NAXIS"]
    if naxis not in (2, 3):
        raise ValueError("NAXIS %d not supported" % naxis)
    dims = [int(h["NAXIS%d" % (k + 1)]) for k in range(naxis)]                  # (nx, ny[, nc]) = Julia shape
    need = offset + int(np.prod(dims)) * (-bitpix // 8)
    if os.path.getsize(path) < need:
        raise ValueError("truncated FITS data block in %s" % path)
    return h, offset, bitpix, dims


def _want_dtype(bitpix, dtype):
    want = dtype if dtype is not None else (torch.float64 if bitpix == -64 else torch.float32)
    if want == torch.float32 and bitpix != -32:
        raise ValueError("a BITPIX -64 file cannot be read as Float32 without loss; read it as Float64")
    if want not in (torch.float32, torch.float64):
        raise TypeError("maps are Float64 or Float32")
    return want


def _resolve_polcconv(h, out, comps, verbose, mode="reference"):
    """IAU -> COSMO sign flip of a polarisation map, enmap.jl:178-195 called from read_map (:203-209): only when some
    header value is "STOKES" and POLCCONV == "IAU", and only on the axis i whose CTYPEi == "STOKES".
    mode="reference" (default) reproduces what the reference DOES: `signs = ones(1, 1, 3); signs[signs_size] .= -1`
    with signs_size = [1, 1, 3] is linear indexing, so planes 1 AND 3 (I and U) change sign; mode="u_only" flips only
    U (the physical IAU <-> COSMO convention, probably what was meant).  comps = 0-based file component of each plane
    of `out` (a selection along the Stokes axis selects the same entries of `signs`, :189-190)."""
    if "STOKES" not in [v for v in h.values() if isinstance(v, str)] or h.get("POLCCONV", "COSMO") != "IAU":
        return
    if mode not in ("reference", "u_only"):
        raise ValueError("polcconv must be 'reference' or 'u_only'")
    naxis = int(h["NAXIS"])
    for i in range(1, naxis + 1):
        if h.get("CTYPE%d" % i, "") != "STOKES":
            continue
        if i != 3 or out.dim() != 3:
            raise NotImplementedError("STOKES on axis %d: only the component axis (3) of an (nx, ny, nc) map is supported" % i)
        if int(h["NAXIS3"]) != 3:
            raise ValueError("POLCCONV=IAU with %d Stokes planes: the reference's (1, 1, 3) sign array does not broadcast "
                             "(DimensionMismatch there)" % int(h["NAXIS3"]))
        if verbose:
            print("convert to IAU: flip U in axis %d" % i)
        flip = (0, 2) if mode == "reference" else (2,)
        if mode == "reference" and 0 in comps:
            import warnings
            warnings.warn("POLCCONV=IAU: following the reference (enmap.jl:186-187, `signs[signs_size] .= -1` is a linear "
                          "index), Stokes I is negated as well as U; pass polcconv='u_only' to flip U alone", stacklevel=3)
        for plane, c in enumerate(comps):
            if c in flip:
                out[plane].neg_()


def read_map_rows(path, row0, nrows, device="cuda", comps=None, dtype=None, verbose=False, polcconv="reference"):
    """Rows [row0, row0 + nrows) (0-based) of every component of a FITS map, straight into HBM: the declination
    strip of one rank of a sharded job (rows are contiguous on disk, so only those bytes are read).  Returns
    (tensor ([nc,] nrows, nx), full Julia shape, WCS of the FULL map) -- windows into the full geometry, like
    the sharded operator wants them (sharding.py)."""
    h, offset, bitpix, dims = _file_layout(path)
    nx, ny = dims[0], dims[1]
    nc_file = dims[2] if len(dims) == 3 else 1
    if row0 < 0 or nrows < 0 or row0 + nrows > ny:
        raise ValueError("rows [%d, %d) are outside the map (%d rows)" % (row0, row0 + nrows, ny))
    comps = list(range(nc_file)) if comps is None else [int(c) for c in comps]
    if any(c < 0 or c >= nc_file for c in comps):
        raise ValueError("component outside the file")
    dev = torch.device(device)
    want = _want_dtype(bitpix, dtype)
    esz = -bitpix // 8
    shape = (len(comps), nrows, nx) if len(dims) == 3 else (nrows, nx)
    out = torch.empty(shape, dtype=want, device=dev)
    spans = [(offset + (c * ny + row0) * nx * esz, nrows * nx, k * nrows * nx) for k, c in enumerate(comps)]
    # neighbouring components of a full-height read are one contiguous run
    merged = []
    for sp in spans:
        if merged and merged[-1][0] + merged[-1][1] * esz == sp[0] and merged[-1][2] + merged[-1][1] == sp[2]:
            merged[-1] = (merged[-1][0], merged[-1][1] + sp[1], merged[-1][2])
        else:
            merged.append(sp)
    _read_spans(path, merged, out, bitpix, dev)
    _resolve_polcconv(h, out, comps, verbose, polcconv)
    return out, tuple(dims), wcs_from_header(h)


def read_map(path, device="cuda", sel=None, verbose=False, dtype=None, polcconv="reference"):
    """read_map(path; sel) -> Enmap on the device.  sel = (sel_x, sel_y[, sel_c]) with the 1-based inclusive
    selections of geometry.slice_geometry (e.g. ((11, 20), (21, 40), (1, 2)) for 11:20, 21:40, 1:2).
    The element type follows the file (BITPIX -64 -> Float64, -32 -> Float32, like the reference's read);
    dtype=torch.float64 widens a Float32 file on the device.  The data block is streamed (see _read_spans); a
    selection of whole rows (all of RA, a unit-step DEC range, a unit-step component range) reads only its bytes."""
    from .geometry import _as_range, slice_geometry
    h, offset, bitpix, dims = _file_layout(path)
    nx, ny = dims[0], dims[1]
    nc_file = dims[2] if len(dims) == 3 else 1
    rx = ry = rc = None
    if sel is not None:
        rx, ry = _as_range(sel[0], nx), _as_range(sel[1], ny)
        rc = _as_range(sel[2], nc_file) if len(sel) > 2 and len(dims) == 3 else None
    whole_rows = sel is not None and rx.first == 1 and rx.step == 1 and rx.length == nx and ry.step == 1 and ry.length > 0 \
        and (rc is None or (rc.step == 1 and rc.length > 0))
    if sel is None or whole_rows:
        row0, nrows = (0, ny) if sel is None else (ry.first - 1, ry.length)
        comps = None if rc is None else list(range(rc.first - 1, rc.last))
        data, _, wcs = read_map_rows(path, row0, nrows, device=device, comps=comps, dtype=dtype, verbose=verbose, polcconv=polcconv)
        if sel is not None:
            _, wcs = slice_geometry((nx, ny), wcs, rx, ry)
        return Enmap(data, wcs)
    data, _, wcs = read_map_rows(path, 0, ny, device=device, dtype=dtype, verbose=verbose, polcconv=polcconv)
    return Enmap(data, wcs).getindex(*sel)


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


def write_map(path, m: Enmap, extra_cards=()):
    """write_map(fname, emap) -- enmap.jl:225-237: BITPIX -64 primary HDU + the CAR WCS cards, degrees.
    extra_cards: (key, value) pairs appended to the header (e.g. CTYPE3 / POLCCONV of a polarisation map; the
    reference writes none)."""
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
              _card("CRVAL1", wcs.crval[0] * scale), _card("CRVAL2", wcs.crval[1] * scale)]
    cards += [_card(k, v) for k, v in extra_cards] + ["%-80s" % "END"]
    header = "".join(cards)
    header += " " * (-len(header) % BLOCK)
    n = data.numel()
    esz = 4 if f32 else 8
    dev = data.device
    flat = data.reshape(-1)
    lib = _lib.load()
    from concurrent.futures import ThreadPoolExecutor
    nthr = _io_threads()
    hbytes = header.encode("ascii")
    with open(path, "wb", buffering=0) as f, ThreadPoolExecutor(max_workers=nthr) as pool:
        f.write(hbytes)
        fpos = len(hbytes)                          # file offset of the next data byte
        if n:
            chunk = min(_chunk_bytes() // esz, n)
            st = _Stager(dev, chunk * esz)
            cur = torch.cuda.current_stream(dev)
            s = C.c_void_p(cur.cuda_stream)
            pending = None                          # (buffer, bytes, event) of the chunk on its way to the host
            with torch.cuda.device(dev):
                for i, (_, m, at) in enumerate(_spans_to_chunks([(0, n, 0)], esz, chunk)):
                    b = i & 1
                    src = flat.data_ptr() + at * esz
                    if f32:
                        _lib.check(lib.pxl_fits_swap_f32(C.c_void_p(src), C.c_void_p(st.raw[b].data_ptr()), m, s))
                    else:
                        _lib.check(lib.pxl_fits_encode_f64(C.c_void_p(src), C.c_void_p(st.raw[b].data_ptr()), m, s))
                    encoded = torch.cuda.Event()
                    encoded.record(cur)
                    with torch.cuda.stream(st.copy_stream):
                        st.copy_stream.wait_event(encoded)
                        st.host[b][:m * esz].copy_(st.raw[b][:m * esz], non_blocking=True)
                        arrived = torch.cuda.Event()
                        arrived.record(st.copy_stream)
                    if pending is not None:         # write chunk i-1 while chunk i is encoded and copied
                        pb, pbytes, pev = pending
                        pev.synchronize()
                        if _parallel_io(pool, os.pwritev, f.fileno(), st.view[pb], fpos, pbytes, nthr) != pbytes:
                            raise OSError("short write to %s" % path)
                        fpos += pbytes
                    pending = (b, m * esz, arrived)
                pb, pbytes, pev = pending
                pev.synchronize()
                if _parallel_io(pool, os.pwritev, f.fileno(), st.view[pb], fpos, pbytes, nthr) != pbytes:
                    raise OSError("short write to %s" % path)
                fpos += pbytes
        os.pwrite(f.fileno(), b"\0" * (-(n * esz) % BLOCK), fpos)
