"""ctypes binding of libpixell_hip.so -- the C ABI declared in include/pixell_hip.h.

There is NO CPU fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PXL_LIB_PATH", os.path.join(_HERE, "libpixell_hip.so"))   # override: A/B builds

WRAP_NONE, WRAP_REWIND, WRAP_UNWIND = 0, 1, 2
FORM_RECIP, FORM_DIV, FORM_RECIP_AV = 0, 1, 2


class PixellHipError(RuntimeError):
    """A libpixell_hip.so entry returned a non-zero code (the reference raises Julia exceptions)."""

    def __init__(self, code, message):
        super().__init__("libpixell_hip error %d: %s" % (code, message))
        self.code = code


class CarWCSStruct(C.Structure):
    """struct pxl_car_wcs -- CarClenshawCurtis{Float64} layout, projections/car_proj.jl:7-12."""
    _fields_ = [("cdelt", C.c_double * 2), ("crpix", C.c_double * 2), ("crval", C.c_double * 2),
                ("unit", C.c_double)]


_P = C.c_void_p
_I64 = C.c_int64
_WP = C.POINTER(CarWCSStruct)
_SHP = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/pixell_hip.h declares
SIGNATURES = {
    "pxl_version": (C.c_int, []),
    "pxl_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "pxl_device_count": (C.c_int, []),
    "pxl_release_scratch": (C.c_int, []),
    "pxl_pix2sky_car_f64": (C.c_int, [_WP, _I64, _P, _P, C.c_int, _P]),
    "pxl_rewind_f64": (C.c_int, [_P, _I64, C.c_double, C.c_double, _P]),
    "pxl_unwind_f64": (C.c_int, [_P, _I64, C.c_int, C.c_double, C.c_double, _P]),
    "pxl_pix2sky_car_soa_f64": (C.c_int, [_WP, _I64, _P, _P, _P, _P, C.c_int, _P]),
    "pxl_sky2pix_car_f64": (C.c_int, [_WP, _SHP, _I64, _P, _P, C.c_int, C.c_int, _P]),
    "pxl_sky2pix_car_soa_f64": (C.c_int, [_WP, _SHP, _I64, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "pxl_posmap_car_f64": (C.c_int, [_WP, _SHP, _I64, _I64, _P, _P, C.c_int, _P]),
    "pxl_pixareamap_car_f64": (C.c_int, [_WP, _SHP, _I64, _I64, _P, _P]),
    "pxl_sky2pix_tan_f64": (C.c_int, [_WP, _I64, _P, _P, _P, _P, _P]),
    "pxl_pix2sky_tan_f64": (C.c_int, [_WP, _I64, _P, _P, _P, _P, _P]),
    "pxl_posmap_tan_f64": (C.c_int, [_WP, _SHP, _I64, _I64, _P, _P, _P]),
    "pxl_reproject_plan_create": (C.c_int, [_WP, _SHP, _I64, _I64, _WP, _SHP, _I64, _I64, C.POINTER(_P)]),
    "pxl_reproject_execute": (C.c_int, [_P, _P, _P, _P]),
    "pxl_reproject_build_tables": (C.c_int, [_P, _P]),
    "pxl_reproject_execute_rows": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "pxl_reproject_execute_f32": (C.c_int, [_P, _P, _P, _P]),
    "pxl_reproject_execute_rows_f32": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "pxl_reproject_plan_src_rows": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "pxl_reproject_plan_rows_covered": (C.c_int, [_P, _I64, _I64, C.POINTER(_I64), C.POINTER(_I64)]),
    "pxl_reproject_plan_set_variant": (C.c_int, [_P, C.c_int]),
    "pxl_reproject_plan_destroy": (C.c_int, [_P]),
    "pxl_reproject_sharded_step_f64": (C.c_int, [_P, _P, _P, _I64, _I64, _P, C.c_int, _P, C.c_int, _P, _P]),
    "pxl_reproject_sharded_step_f32": (C.c_int, [_P, _P, _P, _I64, _I64, _P, C.c_int, _P, C.c_int, _P, _P]),
    "pxl_comm_unique_id": (C.c_int, [_P]),
    "pxl_comm_init_rank": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "pxl_comm_destroy": (C.c_int, [_P]),
    "pxl_comm_backend": (C.c_char_p, []),
    "pxl_reproject_car_bilinear_f64": (C.c_int, [_WP, _SHP, _P, _WP, _SHP, _P, _P]),
    "pxl_reproject_generic_bilinear_f64": (C.c_int, [_WP, C.c_int, _SHP, _P, _WP, C.c_int, _SHP, _P, _P]),
    "pxl_reproject_generic_last_tiles": (C.c_int, [C.POINTER(_I64), C.POINTER(_I64), _P]),
    "pxl_generic_plan_create": (C.c_int, [_WP, C.c_int, _SHP, _WP, C.c_int, _SHP, _P, C.POINTER(_P)]),
    "pxl_generic_plan_execute": (C.c_int, [_P, _I64, _P, _P, _P]),
    "pxl_generic_plan_tiles": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "pxl_generic_plan_destroy": (C.c_int, [_P]),
    "pxl_sample_car_bilinear_f64": (C.c_int, [_WP, _SHP, _P, _I64, _I64, _I64, _P, _P, _P]),
    "pxl_sample_car_bilinear_f32": (C.c_int, [_WP, _SHP, _P, _I64, _I64, _I64, _P, _P, _P]),
    "pxl_sample_pairs_elems": (_I64, [_SHP, _I64]),
    "pxl_sample_build_pairs_f64": (C.c_int, [_SHP, _P, _I64, _P, _P]),
    "pxl_sample_build_pairs_f32": (C.c_int, [_SHP, _P, _I64, _P, _P]),
    "pxl_sample_car_bilinear_pairs_f64": (C.c_int, [_WP, _SHP, _P, _I64, _I64, _I64, _P, _P, _P]),
    "pxl_sample_car_bilinear_pairs_f32": (C.c_int, [_WP, _SHP, _P, _I64, _I64, _I64, _P, _P, _P]),
    "pxl_fits_decode_f64": (C.c_int, [_P, _P, _I64, C.c_int, _P]),
    "pxl_fits_encode_f64": (C.c_int, [_P, _P, _I64, _P]),
    "pxl_fits_swap_f32": (C.c_int, [_P, _P, _I64, _P]),
    "pxl_mem_probe_pair": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_float), _P]),
    "pxl_mem_pair_alloc": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, _P, _P]),
    "pxl_mem_pair_free": (C.c_int, [_P]),
    "pxl_mem_alloc_placed": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(_P), _P, _P]),
    "pxl_mem_free": (C.c_int, [_P]),
    "pxl_fill_random_f64": (C.c_int, [_P, _I64, C.c_uint64, C.c_uint64, C.c_int, _P]),
    "pxl_fill_sphere_points_f64": (C.c_int, [_P, _I64, C.c_uint64, C.c_uint64, _P]),
}

class MemPair(C.Structure):
    """struct pxl_mem_pair (include/pixell_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("arena", C.c_void_p), ("src_alloc", C.c_void_p),
                ("arena_bytes", C.c_uint64), ("src_offset", C.c_uint64), ("dst_offset", C.c_uint64),
                ("classes", C.c_int32), ("dst_two_classes", C.c_int32), ("src_own_class", C.c_int32), ("probes", C.c_int32),
                ("separate_tried", C.c_int32), ("reserved_", C.c_int32)]


class MemPlacedInfo(C.Structure):
    """struct pxl_mem_placed_info (include/pixell_hip.h)."""
    _fields_ = [("tries", C.c_int32), ("probes", C.c_int32), ("two_classes", C.c_int32), ("minor_share_pct", C.c_int32),
                ("ballast_bytes", C.c_uint64)]


_lib = None


def load():
    """Load libpixell_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libpixell_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    load().pxl_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise PixellHipError(rc, last_error())


class HaloXfer(C.Structure):
    """struct pxl_halo_xfer"""
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_int32), ("row0", C.c_int64), ("nrows", C.c_int64)]


def xfer_arr(items):
    """(peer, row_lo, row_hi) triples -> a pxl_halo_xfer array (absolute source rows)."""
    arr = (HaloXfer * max(1, len(items)))()
    for k, (peer, lo, hi) in enumerate(items):
        arr[k].peer, arr[k].reserved, arr[k].row0, arr[k].nrows = int(peer), 0, int(lo), int(hi - lo)
    return arr


def shape_arr(vals):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])
