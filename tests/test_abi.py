"""The drop-in boundary: libpixell_hip.so loads, exports every symbol include/pixell_hip.h declares, and
the product never reaches into the oracle.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "pixell_hip.h")
PKG = os.path.join(ROOT, "pixell.jl_amd")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pxl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entries():
    syms = declared_symbols()
    for must in ("pxl_pix2sky_car_f64", "pxl_sky2pix_car_f64", "pxl_posmap_car_f64", "pxl_reproject_plan_create",
                 "pxl_reproject_execute", "pxl_sample_car_bilinear_f64", "pxl_last_error", "pxl_version"):
        assert must in syms


def test_library_exports_every_declared_symbol(pj):
    lib = ctypes.CDLL(pj.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), "libpixell_hip.so does not export %s" % name
    # and the Python binding table covers exactly the header
    assert sorted(pj._lib.SIGNATURES) == declared_symbols()


def _c_prototypes():
    """name -> (return type, [parameter types]) from include/pixell_hip.h, each type reduced to a canonical spelling:
    qualifiers and parameter names dropped, `T name[k]` read as `T*`."""
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {}
    for ret, name, params in re.findall(r"^\s*([A-Za-z_][\w \t\*]*?)\s*\b(pxl_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", header, flags=re.M):
        def canon(decl, is_param):
            decl = decl.strip()
            stars = decl.count("*") + len(re.findall(r"\[[^\]]*\]", decl))
            decl = re.sub(r"\[[^\]]*\]", "", decl).replace("*", " ")
            toks = [t for t in decl.split() if t not in ("const", "struct")]
            if is_param and len(toks) > 1:
                toks = toks[:-1]                      # the parameter's name
            return " ".join(toks) + "*" * stars
        plist = [] if params.strip() in ("", "void") else [canon(q, True) for q in params.split(",")]
        protos[name] = (canon(ret, False), plist)
    return protos


# what each Julia ccall type may stand for on the C side (opaque handles travel as Ptr{Cvoid})
_JULIA_TO_C = {
    "Ref{CarWCS}": {"pxl_car_wcs*"},
    "Int64": {"int64_t"}, "UInt64": {"uint64_t"}, "Cint": {"int"}, "Cdouble": {"double"}, "Csize_t": {"size_t"},
    "Ptr{Cdouble}": {"double*"}, "Ptr{Cfloat}": {"float*"}, "Ptr{Int64}": {"int64_t*"},
    "Ptr{Cvoid}": {"void*", "pxl_reproject_plan*", "pxl_generic_plan*", "pxl_mem_pair*"},
    "Ref{MemPlacedInfo}": {"pxl_mem_placed_info*"},
    "Ptr{Ptr{Cvoid}}": {"void**", "pxl_reproject_plan**", "pxl_generic_plan**"},
    "Ptr{UInt8}": {"char*", "void*"},
    "Ptr{HaloXfer}": {"pxl_halo_xfer*"},
    "Cstring": {"char*"},
}


def _split_julia_types(argtypes):
    out, depth, cur = [], 0, ""
    for ch in argtypes:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_julia_binding_matches_the_c_prototypes(pj):
    """julia/PixellHIP.jl (the binding a Pixell.jl maintainer would add) must ccall entry points that the
    header declares and the library exports, and every ccall's return type and argument TYPES must be the C prototype's
    (Ref{CarWCS} <-> const pxl_car_wcs*, Int64 <-> int64_t, Cint <-> int, Ptr{Cdouble} <-> double*, ...): a Cint where
    the C side takes an int64_t would corrupt the call silently."""
    text = open(os.path.join(ROOT, "julia", "PixellHIP.jl")).read()
    lib = ctypes.CDLL(pj.library_path())
    protos = _c_prototypes()
    assert sorted(protos) == declared_symbols()
    calls = re.findall(r"ccall\(\(:(pxl_[a-z0-9_]+), libpixell_hip\),\s*([\w{}]+),\s*\(([^)]*)\)", text, flags=re.S)
    assert len(calls) >= 28
    for name, jret, argtypes in calls:
        assert name in protos, "PixellHIP.jl calls %s, which include/pixell_hip.h does not declare" % name
        assert hasattr(lib, name)
        cret, cparams = protos[name]
        jargs = _split_julia_types(argtypes)
        assert len(jargs) == len(cparams), "%s: %d Julia argument types for %d C parameters" % (name, len(jargs), len(cparams))
        assert cret in _JULIA_TO_C[jret], "%s returns %s in C, %s in the ccall" % (name, cret, jret)
        for k, (jt, ct) in enumerate(zip(jargs, cparams)):
            assert jt in _JULIA_TO_C, "%s argument %d: unknown Julia type %s" % (name, k + 1, jt)
            assert ct in _JULIA_TO_C[jt], "%s argument %d: C takes %s, the ccall passes %s" % (name, k + 1, ct, jt)
    # the methods INTEGRATION.md's table promises exist, on the reference's own generic functions
    for sig in (r"Pixell\.pix2sky!\(shape, wcs::AbstractCARWCS, pix::DevCoords", r"Pixell\.sky2pix!\(shape, wcs::AbstractCARWCS, sky::DevCoords",
                r"Pixell\.pix2sky\(shape, wcs::AbstractCARWCS, ra_pixel::DevVector, dec_pixel::DevVector",
                r"Pixell\.sky2pix\(shape, wcs::AbstractCARWCS, ra::DevVector, dec::DevVector",
                r"Pixell\.sky2pix\(shape, wcs::Gnomonic, ra::DevVector", r"Pixell\.pix2sky\(shape, wcs::Gnomonic, ra_pixel::DevVector",
                r"posmap_device\(shape::Tuple\{Int,Int\}, wcs::Gnomonic\)", r"function reproject_generic\(",
                r"Pixell\.read_map\(path::String, ::Type\{HIPArray\}", r"Pixell\.write_map\(fname::String, emap::Enmap\{Float64,N,<:HIPArray\}\)",
                r"Pixell\.rewind\(angles::HIPArray", r"Pixell\.unwind\(angles::HIPArray", r"Pixell\.pixareamap!\("):
        assert re.search(sig, text), "PixellHIP.jl lacks the method %s" % sig


def test_julia_structs_mirror_the_c_structs():
    """CarWCS == pxl_car_wcs (7 doubles) and HaloXfer == pxl_halo_xfer (int32, int32, int64, int64), field for field."""
    text = open(os.path.join(ROOT, "julia", "PixellHIP.jl")).read()
    car = re.search(r"struct CarWCS\n(.*?)\nend", text, flags=re.S).group(1)
    assert [ln.split("::")[1].strip() for ln in car.strip().splitlines()] == ["NTuple{2,Cdouble}"] * 3 + ["Cdouble"]
    halo = re.search(r"struct HaloXfer[^\n]*\n(.*?)\nend", text, flags=re.S).group(1)
    fields = [ln.split("#")[0].split("::")[1].strip() for ln in halo.strip().splitlines()]
    assert fields == ["Int32", "Int32", "Int64", "Int64"]
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    chalo = re.search(r"typedef struct pxl_halo_xfer \{(.*?)\}", header, flags=re.S).group(1)
    assert [d.split()[0] for d in chalo.split(";") if d.strip()] == ["int32_t", "int32_t", "int64_t", "int64_t"]


def test_version_and_error_channel_without_gpu(pj):
    lib = pj.load_library()
    assert lib.pxl_version() == 100
    # argument validation happens before any HIP call: a NULL WCS is rejected with a message
    rc = lib.pxl_pix2sky_car_f64(None, 0, None, None, 0, None)
    assert rc == -22
    assert "WCS" in pj._lib.last_error()


def test_mem_pair_struct_mirrors(pj):
    """struct pxl_mem_pair: the header, the ctypes mirror and the Julia mirror list the same fields with the same types."""
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct pxl_mem_pair \{(.*?)\}", header, flags=re.S).group(1)
    cfields = [(t.replace(" ", ""), n) for t, n in re.findall(r"([\w \*]+?)\s*\b(\w+);", body)]
    assert [n for _, n in cfields] == [n for n, _ in pj._lib.MemPair._fields_]
    cmap = {"void*": ctypes.c_void_p, "uint64_t": ctypes.c_uint64, "int32_t": ctypes.c_int32}
    assert [cmap[t] for t, _ in cfields] == [t for _, t in pj._lib.MemPair._fields_]
    text = open(os.path.join(ROOT, "julia", "PixellHIP.jl")).read()
    jbody = re.search(r"mutable struct MemPair[^\n]*\n(.*?)\n    MemPair\(\)", text, flags=re.S).group(1)
    jfields = re.findall(r"^\s*(\w+)::([\w{}]+)", jbody, flags=re.M)
    jmap = {"void*": "Ptr{Cvoid}", "uint64_t": "UInt64", "int32_t": "Int32"}
    assert jfields == [(n, jmap[t]) for t, n in cfields]
    assert ctypes.sizeof(pj._lib.MemPair) == 4 * 8 + 3 * 8 + 6 * 4


def test_mem_placed_info_struct_mirrors(pj):
    """struct pxl_mem_placed_info (the default allocation policy's report): header, ctypes mirror and Julia mirror agree."""
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct pxl_mem_placed_info \{(.*?)\}", header, flags=re.S).group(1)
    cfields = [(t.replace(" ", ""), n) for t, n in re.findall(r"([\w \*]+?)\s*\b(\w+);", body)]
    assert [n for _, n in cfields] == [n for n, _ in pj._lib.MemPlacedInfo._fields_]
    cmap = {"uint64_t": ctypes.c_uint64, "int32_t": ctypes.c_int32}
    assert [cmap[t] for t, _ in cfields] == [t for _, t in pj._lib.MemPlacedInfo._fields_]
    text = open(os.path.join(ROOT, "julia", "PixellHIP.jl")).read()
    jbody = re.search(r"struct MemPlacedInfo[^\n]*\n(.*?)\nend", text, flags=re.S).group(1)
    jfields = re.findall(r"^\s*(\w+)::([\w{}]+)", jbody, flags=re.M)
    jmap = {"uint64_t": "UInt64", "int32_t": "Int32"}
    assert jfields == [(n, jmap[t]) for t, n in cfields]
    assert ctypes.sizeof(pj._lib.MemPlacedInfo) == 4 * 4 + 8


def test_struct_layout_matches_reference_wcs(pj):
    """CarClenshawCurtis{Float64} is 7 Float64 = 56 bytes (car_proj.jl:7-12); the ABI struct must match."""
    assert ctypes.sizeof(pj._lib.CarWCSStruct) == 56
    w = pj.CarClenshawCurtis((-1.0, 1.0), (180.5, 91.0), (0.5, 0.0))
    s = w.to_struct()
    raw = (ctypes.c_double * 7).from_buffer_copy(s)
    assert list(raw) == [-1.0, 1.0, 180.5, 91.0, 0.5, 0.0, w.unit]


def test_missing_library_fails_loudly(pj, monkeypatch):
    monkeypatch.setattr(pj._lib, "_lib", None)
    monkeypatch.setattr(pj._lib, "LIB_PATH", os.path.join(PKG, "does_not_exist.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        pj._lib.load()


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    offenders = []
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                # comments may cite the oracle as the definition of R1; code may not load, link or include it
                if re.search(r"liboracle|from oracle|import oracle|#include\s*[\"<][^\n]*oracle|oracle\.(lib|build)\(", text):
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders
    # the shared library has no dependency on liboracle
    out = subprocess.run(["ldd", os.path.join(PKG, "libpixell_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_header_is_plain_c_and_links(tmp_path, pj):
    """include/pixell_hip.h compiles as C99 and a C program links against the shared library and runs its
    no-GPU paths (gcc, no hipcc, no C++)."""
    exe = str(tmp_path / "abi_c99")
    libdir = os.path.dirname(pj.library_path())
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_c99.c"), "-o", exe, "-L", libdir, "-lpixell_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_c99 ok" in r.stdout
