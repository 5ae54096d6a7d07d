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


def test_julia_binding_only_calls_declared_symbols(pj):
    """julia/PixellHIP.jl (the binding a Pixell.jl maintainer would add) must ccall entry points that the
    header declares and the library exports, with as many argument types as the C prototype has parameters."""
    text = open(os.path.join(ROOT, "julia", "PixellHIP.jl")).read()
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    lib = ctypes.CDLL(pj.library_path())
    calls = re.findall(r"ccall\(\(:(pxl_[a-z0-9_]+), libpixell_hip\),\s*\w+,\s*\(([^)]*)\)", text, flags=re.S)
    assert len(calls) >= 20
    declared = declared_symbols()
    for name, argtypes in calls:
        assert name in declared, "PixellHIP.jl calls %s, which include/pixell_hip.h does not declare" % name
        assert hasattr(lib, name)
        proto = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, header, flags=re.S).group(1)
        nparams = 0 if proto.strip() in ("", "void") else len(proto.split(","))
        nargs = len([a for a in argtypes.split(",") if a.strip()])
        assert nargs == nparams, "%s: %d Julia argument types for %d C parameters" % (name, nargs, nparams)


def test_version_and_error_channel_without_gpu(pj):
    lib = pj.load_library()
    assert lib.pxl_version() == 100
    # argument validation happens before any HIP call: a NULL WCS is rejected with a message
    rc = lib.pxl_pix2sky_car_f64(None, 0, None, None, 0, None)
    assert rc == -22
    assert "WCS" in pj._lib.last_error()


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
