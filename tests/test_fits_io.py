"""FITS interop: header/WCS parsing on the CPU against the reference's own fixture
(test/data/test.fits, test/test_io.jl:1-13, test/test_enmap.jl:143-150) and the device decode/encode."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

FIXTURE = os.path.join(GOLDEN, "test.fits")


def test_header_and_wcs_of_reference_fixture(pj):
    h, offset = pj.read_header(FIXTURE)
    assert (h["BITPIX"], h["NAXIS"], h["NAXIS1"], h["NAXIS2"], h["NAXIS3"]) == (-64, 3, 100, 100, 3)
    assert offset == 2880
    wcs = pj.wcs_from_header(h)
    assert wcs.cdelt == (-1.0, 1.0) and wcs.crval == (0.5, 0.0) and wcs.crpix == (180.5, 91.0)   # test_io.jl:6-7
    assert wcs.naxis == 2 and abs(wcs.unit - np.pi / 180) < 1e-18                                # test_enmap.jl:144
    # the data block, decoded on the host only to pin the fixture: sum(imap) ~ 14967.2985 (test_io.jl:8)
    data = np.frombuffer(open(FIXTURE, "rb").read()[offset:offset + 100 * 100 * 3 * 8], dtype=">f8")
    assert abs(data.sum() - 14967.2985) < 1e-3


def test_non_car_header_is_refused(pj):
    h, _ = pj.read_header(FIXTURE)
    h["CTYPE1"] = "RA---TAN"
    with pytest.raises(AssertionError):
        pj.wcs_from_header(h)


@pytest.mark.gpu
def test_read_map_on_device(pj):
    import torch
    m = pj.read_map(FIXTURE, device="cuda:0")
    assert m.shape == (100, 100, 3)                                           # test_io.jl:4
    assert m.wcs.cdelt == (-1.0, 1.0) and m.wcs.crval == (0.5, 0.0)
    assert abs(float(m.data.sum()) - 14967.2985) < 1e-3                       # test_io.jl:8
    host = np.frombuffer(open(FIXTURE, "rb").read()[2880:2880 + 240000], dtype=">f8").astype("<f8").reshape(3, 100, 100)
    assert np.array_equal(m.data.cpu().numpy(), host)
    sub = pj.read_map(FIXTURE, device="cuda:0", sel=((11, 20), (21, 40), (1, 2)))   # test_io.jl:10-11
    assert sub.shape == (10, 20, 2)
    assert np.array_equal(sub.data.cpu().numpy(), host[0:2, 20:40, 10:20])
    assert sub.wcs.crpix == (180.5 - 10, 91.0 - 20)


@pytest.mark.gpu
def test_write_read_roundtrip(pj, tmp_path):
    import math
    import torch
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 64, dims=(3,))
    m = pj.Enmap(torch.randn((3, shape[1], shape[0]), dtype=torch.float64, device="cuda:0"), wcs)
    path = str(tmp_path / "out.fits")
    pj.write_map(path, m)
    assert os.path.getsize(path) % 2880 == 0
    back = pj.read_map(path, device="cuda:0")
    assert back.shape == shape and torch.equal(back.data, m.data)
    assert back.wcs.crpix == wcs.crpix and back.wcs.crval == wcs.crval and back.wcs.cdelt == wcs.cdelt
    h, _ = pj.read_header(path)
    assert h["CTYPE1"] == "RA---CAR" and h["CUNIT1"] == "deg" and h["NAXIS3"] == 3


@pytest.mark.gpu
def test_float32_files_roundtrip(pj, tmp_path):
    """BITPIX -32 files stay Float32 on the device (as the reference's read keeps the file's element type) and
    can be widened on request; a Float32 map written and read back is unchanged."""
    import math
    import torch
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 64, dims=(2,))
    m = pj.Enmap(torch.randn((2, shape[1], shape[0]), dtype=torch.float32, device="cuda:0"), wcs)
    path = str(tmp_path / "f32.fits")
    pj.write_map(path, m)
    h, off = pj.read_header(path)
    assert h["BITPIX"] == -32 and os.path.getsize(path) % 2880 == 0
    host = np.frombuffer(open(path, "rb").read()[off:off + m.data.numel() * 4], dtype=">f4").astype("<f4")
    assert np.array_equal(host.reshape(m.data.shape), m.data.cpu().numpy())
    back = pj.read_map(path, device="cuda:0")
    assert back.data.dtype == torch.float32 and torch.equal(back.data, m.data) and back.wcs.crpix == wcs.crpix
    wide = pj.read_map(path, device="cuda:0", dtype=torch.float64)
    assert wide.data.dtype == torch.float64 and torch.equal(wide.data, m.data.double())


@pytest.mark.gpu
def test_streamed_io_many_chunks(pj, tmp_path, monkeypatch):
    """The data block moves through two pinned buffers in chunks: with a chunk far smaller than the file, a
    written map, the file bytes and every kind of read (whole, row strip, components, generic selection) agree."""
    import math
    import torch
    monkeypatch.setenv("PXL_FITS_CHUNK_MB", "0.3")
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 1000, dims=(3,))             # 1000 x 501 x 3 = 12 MB
    m = pj.Enmap(torch.randn((3, shape[1], shape[0]), dtype=torch.float64, device="cuda:0"), wcs)
    path = str(tmp_path / "big.fits")
    pj.write_map(path, m)
    h, off = pj.read_header(path)
    host = np.frombuffer(open(path, "rb").read()[off:off + m.data.numel() * 8], dtype=">f8").astype("<f8")
    assert np.array_equal(host.reshape(m.data.shape), m.data.cpu().numpy()) and os.path.getsize(path) % 2880 == 0
    assert torch.equal(pj.read_map(path, device="cuda:0").data, m.data)
    # a declination strip of a sharded job: only those rows are read; windows into the FULL geometry
    strip, full_shape, full_wcs = pj.read_map_rows(path, 100, 37, device="cuda:0")
    assert full_shape == shape and full_wcs.crpix == wcs.crpix and torch.equal(strip, m.data[:, 100:137, :])
    one, _, _ = pj.read_map_rows(path, 0, shape[1], device="cuda:0", comps=[2])
    assert torch.equal(one, m.data[2:3])
    # reference-style selections: whole rows take the strip path, anything else the general one; same answers
    a = pj.read_map(path, device="cuda:0", sel=(None, (101, 137), (2, 3)))
    assert torch.equal(a.data, m.data[1:3, 100:137, :]) and a.wcs.crpix == (wcs.crpix[0], wcs.crpix[1] - 100)
    b = pj.read_map(path, device="cuda:0", sel=((5, 2, 900), (137, -1, 101), None))
    assert torch.equal(b.data, m.getindex((5, 2, 900), (137, -1, 101), None).data) and b.wcs == m.getindex((5, 2, 900), (137, -1, 101), None).wcs
    with pytest.raises(ValueError):
        pj.read_map_rows(path, 400, 200, device="cuda:0")
    # a truncated file is refused before anything is copied
    cut = str(tmp_path / "cut.fits")
    open(cut, "wb").write(open(path, "rb").read()[:off + 1000])
    with pytest.raises(ValueError):
        pj.read_map(cut, device="cuda:0")


def test_header_units(pj):
    """getunit of the header's CUNIT (arbitrary_wcs.jl:16-34; test_enmap.jl:155-162): deg, rad, arcmin, arcsec, mas,
    unknown -> degrees; mixed units are refused (the reference asserts)."""
    h, _ = pj.read_header(FIXTURE)
    for cunit, unit in (("deg", np.pi / 180), ("rad", 1.0), ("arcmin", np.pi / 180 / 60), ("arcsec", np.pi / 180 / 60 / 60),
                        ("mas", np.pi / 180 / 60 / 60 / 1000), ("furlong", np.pi / 180)):
        h2 = dict(h, CUNIT1=cunit, CUNIT2=cunit)
        assert abs(pj.wcs_from_header(h2).unit - unit) <= 1e-18 * max(1.0, unit)
    with pytest.raises(AssertionError):
        pj.wcs_from_header(dict(h, CUNIT1="deg", CUNIT2="rad"))


@pytest.mark.gpu
def test_polcconv_iau_follows_the_reference(pj, tmp_path):
    """POLCCONV = IAU files (enmap.jl:178-195, :203-209): the reference multiplies by `signs` built with
    `signs[signs_size] .= -1`, signs_size = [1, 1, 3] -- linear indices 1 and 3, i.e. planes I and U change sign.
    The default reproduces that; polcconv="u_only" is the physical convention.  Gated on CTYPE3 == "STOKES"."""
    import math
    import torch
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 32, dims=(3,))
    m = pj.Enmap(torch.randn((3, shape[1], shape[0]), dtype=torch.float64, device="cuda:0"), wcs)
    iau, cosmo, notstokes = (str(tmp_path / n) for n in ("iau.fits", "cosmo.fits", "other.fits"))
    pj.write_map(iau, m, extra_cards=[("CTYPE3", "STOKES"), ("POLCCONV", "IAU")])
    pj.write_map(cosmo, m, extra_cards=[("CTYPE3", "STOKES"), ("POLCCONV", "COSMO")])
    pj.write_map(notstokes, m, extra_cards=[("CTYPE3", "FREQ"), ("POLCCONV", "IAU")])
    ref = pj.read_map(iau, device="cuda:0")
    assert torch.equal(ref.data[0], -m.data[0]) and torch.equal(ref.data[1], m.data[1]) and torch.equal(ref.data[2], -m.data[2])
    phys = pj.read_map(iau, device="cuda:0", polcconv="u_only")
    assert torch.equal(phys.data[0], m.data[0]) and torch.equal(phys.data[1], m.data[1]) and torch.equal(phys.data[2], -m.data[2])
    assert torch.equal(pj.read_map(cosmo, device="cuda:0").data, m.data)
    assert torch.equal(pj.read_map(notstokes, device="cuda:0").data, m.data)        # no axis is STOKES: untouched
    # a selection along the Stokes axis selects the same signs (enmap.jl:189-190): planes 2:3 -> (+Q, -U)
    sub = pj.read_map(iau, device="cuda:0", sel=(None, None, (2, 3)))
    assert torch.equal(sub.data[0], m.data[1]) and torch.equal(sub.data[1], -m.data[2])
