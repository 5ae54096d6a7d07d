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
