"""Generate wcslib golden vectors for the CAR fast path (mirrors /root/reference test/test_geometry.jl:66-80,
where the reference itself cross-checks its CAR evaluators against wcslib through WCS.jl).

Run ONCE in the authoring container with the conda interpreter that bundles astropy 4.3.1 / wcslib 7.6:
    /opt/conda/bin/python3.9 tests/golden/gen_wcslib_vectors.py
Only the resulting JSON (data) is committed and used by tests; nothing here runs on the GPU box.
"""
import json
import os

import numpy as np

for _n, _v in (("asscalar", lambda a: a.item()), ("alen", len), ("float", float), ("int", int),
               ("bool", bool), ("object", object), ("complex", complex), ("str", str)):
    if not hasattr(np, _n):
        setattr(np, _n, _v)          # names astropy 4.3 expects and numpy >= 1.24 removed
import astropy.wcs as awcs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hexs(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def main():
    rng = np.random.default_rng(20261003)
    out = {"wcslib_version": awcs._wcs.__version__, "cases": []}
    # 1-degree full-sky CC geometry (the one test_geometry.jl:50 uses) and a 0.5' box geometry (test_geometry.jl:22-27)
    geoms = [
        dict(name="fullsky_1deg", shape=[360, 181], cdelt=[-1.0, 1.0], crpix=[180.5, 91.0], crval=[0.5, 0.0]),
        dict(name="box_0.5arcmin", shape=[2400, 1200], cdelt=[-0.008333333333333333, 0.008333333333333333],
             crpix=[1201.0, 601.0], crval=[0.0, 0.0]),
    ]
    for g in geoms:
        W = awcs.WCS(naxis=2)
        W.wcs.ctype = ["RA---CAR", "DEC--CAR"]
        W.wcs.cdelt = g["cdelt"]; W.wcs.crpix = g["crpix"]; W.wcs.crval = g["crval"]
        nx, ny = g["shape"]
        n = 1024
        pix = np.stack([1 + (nx - 1) * rng.random(n), 1 + (ny - 1) * rng.random(n)], axis=1)
        sky_deg = W.wcs_pix2world(pix, 1)
        # on-sky points that land inside the map, so wcslib's own wrapping conventions never kick in
        sky2_deg = W.wcs_pix2world(np.stack([1 + (nx - 1) * rng.random(n), 1 + (ny - 1) * rng.random(n)], axis=1), 1)
        pix2 = W.wcs_world2pix(sky2_deg, 1)
        # the reference's own draw (test_geometry.jl:67,75): pi .* rand(2, 1024), no wrap is ever involved
        pix_small = np.pi * rng.random((n, 2))
        sky_small_deg = W.wcs_pix2world(pix_small, 1)
        keep = np.all(np.isfinite(sky_small_deg), axis=1)      # astropy's wcslib returns NaN below the pole
        pix_small, sky_small_deg = pix_small[keep], sky_small_deg[keep]
        world_small_deg = np.rad2deg(np.stack([np.pi * rng.random(n), 0.5 * np.pi * rng.random(n)], axis=1))
        pix_from_small = W.wcs_world2pix(world_small_deg, 1)
        assert np.all(np.isfinite(sky_deg)) and np.all(np.isfinite(pix2))
        assert np.all(np.isfinite(sky_small_deg)) and np.all(np.isfinite(pix_from_small))
        out["cases"].append(dict(geom=g, pix=hexs(pix), pix2world_deg=hexs(sky_deg),
                                 world_deg=hexs(sky2_deg), world2pix=hexs(pix2),
                                 pix_small=hexs(pix_small), pix2world_small_deg=hexs(sky_small_deg),
                                 world_small_deg=hexs(world_small_deg), world2pix_small=hexs(pix_from_small)))
    # Gnomonic: the reference pins its TAN evaluators against wcslib too (test_geometry.jl:92-119: scalars with
    # `≈`, and the L1 difference of a full 1827x1825 posmap below 1e-9)
    tan = dict(shape=[1827, 1825], cdelt=[0.008333333333333333, 0.008333333333333333],
               crpix=[913.3649509696, 921.0316523678962], crval=[97.50416559979826, -7.45833685170031])
    T = awcs.WCS(naxis=2)
    T.wcs.ctype = ["RA---TAN", "DEC--TAN"]
    T.wcs.cdelt = tan["cdelt"]; T.wcs.crpix = tan["crpix"]; T.wcs.crval = tan["crval"]
    tpix = np.stack([1 + 1826 * rng.random(2048), 1 + 1824 * rng.random(2048)], axis=1)
    tsky = T.wcs_pix2world(tpix, 1)
    tback = T.wcs_world2pix(tsky, 1)
    assert np.all(np.isfinite(tsky)) and np.max(np.abs(tback - tpix)) < 1e-6
    out["tan"] = dict(geom=tan, pix=hexs(tpix), pix2world_deg=hexs(tsky))
    # An independent end-to-end reprojection: wcslib for both coordinate steps + scipy order-1 interpolation
    # (python-pixell's `project(..., order=1)` pipeline, which Pixell.jl says it mirrors): a 64x33 full-sky CC map
    # with analytic content onto a half-pixel-shifted 96x49 grid.
    from scipy.ndimage import map_coordinates
    def cc(nx):
        ny = nx // 2 + 1
        W = awcs.WCS(naxis=2)
        W.wcs.ctype = ["RA---CAR", "DEC--CAR"]
        W.wcs.cdelt = [-360.0 / nx, 180.0 / (ny - 1)]
        W.wcs.crpix = [nx // 2 + 0.5, (ny + 1) / 2]
        W.wcs.crval = [(np.pi / (ny - 1)) * 90 / np.pi, 0.0]
        return W, nx, ny
    Win, nxi, nyi = cc(64)
    Wout, nxo, nyo = cc(96)
    Wout.wcs.crpix = [Wout.wcs.crpix[0] + 0.37, Wout.wcs.crpix[1]]
    jj, ii = np.meshgrid(np.arange(1, nyi + 1, dtype=float), np.arange(1, nxi + 1, dtype=float), indexing="ij")
    src = np.sin(0.2 * ii) * np.cos(0.3 * jj) + 0.01 * ii
    oj, oi = np.meshgrid(np.arange(2, nyo, dtype=float), np.arange(1, nxo + 1, dtype=float), indexing="ij")  # skip pole rows
    world = Wout.wcs_pix2world(np.stack([oi.ravel(), oj.ravel()], axis=1), 1)
    pin = Win.wcs_world2pix(world, 1)
    assert np.all(np.isfinite(pin))
    ref = map_coordinates(src, [pin[:, 1] - 1, pin[:, 0] - 1], order=1, mode="grid-wrap")
    out["reproject_wcslib_scipy"] = dict(
        geom_in=dict(shape=[nxi, nyi], cdelt=list(Win.wcs.cdelt), crpix=list(Win.wcs.crpix), crval=list(Win.wcs.crval)),
        geom_out=dict(shape=[nxo, nyo], cdelt=list(Wout.wcs.cdelt), crpix=list(Wout.wcs.crpix), crval=list(Wout.wcs.crval)),
        src=hexs(src), rows=[2, nyo - 1], expected=hexs(ref))
    with open(os.path.join(HERE, "wcslib_car_vectors.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
