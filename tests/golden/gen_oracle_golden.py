"""Freeze oracle outputs for the build-defined bilinear reprojection (SURVEY 8(c) 'golden fixtures to commit'):
a 64x33 Clenshaw-Curtis map with analytic content m[i,j] = ((j-1)*nx + i)^2 (the pattern of the reference's
test/test_transforms.jl:3-9) reprojected by the oracle (a) onto the 2x-refined grid and (b) onto the same
grid shifted by half a pixel, plus 1024 seeded pix<->sky pairs.  Stored as hex floats (bit-exact).

    python tests/golden/gen_oracle_golden.py        # run from the repo root; rewrites oracle_golden.json
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def hexs(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def main():
    shape, w = O.fullsky_geometry(2 * math.pi / 64)
    nx, ny = shape
    assert shape == (64, 33)
    jj, ii = np.meshgrid(np.arange(1, ny + 1, dtype=float), np.arange(1, nx + 1, dtype=float), indexing="ij")
    src = (((jj - 1) * nx + ii) ** 2)[None]
    shape2, w2 = O.fullsky_geometry(2 * math.pi / 128)

    class Shift:
        cdelt, crval, unit = tuple(w.cdelt), tuple(w.crval), w.unit
        crpix = (w.crpix[0] + 0.5, w.crpix[1] + 0.5)
    rng = np.random.default_rng(20261003)
    pix = np.stack([rng.uniform(-10, nx + 10, 1024), rng.uniform(-5, ny + 5, 1024)], axis=1)
    sky = O.pix2sky(w, pix, O.WRAP_NONE)
    out = {
        "geometry": {"shape": list(shape), "cdelt": list(w.cdelt), "crpix": list(w.crpix), "crval": list(w.crval)},
        "refined_shape": list(shape2),
        "refined": hexs(O.reproject(w, (nx, ny, 1), src, w2, shape2)),
        "shifted": hexs(O.reproject(w, (nx, ny, 1), src, Shift, shape)),
        "pix": hexs(pix),
        "pix2sky_unsafe": hexs(sky),
        "pix2sky_rewind": hexs(O.pix2sky(w, pix, O.WRAP_REWIND)),
        "pix2sky_unwind": hexs(O.pix2sky(w, pix, O.WRAP_UNWIND)),
        "sky2pix_safe_recip": hexs(O.sky2pix(w, shape, sky, safe=True, form=O.FORM_RECIP)),
        "sky2pix_safe_div": hexs(np.stack(O.sky2pix_soa(w, shape, sky[:, 0], sky[:, 1], safe=True, form=O.FORM_DIV), axis=1)),
    }
    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote oracle_golden.json")


if __name__ == "__main__":
    main()
