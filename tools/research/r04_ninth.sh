#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r04_ninth
mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -s -k "gnomonic or tan" > $out/pytest_sel.log 2>&1; echo rc=$? >> $out/pytest_sel.log; tail -8 $out/pytest_sel.log | cut -c1-300; grep "polar patch" $out/pytest_sel.log
echo "== tan evaluators, per-pixel posmap"; PXL_TAN_GRID=0 timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep posmap
echo "== tan evaluators, grid posmap"; timeout -k 10 200 python3 tools/bench_tan_evaluators.py 2>&1 | grep posmap
python3 - <<'PY'
import sys, math, numpy as np, torch
sys.path.insert(0, '.')
import pixell_jl_amd as pj
from oracle import oracle as O
import json, os
dev = torch.device("cuda:0")
g = json.load(open("tests/golden/reference_literals.json"))["gnomonic"]
for name, wcs, shape in (("reference patch", pj.Gnomonic(g["cdelt"], g["crpix"], g["crval"]), tuple(g["shape"])),
                         ("0.5' 4096^2 at dec 40", pj.Gnomonic((-0.5/60, 0.5/60), (2048.5, 2048.5), (10.0, 40.0)), (4096, 4096)),
                         ("1' 2048^2 at dec -60", pj.Gnomonic((-1/60, 1/60), (1024.5, 1024.5), (200.0, -60.0)), (2048, 2048))):
    jj, ii = np.meshgrid(np.arange(1, shape[1] + 1, dtype=float), np.arange(1, shape[0] + 1, dtype=float), indexing="ij")
    era, edec = O.pix2sky_tan(wcs, ii.ravel(), jj.ravel())
    for grid in ("0", "1"):
        os.environ["PXL_TAN_GRID"] = grid
        ra, dec = pj.posmap(shape, wcs, device=dev)
        ra, dec = ra.data.cpu().numpy().ravel(), dec.data.cpu().numpy().ravel()
        print(name, "grid" if grid == "1" else "per pixel", "L1 ra %.3e dec %.3e  max ra %.2e dec %.2e" % (np.abs(ra - era).sum(), np.abs(dec - edec).sum(), np.abs(ra - era).max(), np.abs(dec - edec).max()))
PY
echo "== unwind two-pass"; PXL_UNWIND_ONEPASS=0 timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2
for v in "" _u4w4 _u4w8 _u4w16; do echo "== unwind one-pass $v"; PXL_LIB_PATH=$R/pixell.jl_amd/libpixell_hip$v.so timeout -k 10 200 python3 tools/prof_unwind.py 2>&1 | grep "out-of-place" | tail -2; done
