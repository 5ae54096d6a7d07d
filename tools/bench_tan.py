#!/usr/bin/env python3
"""Timings of the Gnomonic rows: posmap on the reference's 1827x1825 TAN patch
(test_geometry.jl:92-119) and CAR<->TAN bilinear reprojection at 0.5 arcmin."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
DEG = math.pi / 180
tan_wcs = pj.Gnomonic((0.008333333333333333, 0.008333333333333333), (913.3649509696, 921.0316523678962),
                      (97.50416559979826, -7.45833685170031))
tan_shape = (1827, 1825)
def t(f, reps=10):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
npx = tan_shape[0] * tan_shape[1]
ms = t(lambda: pj.posmap(tan_shape, tan_wcs, device=dev))
print("posmap TAN 1827x1825           : %.3f ms  %.1f Mpix/s" % (ms, npx / ms / 1e3))
car_shape, car_wcs = pj.geometry([[106 * DEG, 89 * DEG], [-16 * DEG, 1 * DEG]], 0.5 * DEG / 60)
car = pj.Enmap(torch.randn((car_shape[1], car_shape[0]), dtype=torch.float64, device=dev), car_wcs)
ms = t(lambda: pj.reproject(car, tan_shape, tan_wcs))
print("reproject CAR %dx%d -> TAN   : %.3f ms  %.1f Mpix/s (output)" % (car_shape[0], car_shape[1], ms, npx / ms / 1e3))
tan = pj.reproject(car, tan_shape, tan_wcs)
ms = t(lambda: pj.reproject(tan, car_shape, car_wcs))
print("reproject TAN -> CAR %dx%d   : %.3f ms  %.1f Mpix/s (output)" % (car_shape[0], car_shape[1], ms, car_shape[0] * car_shape[1] / ms / 1e3))
