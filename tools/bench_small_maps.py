#!/usr/bin/env python3
"""Latency of reproject() on small maps: one-shot (plan created and destroyed per call) vs a kept plan."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
def wall(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6
for nx in (256, 1024, 4096):
    (si, wi), (so, wo) = pj.fullsky_geometry(2 * math.pi / nx), pj.fullsky_geometry(2 * math.pi / (2 * nx))
    m = pj.Enmap(torch.randn((si[1], si[0]), dtype=torch.float64, device=dev), wi)
    out = pj.Enmap(torch.empty((so[1], so[0]), dtype=torch.float64, device=dev), wo)
    plan = pj.ReprojectPlan(si, wi, so, wo, device=dev)
    t_one = wall(lambda: pj.reproject(m, so, wo, out=out))
    t_plan = wall(lambda: pj.reproject(m, so, wo, out=out, plan=plan))
    t_create = wall(lambda: pj.ReprojectPlan(si, wi, so, wo, device=dev).close())
    print("%5dx%-5d -> 2x: one-shot %7.1f us | kept plan %7.1f us | plan create+destroy %7.1f us" % (si[0], si[1], t_one, t_plan, t_create), flush=True)
