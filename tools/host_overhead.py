#!/usr/bin/env python3
"""How much host time does one sharded step cost to ENQUEUE (everything except the RCCL calls), compared
with the GPU time of a 1/8 strip of config 4?  If host < GPU, the GPU never starves at N = 8."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import pixell_jl_amd as pj

dev = torch.device("cuda:0")
shape_in, wcs_in, shape_out, wcs_out, _ = bench.workload_geometry("cfg4")
L = pj.sharding.DecStripLayout(shape_in, wcs_in, shape_out, wcs_out, 3, 8)
plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, src_rows=L.src_window, dst_rows=L.dst_window, device=dev)
src = torch.zeros(L.src_tensor_shape(), dtype=torch.float64, device=dev)
dst = torch.empty(L.dst_tensor_shape(), dtype=torch.float64, device=dev)
pj.fill_random_(src, 1)
send, recv = L.make_staging(src)
i_lo, i_hi = L.interior
n = L.dst_window[1]

def step():
    for (peer, lo, hi), buf in zip(L.sends, send):
        buf.copy_(src[:, lo - L.buf_lo:hi - L.buf_lo, :])
    plan.build_tables()
    plan.execute_rows(src, dst, i_lo, i_hi - i_lo)
    for (peer, lo, hi), buf in zip(L.recvs, recv):
        src[:, lo - L.buf_lo:hi - L.buf_lo, :].copy_(buf)
    if i_lo > 0:
        plan.execute_rows(src, dst, 0, i_lo)
    if i_hi < n:
        plan.execute_rows(src, dst, i_hi, n - i_hi)

for _ in range(5):
    step()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("strip rows %d (interior %d), sends %d recvs %d" % (n, i_hi - i_lo, len(L.sends), len(L.recvs)))
print("host enqueue per step: %.1f us; GPU per step: %.1f us" % (t_host / K * 1e6, t_all / K * 1e6))
