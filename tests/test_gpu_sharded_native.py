"""The library's own sharded step (pxl_reproject_sharded_step_*) over a real RCCL communicator.

The development boxes have ONE GPU and RCCL refuses two ranks on one device, so the exchange is exercised in
loopback: a single-rank communicator, with the rank sending rows it owns to ITSELF and receiving them into its
halo slots (RCCL pairs a grouped send and recv to the same peer in order).  The map is built so that the rows
travelling are exactly what a neighbour would have sent, so the result must equal the oracle on the full map --
every piece of the entry runs: communicator checks, the message per component plane straight from/into the
resident buffer, the plan-owned exchange stream and its two events, interior rows before / boundary rows after."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import bits_equal

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    pg = dist.distributed_c10d._get_default_group()
    comm = int(pg._get_backend(dev)._comm_ptr())
    assert comm != 0
    yield dev, comm
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("f32,own_comm", [(False, False), (True, False), (False, True)])
def test_sharded_step_loopback(pj, O, rccl, f32, own_comm):
    dev, comm = rccl
    lib = pj.load_library()
    made = None
    if own_comm:            # the communicator a Julia / C host would make through the ABI (same RCCL instance as torch's here)
        ident = C.create_string_buffer(128)
        assert lib.pxl_comm_unique_id(ident) == 0, pj._lib.last_error()
        made = C.c_void_p()
        assert lib.pxl_comm_init_rank(ident, 0, 1, C.byref(made)) == 0, pj._lib.last_error()
        assert lib.pxl_comm_init_rank(ident, 1, 1, C.byref(C.c_void_p())) == -22          # rank outside the communicator
        comm = made.value
        assert b"already loaded" in lib.pxl_comm_backend()
    # source 600 x 120, output 2x refined in RA and DEC; this "rank" owns source rows [41, 78) and output rows [80, 160)
    shape_in, wcs_in = (600, 120), pj.CarClenshawCurtis((-0.6, 0.5), (300.5, 60.0), (0.3, 0.0))
    shape_out, wcs_out = (1200, 240), pj.CarClenshawCurtis((-0.3, 0.25), (600.25, 120.3), (0.3, 0.0))
    nc = 2
    d0, dn = 80, 80
    s_lo, s_hi = O.reproject_src_rows(wcs_in, shape_in, wcs_out, shape_out, d0, dn)
    own_lo, own_hi = 41, 78
    assert s_lo < own_lo and s_hi > own_hi, "the strip needs halo rows on both sides"
    rng = np.random.default_rng(5)
    full = rng.normal(size=(nc, shape_in[1], shape_in[0]))
    # the halo rows are copies of owned rows, so that a self-exchange delivers what a neighbour would
    below, above = list(range(s_lo, own_lo)), list(range(own_hi, s_hi))
    for k, r in enumerate(below):
        full[:, r] = full[:, own_lo + 3 + k]
    for k, r in enumerate(above):
        full[:, r] = full[:, own_lo + 13 + k]
    full = full.astype(np.float32) if f32 else full
    ref = (O.reproject_f32 if f32 else O.reproject)(wcs_in, (600, 120, nc), full, wcs_out, shape_out, dst_row0=d0, dst_nrows=dn)
    # resident buffer: rows [s_lo, s_hi), halo slots poisoned
    resident = full[:, s_lo:s_hi].copy()
    resident[:, :own_lo - s_lo] = np.nan
    resident[:, own_hi - s_lo:] = np.nan
    d_src = torch.from_numpy(resident).to(dev)
    d_dst = torch.full((nc, dn, shape_out[0]), float("nan"), dtype=d_src.dtype, device=dev)
    plan = pj.ReprojectPlan((600, 120, nc), wcs_in, shape_out, wcs_out, src_rows=(s_lo, s_hi - s_lo), dst_rows=(d0, dn), device=dev)
    sends = [(0, own_lo + 3, own_lo + 3 + len(below)), (0, own_lo + 13, own_lo + 13 + len(above))]
    recvs = [(0, s_lo, own_lo), (0, own_hi, s_hi)]
    fn = lib.pxl_reproject_sharded_step_f32 if f32 else lib.pxl_reproject_sharded_step_f64
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    xs, xr = pj._lib.xfer_arr(sends), pj._lib.xfer_arr(recvs)
    for _ in range(3):                           # repeated steps reuse the plan's stream/events and the cached interior
        rc = fn(plan._h, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_dst.data_ptr()), own_lo, own_hi - own_lo,
                C.cast(xs, C.c_void_p), 2, C.cast(xr, C.c_void_p), 2, C.c_void_p(comm), s)
        assert rc == 0, pj._lib.last_error()
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert np.isfinite(got).all()
    if f32:
        assert np.array_equal(got.view(np.int32), ref.view(np.int32))
    else:
        assert bits_equal(got, ref)
    assert np.array_equal(d_src.cpu().numpy(), full[:, s_lo:s_hi])      # the halo slots now hold the neighbour rows
    # argument checks
    bad = pj._lib.xfer_arr([(1, own_lo, own_lo + 1)])                     # peer outside the communicator
    assert fn(plan._h, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_dst.data_ptr()), own_lo, own_hi - own_lo,
              C.cast(bad, C.c_void_p), 1, None, 0, C.c_void_p(comm), s) == -22
    bad = pj._lib.xfer_arr([(0, s_lo, s_lo + 1)])                         # sending a row the rank does not own
    assert fn(plan._h, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_dst.data_ptr()), own_lo, own_hi - own_lo,
              C.cast(bad, C.c_void_p), 1, None, 0, C.c_void_p(comm), s) == -22
    # no transfers: plain build_tables + execute, no communicator needed
    d_full = torch.from_numpy(full[:, s_lo:s_hi].copy()).to(dev)
    d_dst.fill_(float("nan"))
    assert fn(plan._h, C.c_void_p(d_full.data_ptr()), C.c_void_p(d_dst.data_ptr()), s_lo, s_hi - s_lo,
              None, 0, None, 0, None, s) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy().view(np.int32 if f32 else np.int64), ref.view(np.int32 if f32 else np.int64))
    plan.close()
    if made is not None:
        assert lib.pxl_comm_destroy(made) == 0, pj._lib.last_error()
        assert lib.pxl_comm_destroy(made) == -22                                           # already gone
