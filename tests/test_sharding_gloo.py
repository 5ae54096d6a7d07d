"""Dec-strip sharding over torch.distributed (gloo, CPU, world_size 2 and 3): ownership, halo exchange and
windows.  The per-rank compute here is the ORACLE (this is a test of the decomposition, on a box without a
GPU): each rank reprojects its output strip from its own rows + the halo it received, and the
concatenation must be bit-identical to the unsharded result.  The same DecStripLayout code drives RCCL
send/recv on the GPUs (DecStripReprojector)."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, bits_equal

DEG = math.pi / 180


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cases():
    import pixell_jl_amd as pj
    fs = pj.fullsky_geometry(2 * math.pi / 96)
    fs2 = pj.fullsky_geometry(2 * math.pi / 192)
    shifted = (fs[0], pj.CarClenshawCurtis(fs[1].cdelt, (fs[1].crpix[0] + 0.5, fs[1].crpix[1] + 0.5), fs[1].crval))
    flipped = pj.geometry([[-60 * DEG, 60 * DEG], [40 * DEG, -40 * DEG]], 2 * math.pi / 192)
    return {"same_res_half_pixel_shift_iqu": (fs, shifted, 3),        # BASELINE config 4 pattern
            "refine2x": (fs, fs2, 1),                                  # config 3 pattern
            "coarsen2x": (fs2, fs, 2),
            "fullsky_to_flipped_box": (fs2, flipped, 1)}               # source rows run backwards


def _worker(rank, world, port, case_name, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pixell_jl_amd as pj
        from oracle import oracle as O
        (shape_in, wcs_in), (shape_out, wcs_out), nc = _cases()[case_name]
        nx, ny = shape_in[:2]
        layout = pj.sharding.DecStripLayout((nx, ny, nc), wcs_in, shape_out, wcs_out, rank, world)
        # every rank can build the full map deterministically, but only keeps the rows it OWNS
        full = np.random.default_rng(99).normal(size=(nc, ny, nx))
        src = torch.full(layout.src_tensor_shape(), float("nan"), dtype=torch.float64)
        own_lo, own_hi = layout.own[rank]
        src[:, layout.own_slice(), :] = torch.from_numpy(full[:, own_lo:own_hi])
        staging = layout.make_staging(src)
        works = layout.start_halo_exchange(src, staging)
        layout.finish_halo_exchange(src, staging, works)
        # the resident buffer now equals the corresponding rows of the full map wherever they are needed
        n_lo, n_hi = layout.need[rank]
        if n_hi > n_lo:
            got = src[:, n_lo - layout.buf_lo:n_hi - layout.buf_lo].numpy()
            assert bits_equal(got, full[:, n_lo:n_hi]), "halo rows wrong on rank %d" % rank
        s_row0, s_nrows = layout.src_window
        d_row0, d_nrows = layout.dst_window
        buf = np.nan_to_num(src.numpy(), nan=1e300)            # rows never needed must not matter
        part = O.reproject(wcs_in, (nx, ny, nc), buf, wcs_out, shape_out, src_row0=s_row0, src_nrows=s_nrows,
                           dst_row0=d_row0, dst_nrows=d_nrows)
        # interior rows need no halo: recompute them from owned rows only
        i_lo, i_hi = layout.interior
        if i_hi > i_lo:
            own_only = O.reproject(wcs_in, (nx, ny, nc), full[:, own_lo:own_hi], wcs_out, shape_out,
                                   src_row0=own_lo, src_nrows=own_hi - own_lo, dst_row0=d_row0 + i_lo,
                                   dst_nrows=i_hi - i_lo)
            assert bits_equal(own_only, part[:, i_lo:i_hi]), "interior rows depend on the halo on rank %d" % rank
        np.save(os.path.join(out_dir, "part_%d.npy" % rank), part)
        np.save(os.path.join(out_dir, "meta_%d.npy" % rank),
                np.array([d_row0, d_nrows, layout.halo_bytes(), len(layout.sends), len(layout.recvs)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case_name", ["same_res_half_pixel_shift_iqu", "refine2x", "coarsen2x",
                                       "fullsky_to_flipped_box"])
def test_sharded_equals_unsharded(tmp_path, world, case_name):
    from oracle import oracle as O
    O.lib()                                                        # build once, before forking
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case_name, str(tmp_path)), nprocs=world, join=True)
    (shape_in, wcs_in), (shape_out, wcs_out), nc = _cases()[case_name]
    nx, ny = shape_in[:2]
    full = np.random.default_rng(99).normal(size=(nc, ny, nx))
    expect = O.reproject(wcs_in, (nx, ny, nc), full, wcs_out, shape_out)
    rows = 0
    total_halo = 0
    for r in range(world):
        part = np.load(tmp_path / ("part_%d.npy" % r))
        d_row0, d_nrows, halo, nsend, nrecv = np.load(tmp_path / ("meta_%d.npy" % r))
        assert d_row0 == rows
        assert bits_equal(part, expect[:, d_row0:d_row0 + d_nrows]), "rank %d strip differs" % r
        rows += d_nrows
        total_halo += halo
    assert rows == shape_out[1]
    if case_name == "same_res_half_pixel_shift_iqu":
        # same DEC boundaries: exactly one halo row per internal boundary
        assert total_halo == (world - 1) * nx * nc * 8


def test_layout_is_consistent_across_ranks():
    """Every send has a matching recv (same rows, same peer) -- no rank can block on a message nobody sends."""
    import pixell_jl_amd as pj
    for case_name, ((shape_in, wcs_in), (shape_out, wcs_out), nc) in _cases().items():
        for world in (1, 2, 4, 8):
            layouts = [pj.sharding.DecStripLayout((shape_in[0], shape_in[1], nc), wcs_in, shape_out, wcs_out, r, world)
                       for r in range(world)]
            sends = {(r, q, lo, hi) for r, L in enumerate(layouts) for q, lo, hi in L.sends}
            recvs = {(q, r, lo, hi) for r, L in enumerate(layouts) for q, lo, hi in L.recvs}
            assert sends == recvs, (case_name, world)
            for r, L in enumerate(layouts):
                n_lo, n_hi = L.need[r]
                assert L.buf_lo <= min(n_lo, L.own[r][0]) and L.buf_hi >= max(n_hi, L.own[r][1])
                covered = set(range(*L.own[r]))
                for _, lo, hi in L.recvs:
                    covered |= set(range(lo, hi))
                assert set(range(n_lo, n_hi)) <= covered, (case_name, world, r)
            assert sum(L.dst_window[1] for L in layouts) == shape_out[1]


def test_headline_config_layout_at_8_ranks():
    """BASELINE config 4 (43200 x 21601 x 3, half-pixel shift) on 8 ranks: one halo row per neighbour,
    1.04 MB per message ('distributed communication backend')."""
    import pixell_jl_amd as pj
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 43200, dims=(3,))
    wcs_out = pj.CarClenshawCurtis(wcs_in.cdelt, (wcs_in.crpix[0] + 0.5, wcs_in.crpix[1] + 0.5), wcs_in.crval)
    for r in range(8):
        L = pj.sharding.DecStripLayout(shape_in, wcs_in, shape_in[:2], wcs_out, r, 8)
        for peer, lo, hi in L.recvs + L.sends:
            assert hi - lo == 1 and abs(peer - r) == 1
            assert (hi - lo) * 43200 * 3 * 8 == 1036800
        assert len(L.recvs) == (0 if r == 0 else 1)         # y = j - 0.5: each strip needs one row from below
        i_lo, i_hi = L.interior
        assert (i_hi - i_lo) >= L.dst_window[1] - 1
