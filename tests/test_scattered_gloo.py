"""BASELINE config 5 with more than one rank: the map replicated, the points split evenly, no data-path
collective, the per-rank oracle checks reduced to the job's worst case.  torch.distributed over gloo on the CPU, world sizes 2
and 3.  The per-rank compute here is the ORACLE (this is a test of the decomposition and of bench.py's reduction, on a box
without a GPU); on the GPU box `tests/test_gpu_entrypoints.py::test_bench_self_launch_two_ranks_cfg5` runs the same flow through
bench.py with the HIP sampler."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, bits_equal


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _points(n):
    """The benchmark's distribution (uniform on the sphere: ra = 2 pi u1 - pi, dec = asin(2 u2 - 1)), one seeded sequence."""
    rng = np.random.default_rng(42)
    u = rng.random((n, 2))
    return np.stack([2 * np.pi * u[:, 0] - np.pi, np.arcsin(2 * u[:, 1] - 1)], axis=1)


def _worker(rank, world, port, npts, out_dir, bad_rank):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import pixell_jl_amd as pj
        from oracle import oracle as O
        shape, wcs = pj.fullsky_geometry(2 * math.pi / 720)
        nx, ny = shape
        m = np.random.default_rng(1234).normal(size=(1, ny, nx))           # the replicated map: every rank holds all of it
        lo, hi = pj.strip_bounds(npts, world, rank)
        pts = _points(npts)[lo:hi]                                          # rank r's share of the ONE sequence
        out = O.sample_bilinear(wcs, (nx, ny, 1), m, pts)[0]
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), out)
        np.save(os.path.join(out_dir, "bounds_%d.npy" % rank), np.array([lo, hi]))
        # bench.py's reduction of the per-rank checks: one rank reports a mismatch, every rank must see the job as failed
        chk = {"points_checked": int(hi - lo), "max_abs_err": 0.25 if rank == bad_rank else 0.0, "bit_identical": rank != bad_rank}
        red = bench.reduce_check(chk, world, None)
        assert red["ranks_checked"] == world, red
        assert red["bit_identical"] == (bad_rank is None or bad_rank < 0), (rank, red)
        assert red["max_abs_err"] == (0.25 if (bad_rank is not None and bad_rank >= 0) else 0.0), red
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bad_rank", [(2, -1), (2, 1), (3, 0)])
def test_points_sharded_map_replicated(tmp_path, world, bad_rank):
    import pixell_jl_amd as pj
    from oracle import oracle as O
    npts = 20011                                  # not divisible by 2 or 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, npts, str(tmp_path), bad_rank), nprocs=world, join=True)
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 720)
    nx, ny = shape
    m = np.random.default_rng(1234).normal(size=(1, ny, nx))
    whole = O.sample_bilinear(wcs, (nx, ny, 1), m, _points(npts))[0]
    bounds = [np.load(os.path.join(str(tmp_path), "bounds_%d.npy" % r)) for r in range(world)]
    # the shares tile [0, npts) exactly, in rank order, sizes within one point of each other
    assert bounds[0][0] == 0 and bounds[-1][1] == npts
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    sizes = [int(b[1] - b[0]) for b in bounds]
    assert max(sizes) - min(sizes) <= 1
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out_%d.npy" % r)) for r in range(world)])
    assert bits_equal(got, whole)


def test_strip_bounds_tiles_any_count():
    import pixell_jl_amd as pj
    for n in (0, 1, 7, 8, 9, 1000, 10**9, 10**9 + 7):
        for world in (1, 2, 3, 4, 8):
            b = [pj.strip_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
