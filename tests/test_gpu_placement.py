"""Class-aware placement (pixell.jl_amd/placement.py over pxl_mem_probe_pair): the probe runs and tells two rates apart or
reports one class, `place_pair` returns disjoint, correctly shaped views of one allocation, and a reprojection on placed maps
gives the bits of a reprojection on plain ones."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import pixell_jl_amd as pj
    pj.load_library()
    return torch.device("cuda:0")


def test_probe_argument_checks(pj, dev):
    import ctypes as C
    lib = pj.load_library()
    us = C.c_float()
    a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    assert lib.pxl_mem_probe_pair(None, C.c_void_p(a.data_ptr()), 128 << 20, 1, C.byref(us), None) == -22
    assert lib.pxl_mem_probe_pair(C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr()), 1 << 20, 1, C.byref(us), None) == -22     # too small
    assert lib.pxl_mem_probe_pair(C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr() + (128 << 20)), 128 << 20, 0, C.byref(us), None) == -22
    a.fill_(7)
    assert lib.pxl_mem_probe_pair(C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr() + (128 << 20)), 128 << 20, 2, C.byref(us), None) == 0
    assert us.value > 0 and int(a.max()) == 0            # the probe writes zeros over both windows


def test_map_classes_and_place_pair(pj, O, dev):
    arena = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
    offs, labels, info = pj.map_classes(arena, step_gib=2)
    assert len(offs) == len(labels) == 6 and labels[0] == 0 and 1 <= info["classes"] <= 6
    # within a class the probe runs at 5.6-5.9 TB/s, across classes at 6.6-7.0 TB/s: 2 GiB in 300-400 us
    assert 150 < info["probe_us_same_class"] < 1500          # order of magnitude only: a timing, on whatever box runs the test
    del arena
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 2400, dims=(2,))
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 4800)
    sshape, dshape = (2, shape_in[1], shape_in[0]), (2, shape_out[1], shape_out[0])
    src, dst, pinfo = pj.place_pair(sshape, dshape, device=dev, headroom_gib=40)
    assert tuple(src.shape) == sshape and tuple(dst.shape) == dshape and src.dtype == dst.dtype == torch.float64
    assert src.is_contiguous() and dst.is_contiguous() and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0
    lo_s, hi_s = src.data_ptr(), src.data_ptr() + src.numel() * 8
    lo_d, hi_d = dst.data_ptr(), dst.data_ptr() + dst.numel() * 8
    assert hi_s <= lo_d or hi_d <= lo_s                                    # disjoint
    a0 = pinfo["arena"].data_ptr()
    assert a0 <= lo_d and hi_d <= a0 + pinfo["arena"].numel()
    # the source is inside the allocation too, unless no third class was there and it was found in a separate allocation
    assert (a0 <= lo_s and hi_s <= a0 + pinfo["arena"].numel()) or "separate allocation" in pinfo["source"]
    assert float(src.abs().max()) == 0.0
    assert "destination" in pinfo["placement"] and pinfo["classes"] >= 1
    pj.fill_random_(src, 99)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    plan.execute(src, dst)
    plain = torch.empty(dshape, dtype=torch.float64, device=dev)
    plan.execute(src.clone(), plain)
    assert torch.equal(dst.view(torch.int64), plain.view(torch.int64))
    exp = O.reproject(wcs_in, shape_in, src[:, 100:104, :].cpu().numpy(), wcs_out, shape_out, src_row0=100, src_nrows=4, dst_row0=201, dst_nrows=4)
    assert np.array_equal(dst[:, 201:205, :].cpu().numpy().view(np.int64), exp.view(np.int64))
    plan.close()


def test_place_streams(pj, dev):
    """Buffers for kernels that write several streams at once (posmap's RA and DEC maps): disjoint views of one allocation, each in a
    memory class of its own when the allocation has more than one, and posmap through them gives the bits of posmap on plain ones."""
    shape, wcs = pj.fullsky_geometry(2 * math.pi / 4800)
    (ra, dec), info = pj.place_streams([(shape[1], shape[0])] * 2, device=dev, headroom_gib=40)
    assert tuple(ra.shape) == tuple(dec.shape) == (shape[1], shape[0]) and ra.dtype == torch.float64
    lo_a, hi_a, lo_b, hi_b = ra.data_ptr(), ra.data_ptr() + ra.numel() * 8, dec.data_ptr(), dec.data_ptr() + dec.numel() * 8
    assert hi_a <= lo_b or hi_b <= lo_a
    assert len(info["buffers"]) == 2 and info["classes"] >= 1
    if info["classes"] > 1:
        assert info["buffers"][0]["class"] != info["buffers"][1]["class"]
    import ctypes as C
    lib = pj.load_library()
    sh2 = (C.c_int64 * 2)(shape[0], shape[1])
    pj._lib.check(lib.pxl_posmap_car_f64(C.byref(wcs.to_struct()), sh2, 0, shape[1], C.c_void_p(ra.data_ptr()), C.c_void_p(dec.data_ptr()), 1, None))
    ra0, dec0 = pj.posmap(shape, wcs, device=dev)
    torch.cuda.synchronize()
    assert torch.equal(ra.view(torch.int64), ra0.data.view(torch.int64)) and torch.equal(dec.view(torch.int64), dec0.data.view(torch.int64))


def test_place_pair_compact(pj, dev):
    """The head-room-free variant: after it returns exactly the two maps are allocated, they are disjoint and usable, and the
    record says how many candidates were tried and how the destination is split over classes."""
    torch.cuda.empty_cache()
    before = torch.cuda.memory_reserved(dev)
    sshape, dshape = (1, 5400, 21600), (1, 10801, 43200)          # 0.87 GiB source, 3.48 GiB destination
    src, dst, info = pj.place_pair_compact(sshape, dshape, device=dev, budget_gib=40)
    assert tuple(src.shape) == sshape and tuple(dst.shape) == dshape and float(src.abs().max()) == 0.0
    grown = torch.cuda.memory_reserved(dev) - before
    # no ballast left behind: every rejected candidate is a destination-sized (3.5 GiB) allocation; allow the allocator's own slack
    assert grown <= (src.numel() + dst.numel()) * 8 + (3 << 29), grown
    assert len(info["candidates_minor_share"]) >= 1 and "destination" in info["placement"]
    dst.fill_(1.0)
    src.fill_(2.0)
    assert float(dst.min()) == 1.0 and float(src.max()) == 2.0


def test_separate_allocation_in_another_class(pj, dev):
    """The fallback place_pair uses when its allocation holds no third class for the source: allocate separately until a candidate
    is in none of the reference windows' classes (or the budget is spent); nothing may stay allocated but the result."""
    from pixell_jl_amd import placement as P
    arena = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
    torch.cuda.empty_cache()
    before = torch.cuda.memory_reserved(dev)
    found, tried = P._separate_in_other_class((3 << 29) // 8, torch.float64, dev, [arena.data_ptr(), arena.data_ptr() + (2 << 30)], 12 << 30)
    assert tried >= 1
    grown = torch.cuda.memory_reserved(dev) - before
    if found is not None:
        assert found.numel() == (3 << 29) // 8 and grown <= (3 << 29) + (64 << 20)
        found.fill_(1.0)
    else:
        assert grown <= (64 << 20)


def test_native_pair_alloc_through_the_abi(pj, dev):
    """pxl_mem_pair_alloc / pxl_mem_pair_free (what a Julia or C host calls): the pair is usable, disjoint, inside the allocation
    it reports, the source zero-filled; a reprojection into it equals the one into plain tensors bit for bit; freeing returns the
    memory; bad arguments are refused."""
    import ctypes as C
    import math
    lib = pj.load_library()
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 10800)
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 21600)
    nx, ny = shape_in
    free0, _ = torch.cuda.mem_get_info(dev)
    src, dst, info = pj.place_pair_native((1, ny, nx), (1, shape_out[1], shape_out[0]), device=dev, headroom_gib=40)
    p = info["owner"].pair
    bs, bd = src.numel() * 8, dst.numel() * 8
    assert info["classes"] >= 1 and info["probes"] >= 1      # (a window that straddles a boundary gets a label of its own)
    assert p.dst >= p.arena and p.dst + bd <= p.arena + p.arena_bytes and p.dst % (2 << 20) == 0 and p.src % (2 << 20) == 0
    if not p.src_alloc:
        assert p.src >= p.arena and p.src + bs <= p.arena + p.arena_bytes
        assert p.src + bs <= p.dst or p.dst + bd <= p.src
    assert src.data_ptr() == p.src and dst.data_ptr() == p.dst
    assert float(src.abs().max()) == 0.0
    pj.fill_random_(src, 5)
    plan = pj.ReprojectPlan((nx, ny, 1), wcs_in, shape_out, wcs_out, device=dev)
    plan.execute(src, dst)
    ref = torch.empty(plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
    plan.execute(src.clone(), ref)
    assert torch.equal(dst, ref)
    del src, dst, ref, plan, p
    info.clear()
    torch.cuda.empty_cache()
    import gc
    gc.collect()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 >= free0 - (1 << 30)                                   # the allocation went back to the driver
    # a pair too small to map (no head-room, below 2 GiB): the plain layout, source first, both 2 MiB aligned, freed again
    tiny = pj._lib.MemPair()
    assert lib.pxl_mem_pair_alloc(3 << 20, (5 << 20) + 8, 0, C.byref(tiny), None) == 0
    assert tiny.src == tiny.arena and tiny.dst == tiny.arena + tiny.dst_offset and tiny.dst_offset >= (3 << 20)
    assert tiny.arena_bytes == (4 << 20) + (6 << 20) and tiny.classes == 1 and tiny.dst_two_classes == 0 and tiny.probes == 0
    assert lib.pxl_mem_pair_free(C.byref(tiny)) == 0 and not tiny.arena and not tiny.src
    bad = pj._lib.MemPair()
    assert lib.pxl_mem_pair_alloc(0, 1 << 20, 0, C.byref(bad), None) != 0
    assert lib.pxl_mem_pair_alloc(1 << 20, 1 << 20, 0, None, None) != 0
    assert lib.pxl_mem_pair_free(None) != 0


def test_default_allocation_policy_keeps_no_headroom(pj, O, dev):
    """pj.reproject(m, shape_out, wcs_out) without `out=` allocates its output through placement.empty_map: by default a
    destination of 3 GiB or more is looked for across a boundary between two memory classes, the rejected candidates are held
    only while the search runs, and afterwards exactly the map is allocated (no 144 GiB of head-room).
    The policy never changes a bit of the result."""
    torch.cuda.empty_cache()
    assert pj.allocation_policy() == "class-aware"
    base = torch.cuda.memory_reserved(dev)
    shape = (2, 8192, 32768)                                   # 4 GiB
    t, info = pj.empty_map(shape, device=dev, budget_gib=40)
    assert tuple(t.shape) == shape and t.dtype == torch.float64 and t.is_contiguous() and t.data_ptr() % 256 == 0
    assert info["policy"] == "class-aware" and info["tries"] >= 1 and len(info["candidates_minor_share"]) == info["tries"]
    assert info["transient_ballast_GiB"] <= 40.0
    held = torch.cuda.memory_reserved(dev) - base
    assert held <= t.numel() * 8 * 1.10 + (64 << 20), (held, info)   # nothing but the map (torch rounds blocks to 2 MiB)
    # the same block asked for again is not probed a second time (labels cached per block while the allocator has freed nothing)
    ptr = t.data_ptr()
    del t
    t2, info2 = pj.empty_map(shape, device=dev, budget_gib=40)
    if t2.data_ptr() == ptr and info2["tries"] == 1:
        assert info2["probes"] == 0, info2
    del t2
    small, sinfo = pj.empty_map((1024, 1024), device=dev)
    assert "plain" in sinfo["policy"] and tuple(small.shape) == (1024, 1024)
    old = pj.set_allocation_policy("plain")
    try:
        t3, info3 = pj.empty_map(shape, device=dev)
        assert info3["policy"] == "plain" and info3["tries"] == 1
        del t3
    finally:
        pj.set_allocation_policy(old)
    with pytest.raises(ValueError):
        pj.set_allocation_policy("fastest")
    # through the API: a 2x refinement whose output is 3.2 GiB, allocated by the library, against the oracle
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 14400)
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 28800)
    src = torch.empty((shape_in[1], shape_in[0]), dtype=torch.float64, device=dev)
    pj.fill_random_(src, 77)
    out = pj.reproject(pj.Enmap(src, wcs_in), shape_out, wcs_out)
    assert tuple(out.data.shape) == (shape_out[1], shape_out[0])
    for r in (0, 5000, shape_out[1] - 1):
        s_lo, s_hi = O.reproject_src_rows(wcs_in, (shape_in[0], shape_in[1], 1), wcs_out, shape_out, r, 1)
        exp = O.reproject(wcs_in, (shape_in[0], shape_in[1], 1), src[s_lo:s_hi].cpu().numpy()[None], wcs_out, shape_out,
                          src_row0=s_lo, src_nrows=s_hi - s_lo, dst_row0=r, dst_nrows=1)
        got = out.data[r:r + 1].cpu().numpy()[None]
        assert np.array_equal(got.view(np.int64), exp.view(np.int64)), r


def test_strip_reprojector_alloc_maps(pj, dev):
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 2400, dims=(2,))
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 4800)
    sh = pj.DecStripReprojector(shape_in, wcs_in, shape_out, wcs_out, 0, 1, dev)
    for policy in ("class-aware", "compact", "plain"):
        src, dst, info = sh.alloc_maps(policy=policy)
        assert tuple(src.shape) == tuple(sh.src_tensor_shape()) and tuple(dst.shape) == tuple(sh.dst_tensor_shape())
        assert info["policy"].startswith(policy if policy != "compact" else "class-aware") and float(src.abs().max()) == 0.0
        lo_s, hi_s, lo_d, hi_d = src.data_ptr(), src.data_ptr() + src.numel() * 8, dst.data_ptr(), dst.data_ptr() + dst.numel() * 8
        assert hi_s <= lo_d or hi_d <= lo_s
        del src, dst, info


def test_place_pair_shifted_keeps_nothing_but_the_pair(pj, dev):
    """placement.place_pair_shifted (what alloc_maps uses by default): a pair big enough to be placed (0.87 GiB source, 3.48 GiB
    destination) comes back zero-filled / writable, non-overlapping, with its report filled in, and after the call exactly the
    pair is allocated: the scout allocation and the ballast are back with the driver."""
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    sshape, dshape = (1, 5400, 21600), (1, 10801, 43200)
    src, dst, info = pj.place_pair_shifted(sshape, dshape, device=dev, scout_gib=40)
    assert tuple(src.shape) == sshape and tuple(dst.shape) == dshape and float(src.abs().max()) == 0.0
    lo_s, hi_s, lo_d, hi_d = src.data_ptr(), src.data_ptr() + src.numel() * 8, dst.data_ptr(), dst.data_ptr() + dst.numel() * 8
    assert hi_s <= lo_d or hi_d <= lo_s
    assert "placement" in info and info["seconds"] >= 0 and info["policy"].startswith("class-aware")
    dst.fill_(1.0)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    pair = (src.numel() + dst.numel()) * 8
    assert free0 - free1 <= pair + (3 << 29), ((free0 - free1) / 2**30, info)        # 1.5 GiB: the allocator's granularity, as for place_pair_compact
    # a pair too small to place is a plain allocation
    s2, d2, i2 = pj.place_pair_shifted((1, 100, 200), (1, 200, 400), device=dev)
    assert i2["placement"].startswith("plain") and float(s2.abs().max()) == 0.0


def test_native_alloc_placed(pj, dev):
    """pxl_mem_alloc_placed / pxl_mem_free: the default allocation policy through the C ABI (what the Julia HIPArray constructor
    calls): a 4 GiB buffer comes back usable, with its report filled in and nothing else left allocated; small buffers are plain."""
    import ctypes as C
    lib = pj.load_library()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    ptr = C.c_void_p()
    info = pj._lib.MemPlacedInfo()
    nbytes = 4 << 30
    pj._lib.check(lib.pxl_mem_alloc_placed(nbytes, 40 << 30, C.byref(ptr), C.byref(info), None))
    assert ptr.value and ptr.value % 256 == 0
    assert 1 <= info.tries <= 24 and 0 <= info.minor_share_pct <= 100 and info.two_classes in (0, 1)
    assert info.ballast_bytes <= 40 << 30
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 <= nbytes + (256 << 20), (free0 - free1, info.tries)        # the ballast is gone: only the buffer remains
    # the memory is real: the probe writes to it
    us = C.c_float()
    pj._lib.check(lib.pxl_mem_probe_pair(ptr, C.c_void_p(ptr.value + (2 << 30)), 1 << 30, 1, C.byref(us), None))
    assert us.value > 0
    pj._lib.check(lib.pxl_mem_free(ptr))
    small = C.c_void_p()
    pj._lib.check(lib.pxl_mem_alloc_placed(1 << 20, 0, C.byref(small), C.byref(info), None))
    assert small.value and info.tries == 1 and info.probes == 0
    pj._lib.check(lib.pxl_mem_free(small))
    assert lib.pxl_mem_alloc_placed(0, 0, C.byref(small), None, None) == -22
    assert lib.pxl_mem_free(None) == 0
