"""Raw C-ABI calls (ctypes straight into libpixell_hip.so, torch only owns the memory): the one-shot
reproject entry, explicit streams, HIP-graph capture of a plan execute (the launch path must not allocate
or synchronise), and error codes."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import bits_equal

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    import pixell_jl_amd as pj
    return pj, pj.load_library(), torch.device("cuda:0")


def P(t):
    return C.c_void_p(t.data_ptr())


def test_device_count_and_version(env):
    pj, lib, dev = env
    assert lib.pxl_device_count() >= 1
    assert lib.pxl_version() == 100


def test_one_shot_reproject_entry(env, O):
    pj, lib, dev = env
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 200, dims=(2,))
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 300)
    src = np.random.default_rng(0).normal(size=(2, shape_in[1], shape_in[0]))
    d_src = torch.from_numpy(src).to(dev)
    d_dst = torch.empty((2, shape_out[1], shape_out[0]), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    win, wout = wcs_in.to_struct(), wcs_out.to_struct()
    rc = lib.pxl_reproject_car_bilinear_f64(C.byref(win), pj._lib.shape_arr(shape_in), P(d_src), C.byref(wout),
                                            pj._lib.shape_arr(shape_out), P(d_dst), None)
    assert rc == 0, pj._lib.last_error()
    # the one-shot entry synchronises its stream before returning
    assert bits_equal(d_dst.cpu().numpy(), O.reproject(wcs_in, shape_in, src, wcs_out, shape_out))


def test_explicit_stream(env, O):
    pj, lib, dev = env
    shape, wcs = pj.fullsky_geometry(math.radians(1))
    pix = np.random.default_rng(1).uniform(0, 300, (100000, 2))
    side = torch.cuda.Stream(device=dev)
    d_pix = torch.from_numpy(pix).to(dev)
    d_sky = torch.empty_like(d_pix)
    torch.cuda.synchronize()
    w = wcs.to_struct()
    rc = lib.pxl_pix2sky_car_f64(C.byref(w), d_pix.shape[0], P(d_pix), P(d_sky), 0, C.c_void_p(side.cuda_stream))
    assert rc == 0
    side.synchronize()
    assert bits_equal(d_sky.cpu().numpy(), O.pix2sky(wcs, pix, O.WRAP_NONE))


def test_plan_execute_is_graph_capturable(env):
    """No hipMalloc / synchronisation in the launch path (cdna_hip_programming.md Guideline 9)."""
    pj, lib, dev = env
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 512, dims=(3,))
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 1024)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    src = torch.randn(plan.src_tensor_shape(), dtype=torch.float64, device=dev)
    ref = torch.empty(plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
    plan.execute(src, ref)
    torch.cuda.synchronize()
    dst = torch.zeros_like(ref)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.execute(src, dst)
    assert float(dst.abs().max()) == 0.0          # capture does not run anything
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(dst, ref)
    src.mul_(2.0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(dst, 2.0 * ref)


def test_error_codes(env):
    pj, lib, dev = env
    shape, wcs = pj.fullsky_geometry(math.radians(1))
    w = wcs.to_struct()
    buf = torch.zeros((8, 2), dtype=torch.float64, device=dev)
    assert lib.pxl_pix2sky_car_f64(C.byref(w), -1, P(buf), P(buf), 0, None) == -22
    assert lib.pxl_pix2sky_car_f64(C.byref(w), 8, P(buf), P(buf), 7, None) == -22
    assert "wrap_mode" in pj._lib.last_error()
    assert lib.pxl_pix2sky_car_f64(C.byref(w), 8, None, P(buf), 0, None) == -22
    odd = C.c_void_p(buf.data_ptr() + 8)           # 2xN buffers must be 16-byte aligned
    assert lib.pxl_pix2sky_car_f64(C.byref(w), 4, odd, P(buf), 0, None) == -22
    assert lib.pxl_sky2pix_car_f64(C.byref(w), pj._lib.shape_arr(shape), 8, P(buf), P(buf), 1, 9, None) == -22
    assert lib.pxl_posmap_car_f64(C.byref(w), pj._lib.shape_arr(shape), 100, 500, P(buf), P(buf), 1, None) == -22
    h = C.c_void_p()
    bad_shape = pj._lib.shape_arr((0, 5, 1))
    assert lib.pxl_reproject_plan_create(C.byref(w), bad_shape, 0, 5, C.byref(w), pj._lib.shape_arr(shape), 0, 181,
                                         C.byref(h)) == -22
    assert not h.value
    # empty work is a successful no-op
    assert lib.pxl_pix2sky_car_f64(C.byref(w), 0, None, None, 0, None) == 0
    assert lib.pxl_sample_car_bilinear_f64(C.byref(w), pj._lib.shape_arr((360, 181, 1)), None, 0, 0, 0, None, None, None) == 0


def test_concurrent_host_threads_and_streams(env, O):
    """The boundary claims thread safety (thread-local error text, no shared mutable state): four host threads,
    each with its own stream, plan and buffers, reproject different geometries at the same time."""
    import threading
    pj, lib, dev = env
    cases = []
    rng = np.random.default_rng(9)
    for n_in, n_out in ((96, 192), (200, 150), (128, 128), (90, 360)):
        shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / n_in)
        shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / n_out)
        src = rng.normal(size=(1, shape_in[1], shape_in[0]))
        cases.append((shape_in, wcs_in, shape_out, wcs_out, src, O.reproject(wcs_in, shape_in, src, wcs_out, shape_out)))
    results, errors = [None] * len(cases), []

    def work(k):
        try:
            shape_in, wcs_in, shape_out, wcs_out, src, _ = cases[k]
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                d_src = torch.from_numpy(src).to(dev, non_blocking=False)
                for _ in range(20):                       # plan churn + execution on this thread's stream
                    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
                    dst = torch.empty(plan.dst_tensor_shape(), dtype=torch.float64, device=dev)
                    plan.execute(d_src, dst)
                    st.synchronize()
                    plan.close()
                results[k] = dst.cpu().numpy()
            # a failing call on this thread leaves ITS message, not another thread's
            rc = lib.pxl_pix2sky_car_f64(None, 0, None, None, 0, None)
            assert rc == -22 and "WCS" in pj._lib.last_error()
        except Exception as e:                            # pragma: no cover
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, case in enumerate(cases):
        assert bits_equal(results[k], case[5]), k


def test_scratch_pool_is_kept_and_released(env):
    """unwind scratch comes from a library-owned pool that survives between calls; pxl_release_scratch hands it back."""
    pj, lib, dev = env
    n = 2_000_000
    g = ((10, 10), pj.CarClenshawCurtis((1.0, 1.0), (0.0, 0.0), (0.0, 0.0), 1.0))
    pix = torch.rand((n, 2), dtype=torch.float64, device=dev) * 40 - 20
    first = pj.pix2sky(g, pix, safe=True)
    torch.cuda.synchronize()
    assert lib.pxl_release_scratch() == 0
    again = pj.pix2sky(g, pix, safe=True)           # the pool re-grows on demand
    assert torch.equal(first, again)
    torch.cuda.synchronize()
    assert lib.pxl_release_scratch() == 0


def test_no_device_memory_leak(env):
    """Plans (tables, zero page, exchange stream and events), one-shot entries and the unwind scratch pool give their
    device memory back: free memory after a few hundred create/execute/destroy cycles is what it was before."""
    pj, lib, dev = env
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 400, dims=(2,))
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 600)
    src = torch.randn((2, shape_in[1], shape_in[0]), dtype=torch.float64, device=dev)
    dst = torch.empty((2, shape_out[1], shape_out[0]), dtype=torch.float64, device=dev)
    pix = torch.rand((50_000, 2), dtype=torch.float64, device=dev) * 300
    g = (shape_in[:2], wcs_in)

    def cycle(n):
        for _ in range(n):
            plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
            plan.execute(src, dst)
            plan.close()
            pj.pix2sky(g, pix, safe=True)
        torch.cuda.synchronize()
        assert lib.pxl_release_scratch() == 0

    cycle(20)                                           # warm every lazy allocation (module load, pools, RNG state)
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info(dev)
    cycle(300)
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert abs(free1 - free0) <= 64 << 20, (free0, free1)
