"""The driver's entry points, exercised the way the driver does: __graft_entry__.smoke() and one short
bench.py run as a child process (JSON contract fields present, parity check on)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` from a bare shell: the parent starts the two ranks itself (rehearsal: both share this
    GPU over the host-staged gloo transport), relays their one JSON line and their exit code."""
    env = dict(os.environ, PXL_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--workload", "cfg2"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "must be launched with" not in r.stderr and "starting 2 ranks" in r.stderr
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["check"]["bit_identical"] and d["check"]["ranks_checked"] == 2
    assert "gloo" in d["config"]["parallelism"]


def _two_rank_rehearsal(workload, extra_env):
    env = dict(os.environ, PXL_BENCH_SHARE_GPU="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--workload", workload], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launch_two_ranks_cfg5():
    """BASELINE config 5 with two ranks: the map replicated on both, the points split by pj.strip_bounds (rank r
    generates points [lo, hi) of the one seeded sequence), no data-path collective, every rank's points checked against the oracle
    and the checks reduced.  Rehearsal: both ranks share this GPU, control plane on gloo."""
    d = _two_rank_rehearsal("cfg5", {"PXL_BENCH_POINTS": "2e6"})
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["unit"] == "Mpts/s"
    assert "points sharded x2" in d["config"]["parallelism"] and "no collective" in d["config"]["parallelism"]
    assert d["check"]["bit_identical"] and d["check"]["ranks_checked"] == 2 and d["check"]["points_checked"] > 0
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 56.0 * 1e6          # each rank's launch covers ITS million points


def test_bench_two_ranks_line_carries_both_multi_gpu_configs():
    """With N > 1 the default workload's line carries cfg4 (dec strips + halo) as the headline and, after it and outside its timed
    region, configs.cfg5 (points sharded, map replicated) -- both multi-GPU BASELINE configs in one driver line."""
    d = _two_rank_rehearsal("cfg4", {"PXL_BENCH_POINTS": "2e6"})
    assert d["n_gpus"] == 2 and d["check"]["bit_identical"] and d["check"]["ranks_checked"] == 2
    c5 = d["configs"]["cfg5"]
    assert c5["n_gpus"] == 2 and c5["Mpts_s"] > 0 and "points sharded x2" in c5["parallelism"]
    assert c5["check"]["bit_identical"] and c5["check"]["ranks_checked"] == 2


@pytest.mark.parametrize("workload", ["cfg2", "cfg2:placed", "cfg5", "cfg5:pairs"])
def test_bench_contract(workload):
    env = dict(os.environ, PXL_BENCH_POINTS="2e6")
    extra = []
    if workload.endswith(":pairs"):          # the sampler the default 1e9-point run takes (row-pair copy rebuilt inside every step)
        workload, env["PXL_BENCH_SAMPLER"] = "cfg5", "pairs"
    if workload.endswith(":placed"):         # class-aware placement of the maps (pj.place_pair), as the default run's configs block uses it
        workload, extra = "cfg2", ["--placed", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3",
                        "--warmup", "1", "--check"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["value"] > 0
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    if env.get("PXL_BENCH_SAMPLER") == "pairs":
        assert rf["kernel"] == "k_sample_pairs" and "row-pair" in d["config"]["sampler"]
        v = d["variants"]           # the same batch with the map fixed (copy reused) and through the direct sampler
        assert v["map_fixed"]["check"]["bit_identical"] and v["direct"]["check"]["bit_identical"]
        assert v["map_fixed"]["ms_per_step"] > 0 and v["direct"]["ms_per_step"] > 0
    if workload == "cfg5":
        assert d["check"]["bit_identical"] and d["check"]["points_checked"] > 0
    if workload == "cfg2":
        assert d["check"]["bit_identical"] and d["check"]["max_abs_err"] == 0.0
        if extra:
            al = d["config"]["buffer_placement"]["allocation"]
            assert al["classes"] >= 1 and "destination" in al["placement"] and al["class_runs_label_from_to_GiB"]
        else:
            cb = d["cpu_baseline"]
            assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_native_cpp_host(tmp_path):
    """A host with no Python and no torch in it: tools/native/native_bench.cpp drives the C ABI with hipMalloc'ed
    buffers (built with hipcc here, run as a child process)."""
    import pixell_jl_amd as pj
    libdir = os.path.dirname(pj.library_path())
    exe = str(tmp_path / "native_bench")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "native", "native_bench.cpp"), "-L", libdir, "-lpixell_hip",
           "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for args in (["2048", "2", "same", "3"], ["1024", "1", "refine", "3"], ["8192", "1", "refine", "3", "placed", "24"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["native"] and d["kernel_ms"] > 0 and d["unity_err"] < 1e-13
        assert d["placed"] == (len(args) > 4) and (not d["placed"] or d["classes"] >= 1)


def test_native_host_makes_its_own_communicator(tmp_path):
    """A host with no torch and no RCCL binding (what a Julia or C program is) shards through the C ABI alone: the
    communicator comes from pxl_comm_unique_id / pxl_comm_init_rank (the library loads librccl itself), the sharded
    step runs over it in loopback, and the strip is bit-identical to the unsharded map; a communicator the library
    did not create is refused (tools/native/native_sharded.cpp)."""
    import pixell_jl_amd as pj
    libdir = os.path.dirname(pj.library_path())
    exe = str(tmp_path / "native_sharded")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "native", "native_sharded.cpp"), "-L", libdir, "-lpixell_hip",
           "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PXL_RCCL_LIB", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "native_sharded ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "loaded by libpixell_hip" in r.stdout          # the library's own RCCL instance, not one found in the process


def test_plan_execute_is_graph_capturable():
    """the launch path of a reprojection plan allocates nothing and never synchronises, so a whole step
    (table build + interior + boundary launches) can be captured into ONE HIP graph and replayed; the replay must
    give the bits of the eager launches, also after the source changed."""
    import math
    import torch
    import pixell_jl_amd as pj
    dev = torch.device("cuda:0")
    shape_in, wcs_in = pj.fullsky_geometry(2 * math.pi / 1200, dims=(2,))
    shape_out, wcs_out = pj.fullsky_geometry(2 * math.pi / 2400)
    plan = pj.ReprojectPlan(shape_in, wcs_in, shape_out, wcs_out, device=dev)
    src = torch.randn((2, shape_in[1], shape_in[0]), dtype=torch.float64, device=dev)
    eager = torch.empty((2, shape_out[1], shape_out[0]), dtype=torch.float64, device=dev)
    plan.build_tables()
    plan.execute_rows(src, eager, 0, shape_out[1])
    torch.cuda.synchronize()
    replayed = torch.full_like(eager, float("nan"))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    n = shape_out[1]
    with torch.cuda.stream(s):
        def step():
            plan.build_tables()
            plan.execute_rows(src, replayed, 1, n - 2)          # interior rows ...
            plan.execute_rows(src, replayed, 0, 1)              # ... then the two boundary rows
            plan.execute_rows(src, replayed, n - 1, 1)
        step()
        torch.cuda.synchronize()
        replayed.fill_(float("nan"))
        with torch.cuda.graph(g, stream=s):
            step()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(replayed.view(torch.int64), eager.view(torch.int64))
    src.mul_(-0.5)                                              # same buffers, new contents: replay again
    plan.execute_rows(src, eager, 0, n)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(replayed.view(torch.int64), eager.view(torch.int64))
    plan.close()
