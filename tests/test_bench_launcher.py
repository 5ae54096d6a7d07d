"""`python bench.py --gpus N` from a bare shell: with no WORLD_SIZE in the environment the process
starts the N ranks itself (a child `torch.distributed.run` job, before anything touches the GPU), relays the JSON line and
the exit code.  On this CPU-only box the ranks then stop at "needs an MI355X" -- which is exactly what shows they were
started; the 2-rank flow on a real GPU is tests/test_gpu_entrypoints.py::test_bench_self_launch_two_ranks."""
import os
import subprocess
import sys

from conftest import ROOT


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_n_without_torchrun_starts_its_own_ranks():
    import torch
    env = _clean_env()
    env["PXL_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--workload", "cfg2"], capture_output=True, text=True, timeout=600, env=env)
    assert "must be launched with" not in r.stderr + r.stdout
    assert "starting 2 ranks" in r.stderr
    if not torch.cuda.is_available():
        # the ranks ran bench.py's main() and refused to run without the GPU; the job's code came back.  (torchrun tears the job
        # down as soon as one rank fails, so the second rank does not always get to print its own refusal.)
        assert r.returncode != 0
        assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_still_refused():
    env = _clean_env()
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not match --gpus" in r.stderr
