"""`python bench.py --gpus N` with no launcher around it: the parent starts N fresh rank processes (before anything touches torch or the
GPU), and a rank that fails -- or ranks that never finish -- end the run with a non-zero exit code and the failing rank's stderr instead
of a hang.  (The successful several-rank run needs a GPU: tests/test_gpu_multirank.py::test_bench_launches_its_own_ranks.)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = [sys.executable, os.path.join(ROOT, "bench.py")]
ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_a_failing_rank_ends_the_run_with_its_code_and_stderr():
    t0 = time.time()
    r = subprocess.run(BENCH + ["--gpus", "3", "--backend", "gloo", "--single-device", "--config", "small3", "--fail-rank", "1"], cwd=ROOT, env=ENV,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank(s) [1] failed" in r.stderr and "rank 1 asked to fail" in r.stderr, r.stderr[-2000:]
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), "no JSON line from a failed run"
    assert time.time() - t0 < 120, "the other ranks were ended, not waited for"


def test_ranks_that_do_not_finish_are_ended_by_the_time_limit():
    r = subprocess.run(BENCH + ["--gpus", "2", "--backend", "gloo", "--single-device", "--config", "small3", "--rank-timeout", "0"], cwd=ROOT, env=ENV,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no result within 0 s" in r.stderr, (r.returncode, r.stderr[-2000:])


def test_a_world_size_that_contradicts_gpus_is_refused():
    r = subprocess.run(BENCH + ["--gpus", "2", "--config", "small3"], cwd=ROOT, env=dict(ENV, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, (r.returncode, r.stderr[-2000:])


def test_without_a_gpu_every_rank_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run")
    r = subprocess.run(BENCH + ["--gpus", "2", "--backend", "gloo", "--single-device", "--config", "small3", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=ENV,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "failed" in r.stderr and "stderr (tail)" in r.stderr, (r.returncode, r.stderr[-2000:])
