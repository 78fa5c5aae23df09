"""The N > 1 path of bench.py (t-slab partition, halo slice, global factors, records merged on rank 0 and traced there) exercised
on ONE GPU: two and three ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device), everything else is the
code the driver runs with `--gpus N`.  The merged record set and the curves traced from it must not depend on the number of
slabs: bit for bit the single-rank result."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _bench(n, cfg, extra=(), dump=None, env=None):
    common = ["bench.py", "--gpus", str(n), "--config", cfg, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-streaming-tracker", *extra]
    if dump:
        common += ["--dump-merged", str(dump)]
    if n == 1:
        cmd = [sys.executable, *common]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), *common, "--backend", "gloo", "--single-device"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("cfg", ["small3", "small2"])
def test_slab_count_does_not_change_the_result(cfg, tmp_path):
    one = _bench(1, cfg, dump=tmp_path / "one.npz")
    assert one["n_gpus"] == 1 and one["check"]["hits"] > 0
    ref = np.load(tmp_path / "one.npz")
    assert one["pass2"]["records"] == len(ref["records"]) == one["check"]["hits"] and one["pass2"]["curves"] == len(ref["curve_loop"]) > 0
    # host-driven batch with the whole slice / with the compact halo's host-side protocol; the device-driven slab pass (the default);
    # the slab pass with requests so small that the halo goes as a whole slice after all (both sides learn it from the same number)
    for n, extra, env in ((2, ("--full-halo",), None), (3, ("--full-halo",), None), (2, ("--full-halo", "--halo-in-loop"), None), (2, ("--compact-halo", "--host-driven"), None),
                          (2, ("--compact-halo",), None), (3, (), None), (2, ("--no-pipeline",), None), (3, (), {"FTKX_DIST_CELLS": "2"})):
        many = _bench(n, cfg, ("--no-other-scaling",) + extra, dump=tmp_path / f"many{n}.npz", env=env)
        got = np.load(tmp_path / f"many{n}.npz")
        # the merged record set (72-byte records, bit for bit) and the curves traced from it: identical to the single-rank run
        assert got["records"].tobytes() == ref["records"].tobytes()
        for k in ("curve_offsets", "curve_indices", "curve_loop"):
            assert np.array_equal(got[k], ref[k]), k
        assert many["pass2"]["curves"] == one["pass2"]["curves"] and many["pass2"]["trajectories_after_post_process"] == one["pass2"]["trajectories_after_post_process"]
        assert many["n_gpus"] == n and many["scaling"] == "strong"
        assert many["check"]["hits"] == one["check"]["hits"]
        assert many["config"]["simplices_per_step"] == one["config"]["simplices_per_step"]
        if "--full-halo" not in extra:         # (the compact halo is the default)
            # the boundary slice travelled as sign masks + patches around the surviving cells: a small fraction of its bytes
            h = many["halo_exchange"]
            moved = h["bytes_sent_per_pass_this_rank"] + h["bytes_received_per_pass_this_rank"]
            assert h["compact"] and moved > 0
            slab = "--host-driven" not in extra
            assert ("slab pass" in many["config"]["pass"]) == slab
            if slab and env is None:      # the device-driven form was what ran on rank 0, every pass
                assert many["check"]["device_driven"] and many["check"]["ok"], many["check"]
            if env is not None:     # requests of two cells: the whole slice after all, on every pass
                assert h["passes_that_fell_back_to_the_whole_slice"] > 0
            elif cfg == "small3":   # smooth 3D data: a few cells survive at the boundary -- a small fraction of the slice's bytes
                assert moved < 0.25 * h["full_slice_bytes"] and h["cells_requested_per_pass_this_rank"] > 0 and h["passes_that_fell_back_to_the_whole_slice"] == 0
            elif not slab:          # hit-dense 2D data on small slices: patches would be more bytes than the slice -> the host-side protocol falls back to it
                assert h["passes_that_fell_back_to_the_whole_slice"] > 0
        else:
            assert many["halo_exchange"]["in_timed_region"] == ("--halo-in-loop" in extra) and many["halo_exchange"]["bytes_per_rank"] > 0
        assert many["config"]["nbits"] == one["config"]["nbits"]


def test_slab_pass_with_many_records_and_more_ranks(tmp_path):
    """The device-driven slab pass where a rank's records leave through the copy kernel of a pipelined pass (more than 4 096 per rank), and
    with four ranks on smooth 3D data (five and more: tests/test_tslab.py, on the CPU): the merged records and curves are the single-rank ones, bit for bit."""
    for cfg, ranks in (("mid2", (2, 3)), ("small3", (4,))):      # (at most 6 processes may hold the GPU at once: this one + 4 ranks + a margin)
        one = _bench(1, cfg, dump=tmp_path / f"one_{cfg}.npz")
        ref = np.load(tmp_path / f"one_{cfg}.npz")
        if cfg == "mid2":
            assert one["check"]["hits"] > 3 * 4096
        for n in ranks:
            many = _bench(n, cfg, ("--no-other-scaling",), dump=tmp_path / f"many_{cfg}_{n}.npz")
            got = np.load(tmp_path / f"many_{cfg}_{n}.npz")
            assert got["records"].tobytes() == ref["records"].tobytes() and len(ref["records"]) > 0, (cfg, n)
            for k in ("curve_offsets", "curve_indices", "curve_loop"):
                assert np.array_equal(got[k], ref[k]), (cfg, n, k)
            assert "slab pass" in many["config"]["pass"] and many["check"]["device_driven"], (cfg, n, many["config"]["pass"], many["check"])


def test_strong_scaling_with_the_compact_halo_is_the_default_and_weak_scaling_is_reported_beside_it(tmp_path):
    """`bench.py --gpus N` without flags (what the driver runs): BASELINE's literal configuration cut over the ranks (strong scaling),
    the slab boundary exchanged as sign masks + patches inside the timed region; a few weak-scaling passes are reported as `other`.
    `--scaling weak`: every rank sweeps a slab of the configuration's length, i.e. the series is N times as long -- and the merged
    records / curves are those of that long series swept by one rank."""
    many = _bench(2, "small3")
    assert many["scaling"] == "strong" and many["n_gpus"] == 2
    assert many["halo_exchange"]["compact"] is True and many["halo_exchange"]["in_timed_region"] is True
    assert many["other"]["scaling"] == "weak" and many["other"]["value"] > 0 and "x16 " in many["other"]["config"]["workload"]
    nt = 8                                                        # small3
    one = _bench(1, "small3", ("--timesteps", str(3 * nt)), dump=tmp_path / "one.npz")
    weak = _bench(3, "small3", ("--scaling", "weak"), dump=tmp_path / "many.npz")
    assert weak["scaling"] == "weak" and weak["n_gpus"] == 3
    assert weak["config"]["simplices_per_step"] == one["config"]["simplices_per_step"] > 0
    ref, got = np.load(tmp_path / "one.npz"), np.load(tmp_path / "many.npz")
    assert got["records"].tobytes() == ref["records"].tobytes() and len(ref["records"]) > 0
    for k in ("curve_offsets", "curve_indices", "curve_loop"):
        assert np.array_equal(got[k], ref[k]), k


def test_the_collectives_run_on_rccl(tmp_path):
    """One rank, `--force-dist --backend nccl`: init_process_group("nccl") (RCCL), the all_gather_into_tensor of the reductions, the
    barriers, the all_reduce of the timings and the record gather execute on RCCL -- with nobody to talk to, but through the very calls
    the driver's 8-GPU run makes -- and the result is the plain single-rank one."""
    one = _bench(1, "small3", dump=tmp_path / "one.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--config", "small3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--force-dist", "--backend", "nccl",
           "--dump-merged", str(tmp_path / "rccl.npz")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["halo_exchange"]["compact"] is True
    # the device-driven slab pass: the all_gather of the contributions queued on the stream between its stages (nobody to exchange a halo with)
    assert "slab pass" in out["config"]["pass"] and out["check"]["device_driven"] and out["check"]["ok"], (out["config"]["pass"], out["check"])
    ref, got = np.load(tmp_path / "one.npz"), np.load(tmp_path / "rccl.npz")
    assert got["records"].tobytes() == ref["records"].tobytes() and len(ref["records"]) > 0
    for full in (("--full-halo",), ("--full-halo", "--halo-in-loop"), ("--compact-halo", "--host-driven")):
        r = subprocess.run(cmd[:-2] + list(full), cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]




def test_bench_launches_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 3 --backend gloo --single-device --config small3` with NO launcher around it (the shape of the command the
    driver runs for N = 1): the parent starts three fresh rank processes, rank 0's JSON line comes back with every rank accounted for, and
    the merged records are the single-rank ones byte for byte.  A rank that dies ends the run at once, the others are not waited for."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = _bench(1, "small3", dump=tmp_path / "one.npz")
    cmd = [sys.executable, "bench.py", "--gpus", "3", "--backend", "gloo", "--single-device", "--config", "small3", "--steps", "2", "--warmup", "1",
           "--no-other-scaling", "--dump-merged", str(tmp_path / "three.npz")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "ONE JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["ranks"]["ranks_seen"] == 3 and out["ranks"]["backend"] == "gloo"
    per = out["ranks"]["per_rank"]
    assert [e["rank"] for e in per] == [0, 1, 2] and len(set(e["pid"] for e in per)) == 3, per
    assert all(e["ms_per_step"] > 0 and e["pci_bus_id"] for e in per) and sum(e["timesteps"] for e in per) == 8
    assert max(e["ms_per_step"] for e in per) <= out["ms_per_step"] * 1.0001, "the line's time is the slowest rank's"
    assert all(e["slab_fallbacks"] == 0 for e in per) and "slab pass" in out["config"]["pass"] and out["check"]["ok"], (per, out["check"])
    ref, got = np.load(tmp_path / "one.npz"), np.load(tmp_path / "three.npz")
    assert got["records"].tobytes() == ref["records"].tobytes() and len(ref["records"]) == one["check"]["hits"] > 0
    for k in ("curve_offsets", "curve_indices", "curve_loop"):
        assert np.array_equal(got[k], ref[k]), k
    # a rank that dies before it joins the process group: the other two wait in the rendezvous -- ended by the parent, exit code = the rank's
    import time
    t0 = time.time()
    r = subprocess.run(cmd + ["--fail-rank", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 7 and "rank(s) [2] failed" in r.stderr and time.time() - t0 < 120, (r.returncode, r.stderr[-2000:])
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
