"""BASELINE's 3D configurations at FULL size against the ORACLE (oracle/ftk_oracle.c, pinned to the reference) -- not only against the analytic
trajectory and this repository's own exact_only path: moving_extremum_3d 256^3 x 16 (C3) and two slices of 512^3 (C4's slice size, its ZPlan
pieces, its tile placement; the oracle over the 160 planes around the extremum's path), records bit-identical, factors equal (critical_point_tracker_3d_regular.hh:150-308, 425-514).

The oracle needs 1.5e10 + 8.8e9 simplices' worth of host time (about four minutes of the GPU box's CPU share).  It runs in a BACKGROUND thread
from the start of the session (tests/conftest.py: pytest_collection_finish) on arrays generated on the host, while the other GPU tests run;
this module sorts last and only waits for what is left.  The host-generated arrays are checked to be the arrays the GPU swept, bit for bit."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIG_HOST = (os.cpu_count() or 1) >= 128
CASES = {"c3": ((256, 256, 256), 16), "c4x2": ((512, 512, 512), 2)}
_JOB = {"thread": None, "out": {}, "err": None}


def _host_steps(dims, nt, nt_of_series):
    import pyoracle
    from ftk_amd import synthetic
    x0, dv = synthetic.moving_extremum_params(dims)
    return [pyoracle.synthetic("moving_extremum_3d", list(dims), k, nt_of_series, list(x0), list(dv)) for k in range(nt)]


C4_CORE_Z = (176, 160)      # the oracle sweeps the planes [176, 336) of the 512^3 slices: the extremum's path lies inside; a third of the simplices


def _work():
    try:
        import time
        import pyoracle
        pyoracle.build()
        nth = os.cpu_count() or 1
        for name, (dims, nt) in CASES.items():
            steps = _host_steps(dims, nt, nt)
            t0 = time.time()
            if name == "c3":
                ref, rf, _ = pyoracle.track(steps, 3, 1, tag_mode=pyoracle.TAG_EXACT64, nthreads=nth)
                rf = [int(f) for f in rf]
            else:
                # two 512^3 slices: the reference's step by hand (critical_point_tracker_3d_regular.hh:150-308) -- V, J derived, the sticky factor,
                # ordinal + interval sweep of step 0 and the ordinal sweep of step 1 -- over a CORE of 160 planes around the extremum's path
                V = [pyoracle.gradient3D(s) for s in steps]
                J = [pyoracle.jacobian3D(v) for v in V]
                dom = ([2] * 3, [d - 3 for d in dims])
                core = ([2, 2, C4_CORE_Z[0]], [dims[0] - 3, dims[1] - 3, C4_CORE_Z[1]])
                res, parts, rf = 1.7976931348623157e308, [], []
                for t in range(nt):
                    for u in (t, t + 1):
                        if u < nt:
                            res = min(res, pyoracle.resolution(V[u]))
                    factor, _ = pyoracle.scaling_factor(res)
                    rf.append(int(factor))
                    for scope in ((1, 2) if t + 1 < nt else (1,)):
                        nxt = t + 1 if scope == 2 else None
                        parts.append(pyoracle.sweep(3, scope, t, dom, core, ([0] * 3, list(dims)), (V[t], V[nxt] if nxt is not None else None),
                                                    (J[t], J[nxt] if nxt is not None else None), (steps[t], steps[nxt] if nxt is not None else None), factor,
                                                    jacobian_symmetric=True, tag_mode=pyoracle.TAG_EXACT64, nthreads=nth))
                ref = np.concatenate(parts)
                del V, J
            ref = ref[np.argsort(ref["tag"], kind="stable")]
            _JOB["out"][name] = (ref, rf, [s[::37, ::41, ::43].copy() for s in steps], [float(s.sum()) for s in steps], time.time() - t0)
            del steps
    except BaseException as e:      # noqa: BLE001
        _JOB["err"] = e


def start_background():
    """called once, when the session has collected tests of this module (tests/conftest.py)"""
    if BIG_HOST and _JOB["thread"] is None:
        _JOB["thread"] = threading.Thread(target=_work, daemon=True)
        _JOB["thread"].start()


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _oracle_result(name):
    start_background()
    t = _JOB["thread"]
    while name not in _JOB["out"] and _JOB["err"] is None and t.is_alive():
        t.join(1.0)
    if _JOB["err"] is not None:
        raise _JOB["err"]
    return _JOB["out"][name]


@pytest.mark.skipif(not BIG_HOST, reason="the oracle over 1e10 simplices needs the GPU box's host threads")
@pytest.mark.parametrize("name", ["c3", "c4x2"])
def test_fullsize_3d_series_vs_oracle(gpu, name):
    """test_c3_series_vs_oracle / test_c4_two_slices_vs_oracle: the timed path (two passes in flight) at full size, bit for bit the oracle"""
    from test_gpu_fullsize_series import Resident, SERIES_EARLY, _bytes_equal
    dims, nt = CASES[name]
    R = Resident(gpu, "moving_extremum_3d", dims, nt, 1, keep_host=False)
    try:
        runs = R.pipelined(3)
        recs, f, path = runs[-1]
        # (two in flight, sparse, 2 GB and more per mask launch: the split pass, path 5; 512^3 x 2 is one 1 GB slice per launch short of that: the fused tail)
        assert path in ((2, SERIES_EARLY), (5, 0)) and all(_bytes_equal(r, recs) and p in ((2, SERIES_EARLY), (5, 0)) for r, _, p in runs)
        # the arrays the oracle was given are the arrays the GPU swept: a strided sample and the sum of every slice, bit for bit
        samples = [a[::37, ::41, ::43].cpu().numpy() for a in R.keep]
        sums = [float(a.cpu().numpy().sum()) for a in R.keep]
    finally:
        R.close()
    ref, rf, ref_samples, ref_sums, secs = _oracle_result(name)
    for t in range(nt):
        assert np.array_equal(samples[t], ref_samples[t]) and sums[t] == ref_sums[t], (name, t, "host-generated slice differs from the device's")
    assert rf == f, (name, rf, f)
    if name == "c4x2":      # (the oracle swept a core of planes: every record of the GPU lies inside it -- the one extremum -- and is the oracle's)
        z = recs["x"][:, 2]
        assert np.all((z >= C4_CORE_Z[0]) & (z < C4_CORE_Z[0] + C4_CORE_Z[1])), (z.min(), z.max())
    assert len(ref) == len(recs) > 0, (name, len(ref), len(recs))
    assert np.array_equal(ref["tag"], recs["tag"]) and np.array_equal(ref["type"], recs["type"]), name
    assert np.array_equal(ref["ordinal"].astype(np.uint32), recs["aux"] & 1) and np.array_equal(ref["timestep"].astype(np.uint32), recs["aux"] >> 1), name
    for fld in ("x", "t"):
        assert np.array_equal(ref[fld], recs[fld]), (name, fld)           # bit-identical (north_star asks for 1e-6)
    assert np.array_equal(ref["scalar"][:, 0], recs["scalar"][:, 0]), name
    print(f"{name}: {len(recs)} records identical to the oracle's ({secs:.0f} s of oracle work in the background)")
