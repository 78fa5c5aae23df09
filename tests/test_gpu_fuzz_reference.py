"""Randomised parity against THE REFERENCE ITSELF: oracle/_ref/ftk_ref_driver (hguo/ftk's own trackers, compiled from its headers by
oracle/Makefile; the binary travels to the GPU box, the sources do not) tracks seeded random series in `file` mode; the HIP path
must produce the same records (tags in the reference's int32-wrapping form, types, ordinal / timestep, coordinates and scalars bit
for bit), the same quantisation factors, -- after pass 2 on the host -- the same traced curves and the same post-processed
trajectories (per point: tag, smoothed type, adjusted time), and in enable_streaming_trajectories mode the same trajectories in the
same order of birth.  Skipped where the binary was
not built (no /root/reference at build time)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import assert_records_equal
from refdump import read_dump, write_input
from test_gpu_fuzz import _field, _vector_series, KINDS, FUZZ_OFFSET

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ftk_ref_driver")
ABORTED = []
COMPARED = {"cases": 0}


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _reference(steps, nd, nv, env):
    with tempfile.TemporaryDirectory() as tmp:
        inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        write_input(inp, steps, nd, nv)
        e = dict(os.environ); e.update(env)
        subprocess.run([DRIVER, "file", inp, out, "8"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=e, timeout=300)
        return read_dump(out)


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/ftk_ref_driver not built")
@pytest.mark.parametrize("seed", range(25))
def test_random_series_equal_the_real_reference(gpu, seed):
    from gpu_common import run_tracker
    rng = np.random.default_rng(5000 + FUZZ_OFFSET + seed)
    for case in range(4):
        nd = int(rng.choice([2, 3]))
        nv = int(rng.choice([1, nd]))
        nt = int(rng.integers(2, 7))
        if nd == 2:
            dims = (int(rng.choice([16, 24, 40, 64, 130])) + int(rng.integers(0, 2)), int(rng.integers(9, 60)))
        else:
            dims = (int(rng.choice([8, 16, 24, 40])) + int(rng.integers(0, 2)), int(rng.integers(7, 30)), int(rng.integers(7, 18)))
        sp = tuple(reversed(dims))
        kind = str(rng.choice(KINDS))
        steps = _field(rng, (nt,) + sp, kind) if nv == 1 else _vector_series(rng, nt, sp, kind)
        robust = bool(rng.random() < 0.85) or nd == 2
        type_filter = int(rng.choice([1, 2, 4, 8, 16, 6, 24])) if (nd == 2 and rng.random() < 0.25) else None
        env = {}
        if not robust:
            env["FTK_REF_NO_ROBUST"] = "1"
        if type_filter is not None:
            env["FTK_REF_TYPE_FILTER"] = str(type_filter)
        what = f"seed {seed} case {case}: nd {nd} nv {nv} dims {dims} nt {nt} {kind} robust {robust} filter {type_filter}"
        streaming = bool(rng.random() < 0.3)             # enable_streaming_trajectories: trajectories grown while the sweep streams
        if streaming:
            env["FTK_REF_STREAMING"] = "1"
            what += " streaming"
        try:
            ref = _reference(steps, nd, nv, env)
        except subprocess.CalledProcessError:
            # the reference aborts on this input (streaming mode asserts `linear_graphs.size() == 1` in trace_critical_points_online,
            # critical_point_tracker.hh, when new points form a branching component): nothing to compare with; the product
            # must still come back -- with trajectories or with an error, not with a crash
            assert streaming, what + ": the reference failed outside streaming mode"
            try:
                run_tracker(steps, nd, nv, robust=robust, type_filter=type_filter, streaming=True)
            except Exception as e:   # noqa: BLE001
                assert "ftkx" in type(e).__name__.lower() or "Ftkx" in type(e).__name__, repr(e)
            ABORTED.append(what)
            COMPARED["cases"] += 1
            continue
        state = {}
        COMPARED["cases"] += 1

        def after(tr):
            if streaming:
                tr.finalize()
                state["curves"] = tr.get_traced_critical_points()
            tr.post_process()
            state["pp"] = tr.get_traced_trajectories()
        out = run_tracker(steps, nd, nv, robust=robust, type_filter=type_filter, want_curves=not streaming, streaming=streaming,
                          device=bool(rng.random() < 0.5), after=after)
        got, gf = out[0], out[1]
        assert np.array_equal(np.asarray(gf, dtype=np.uint64), ref["factors"]), what + f": factors {gf} vs {ref['factors']}"
        if streaming:
            # same trajectories in the same order of birth; what is left as discrete points: the last step's ordinal points
            curves, loop = state["curves"]
            mine = [(tuple(c.tolist()), int(l)) for c, l in zip(curves, loop)]
            theirs = [(tuple(t.tolist()), int(l)) for l, t in ref["curves"]]
            assert mine == theirs, what + f": {len(mine)} trajectories, the reference grew {len(theirs)}"
            assert np.array_equal(np.sort(got["tag"]), np.sort(ref["records"]["tag"])), what + ": discrete points left behind"
        else:
            assert_records_equal(got, ref["records"], coord_tol=0.0, what=what)
            curves, loop = out[3]
            mine = sorted((tuple(c.tolist()), int(l)) for c, l in zip(curves, loop))
            theirs = sorted((tuple(t.tolist()), int(l)) for l, t in ref["curves"])
            assert mine == theirs, what + f": {len(mine)} curves, the reference traced {len(theirs)}"
        # json_interface::post_process with its defaults: per point tag, smoothed type, adjusted time
        mine = sorted((tuple(tg.tolist()), tuple(ty.tolist()), tuple(tt.tolist()), lp) for tg, ty, tt, lp, _ in state["pp"])
        theirs = sorted((tuple(pts["tag"].tolist()), tuple(pts["type"].tolist()), tuple(pts["t"].tolist()), lp) for lp, pts in ref["pp"])
        assert mine == theirs, what + f": {len(mine)} post-processed trajectories, the reference has {len(theirs)}"


def test_the_reference_aborted_on_few_series():
    """(runs after the seeds above) the series skipped because the REFERENCE aborts on them in streaming mode stay a small minority: the
    comparison above is not quietly hollowed out"""
    import pytest
    if not COMPARED.get("cases"):
        pytest.skip("the seeds above did not run in this process")
    assert len(ABORTED) <= max(3, COMPARED["cases"] // 8), (len(ABORTED), COMPARED["cases"], ABORTED[:5])
