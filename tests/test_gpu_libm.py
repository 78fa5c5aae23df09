"""The one dependency of the record path on the C library, pinned.

The class of a 3D record whose Hessian has an eigenvalue that is zero up to rounding hangs on the last bits of pow / acos / cos
(numeric/eigen_solver3.hh:20-47): the kernels flag such records and the HOST classifies them with its libm (cp_device.hpp classify3,
`ftkx_stats.reclassified`).  The fixtures tests/golden/singular_*.npz were made by the reference itself in the build container
(tests/golden/make_golden_singular.py; the container's glibc version is stored with them) out of fields that produce such records
by the thousand.  Here, on the GPU box: the path is really taken (reclassified > 0, over 10^4 in total) and every type equals the
fixture's -- a box whose libm rounds differently from the reference run fails this test by name instead of drifting silently."""
import ctypes

import numpy as np
import pytest

from common import assert_records_equal, golden_names, load_golden

pytestmark = pytest.mark.gpu

SINGULAR = [n for n in golden_names() if n.startswith("singular_")]
TOTAL = {"reclassified": 0, "records": 0, "fixtures": 0}


def _glibc():
    f = ctypes.CDLL(None).gnu_get_libc_version
    f.restype = ctypes.c_char_p
    return f().decode()


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


@pytest.mark.parametrize("name", SINGULAR)
def test_near_singular_hessians_get_the_reference_class(gpu, name):
    from gpu_common import run_tracker
    import os
    g = load_golden(name)
    made_with = str(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))["glibc"])
    here = _glibc()
    recs, factors, stats = run_tracker(g["steps"], 3, 1)
    re = sum(int(s["reclassified"]) for s in stats)
    what = f"{name}: fixture made with glibc {made_with}, this box has glibc {here}; {re} of {len(recs)} records classified with the host's libm"
    assert np.array_equal(factors, g["factors"]), what
    assert re > 0, what + " -- the fixture does not exercise the host classification here"
    got_t = recs[np.argsort(recs["tag"], kind="stable")]["type"]
    ref = g["records"][np.argsort(g["records"]["tag"], kind="stable")]
    assert len(recs) == len(ref), what
    bad = int((got_t != ref["type"]).sum())
    assert bad == 0, what + f": {bad} types differ from the reference run" + (" -- the C libraries differ: pow / acos / cos round differently" if here != made_with else "")
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=what)
    TOTAL["reclassified"] += re; TOTAL["records"] += len(recs); TOTAL["fixtures"] += 1
    print(what)


def test_the_pin_is_not_vacuous(gpu):
    if TOTAL["fixtures"] != len(SINGULAR):
        pytest.skip("the fixtures above did not all run in this process")
    print(TOTAL, "glibc here:", _glibc())
    assert TOTAL["reclassified"] >= 10000, TOTAL
