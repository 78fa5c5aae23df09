"""The oracle against THE REFERENCE ITSELF on seeded random series (not gpu): oracle/_ref/ftk_ref_driver -- hguo/ftk's own trackers
compiled from its headers by oracle/Makefile -- tracks each series in `file` mode; the C restatement must give the same records
(bit for bit), factors included.  Pins the oracle beyond the committed fixtures; skipped where the binary was not built."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import assert_records_equal
from refdump import read_dump, write_input
from test_gpu_fuzz import _field, _vector_series, KINDS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ftk_ref_driver")


@pytest.fixture(scope="module")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/ftk_ref_driver not built")
@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_the_real_reference_on_random_series(oracle, seed):
    rng = np.random.default_rng(7000 + seed)
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
        degrees = bool(nd == 2 and rng.random() < 0.15)
        env = {}
        if not robust:
            env["FTK_REF_NO_ROBUST"] = "1"
        if type_filter is not None:
            env["FTK_REF_TYPE_FILTER"] = str(type_filter)
        if degrees:
            env["FTK_REF_DEGREES"] = "1"
        with tempfile.TemporaryDirectory() as tmp:
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            write_input(inp, steps, nd, nv)
            e = dict(os.environ); e.update(env)
            subprocess.run([DRIVER, "file", inp, out, "8"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=e, timeout=300)
            ref = read_dump(out)
        got, gf, _ = oracle.track(steps, nd, nv, robust=robust, type_filter=type_filter, compute_degrees=degrees, tag_mode=oracle.TAG_REFERENCE, nthreads=8)
        what = f"seed {seed} case {case}: nd {nd} nv {nv} dims {dims} nt {nt} {kind} robust {robust} filter {type_filter} degrees {degrees}"
        assert np.array_equal(gf, ref["factors"]), what + f": factors {gf} vs {ref['factors']}"
        assert_records_equal(got, ref["records"], coord_tol=0.0, what=what)
