#!/usr/bin/env python3
"""Replays one case of tests/test_gpu_fuzz.py (seed, case) through every call path, with the straight-line and the general record
gather, and prints the records that differ from the oracle's.  usage: python tools/fuzz_case.py SEED CASE"""
import importlib.util
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import pyoracle as oracle  # noqa: E402
import ftk_amd as gpu  # noqa: E402
from common import by_tag  # noqa: E402
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
for case in range(6):
    nd = int(rng.choice([2, 3])); nv = int(rng.choice([1, nd])); nt = int(rng.integers(2, 6))
    if nd == 2:
        dims = (int(rng.choice([16, 24, 40, 64, 130, 136, 257])) + int(rng.integers(0, 2)), int(rng.integers(9, 70)))
    else:
        dims = (int(rng.choice([8, 16, 24, 40, 130])) + int(rng.integers(0, 2)), int(rng.integers(7, 36)), int(rng.integers(7, 20)))
    sp = tuple(reversed(dims)); kind = str(rng.choice(fz.KINDS))
    steps = fz._field(rng, (nt,) + sp, kind) if nv == 1 else fz._vector_series(rng, nt, sp, kind)
    robust = bool(rng.random() < 0.85) or nd == 2
    type_filter = int(rng.choice([1, 2, 4, 8, 16, 6, 24])) if (nd == 2 and rng.random() < 0.25) else None
    degrees = bool(nd == 2 and rng.random() < 0.15)
    tag_mode = oracle.TAG_REFERENCE if rng.random() < 0.5 else oracle.TAG_EXACT64
    mode = str(rng.choice(["tracker", "exact_prepass", "one_pass", "announced"]))
    if mode == "tracker":
        if rng.random() < 0.3:
            rng.uniform(-3.0, 5.0, size=2 * nd)   # (approximately: the bounds draw)
        rng.random()
    if case != want:
        continue
    print("case", nd, nv, dims, nt, kind, robust, type_filter, degrees, tag_mode, mode)
    ref, rf, _ = oracle.track(steps, nd, nv, robust=robust, type_filter=type_filter, compute_degrees=degrees, tag_mode=tag_mode, nthreads=8)
    ref = by_tag(ref)
    for general in ("0", "1"):
        os.environ["FTKX_RECORD_GENERAL"] = general
        for m in ("exact_prepass", "one_pass", "announced"):
            got, gf = fz._context_run(gpu, steps, nd, nv, dims, m, tag_mode, robust, type_filter, degrees)
            got = by_tag(got)
            same_tags = len(got) == len(ref) and np.array_equal(got["tag"], ref["tag"])
            bad = np.nonzero(got["type"] != ref["type"])[0] if same_tags else []
            print(f"general {general} {m}: {len(got)} records, tags equal {same_tags}, type mismatches {len(bad)}, x equal {same_tags and np.array_equal(got['x'], ref['x'], equal_nan=True)}")
            for i in bad[:3]:
                print("   tag", int(got["tag"][i]), "gpu type", int(got["type"][i]), "oracle type", int(ref["type"][i]), "x", got["x"][i], "t", got["t"][i], "ordinal", int(got["ordinal"][i]), "ts", int(got["timestep"][i]))
