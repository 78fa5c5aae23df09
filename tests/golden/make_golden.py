#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference CPU path (oracle/_ref/ftk_ref_driver, built by
oracle/Makefile from the headers under /root/reference).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden.py

Each fixture holds DATA only: the input time series exactly as handed to the reference tracker API, the
quantisation factor in force at every sweep, and the reference's discrete critical-point records
(tag, type, ordinal, timestep, x[3], t, scalar) in the reference's own std::map order.
The adversarial_* cases feed numpy-generated fields (plateaus, exact zeros, NaN/Inf, huge values) through the
reference in `file` mode."""
import os
import subprocess
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refdump import read_dump, write_input  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ftk_ref_driver")

SYNTHETIC = [
    # name, case, DW, DH, DD, DT, x0dir
    ("woven_31x37x32", "woven", 31, 37, 1, 32, None),                 # tests/test_critical_point_tracking_woven.cpp (56 trajectories)
    ("woven_128x128x10", "woven", 128, 128, 1, 10, None),             # BASELINE config 1
    ("double_gyre_64x32x50", "double_gyre", 64, 32, 1, 50, None),     # tests/test_critical_point_tracking_double_gyre.cpp
    ("merger_2d_32x32x100", "merger_2d", 32, 32, 1, 100, None),       # tests/test_critical_point_tracking_merger_2d.cpp
    ("moving_extremum_2d_21x21x32", "moving_extremum_2d", 21, 21, 1, 32, None),
    ("moving_extremum_2d_21x21x9_aligned", "moving_extremum_2d", 21, 21, 1, 9, [10, 10, 0, 0.5, 0.25, 0]),   # lattice-aligned: SoS terms 2-5
    ("moving_extremum_3d_21x21x21x32", "moving_extremum_3d", 21, 21, 21, 32, None),
    ("moving_extremum_3d_21x21x21x4_overflow", "moving_extremum_3d", 21, 21, 21, 4,
     [10 + 1e-7, 10 + 2e-7, 10 + 3e-7, 0.1, 0.11, 0.1]),                                                     # nbits 21: int64 overflow regime
    ("moving_extremum_3d_32x32x32x8_dyadic", "moving_extremum_3d", 32, 32, 32, 8,
     [16.25, 16.375, 16.125, 0.5, 0.25, 0.125]),                                                             # the bench parameterisation
    ("moving_extremum_3d_12x10x9x5_aligned", "moving_extremum_3d", 12, 10, 9, 5, [5, 5, 4, 0.5, 0.25, 0.5]), # lattice aligned 3D: SoS terms 2-15
]


def run_synthetic(out, case, DW, DH, DD, DT, x0dir, env=None):
    cmd = [DRIVER, "synthetic", case, str(DW), str(DH), str(DD), str(DT), out]
    if x0dir is not None:
        cmd += [repr(float(v)) for v in x0dir]
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=e)
    return r.stdout.decode()


def run_file(out, steps, nd, nv, env=None):
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        inp = f.name
    write_input(inp, steps, nd, nv)
    e = dict(os.environ)
    e.update(env or {})
    subprocess.run([DRIVER, "file", inp, out], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=e)
    os.unlink(inp)


def save(name, d, extra=None):
    meta = dict(nd=d["nd"], nv=d["nv"], dims=np.array(d["dims"]), DT=d["DT"])
    meta.update(extra or {})
    if d.get("curves") is not None:
        # traced curves of the reference's pass 2: concatenated tag lists + offsets + loop flags
        meta["curve_tags"] = np.concatenate([c[1] for c in d["curves"]]) if d["curves"] else np.zeros(0, dtype=np.uint64)
        meta["curve_offsets"] = np.cumsum([0] + [len(c[1]) for c in d["curves"]]).astype(np.int64)
        meta["curve_loop"] = np.array([c[0] for c in d["curves"]], dtype=np.int32)
    if d.get("pp") is not None:
        # post-processed trajectories (json_interface::post_process defaults): per point tag, type, t
        meta["pp_offsets"] = np.cumsum([0] + [len(c[1]) for c in d["pp"]]).astype(np.int64)
        allp = np.concatenate([c[1] for c in d["pp"]]) if d["pp"] else np.zeros(0, dtype=[("tag", "<u8"), ("type", "<u4"), ("_pad", "<u4"), ("t", "<f8")])
        meta["pp_tags"] = allp["tag"]; meta["pp_types"] = allp["type"]; meta["pp_t"] = allp["t"]
        meta["pp_loop"] = np.array([c[0] for c in d["pp"]], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), steps=d["steps"], factors=d["factors"], records=d["records"], **meta)
    types, counts = np.unique(d["records"]["type"], return_counts=True)
    print(f"{name}: {len(d['records'])} records ({int(d['records']['ordinal'].sum())} ordinal) types "
          f"{dict(zip(types.tolist(), counts.tolist()))} factors {sorted(set(d['factors'].tolist()))} "
          f"curves {len(d['curves']) if d.get('curves') is not None else None} post-processed {len(d['pp']) if d.get('pp') is not None else None}")


def adversarial_2d(rng, DW, DH, DT, nv):
    steps = []
    for k in range(DT):
        if nv == 1:
            a = rng.integers(-3, 4, size=(DH, DW)).astype(np.float64)     # plateaus + exact ties everywhere
            a += 0.25 * rng.integers(-2, 3, size=(DH, DW))
        else:
            a = rng.integers(-2, 3, size=(DH, DW, 2)).astype(np.float64) * 0.5  # many exact zeros
        steps.append(a)
    return steps


def adversarial_3d(rng, D, DT, nv):
    steps = []
    for k in range(DT):
        if nv == 1:
            a = rng.integers(-2, 3, size=(D, D, D)).astype(np.float64)
        else:
            a = rng.integers(-1, 2, size=(D, D, D, 3)).astype(np.float64) * 0.25
        steps.append(a)
    return steps


def main():
    if not os.path.exists(DRIVER):
        sys.exit("build the reference driver first: make -C oracle ref")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "o.bin")
        for name, case, DW, DH, DD, DT, x0dir in SYNTHETIC:
            run_synthetic(out, case, DW, DH, DD, DT, x0dir)
            save(name, read_dump(out), dict(case=case, x0dir=np.array(x0dir if x0dir is not None else [], dtype=float)))

        rng = np.random.default_rng(20251003)
        run_file(out, adversarial_2d(rng, 17, 13, 6, 1), 2, 1); save("adversarial_2d_scalar_17x13x6", read_dump(out))
        run_file(out, adversarial_2d(rng, 15, 12, 5, 2), 2, 2); save("adversarial_2d_vector_15x12x5", read_dump(out))
        run_file(out, adversarial_3d(rng, 9, 4, 1), 3, 1); save("adversarial_3d_scalar_9x9x9x4", read_dump(out))
        run_file(out, adversarial_3d(rng, 8, 3, 3), 3, 3); save("adversarial_3d_vector_8x8x8x3", read_dump(out))

        # smooth random fields (generic position, irrational values)
        s2 = [np.cumsum(np.cumsum(rng.standard_normal((24, 29)), 0), 1) * 0.01 for _ in range(6)]
        run_file(out, s2, 2, 1); save("random_2d_scalar_29x24x6", read_dump(out))
        v2 = [rng.standard_normal((20, 23, 2)) for _ in range(5)]
        run_file(out, v2, 2, 2); save("random_2d_vector_23x20x5", read_dump(out))
        s3 = [rng.standard_normal((11, 12, 13)) for _ in range(4)]
        run_file(out, s3, 3, 1); save("random_3d_scalar_13x12x11x4", read_dump(out))
        # huge values: determinant overflow in 2D as well, NaN/Inf rejection
        h2 = [rng.standard_normal((14, 16, 2)) * 3e4 for _ in range(4)]
        h2[1][3, 4, 0] = np.nan; h2[2][5, 6, 1] = np.inf; h2[0][7, 7, 0] = 1e-9
        run_file(out, h2, 2, 2); save("huge_nan_2d_vector_16x14x4", read_dump(out))
        h3 = [rng.standard_normal((9, 9, 10)) * 50 for _ in range(3)]
        h3[1][4, 4, 4] = np.nan; h3[0][3, 3, 3] += 1e-8
        run_file(out, h3, 3, 1); save("huge_nan_3d_scalar_10x9x9x3", read_dump(out))
        # non-robust 3D path and the 2D type filter
        run_file(out, s3, 3, 1, env={"FTK_REF_NO_ROBUST": "1"}); save("random_3d_scalar_13x12x11x4_norobust", read_dump(out), dict(robust=0))
        run_file(out, s2, 2, 1, env={"FTK_REF_TYPE_FILTER": "4"}); save("random_2d_scalar_29x24x6_saddles", read_dump(out), dict(type_filter=4))

        a3 = adversarial_3d(rng, 9, 3, 1)
        run_file(out, a3, 3, 1, env={"FTK_REF_NO_ROBUST": "1"}); save("adversarial_3d_scalar_9x9x9x3_norobust", read_dump(out), dict(robust=0))
        run_file(out, a3, 3, 1); save("adversarial_3d_scalar_9x9x9x3_b", read_dump(out))

        # enable_computing_degrees (2d:653-662) and REGULAR_COORDS_BOUNDS (2d:504-510, 3d:358-365)
        run_synthetic(out, "woven", 31, 37, 1, 6, None, env={"FTK_REF_DEGREES": "1"})
        save("woven_31x37x6_degrees", read_dump(out), dict(case="woven", degrees=1))
        run_synthetic(out, "woven", 40, 33, 1, 5, None, env={"FTK_REF_BOUNDS": "-1.5,2.25,10,11.5"})
        save("woven_40x33x5_bounds", read_dump(out), dict(case="woven", bounds=np.array([-1.5, 2.25, 10, 11.5])))
        run_synthetic(out, "moving_extremum_3d", 14, 13, 12, 4, [6.25, 6.375, 5.125, 0.5, 0.25, 0.125], env={"FTK_REF_BOUNDS": "0,1,-2,2,100,130"})
        save("moving_extremum_3d_14x13x12x4_bounds", read_dump(out), dict(case="moving_extremum_3d", bounds=np.array([0, 1, -2, 2, 100, 130.0]),
                                                                          x0dir=np.array([6.25, 6.375, 5.125, 0.5, 0.25, 0.125])))

        subprocess.run([DRIVER, "tables", os.path.join(HERE, "unit_simplex_tables.txt")], check=True, stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    main()
