#!/usr/bin/env python3
"""Generates the fixtures for REGULAR_COORDS_RECTILINEAR / _EXPLICIT (include/ftk/filters/regular_tracker.hh:39-40; simplex_coordinates
2d:494-527, 3d:342-378) from the REAL reference (oracle/_ref/ftk_ref_driver with FTK_REF_COORDS).  The coordinate arrays follow the
closed forms in oracle/ref_driver.cpp (exactly representable) and are stored in the fixture next to the records.  Build container only:

    make -C oracle ref && python tests/golden/make_golden_coords.py"""
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from refdump import read_dump  # noqa: E402


def rect(D):
    return [np.array([0.5 * i + 0.0625 * ((i * (d + 3)) % 5) + d for i in range(n)], dtype=np.float64) for d, n in enumerate(D)]


def explicit(DW, DH, nc):
    e = np.zeros((DH, DW, nc), dtype=np.float64)          # ndarray (nc, DW, DH): first index fastest
    for y in range(DH):
        for x in range(DW):
            for c in range(nc):
                e[y, x, c] = 0.75 * (x if c == 0 else y if c == 1 else 1.0) + 0.03125 * ((3 * x + 5 * y + c) % 11)
    return e


def main():
    if not os.path.exists(mg.DRIVER):
        sys.exit("build the reference driver first: make -C oracle ref")
    src2 = np.load(os.path.join(HERE, "random_2d_scalar_29x24x6.npz"))
    src3 = np.load(os.path.join(HERE, "random_3d_scalar_13x12x11x4.npz"))
    s2 = [np.array(s) for s in src2["steps"]]; D2 = [int(v) for v in src2["dims"]]
    s3 = [np.array(s) for s in src3["steps"]]; D3 = [int(v) for v in src3["dims"]]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "o.bin")
        for mode in ("rect", "explicit2", "explicit3"):
            mg.run_file(out, s2, 2, 1, env={"FTK_REF_COORDS": mode})
            extra = dict(coords=mode)
            if mode == "rect":
                extra.update({f"rect{d}": a for d, a in enumerate(rect(D2))})
            else:
                extra["explicit"] = explicit(D2[0], D2[1], int(mode[-1]))
            mg.save("random_2d_scalar_29x24x6_" + mode, read_dump(out), extra)
        for mode in ("rect", "explicit3"):
            mg.run_file(out, s3, 3, 1, env={"FTK_REF_COORDS": mode})
            extra = dict(coords=mode)
            if mode == "rect":
                extra.update({f"rect{d}": a for d, a in enumerate(rect(D3))})
            else:
                extra["explicit"] = explicit(D3[0], D3[1], 3)
            mg.save("random_3d_scalar_13x12x11x4_" + mode, read_dump(out), extra)


if __name__ == "__main__":
    main()
