#!/usr/bin/env python3
"""Fixtures in which the reference's int arithmetic WRAPS: element::to_integer (mesh/simplicial_regular_mesh.hh:496-502, int products
(corner - lb) * dimprod) and simplex_indices (filters/regular_tracker.hh:188-194, lattice id truncated to int).  On a mesh small
enough for the CPU that needs a large time coordinate, so the series starts at tracker::set_current_timestep(T0)
(filters/tracker.hh:40; oracle/ref_driver.cpp reads FTK_REF_T0).  From the REAL reference, build container only:

    make -C oracle ref && python tests/golden/make_golden_wrap.py"""
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from refdump import read_dump  # noqa: E402


def main():
    if not os.path.exists(mg.DRIVER):
        sys.exit("build the reference driver first: make -C oracle ref")
    rng = np.random.default_rng(20261003)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "o.bin")
        # 2D: mesh 57 x 42 vertices -> dimprod[2] = 2394; T0 = 1e6: 2.39e9 > 2^31
        s2 = [np.cumsum(np.cumsum(rng.standard_normal((45, 60)), 0), 1) * 0.01 for _ in range(4)]
        t0 = 1000000
        mg.run_file(out, s2, 2, 1, env={"FTK_REF_T0": str(t0)})
        mg.save("wrap_2d_scalar_60x45x4_t1000000", read_dump(out), dict(t0=t0))
        # 3D: mesh 13 x 12 x 11 -> dimprod[3] = 1716; T0 = 2e6: 3.4e9
        s3 = [rng.standard_normal((14, 15, 16)) for _ in range(3)]
        t0 = 2000000
        mg.run_file(out, s3, 3, 1, env={"FTK_REF_T0": str(t0)})
        mg.save("wrap_3d_scalar_16x15x14x3_t2000000", read_dump(out), dict(t0=t0))
        # 2D vector input (domain [1, D-2]): 62 x 48 -> 2976; T0 = 1.5e6: 4.46e9 > 2^32 as well
        v2 = [rng.standard_normal((50, 64, 2)) for _ in range(3)]
        t0 = 1500000
        mg.run_file(out, v2, 2, 2, env={"FTK_REF_T0": str(t0)})
        mg.save("wrap_2d_vector_64x50x3_t1500000", read_dump(out), dict(t0=t0))


if __name__ == "__main__":
    main()
