#!/usr/bin/env python3
"""Fixtures of the reference's enable_streaming_trajectories mode (critical_point_tracker.hh:38, 523-639; update_timestep 2d:326-330,
3d:197-201): the trajectories trace_critical_points_online grows while the sweep streams, as tag sequences + loop flags, and the
discrete points it leaves behind (the last step's ordinal points, which no interval sweep follows).  The inputs and the sweep's
records are those of the corresponding non-streaming fixture.  From the REAL reference (FTK_REF_STREAMING=1), build container only:

    make -C oracle ref && python tests/golden/make_golden_streaming.py"""
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from refdump import read_dump  # noqa: E402

CASES = [("woven_31x37x32", "woven", 31, 37, 1, 32, None), ("woven_128x128x10", "woven", 128, 128, 1, 10, None),
         ("double_gyre_64x32x50", "double_gyre", 64, 32, 1, 50, None), ("merger_2d_32x32x100", "merger_2d", 32, 32, 1, 100, None),
         ("moving_extremum_3d_21x21x21x32", "moving_extremum_3d", 21, 21, 21, 32, None)]


def main():
    if not os.path.exists(mg.DRIVER):
        sys.exit("build the reference driver first: make -C oracle ref")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "o.bin")
        for name, case, DW, DH, DD, DT, x0dir in CASES:
            mg.run_synthetic(out, case, DW, DH, DD, DT, x0dir, env={"FTK_REF_STREAMING": "1"})
            d = read_dump(out)
            np.savez_compressed(os.path.join(HERE, "streaming_" + name + ".npz"), of=name,
                                curve_tags=np.concatenate([c[1] for c in d["curves"]]), curve_offsets=np.cumsum([0] + [len(c[1]) for c in d["curves"]]).astype(np.int64),
                                curve_loop=np.array([c[0] for c in d["curves"]], dtype=np.int32), leftover_tags=np.sort(d["records"]["tag"]),
                                pp_count=len(d["pp"]))
            print(name, len(d["curves"]), "trajectories,", int(sum(len(c[1]) for c in d["curves"])), "points,", len(d["records"]), "discrete points left,", len(d["pp"]), "after post_process")


if __name__ == "__main__":
    main()
