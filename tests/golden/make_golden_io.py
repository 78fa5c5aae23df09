#!/usr/bin/env python3
"""Generates tests/golden/io_*.npz: the files the REAL reference writes for one run -- discrete critical points and traced
(post-processed) trajectories, each as json / diy-binary / text -- through its own writers
(filters/critical_point_tracker.hh:106-108, 339-343, 354-373, 392-396, 475-484), driven by oracle/_ref/ftk_ref_driver with
FTK_REF_WRITE_PREFIX.  Build container only:

    make -C oracle ref && python tests/golden/make_golden_io.py

A fixture is DATA: the six output files as byte arrays, next to the name of the record fixture (same run) they belong to."""
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

KINDS = ["discrete.json", "discrete.bin", "discrete.txt", "traced.json", "traced.bin", "traced.txt"]


def collect(prefix, name, of):
    d = {k.replace(".", "_"): np.frombuffer(open(prefix + "." + k, "rb").read(), dtype=np.uint8) for k in KINDS}
    np.savez_compressed(os.path.join(HERE, "io_" + name + ".npz"), records_fixture=of, **d)
    print(name, {k: len(v) for k, v in d.items()})


def main():
    if not os.path.exists(mg.DRIVER):
        sys.exit("build the reference driver first: make -C oracle ref")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "o.bin")
        for name, case, DW, DH, DD, DT, x0dir in mg.SYNTHETIC:
            if name not in ("merger_2d_32x32x100", "moving_extremum_3d_21x21x21x32"):
                continue
            prefix = os.path.join(tmp, name)
            mg.run_synthetic(out, case, DW, DH, DD, DT, x0dir, env={"FTK_REF_WRITE_PREFIX": prefix})
            collect(prefix, name, name)
        # irrational values (digit-generation stress for the JSON numbers): the inputs of an existing record fixture
        name = "random_2d_scalar_29x24x6"
        z = np.load(os.path.join(HERE, name + ".npz"))
        steps = [np.array(s) for s in z["steps"]]
        prefix = os.path.join(tmp, name)
        mg.run_file(out, steps, 2, 1, env={"FTK_REF_WRITE_PREFIX": prefix})
        collect(prefix, name, name)


if __name__ == "__main__":
    main()
