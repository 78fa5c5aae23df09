"""The oracle itself (oracle/ftk_oracle.c built with ASan + UBSan; -fwrapv keeps its int64 wrap-around defined, as in the normal
build) over every record fixture, multi-threaded.  A checker with a memory bug would be no checker."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle  # noqa: E402
from common import golden_names, load_golden, assert_records_equal  # noqa: E402

for name in golden_names():
    g = load_golden(name)
    recs, factors, _ = pyoracle.track(g["steps"], g["nd"], g["nv"], robust=g["robust"], type_filter=g["type_filter"], compute_degrees=g["degrees"],
                                      bounds=g["bounds"], tag_mode=pyoracle.TAG_REFERENCE, nthreads=4, rectilinear=g["rectilinear"], explicit=g["explicit"])
    assert np.array_equal(factors, g["factors"]), name
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)
print("oracle sanitizer run complete")
