#!/usr/bin/env python3
"""Phase stamps of the fused tail (series_small_kernel) -- needs a build with FTKX_EXTRA_CFLAGS=-DFTKX_SMALL_STAMPS.
usage: python tools/small_stamps.py [c3|c4s|...]    prints, per pass, the latest passage of every phase boundary in us after the kernel's earliest start"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ftk_amd
from ftk_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
CONFIGS = {"c3": ((256, 256, 256), 16), "c4s": ((512, 512, 512), 1), "c4n4": ((512, 512, 512), 4)}
dims, nt = CONFIGS[cfg]
dev = torch.device("cuda", 0)
ctx = ftk_amd.Context(3)
dom = ([2] * 3, [d - 3 for d in dims])
ctx.set_mesh(dom, dom, ([0] * 3, list(dims)))
ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
keep = []
for t in range(nt):
    a = synthetic.generate("moving_extremum_3d", dims, t, nt, torch, dev); torch.cuda.synchronize(); keep.append(a)
    ctx.push_scalar_slice(t, a)
scopes = [ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL for t in range(nt)]
L = ctx._L
L.ftkx_debug_small_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
names = ["start", "setup", "refine", "exact", "records", "counters", "arrived(last)", "fence", "ranked", "copied", "flag"]
for rows in ("4", "16", "4", "16"):
    os.environ["FTKX_U_ROWS"] = rows
    for rep in range(3):
        ctx.invalidate_masks()
        L.ftkx_debug_small_stamps(buf, 1)
        recs, f, _ = ctx.sweep_series(range(nt), scopes)
        L.ftkx_debug_small_stamps(buf, 0)
    v = [int(x) for x in buf]
    t0 = v[0]
    st = ctx.stats()
    print("u_rows", rows, "path", ctx.series_last_path(), "cells", st["cells_survived"], "tested", st["simplices_tested"],
          " ".join("%s %.1f" % (names[k] if k < len(names) else str(k), (v[k] - t0) / 100.0) for k in range(1, 10) if v[k]))
