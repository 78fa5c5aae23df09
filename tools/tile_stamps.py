#!/usr/bin/env python3
"""Phase breakdown of tile_kernel by in-kernel cycle stamps (a diagnostic build: -DFTKX_TILE_STAMPS).
    rm -f ftk_amd/csrc/build/tile_kernels.hip.o && FTKX_EXTRA_CFLAGS=-DFTKX_TILE_STAMPS python3 -m ftk_amd.build && python3 tools/tile_stamps.py [--config c3] [--nt 4]
prints cycles per wavefront by phase (S block | vertices | flags | fan | unsure + hits + pairs | statistics) and the pass time.  Rebuild without
the flag afterwards (the stamps cost registers)."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="moving_extremum_3d")
    ap.add_argument("--dims", type=int, nargs="+", default=[256, 256, 256])
    ap.add_argument("--nt", type=int, default=4)
    ap.add_argument("--passes", type=int, default=3)
    a = ap.parse_args()
    import torch
    import ftk_amd
    from ftk_amd import synthetic, _lib
    dev = torch.device("cuda", 0)
    nd = len(a.dims)
    ctx = ftk_amd.Context(nd)
    dom = ([2] * nd, [d - 3 for d in a.dims])
    ctx.set_mesh(dom, dom, ([0] * nd, list(a.dims)))
    ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64, exact_only=1)
    keep = []
    for t in range(a.nt):
        s = synthetic.generate(a.case, a.dims, t, 16, torch, dev)
        torch.cuda.synchronize(); keep.append(s)
        ctx.push_scalar_slice(t, s)
    scopes = [ftk_amd.SCOPE_BOTH if t + 1 < a.nt else ftk_amd.SCOPE_ORDINAL for t in range(a.nt)]
    L = _lib.load()
    out = (ctypes.c_ulonglong * 8)()
    have = hasattr(L, "ftkx_debug_tile_stamps")
    ctx.sweep_series(range(a.nt), scopes)
    if have:
        L.ftkx_debug_tile_stamps(out, 1)
    t0 = time.perf_counter()
    for _ in range(a.passes):
        recs, f, _r = ctx.sweep_series(range(a.nt), scopes)
    dt = (time.perf_counter() - t0) / a.passes
    nsimp = ctx.stats()["simplices_tested"]
    print("pass %.3f ms, %d records, %.3e simplices tested -> %.3e simplices/s" % (dt * 1e3, len(recs), nsimp, nsimp / dt))
    if have:
        L.ftkx_debug_tile_stamps(out, 1)
        v = list(out); n = max(v[7], 1)
        names = ["S block", "vertices", "flags", "fan", "unsure+hits+pairs", "statistics"]
        print("wavefronts", v[7], "cycles per wavefront:", {k: round(x / n, 1) for k, x in zip(names, v[:6])}, "sum", round(sum(v[:6]) / n, 1))
    else:
        print("(no stamps in this build)")
    ctx.close()


if __name__ == "__main__":
    main()
