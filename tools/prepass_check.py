import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np, ftk_amd
from ftk_amd import synthetic
for case, dims, nt in (("woven", (1024, 1024), 64), ("moving_extremum_3d", (256, 256, 256), 16)):
    nd = len(dims); dev = torch.device("cuda", 0)
    slices = [synthetic.generate(case, dims, t, nt, torch, dev) for t in range(nt)]
    torch.cuda.synchronize()
    for rep in range(3):
        for batched in (False, True):
            ctx = ftk_amd.Context(nd)
            dom = ([2] * nd, [d - 3 for d in dims])
            ctx.set_mesh(dom, dom, ([0] * nd, list(dims))); ctx.set_options(jacobian_symmetric=1, derive_jacobian=1)
            for t in range(nt): ctx.push_scalar_slice(t, slices[t])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = ctx.slices_resolution(range(nt)) if batched else {t: ctx.slice_resolution(t) for t in range(nt)}
            dt = time.perf_counter() - t0
            print(case, "batched" if batched else "per-slice", "%.3f ms" % (dt * 1e3))
            ctx.close()
