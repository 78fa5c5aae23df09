#!/usr/bin/env python3
"""Where does the host time of bench.py's prepare_and_factors() go?  announce / the C call / the Python around it, per config."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ftk_amd
from ftk_amd import synthetic, tslab
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
nd, nv, case, dims, nt = {"c4": (3, 1, "moving_extremum_3d", (512, 512, 512), 32), "c3": (3, 1, "moving_extremum_3d", (256, 256, 256), 16),
                          "c2": (2, 1, "woven", (1024, 1024), 64), "c5": (2, 2, "double_gyre", (2048, 1024), 128)}[cfg]
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = ftk_amd.Context(nd); ctx.set_stream(stream.cuda_stream)
scalar = nv == 1
lo = 2 if scalar else 1
dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
ctx.set_options(jacobian_symmetric=scalar, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
keep = []
for t in range(nt):
    a = synthetic.generate(case, dims, t, nt, torch, dev); torch.cuda.synchronize(); keep.append(a)
    (ctx.push_scalar_slice if scalar else ctx.push_slice)(t, a)
own = list(range(nt))
ann_ts = np.array(own, dtype=np.int32); ann_sc = np.array([ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL for t in own], dtype=np.int32)
ctx.set_profiling(True)
acc = np.zeros(6)
for rep in range(12):
    ctx.invalidate_masks()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.sweep_announce(ann_ts, ann_sc)
    t1 = time.perf_counter(); rm = ctx.slices_prepare(own, 0)
    t2 = time.perf_counter(); res = [rm[t][0] for t in range(nt)]
    t3 = time.perf_counter(); f = tslab.factors_from_resolutions(res)
    t4 = time.perf_counter(); ctx.sweep_enqueue_many(ann_ts, ann_sc, f)
    t5 = time.perf_counter(); recs = ctx.sweep_collect(copy=False)
    t6 = time.perf_counter()
    if rep >= 2:
        acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5]) * 1e3
acc /= 10
kt = ctx.kernel_times()
print(cfg, "announce %.4f  slices_prepare %.4f (mask kernel %.4f)  dict %.4f  factors %.4f  enqueue %.4f  collect %.4f ms" % (*acc[:2], kt["mask_kernel"][0] / max(1, kt["mask_kernel"][1]), *acc[2:]))
