#!/usr/bin/env python3
"""Does the placement of the 32 slices in memory matter to the mask kernel?  Separate allocations (what bench.py does) vs one
contiguous 32 GiB tensor vs slices padded apart."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ftk_amd
from ftk_amd import synthetic, tslab
dims, nt, case = (512, 512, 512), 32, "moving_extremum_3d"
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "separate"
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = ftk_amd.Context(3); ctx.set_stream(stream.cuda_stream)
dom = ([2] * 3, [d - 3 for d in dims])
ctx.set_mesh(dom, dom, ([0] * 3, list(dims)))
ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
n = dims[0] * dims[1] * dims[2]
if mode == "contiguous":
    big = torch.empty(nt * n, dtype=torch.float64, device=dev)
    views = [big[t * n:(t + 1) * n] for t in range(nt)]
elif mode == "padded":      # slices 1 GiB + 2 MiB + 4 KiB apart: different low address bits per slice
    pad = (2 * 1024 * 1024 + 4096) // 8
    big = torch.empty(nt * (n + pad), dtype=torch.float64, device=dev)
    views = [big[t * (n + pad):t * (n + pad) + n] for t in range(nt)]
else:
    views = [None] * nt
keep = []
for t in range(nt):
    a = synthetic.generate(case, dims, t, nt, torch, dev)
    if views[t] is not None:
        views[t].copy_(a.reshape(-1)); a = views[t]
    torch.cuda.synchronize(); keep.append(a); ctx.push_scalar_slice(t, a)
factors = tslab.factors_from_resolutions([ctx.slice_resolution(t)[0] for t in range(nt)])
ctx.set_profiling(True)
for rep in range(4):
    ctx.invalidate_masks()
    for t in range(nt):
        ctx.sweep_enqueue(t, ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL, factors[t])
    recs = ctx.sweep_collect(copy=False)
kt = ctx.kernel_times()
print(mode, "mask ms/launch %.3f" % (kt["mask_kernel"][0] / kt["mask_kernel"][1]), "hits", len(recs), "ptr0 %x" % keep[0].data_ptr(), "ptr1 %x" % keep[1].data_ptr())
