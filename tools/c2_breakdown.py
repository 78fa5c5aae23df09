#!/usr/bin/env python3
"""Where does a hit-dense pass (woven 1024^2 x 64) spend its wall time?  C call vs the Python copy of the records."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ftk_amd
from ftk_amd import synthetic, tslab, _lib
dims, nt, case = (1024, 1024), 64, "woven"
dev = torch.device("cuda", 0)
ctx = ftk_amd.Context(2)
dom = ([2, 2], [d - 3 for d in dims])
ctx.set_mesh(dom, dom, ([0, 0], list(dims)))
ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
keep = []
for t in range(nt):
    a = synthetic.generate(case, dims, t, nt, torch, dev); torch.cuda.synchronize(); keep.append(a); ctx.push_scalar_slice(t, a)
factors = tslab.factors_from_resolutions([ctx.slice_resolution(t)[0] for t in range(nt)])
L = ctx._L
for rep in range(6):
    ctx.invalidate_masks()
    t0 = time.perf_counter()
    for t in range(nt):
        ctx.sweep_enqueue(t, ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL, factors[t])
    t1 = time.perf_counter()
    out, n = C.c_void_p(), C.c_size_t()
    rc = L.ftkx_sweep_collect(ctx._h, C.byref(out), C.byref(n))
    t2 = time.perf_counter()
    recs = _lib.records_from(out.value, n.value)
    t3 = time.perf_counter()
    print("enqueue %.3f ms  C collect %.3f ms  numpy copy %.3f ms  hits %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n.value))
