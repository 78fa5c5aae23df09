#!/usr/bin/env python3
"""Pass 2 (ftkx_trace_curves) on the hit set of a hit-dense sweep (woven 1024^2 x 64, 62 181 records): time per phase
(FTKX_TRACE_PROF) over host thread counts.  Runs on the GPU box (the records come from the sweep itself)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ftk_amd  # noqa: E402
from ftk_amd import synthetic, tslab  # noqa: E402

dims, nt, case = (1024, 1024), 64, "woven"
dev = torch.device("cuda", 0)
ctx = ftk_amd.Context(2)
dom = ([2, 2], [d - 3 for d in dims])
ctx.set_mesh(dom, dom, ([0, 0], list(dims)))
ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
keep = []
for t in range(nt):
    a = synthetic.generate(case, dims, t, nt, torch, dev); torch.cuda.synchronize(); keep.append(a); ctx.push_scalar_slice(t, a)
rm = ctx.slices_prepare(range(nt), 0)
factors = tslab.factors_from_resolutions([rm[t][0] for t in range(nt)])
for t in range(nt):
    ctx.sweep_enqueue(t, ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL, factors[t])
recs = np.array(ctx.sweep_collect())
print("records", len(recs), "host threads", os.cpu_count())
os.environ["FTKX_TRACE_PROF"] = "1"
for threads in (8, 16, 24, 32, 48, 64):
    os.environ["FTKX_TRACE_THREADS"] = str(threads)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        curves, loop, nsp = ftk_amd.trace_curves(2, dom, recs)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print("threads %2d: trace_curves best of 5 %.3f ms, %d curves" % (threads, best, len(curves)), flush=True)
