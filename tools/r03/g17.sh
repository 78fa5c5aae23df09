#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
FTKX_TRACE_PROF=1 python3 - <<'P' 2>&1 | tail -30
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ftk_amd
from ftk_amd import synthetic
dims, nt = (1024, 1024), 64
dev = torch.device("cuda", 0)
ctx = ftk_amd.Context(2); dom = ([2, 2], [d - 3 for d in dims]); ctx.set_mesh(dom, dom, ([0, 0], list(dims))); ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
keep = []
for t in range(nt):
    a = synthetic.generate("woven", dims, t, nt, torch, dev); torch.cuda.synchronize(); keep.append(a); ctx.push_scalar_slice(t, a)
recs, f, _ = ctx.sweep_series(range(nt), [3] * (nt - 1) + [1])
recs = np.array(recs)
for use in (None, ctx, None, ctx):
    best = (1e9, 1e9)
    for rep in range(6):
        r = ftk_amd.pass2(2, dom, recs, use)
        best = min(best, (r[4], r[5]), key=lambda v: v[0] + v[1])
    print("ctx" if use is not None else "host", "trace %.3f post %.3f ms" % best, flush=True)
P
