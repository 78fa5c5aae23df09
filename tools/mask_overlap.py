#!/usr/bin/env python3
"""PROBE: can the ramp of a one-slice mask launch (a streaming tracker's step at 512^3: 0.193 ms against 0.172 per slice inside a 32-slice
launch) be hidden by launching consecutive steps' mask kernels on two streams?  ftkx_debug_mask_relaunch: the same job `reps` times on one
stream / alternately on two, with and without a small dependent kernel in front (the pass's begin kernel)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ftk_amd  # noqa: E402
from ftk_amd import synthetic, _lib  # noqa: E402

dims = (512, 512, 512)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = ftk_amd.Context(3); ctx.set_stream(stream.cuda_stream)
dom = ([2] * 3, [d - 3 for d in dims])
ctx.set_mesh(dom, dom, ([0] * 3, list(dims)))
ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
a = synthetic.generate("moving_extremum_3d", dims, 0, 4, torch, dev); torch.cuda.synchronize()
ctx.push_scalar_slice(0, a)
ctx.slices_prepare([0], 0)
L = _lib.load()
f = L.ftkx_debug_mask_relaunch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for rnd in range(4):
    row = []
    for ns, wb in ((1, 0), (2, 0), (1, 1), (2, 1)):
        ms = C.c_double(0)
        rc = f(ctx._h, 0, reps, ns, wb, C.byref(ms))
        assert rc == 0, rc
        row.append("streams %d begin %d: %.4f ms" % (ns, wb, ms.value))
    print("round %d  " % rnd + "   ".join(row), flush=True)
