#!/usr/bin/env python3
"""Per-step host time of the streaming tracker (push_scalar_field_snapshot / advance_timestep), device-resident input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ftk_amd
from ftk_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dims, nt, case, nd = {"c3": ((256, 256, 256), 16, "moving_extremum_3d", 3), "c4": ((512, 512, 512), 32, "moving_extremum_3d", 3),
                      "c2": ((1024, 1024), 64, "woven", 2)}[cfg]
dev = torch.device("cuda", 0)
slices = [synthetic.generate(case, dims, t, nt, torch, dev) for t in range(nt)]
torch.cuda.synchronize()
for rep in range(3):
    T = ftk_amd.CriticalPointTracker3DRegular if nd == 3 else ftk_amd.CriticalPointTracker2DRegular
    tr = T()
    tr.set_scalar_field_source(ftk_amd.SOURCE_GIVEN); tr.set_vector_field_source(ftk_amd.SOURCE_DERIVED)
    tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(True)
    tr.set_domain([2] * nd, [d - 3 for d in dims]); tr.set_array_domain([0] * nd, list(dims))
    tr.set_tag_mode(ftk_amd.TAG_EXACT64)
    tr.initialize()
    tp = ta = 0.0
    for k in range(nt):
        t0 = time.perf_counter(); tr.push_scalar_field_snapshot(slices[k]); t1 = time.perf_counter()
        if k != 0: tr.advance_timestep()
        if k == nt - 1: tr.update_timestep()
        t2 = time.perf_counter()
        if k >= 2: tp += t1 - t0; ta += t2 - t1
    tr.close()
    print(cfg, "per step: push %.4f ms, advance (prepare + sweep + pop) %.4f ms" % (tp * 1e3 / (nt - 2), ta * 1e3 / (nt - 2)))
