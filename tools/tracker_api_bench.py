#!/usr/bin/env python3
"""Throughput of the drop-in path a reference user drives: push_scalar_field_snapshot / advance_timestep per timestep
(one sweep per call, hits downloaded every step), device-resident input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ftk_amd
from ftk_amd import synthetic, tslab
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dims, nt, case, nd = {"c3": ((256, 256, 256), 16, "moving_extremum_3d", 3), "c4": ((512, 512, 512), 32, "moving_extremum_3d", 3),
                      "c2": ((1024, 1024), 64, "woven", 2), "c1": ((128, 128), 10, "woven", 2)}[cfg]
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
    t0 = time.perf_counter()
    for k in range(nt):
        tr.push_scalar_field_snapshot(slices[k])
        if k != 0: tr.advance_timestep()
        if k == nt - 1: tr.update_timestep()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    recs, o, ts = tr.get_critical_points()
    t2 = time.perf_counter()
    tr.finalize()
    t3 = time.perf_counter()
    tr.post_process()
    t4 = time.perf_counter()
    tr.close()
    print(cfg, "tracker API: %.3f ms total, %.3f ms per timestep, %d records, %.3e simplices/s; get_critical_points %.3f ms, finalize %.3f ms, post_process %.3f ms"
          % (dt * 1e3, dt * 1e3 / nt, len(recs), tslab.count_simplices(nd, dims, nt) / dt, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
