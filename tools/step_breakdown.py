import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np, ftk_amd
from ftk_amd import synthetic
dims, nt, case, nd = (256, 256, 256), 16, "moving_extremum_3d", 3
dev = torch.device("cuda", 0)
slices = [synthetic.generate(case, dims, t, nt, torch, dev) for t in range(nt)]
torch.cuda.synchronize()
for rep in range(2):
    tr = ftk_amd.CriticalPointTracker3DRegular()
    tr.set_scalar_field_source(ftk_amd.SOURCE_GIVEN); tr.set_vector_field_source(ftk_amd.SOURCE_DERIVED)
    tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(True)
    tr.set_domain([2] * nd, [d - 3 for d in dims]); tr.set_array_domain([0] * nd, list(dims))
    tr.set_tag_mode(ftk_amd.TAG_EXACT64); tr.initialize()
    tp = ta = 0.0
    for k in range(nt):
        t0 = time.perf_counter(); tr.push_scalar_field_snapshot(slices[k]); t1 = time.perf_counter()
        if k != 0: tr.advance_timestep()
        t2 = time.perf_counter()
        tp += t1 - t0; ta += t2 - t1
    print("push %.3f ms/step  advance %.3f ms/step" % (tp / nt * 1e3, ta / (nt - 1) * 1e3))
    tr.close()
