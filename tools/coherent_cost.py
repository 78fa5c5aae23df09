#!/usr/bin/env python3
"""What the record buffer's memory type costs (FTKX_SERIES_OUT_COHERENT=1: fine-grained host memory, the HIP memory model's guarantee for records
read behind a flag; 0: coarse-grained, gfx950 behaviour): the pass of the hit-dense configurations and the host's copy of the records.
    for m in 0 1; do FTKX_SERIES_OUT_COHERENT=$m python3 tools/coherent_cost.py c2 c5; done"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    names = sys.argv[1:] or ["c2", "c5"]
    import torch
    import bench
    import ftk_amd
    from ftk_amd import synthetic, tslab
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {}
    for n in names:
        r = bench.side_config(n, torch, dev, ftk_amd, synthetic, tslab, steps=48, warmup=6)
        # the host's side: one pass, then the records copied out of the library's buffer (what a tracker does with them)
        nd, nv, case, dims, nt = bench.CONFIGS[n]
        scalar = nv == 1
        ctx = ftk_amd.Context(nd)
        lo = 2 if scalar else 1
        dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
        ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
        ctx.set_options(jacobian_symmetric=scalar, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
        keep = []
        for t in range(nt):
            a = synthetic.generate(case, dims, t, nt, torch, dev); keep.append(a); torch.cuda.synchronize()
            (ctx.push_scalar_slice if scalar else ctx.push_slice)(t, a)
        scopes = [ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL for t in range(nt)]
        best = None
        for _ in range(5):
            recs, f, _r = ctx.sweep_series(range(nt), scopes, copy=False)
            t0 = time.perf_counter()
            mine = np.array(recs)                      # 72 bytes a record out of the pinned buffer
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        ctx.close(); del keep; torch.cuda.empty_cache()
        out[n] = {"ms_per_pass": round(r["ms_per_step"], 4), "records": int(len(mine)), "host_copy_ms": round(best * 1e3, 4), "host_copy_GB/s": round(mine.nbytes / best / 1e9, 2)}
    print(json.dumps({"FTKX_SERIES_OUT_COHERENT": os.environ.get("FTKX_SERIES_OUT_COHERENT", "(unset)"), "configs": out}))


if __name__ == "__main__":
    main()
