#!/usr/bin/env python3
"""Hit-dense passes with their tail on a stream of its own (FTKX_SERIES_HOOKS split=3) -- in a fresh process and behind other contexts:
    python3 tools/dense_split.py c2 c5            -> ms per pass of each configuration, run one after the other in THIS process
    FTKX_SERIES_HOOKS=split=3 FTKX_STREAM_POOL=0 python3 tools/dense_split.py c1 c2 c3 c5
(bench.side_config: the driver's own side-configuration loop)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    names = sys.argv[1:] or ["c5"]
    import torch
    import bench
    import ftk_amd
    from ftk_amd import synthetic, tslab
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {}
    for n in names:
        r = bench.side_config(n, torch, dev, ftk_amd, synthetic, tslab, steps=48, warmup=6)
        out[n] = {"ms_per_step": round(r["ms_per_step"], 4), "median_ms": round(r.get("median_ms", 0), 4), "paths": r["series_paths"], "ok": r["check"].get("ok")}
    print(json.dumps({"hooks": os.environ.get("FTKX_SERIES_HOOKS", ""), "pool": os.environ.get("FTKX_STREAM_POOL", "1"), "configs": out}))


if __name__ == "__main__":
    main()
