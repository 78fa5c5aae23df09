#!/usr/bin/env python3
"""The integer regime alone (bench.py's `configs.c3_exact_only` / `configs.c3o` entries) -- for profiling:
    python3 tools/int_regime.py c3_exact_only|c3o          -> one JSON line (bench.integer_config)
    bash tools/pmc_int.sh c3_exact_only c3x                -> kernel trace + VALU counters of the same command into gpurun_out/r06_c3x_*"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3_exact_only"
    import torch
    import bench
    import ftk_amd
    from ftk_amd import synthetic, tslab
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(bench.integer_config(name, torch, dev, ftk_amd, synthetic, tslab)))


if __name__ == "__main__":
    main()
