#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE ...`: streams a known number of bytes (far larger than the 256 MiB Infinity Cache)
with 16-byte-per-lane loads, so that FETCH_SIZE per byte can be calibrated for this access shape."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ftk_amd

GB = 8
a = torch.ones(GB * (1 << 30) // 8, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
ctx = ftk_amd.Context(3)
for _ in range(3):
    ftk_amd._lib.check(ctx._L.ftkx_debug_stream_read(ctx._h, C.c_void_p(a.data_ptr()), a.numel() * 8), ctx._h)
print("streamed bytes per launch:", a.numel() * 8)
ctx.close()
