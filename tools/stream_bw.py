#!/usr/bin/env python3
"""Achievable HBM read bandwidth on this GPU with the mask kernel's load shape (16 B per lane), for reference next to bench.py."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ftk_amd
a = torch.ones(8 * (1 << 30) // 8, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
ctx = ftk_amd.Context(3)
f = lambda: ftk_amd._lib.check(ctx._L.ftkx_debug_stream_read(ctx._h, C.c_void_p(a.data_ptr()), a.numel() * 8), ctx._h)
f(); f()
t0 = time.perf_counter()
for _ in range(10): f()
dt = (time.perf_counter() - t0) / 10
print("stream read: %.3f ms for %.2f GB -> %.2f TB/s" % (dt * 1e3, a.numel() * 8 / 1e9, a.numel() * 8 / dt / 1e12))
b = torch.empty_like(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): b.copy_(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("torch copy : %.3f ms -> %.2f TB/s (read+write)" % (dt * 1e3, 2 * a.numel() * 8 / dt / 1e12))
