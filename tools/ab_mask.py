#!/usr/bin/env python3
"""A/B timing of mask-kernel variants inside ONE process (the library reads FTKX_MASK_PLAN at every launch): the variants
are interleaved round-robin so that box-to-box and minute-to-minute drift hits all of them alike.  Reports per variant the
mean / min of the fused mask kernel's device time (HIP events inside the library) over the rounds.
usage: python tools/ab_mask.py [config] [rounds] -- "TILE=0 YG=1" "TILE=1 YG=4" ...   (names of FTKX_MASK_PLAN; other variables by their full name)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ftk_amd  # noqa: E402
from ftk_amd import synthetic  # noqa: E402

args = sys.argv[1:]
sep = args.index("--") if "--" in args else len(args)
cfg = args[0] if sep > 0 else "c4"
rounds = int(args[1]) if sep > 1 else 6
variants = args[sep + 1:] or ["TILE=0"]
CONFIGS = {"c4": (3, "moving_extremum_3d", (512, 512, 512), 32), "c3": (3, "moving_extremum_3d", (256, 256, 256), 16),
           "c4s": (3, "moving_extremum_3d", (512, 512, 512), 1), "c4n4": (3, "moving_extremum_3d", (512, 512, 512), 4),
           "c2": (2, "woven", (1024, 1024), 64), "c2x4": (2, "woven", (2048, 2048), 64), "c2w": (2, "woven", (4096, 4096), 16), "c2l": (2, "woven", (1024, 1024), 256), "c5": (2, "double_gyre", (2048, 1024), 128)}
nd, case, dims, nt = CONFIGS[cfg]
vector = case == "double_gyre"
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = ftk_amd.Context(nd); ctx.set_stream(stream.cuda_stream)
dom = ([1] * nd, [d - 2 for d in dims]) if vector else ([2] * nd, [d - 3 for d in dims])
ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
ctx.set_options(jacobian_symmetric=0 if vector else 1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
keep = []
for t in range(nt):
    a = synthetic.generate(case, dims, t, nt, torch, dev); torch.cuda.synchronize(); keep.append(a)
    (ctx.push_slice if vector else ctx.push_scalar_slice)(t, a)
ctx.set_profiling(True)
times = {v: [] for v in variants}
for rnd in range(rounds + 1):
    for v in variants:
        old, plan = {}, []
        for kv in v.split():
            k, val = kv.split("=")
            if k.startswith("FTKX_"):                                   # (full names for knobs outside the mask-plan family, e.g. FTKX_U_ROWS)
                old[k] = os.environ.get(k)
                os.environ[k] = val
            else:
                plan.append("%s=%s" % (k.lower(), val))                 # FTKX_MASK_PLAN: "swizzle=..,yg=..,zchunk=..,lmin=..,lcap=..,order=..,lean=.."
        old["FTKX_MASK_PLAN"] = os.environ.get("FTKX_MASK_PLAN")
        os.environ["FTKX_MASK_PLAN"] = ",".join(plan)
        ctx.invalidate_masks()
        ctx.set_profiling(True)      # resets the accumulated kernel times
        ctx.slices_prepare(range(nt), 0)
        ms = ctx.kernel_times()["mask_kernel"][0]
        if rnd > 0:                  # round 0 = warm-up (code object load)
            times[v].append(ms)
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
nbytes = 8.0 * (nd if vector else 1) * float(__import__("numpy").prod(dims)) * nt
for v in variants:
    t = times[v]
    print(f"   -> {nbytes / (min(t) * 1e-3) / 1e12:.2f} TB/s at the minimum, {nbytes / (sum(t) / len(t) * 1e-3) / 1e12:.2f} at the mean")
    print(f"{cfg} {v:28s} mean {sum(t) / len(t):.3f} ms  min {min(t):.3f}  max {max(t):.3f}  ({len(t)} rounds)")
