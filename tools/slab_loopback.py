#!/usr/bin/env python3
"""One rank's slab pass at full size on ONE GPU, with itself as both neighbours (a loopback): the stages of ftkx_series_dist_* run on
resident slices t = 0 .. n-1 as rank 0 of 2, the masks / request / reply that would cross xGMI are handed back to the same context
(the halo slice t = n becomes a masks-only copy of slice 0's masks + the patches asked for).  The records are not those of a real series
(the halo is not the real slice n) -- this measures what every stage costs on the stream at the shape N = 8 gives a rank of C4:
512^3 x 4 slices + the halo.

    python3 tools/slab_loopback.py [--dims 512 512 512] [--slices 4] [--passes 8] [--all-gather]
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/slab_loopback.py ; python3 tools/pass_timeline.py DIR --first series_begin_kernel
--all-gather: a process group of one on RCCL, the contribution really gathered with all_gather_into_tensor on the stream."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[512, 512, 512])
    ap.add_argument("--slices", type=int, default=4)
    ap.add_argument("--passes", type=int, default=8)
    ap.add_argument("--case", default="moving_extremum_3d")
    ap.add_argument("--all-gather", action="store_true")
    ap.add_argument("--no-halo", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="the mask message handed over on the context's stream")
    a = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import ftk_amd
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dist = None
    if a.all_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    nd, n = len(a.dims), a.slices
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = ftk_amd.Context(nd)
    ctx.set_stream(stream.cuda_stream)
    dom = ([2] * nd, [d - 3 for d in a.dims])
    ctx.set_mesh(dom, dom, ([0] * nd, list(a.dims)))
    ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
    nt_total = 8 * n
    keep = []
    for t in range(n):
        s = synthetic.generate(a.case, a.dims, t, nt_total, torch, dev)
        torch.cuda.synchronize()
        keep.append(s)
        ctx.push_scalar_slice(t, s)
    halo = not a.no_halo
    side = None if a.no_side else torch.cuda.Stream(device=dev)
    ts = np.arange(n, dtype=np.int32)
    scopes = np.array([ftk_amd.SCOPE_BOTH] * (n - 1) + [ftk_amd.SCOPE_BOTH if halo else ftk_amd.SCOPE_ORDINAL], dtype=np.int32)
    nbytes, _ = ctx.packed_masks_bytes()
    cells, pd = ctx.series_dist_cells(), ctx.patch_doubles()
    f64, i64, u8 = torch.float64, torch.int64, torch.uint8
    sets = []
    for _ in range(2):
        sets.append(dict(contrib=torch.zeros(4, dtype=f64, device=dev), gathered=torch.zeros(8, dtype=f64, device=dev), one=torch.zeros(4, dtype=f64, device=dev),
                         masks=torch.zeros(nbytes, dtype=u8, device=dev), masks_rx=torch.zeros(nbytes, dtype=u8, device=dev), req=torch.zeros(1 + cells, dtype=i64, device=dev), reply=torch.zeros(cells * pd, dtype=f64, device=dev)))
    print(f"mask message {nbytes / 1e6:.2f} MB, request {8 * (1 + cells)} B ({cells} cells), reply {cells * pd * 8 / 1e6:.3f} MB", flush=True)

    def submit(k):
        b = sets[k & 1]
        ctx.invalidate_masks()
        ctx.series_dist_begin(ts, scopes, None, 0, 2, 1 if halo else None, b["contrib"], b["gathered"], b["masks"] if halo else None, side_stream=side.cuda_stream if (halo and side is not None) else None)
        if halo and side is not None:                                    # (where the masks would cross xGMI: a copy of the message on the side stream)
            with torch.cuda.stream(side):
                b["masks_rx"].copy_(b["masks"], non_blocking=True)
            stream.wait_stream(side)
        if dist is not None:
            dist.all_gather_into_tensor(b["one"], b["contrib"])         # (RCCL, world 1, queued on the stream)
            b["gathered"][:4].copy_(b["one"], non_blocking=True)
        else:
            b["gathered"][:4].copy_(b["contrib"], non_blocking=True)
        b["gathered"][4:].copy_(b["contrib"], non_blocking=True)        # (the "upper neighbour" is this rank again)
        ctx.series_dist_cull((b["masks_rx"] if side is not None else b["masks"]) if halo else None, b["req"] if halo else None)
        ctx.series_dist_serve(b["req"] if halo else None, b["reply"] if halo else None)
        ctx.series_dist_finish(b["reply"] if halo else None)

    def run(k, pipelined):
        t0 = time.perf_counter()
        if pipelined:
            submit(0)
            for i in range(1, k + 1):
                if i < k:
                    submit(i)
                recs, f, _ = ctx.sweep_series_complete(copy=False)
        else:
            for i in range(k):
                submit(i)
                recs, f, _ = ctx.sweep_series_complete(copy=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3, len(recs), ctx.series_last_path(), ctx.series_dist_status(2)[:2]
    run(3, True)
    for pipelined in (True, False):
        ms, nrec, path, st = run(a.passes, pipelined)
        print(f"{'two in flight' if pipelined else 'one at a time'}: {ms:.4f} ms per pass, {nrec} records, path {path}, (asked, served) {st}", flush=True)
    ctx.set_profiling(2)
    run(4, False)
    print("mask kernel (events):", ctx.kernel_times()["mask_kernel"], flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
