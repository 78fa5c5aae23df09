#!/usr/bin/env python3
"""Benchmark of the hot path: the critical-point space-time simplex sweep (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the sweep over the whole synthetic time series: every ordinal sweep and every interval sweep of the
configured lattice, hit records downloaded to the host.  Inputs (scalar slices and the gradient field the tracker API derives
from them at push time) are resident in HBM when the timed region starts; the quantisation-factor pre-pass (a device
reduction per slice + one all_gather) is FUSED into the pass that builds the sign masks and lies inside the timed region.
With N > 1 the lattice is cut into timestep slabs (ftk_amd/tslab.py).  A rank's INPUT is its slab plus the first slice of the next slab (the interval sweep across
the slab boundary reads both); that boundary slice reaches it by one RCCL send/recv over xGMI, which -- like every other step
that makes the inputs resident -- happens before the timed region and is reported on its own (`halo_exchange`: ms, bytes,
GB/s).  The sweeps themselves then shard with no data-path collective.  `--halo-in-loop` re-sends the boundary slice in every
timed pass instead (1 GiB per pass for 512^3: the pass is then bound by one xGMI link, not by the sweep); whichever convention
is timed as `value`, a few passes of the other one are timed as well and reported as `other_halo_convention`.
Scaling: the path shards by timestep slabs with no data-path collective, so the default is WEAK scaling -- every rank owns one
slab of the configuration's length (N = 8 on c4: a 512^3 x 256 series in eight slabs of 32, the single-GPU workload per GPU) and
`value` is the series' simplices over the slowest rank's time.  `--scaling strong` cuts the configuration's own series instead
(N = 8 on c4 is BASELINE.json's literal `moving_extremum_3d 512^3 x 32, 8xMI355X` case: 4 timesteps + 1 halo slice per rank).
`--timesteps T` overrides the series length.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (nd, nv, case, dims, nt)
    "c4": (3, 1, "moving_extremum_3d", (512, 512, 512), 32),   # BASELINE configs[3]: the configuration the metric is quoted on
    "c3": (3, 1, "moving_extremum_3d", (256, 256, 256), 16),   # configs[2]
    "c2": (2, 1, "woven", (1024, 1024), 64),                   # configs[1]
    "c1": (2, 1, "woven", (128, 128), 10),                     # configs[0]
    "c5": (2, 2, "double_gyre", (2048, 1024), 128),            # configs[4]
    "c3o": (3, 1, "moving_extremum_3d_overflow", (256, 256, 256), 4),   # the int64-overflow regime (nbits 21): every cell takes the integer test
    "small3": (3, 1, "moving_extremum_3d", (96, 96, 96), 8),
    "small2": (2, 1, "woven", (256, 256), 8),
    "mid2": (2, 1, "woven", (512, 512), 12),                  # enough records per rank for the copy kernel of a pipelined pass (tests)
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
STREAM_CEILING_GBS = 6290.0   # what a bare streaming read reaches on this part (same guide: achievable HBM read bandwidth).  A yardstick, not a bound:
                              # tools/probe/bw_probe.hip reads 4.3 GB at 6.4-6.6 TB/s on the faster boxes of the pool, where the mask kernel reaches 6.4 as well

# what a pass of each configuration must return (checked in the JSON line, `configs.*.check`):
#   c1: the reference's own count and type histogram (woven 128 x 128 x 10)
#   c2 / c5: the count the oracle gives on the very arrays the GPU sweeps (tests/test_gpu_fullsize_series.py holds record-for-record equality)
#   c3 / c4: the analytic trajectory of the single extremum
EXPECT = {
    "c1": {"hits": 7357, "types": {2: 1861, 4: 3657, 8: 1839}, "source": "reference (BASELINE.md section 3)"},
    "c2": {"hits": 62181, "source": "oracle on the swept arrays (tests/test_gpu_fullsize_series.py::test_c2_series_*)"},
    "c5": {"hits": 56766, "types_only": [4], "source": "oracle on the swept arrays (tests/test_gpu_fullsize_series.py::test_c5_series_*)"},
}


def spread(samples_ms):
    """per-pass wall times (the time between consecutive completions) -> min / median / max.  With two passes in flight the first sample
    holds the filling of the pipeline (two submits before the first completion) and the last one a pass with nothing queued behind it:
    the three figures are taken over the passes in between (all samples are listed)"""
    inner = samples_ms[1:-1] if len(samples_ms) >= 4 else samples_ms
    a = np.sort(np.asarray(inner, dtype=np.float64))
    if not len(a):
        return {}
    return {"ms_per_step_min": float(a[0]), "ms_per_step_median": float(np.median(a)), "ms_per_step_max": float(a[-1]),
            "ms_per_step_samples": [round(float(v), 4) for v in np.asarray(samples_ms, dtype=np.float64)[:64]]}     # (in the order the passes completed)


# passes in flight in the timed loops: two keep the stream fed; the third lets the host run one pass ahead of a split pass's tail, which ends
# behind the mask kernel of the pass after it
IN_FLIGHT = int(os.environ.get("FTKX_BENCH_IN_FLIGHT", "3"))


def check_records(name, case, dims, nt, recs, paths, want_paths):
    """the result of the timed passes against what the configuration must give; -> the `check` object (ok: everything held)"""
    from ftk_amd import synthetic
    asc = bool(np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)) if len(recs) > 1 else True
    out = {"hits": int(len(recs)), "device_driven": all(p in want_paths for p in paths), "paths": sorted(set(str(p) for p in paths)),
           "tags_ascending_unique": asc}
    ok = out["device_driven"] and asc
    exp = EXPECT.get(name)
    if exp:
        out["expected_hits"], out["expected_from"] = exp["hits"], exp["source"]
        ok = ok and len(recs) == exp["hits"]
        if "types" in exp:
            t, c = np.unique(recs["type"], return_counts=True)
            out["types"] = {int(a): int(b) for a, b in zip(t, c)}
            ok = ok and out["types"] == exp["types"]
        if "types_only" in exp:
            out["types"] = sorted(set(int(v) for v in recs["type"]))
            ok = ok and out["types"] == exp["types_only"]
    if case == "moving_extremum_3d" and len(recs):
        x0, dv = synthetic.moving_extremum_params(dims)
        err = max(float(np.abs(recs["x"][:, a] - (x0[a] + dv[a] * recs["t"])).max()) for a in range(3))
        out["max_abs_position_error_vs_analytic"] = err
        out["types"] = sorted(set(int(v) for v in recs["type"]))
        crossed = len(np.unique(recs["aux"] >> 1)) == nt
        out["every_timestep_crossed"] = bool(crossed)
        ok = ok and err < 1e-6 and out["types"] == [2] and crossed and len(recs) >= 2 * nt - 1
    out["ok"] = bool(ok)
    return out


def glibc_version():
    try:
        import ctypes
        f = ctypes.CDLL(None).gnu_get_libc_version
        f.restype = ctypes.c_char_p
        return "glibc " + f().decode()
    except Exception:   # noqa: BLE001
        return "unknown"


def pci_bus_id(torch, dev):
    """the device's PCI address as the HIP runtime names it (hipDeviceGetPCIBusId), e.g. 0000:05:00.0"""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev.index or 0)) == 0:
            return buf.value.decode()
    except Exception:   # noqa: BLE001
        pass
    pr = torch.cuda.get_device_properties(dev)
    return "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))


def me3d_params(dims):
    from ftk_amd import synthetic
    x0, dv = synthetic.moving_extremum_params(dims)
    return list(x0), list(dv)


def cpu_baseline(nd, case, want_seconds=20.0):
    """The reference CPU sweep (oracle/_ref, the real hguo/ftk code) -- or, if that binary did not travel, the oracle port --
    timed on this box's host cores on a bounded sub-volume of the same workload.  The reference starts hardware_concurrency() threads
    (include/ftk/filters/filter.hh:36-39) -- 256 on a GPU box whose cgroup gives this process 16 CPUs -- so it is run TWICE: with its own
    default and with nthreads = the CPUs this process may use; `value` is the faster of the two, both are in `runs`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from ftk_amd import tslab
    ncores = os.cpu_count() or 1
    share = host_cpu_share()
    drv = os.path.join(ROOT, "oracle", "_ref", "ftk_ref_driver")
    if nd == 3:
        dims, nt = (96, 96, 96), 6            # 2.5e8 simplices: 15-20 s of the reference's sweep on this class of host
        x0, dv = me3d_params(dims)
        sample = f"moving_extremum_3d {dims[0]}x{dims[1]}x{dims[2]}x{nt} sub-volume, same dyadic x0 offset / dir"
        extra = [repr(v) for v in x0 + dv]
    else:
        dims, nt = (768, 768), 16             # 1.1e8 simplices
        sample = f"{case} {dims[0]}x{dims[1]}x{nt} sub-volume"
        extra = []
    nsimp = tslab.count_simplices(nd, dims, nt, scalar_input=(case != "double_gyre"))
    out = {}
    if os.path.exists(drv):
        runs = []
        for label, nthreads in (("reference default (hardware_concurrency)", 0), ("nthreads = CPUs this process may use", max(1, int(round(share))))):
            with tempfile.TemporaryDirectory() as tmp:
                cmd = [drv, "synthetic", case, str(dims[0]), str(dims[1]), str(dims[2] if nd == 3 else 1), str(nt), os.path.join(tmp, "o.bin")] + extra
                if nthreads:
                    cmd += [str(nthreads)]        # (the driver reads nthreads behind the output path, or behind the six x0 / dir values)
                try:
                    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
                    j = json.loads(r.stdout.decode().strip().splitlines()[-1])
                    runs.append({"setting": label, "cores": int(j["nthreads"]), "value": nsimp / j["sweep_seconds"], "sweep_seconds": j["sweep_seconds"], "records": j["records"]})
                except Exception as e:   # noqa: BLE001
                    runs.append({"setting": label, "error": repr(e)})
        good = [r for r in runs if "value" in r]
        if good:
            best = max(good, key=lambda r: r["value"])
            out = {"value": best["value"], "unit": "simplices/s", "cores": best["cores"], "kind": "reference",
                   "sample": sample + f" ({nsimp} simplices in {best['sweep_seconds']:.2f} s; {best['setting']}: the faster of the two runs)", "runs": runs}
    # the oracle port (flat arrays, pthreads) on all cores, for orientation
    try:
        import pyoracle
        steps = [pyoracle.synthetic(case, list(dims), k, nt, *(me3d_params(dims) if nd == 3 else (None, None))) for k in range(nt)]
        _, _, secs = pyoracle.track(steps, nd, 2 if case == "double_gyre" else 1, nthreads=ncores)
        port = {"value": nsimp / secs, "unit": "simplices/s", "cores": ncores, "kind": "port", "sample": sample}
    except Exception as e:   # noqa: BLE001
        port = None
    if not out and port:
        out, port = port, None
    # (a container's CPU share can be far below the core count `cores` threads were started on: said beside it)
    for o in (out, port):
        if o:
            o["host_cpu_share"] = share
    return out, port


def host_cpu_share():
    """CPUs this process may use: the cgroup's quota (cpu.max) if there is one, else its affinity mask"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return round(float(q) / float(per), 2)
    except Exception:   # noqa: BLE001
        pass
    try:
        return float(len(os.sched_getaffinity(0)))
    except Exception:   # noqa: BLE001
        return float(os.cpu_count() or 1)


def side_config(name, torch, dev, ftk_amd, synthetic, tslab, steps=5, warmup=2):
    """A few passes of another BASELINE configuration on this GPU (after the timed region of the headline one): the same pass -- resident
    input to records on the host, ftkx_sweep_series -- measured the same way.  -> the entry of `configs` in the JSON line."""
    nd, nv, case, dims, nt = CONFIGS[name]
    scalar_input = nv == 1
    stream = torch.cuda.current_stream()
    ctx = ftk_amd.Context(nd, dev.index or 0)
    ctx.set_stream(stream.cuda_stream)
    lo = 2 if scalar_input else 1
    dom = ([lo] * nd, [d - (3 if scalar_input else 2) for d in dims])
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=scalar_input, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
    keep = []
    for t in range(nt):
        a = synthetic.generate(case, dims, t, nt, torch, dev)
        keep.append(a)
        torch.cuda.synchronize()
        (ctx.push_scalar_slice if scalar_input else ctx.push_slice)(t, a)
    ts = np.arange(nt, dtype=np.int32)
    scopes = np.array([ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL for t in range(nt)], dtype=np.int32)
    paths = {}
    path_list = []
    stamps = []

    def passes(k, count):
        # up to IN_FLIGHT passes in flight, like the headline run (main(): passes)
        submitted = done = 0
        while done < k:
            while submitted < k and submitted - done < IN_FLIGHT:
                ctx.invalidate_masks()
                ctx.sweep_series_submit(ts, scopes)
                submitted += 1
            recs, f, _r = ctx.sweep_series_complete(copy=False)
            done += 1
            if count:
                stamps.append(time.perf_counter())
                p = ctx.series_last_path()
                path_list.append(p)
                paths[str(p)] = paths.get(str(p), 0) + 1
        return recs, f
    recs, f = passes(max(warmup, 26), False)      # (see main(): everything a pass allocates exists before the clock starts)
    # no HIP events inside the timed region of a side configuration: a pair costs the stream ~10 us, which a 0.2 ms pass notices; the
    # dominant kernel is timed in extra passes behind it
    ctx.set_profiling(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs, f = passes(steps, True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    recs = np.array(recs)
    per_pass = np.diff(np.array([t0] + stamps)) * 1e3
    st = ctx.stats()
    ctx.set_profiling(2)            # events around the dominant (mask) kernel only
    passes(4, False)
    torch.cuda.synchronize()
    kt = ctx.kernel_times()
    ctx.set_profiling(0)
    domk = max(kt, key=lambda k: kt[k][0])
    dom_ms, dom_n = kt[domk]
    c = 1 if scalar_input else nd
    alg = 8.0 * c * float(np.prod(dims)) * nt + 72.0 * len(recs)
    nsimp = tslab.count_simplices(nd, dims, nt, scalar_input)
    symbol = "ftkx::%s<%d>" % (domk, nd)
    if domk == "mask_kernel":
        symbol = (ctx._L.ftkx_last_mask_kernel() or b"").decode() or symbol
    want_paths = [(2, 32), (5, 0)] if case == "moving_extremum_3d" else ([(1, 0), (2, 32), (4, 544)] if name == "c1" else [(1, 0), (5, 0)])      # (c1: the one-launch pass for small series; small hit-dense series may also take (1, 64): see job())
    out = {"workload": f"{case} {'x'.join(str(d) for d in dims)}x{nt}", "steps": steps, "ms_per_step": elapsed / steps * 1e3, "value": nsimp * steps / elapsed,
           "simplices_per_step": nsimp, "kernel": symbol, "kernel_avg_launch_ms": dom_ms / max(1, dom_n),
           "frac": alg / (dom_ms / max(1, dom_n) * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_n else None,
           "frac_of_streaming_ceiling": alg / (dom_ms / max(1, dom_n) * 1e-3) / 1e9 / STREAM_CEILING_GBS if dom_n else None,
           "end_to_end_frac": alg / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg,
           "hits": int(len(recs)), "simplices_tested_exactly": int(st["simplices_tested"]), "nbits": int(np.log2(max(int(v) for v in f))),
           "series_paths": paths, "split_pass": ctx.series_split_decision(), "warmup_effective": max(warmup, 26),
           "check": check_records(name, case, dims, nt, recs, path_list, want_paths),
           "kernel_timing": "HIP events around the mask kernel in 4 passes BEHIND the timed region (none inside it)"}
    out.update(spread(per_pass))
    ctx.close()
    del keep
    torch.cuda.empty_cache()
    return out


INT_PROFILES = {"c3_exact_only": "profiles/r06_c3x_valu_summary.json", "c3o": "profiles/r06_c3o_valu_summary.json"}


def integer_config(name, torch, dev, ftk_amd, synthetic, tslab):
    """The INTEGER regime: where the sign cull is illegal (`exact_only`: c3_exact_only = C3 with every simplex through the
    integer test) or useless (c3o: nbits 21 with |V| large enough that determinants may wrap -- most cells survive, the library replays the
    batch through the tile kernel with its in-tile cull), every simplex takes tile_kernel: 64-bit integer / FP64 VALU work on vertices
    staged in LDS, no HBM bound to speak of.  Reference arithmetic: critical_point_tracker_3d_regular.hh:453-464, numeric/sign_det.hh:92-200,
    360-414, numeric/det.hh:16-55.  Roofline of that kernel, `bound: int_valu`, in simplices/s:
      achieved = simplices the kernel tested / its device time (HIP events around tile_kernel alone, live);
      peak     = the same simplices / the time of the predicate arithmetic ALONE -- the fan phase on tiles already staged in LDS, without
                 staging, lists and records: the kernel is run with its fan phase 6 and 2 times per tile (ftkx_debug_tile_repeat), a
                 quarter of the difference is one repetition;
      counters: VALU busy and VALU instructions per wavefront as collected by tools/pmc_int.sh into profiles/ (committed figures, stamped)."""
    base = "c3" if name == "c3_exact_only" else name
    nd, nv, case, dims, nt = CONFIGS[base]
    exact_only = name == "c3_exact_only"
    stream = torch.cuda.current_stream()
    ctx = ftk_amd.Context(nd, dev.index or 0)
    ctx.set_stream(stream.cuda_stream)
    dom = ([2] * nd, [d - 3 for d in dims])
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    opts = dict(jacobian_symmetric=1, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64)
    ctx.set_options(exact_only=1 if exact_only else 0, **opts)
    keep = []
    for t in range(nt):
        a = synthetic.generate(case, dims, t, nt, torch, dev)
        keep.append(a)
        torch.cuda.synchronize()
        ctx.push_scalar_slice(t, a)
    ts = np.arange(nt, dtype=np.int32)
    scopes = np.array([ftk_amd.SCOPE_BOTH if t + 1 < nt else ftk_amd.SCOPE_ORDINAL for t in range(nt)], dtype=np.int32)

    def one_pass(profile):
        ctx.invalidate_masks()
        ctx.set_profiling(1 if profile else 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs, f, _r = ctx.sweep_series(ts, scopes, copy=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kt = ctx.kernel_times() if profile else None
        ctx.set_profiling(0)
        return recs, f, dt, kt
    steps = 3 if exact_only else 2
    one_pass(False); one_pass(False)                     # (the first passes allocate; c3o: the first one finds the cull useless and the rest go to the tile kernel up front)
    wall, tile_ms, launches = [], [], 0
    for _ in range(steps):
        recs, f, dt, kt = one_pass(True)
        wall.append(dt); tile_ms.append(kt["tile_kernel"][0]); launches = kt["tile_kernel"][1]
    st = ctx.stats()
    nrec = int(len(recs))
    tags = np.array(recs["tag"]) if nrec else np.zeros(0, dtype=np.uint64)
    # the predicate arithmetic alone
    rep_ms = {}
    for r in (6, 2):
        ctx.debug_tile_repeat(r)
        _recs, _f, _dt, kt = one_pass(True)
        rep_ms[r] = kt["tile_kernel"][0]
    ctx.debug_tile_repeat(1)
    fan_ms = (rep_ms[6] - rep_ms[2]) / 4.0
    # the check: the other way through the library gives the same records (c3_exact_only: the culled, device-driven pass; c3o: exact_only)
    ctx.set_options(exact_only=0 if exact_only else 1, **opts)
    recs2, f2, _dt, _kt = one_pass(False)
    path2 = ctx.series_last_path()
    same = bool(len(recs2) == nrec and np.array_equal(np.array(recs2["tag"]), tags) and np.array_equal(np.asarray(f2), np.asarray(f)))
    nsimp = tslab.count_simplices(nd, dims, nt, True)
    tested = int(st["simplices_tested"])
    t_ms = float(np.mean(tile_ms))
    prof = None
    try:
        pj = json.load(open(os.path.join(ROOT, INT_PROFILES[name])))
        k = next(v for kk, v in pj["kernels"].items() if "tile_kernel" in kk)
        prof = {"valu_busy": k.get("valu_busy"), "valu_instructions_per_wave": k.get("valu_instructions_per_wave"), "source": INT_PROFILES[name], "kernel_sources_at": pj.get("kernel_sources_at")}
    except Exception:   # noqa: BLE001
        pass
    out = {"workload": f"{case} {'x'.join(str(d) for d in dims)}x{nt}" + (", exact_only (no sign cull: every simplex takes the integer test)" if exact_only else " (nbits 21, determinants may wrap: the cull is useless, every cell goes to the tile kernel)"),
           "steps": steps, "ms_per_step": float(np.mean(wall)) * 1e3, "value": nsimp / float(np.mean(wall)), "simplices_per_step": nsimp,
           "simplices_tested_exactly": tested, "hits": nrec, "nbits": int(np.log2(max(int(v) for v in f))),
           "kernel": "ftkx::tile_kernel<3, %d, false>" % (2 if exact_only else 1), "kernel_launches_per_pass": int(launches), "kernel_ms_per_pass": t_ms,
           "roofline": {"bound": "int_valu", "unit": "simplices/s", "achieved": tested / (t_ms * 1e-3), "peak": tested / (fan_ms * 1e-3) if fan_ms > 0 else None,
                        "frac": fan_ms / t_ms if fan_ms > 0 else None, "fan_phase_ms_per_pass": fan_ms, "traffic": None, "counters": prof,
                        "note": "achieved: simplices tested / device time of tile_kernel (HIP events, live); peak: the same simplices / the time of the kernel's fan phase alone "
                                "(predicate arithmetic on tiles staged in LDS: passes with the phase repeated 6 and 2 times, differenced); the rest of the kernel is staging "
                                "(block of S, gradients, quantisation), the list of degenerate simplices and the hand-over of hits"},
           "check": {"hits": nrec, "same_records_and_factors_the_other_way": same, "other_way": ("culled pass" if exact_only else "exact_only"), "other_way_path": list(path2), "ok": same and nrec > 0}}
    ctx.close()
    del keep
    torch.cuda.empty_cache()
    return out


def boundary_call(torch, dev, ftk_amd, synthetic):
    """The literal drop-in boundary -- ftkx_extract_cp3dt with the reference's argument list (critical_point_tracker_3d_regular.hh:42-56, call
    sites 248-260 / 274-286): HOST V / J / S of the current and the next timestep, one call per scope, records back in a malloc'ed array --
    i.e. what patches/ftk-xl-hip.patch alone gives a user of the reference.  256^3: 1.7 GB (ordinal) / 3.5 GB (interval) cross PCIe per call;
    that, not the sweep, is what the call costs."""
    dims, nt = (256, 256, 256), 16
    DW, DH, DD = dims
    ctx = ftk_amd.Context(3, dev.index or 0)
    host = []
    for t in (0, 1):
        S = synthetic.generate("moving_extremum_3d", dims, t, nt, torch, dev)
        V = torch.empty((DD, DH, DW, 3), dtype=torch.float64, device=dev)
        J = torch.empty((DD, DH, DW, 3, 3), dtype=torch.float64, device=dev)
        ctx.gradient3D(S.data_ptr(), DW, DH, DD, V.data_ptr())
        ctx.jacobian3D(V.data_ptr(), DW, DH, DD, J.data_ptr())
        torch.cuda.synchronize()
        host.append((V.cpu().numpy(), J.cpu().numpy(), S.cpu().numpy()))
        del S, V, J
    ctx.close()
    torch.cuda.empty_cache()
    dom = ([2] * 3, [d - 3 for d in dims])
    opt = ftk_amd.default_options(jacobian_symmetric=1, tag_mode=ftk_amd.TAG_WORK_INDEX)
    out = {"workload": "moving_extremum_3d 256x256x256, timesteps 0 and 1, host V / J / S (ndarray<double> as the reference hands them over)"}
    x0, dv = synthetic.moving_extremum_params(dims)
    for name, scope in (("ordinal", ftk_amd.SCOPE_ORDINAL), ("interval", ftk_amd.SCOPE_INTERVAL)):
        nxt = scope == ftk_amd.SCOPE_INTERVAL
        best, recs = None, None
        for rep in range(2):
            t0 = time.perf_counter()
            recs = ftk_amd.extract_cp3dt(scope, 0, (dom[0] + [0], dom[1] + [2 ** 31 - 1]), (dom[0] + [0], dom[1] + [1]), ([0] * 3, list(dims)),
                                         host[0][0], host[1][0] if nxt else None, host[0][1], host[1][1] if nxt else None, host[0][2], host[1][2] if nxt else None, 256, opt)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        nbytes = sum(a.nbytes for a in host[0]) + (sum(a.nbytes for a in host[1]) if nxt else 0)
        err = max(float(np.abs(recs["x"][:, a] - (x0[a] + dv[a] * recs["t"])).max()) for a in range(3)) if len(recs) else None
        out[name] = {"ms": best * 1e3, "host_bytes_in": nbytes, "GB/s": nbytes / best / 1e9, "records": int(len(recs)), "max_abs_position_error_vs_analytic": err,
                     "ok": bool(len(recs) >= 1 and err is not None and err < 1e-6)}
    out["note"] = "one-shot calls: context, upload of every array, masks, cull, exact test, records, download, teardown inside each call; best of 2"
    return out


def patched_reference_tracker(synthetic, expect_records=None):
    """What a user of hguo/ftk gets from patches/ftk-xl-hip.patch: oracle/_ref/ftk_shim_driver is the REAL
    ftk::critical_point_tracker_3d_regular compiled from the patched headers, told use_accelerator("hip"), driven the way the reference's
    callers drive it (push_scalar_field_snapshot + advance_timestep per timestep, pageable ndarray<double>s made afresh for every step).
    resident (the patch's default): each snapshot crosses PCIe once, 8 bytes per vertex, and stays in HBM for the two steps that read it;
    one_shot (set_hip_resident(false)): host gradient / jacobian per push and V, J, S of both snapshots across PCIe on every call.
    A separate process, after the timed region; never `value`."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ftk_shim_driver")
    if not os.path.exists(drv):
        return {"error": "oracle/_ref/ftk_shim_driver not built (needs the reference tree: build container only)"}
    dims = (256, 256, 256)
    x0, dv = synthetic.moving_extremum_params(dims)
    nvert = dims[0] * dims[1] * dims[2]
    out = {"workload": "moving_extremum_3d 256x256x256, host-fed scalar snapshots (8 B/vertex), the reference's own push / advance_timestep loop",
           "binary": "oracle/_ref/ftk_shim_driver (the reference's trackers + patches/ftk-xl-hip.patch, no override)"}

    def run(nt, oneshot):
        env = {k: v for k, v in os.environ.items() if not k.startswith(("FTK_REF_", "FTK_SHIM_"))}
        env["FTK_REF_PER_CALL"] = "1"
        if oneshot:
            env["FTK_SHIM_ONESHOT"] = "1"
        cmd = [drv, "synthetic", "moving_extremum_3d", str(dims[0]), str(dims[1]), str(dims[2]), str(nt), "/dev/null"] + [repr(float(v)) for v in list(x0) + list(dv)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True, env=env)
        lines = r.stdout.decode().strip().splitlines()
        return json.loads(lines[-2]), json.loads(lines[-1])

    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if v else None

    try:
        info, per = run(16, False)
        push, upd = per["push_ms"][1:], per["update_ms"][1:-1]          # (the first calls create the context and allocate; the last update is ordinal only)
        steady = med(push) + med(upd)
        out["resident"] = {"timesteps": 16, "ms_per_step": steady, "push_ms": med(push), "update_timestep_ms": med(upd),
                           "ms_per_step_mean_incl_first_call": info["loop_seconds"] / info["steps"] * 1e3,
                           "first_push_ms": per["push_ms"][0], "first_update_ms": per["update_ms"][0],
                           "host_GB/s": 8.0 * nvert / (med(push) * 1e-3) / 1e9, "records": info["records"], "hip_resident": info["hip_resident"],
                           "ok": bool(info["hip_resident"] and info["records"] >= 1 and (expect_records is None or info["records"] == expect_records)),
                           "note": "median over steps 1..15 of push (the fresh pageable ndarray staged through pinned pieces by the library's copy threads, ftk_amd/csrc/upload.cpp) + update_timestep (one device-driven pass, records through the reference's from_work_index / to_integer loops)"}
        info1, per1 = run(3, True)
        push1, upd1 = per1["push_ms"][1:], per1["update_ms"][1:-1]
        out["one_shot"] = {"timesteps": 3, "ms_per_step": med(push1) + med(upd1), "push_ms": med(push1), "update_timestep_ms": med(upd1),
                           "records": info1["records"], "hip_resident": info1["hip_resident"],
                           "note": "push = the reference's host gradient3D + jacobian3D; update_timestep = extract_cp3dt_hip twice (V, J, S over PCIe, 104 B/vertex and slice)"}
        out["speedup"] = out["one_shot"]["ms_per_step"] / steady
    except Exception as e:   # noqa: BLE001
        out["error"] = repr(e)
    return out


def streaming_tracker(nd, case, dims, nt_run, torch, dev, ftk_amd, synthetic, host_steps=4):
    """The drop-in's everyday use (critical_point_tracker_regular: push_*_snapshot / advance_timestep per timestep, the records of every
    step on the host before the next push): per-step wall time with device-resident input, and the same fed from HOST arrays -- what the
    reference's accelerator boundary hands over (critical_point_tracker_2d_regular.hh:369-384) -- as a rate against the box's own pinned
    hipMemcpyAsync."""
    T = ftk_amd.CriticalPointTracker3DRegular if nd == 3 else ftk_amd.CriticalPointTracker2DRegular

    def make(deferred=False, depth=1):
        tr = T()
        tr.set_scalar_field_source(ftk_amd.SOURCE_GIVEN); tr.set_vector_field_source(ftk_amd.SOURCE_DERIVED)
        tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(True)
        tr.set_domain([2] * nd, [d - 3 for d in dims]); tr.set_array_domain([0] * nd, list(dims))
        tr.set_tag_mode(ftk_amd.TAG_EXACT64)
        tr.initialize()
        if deferred:
            tr.set_deferred_collection(True, depth)
        return tr

    def drive(tr, snaps):
        n = len(snaps)
        t0 = time.perf_counter()
        for k in range(n):
            tr.push_scalar_field_snapshot(snaps[k])
            if k != 0:
                tr.advance_timestep()
            if k == n - 1:
                tr.update_timestep()
        return time.perf_counter() - t0

    out = {}
    dev_snaps = [synthetic.generate(case, dims, t, nt_run, torch, dev) for t in range(nt_run)]
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        tr = make()
        dt = drive(tr, dev_snaps)
        nrec = len(tr.get_critical_points()[0])
        tr.close()
        best = dt if best is None or dt < best else best
    out["device_resident"] = {"timesteps": nt_run, "ms_per_step": best / nt_run * 1e3, "records": int(nrec),
                              "note": "push (device pointer, borrowed) + advance_timestep per step, every step's records on the host before the next push; best of 3 series"}
    best = None
    for rep in range(3):
        tr = make(deferred=True)
        t0 = time.perf_counter()
        drive(tr, dev_snaps)
        tr.sync()                      # (the steps still out are collected inside the timed series)
        dt = time.perf_counter() - t0
        nrec2 = len(tr.get_critical_points()[0])
        tr.close()
        best = dt if best is None or dt < best else best
    out["device_resident_deferred"] = {"timesteps": nt_run, "ms_per_step": best / nt_run * 1e3, "records": int(nrec2),
                                       "note": "set_deferred_collection(True): step t+1's sweep is queued (continuing on the device from step t's running minimum) before "
                                               "step t's records are collected; same records, visible one step later; sync() inside the timed series; best of 3"}
    best = None
    for rep in range(3):
        tr = make(deferred=True, depth=4)
        t0 = time.perf_counter()
        drive(tr, dev_snaps)
        tr.sync()
        dt = time.perf_counter() - t0
        nrec3 = len(tr.get_critical_points()[0])
        tr.close()
        best = dt if best is None or dt < best else best
    out["device_resident_deferred_4"] = {"timesteps": nt_run, "ms_per_step": best / nt_run * 1e3, "records": int(nrec3),
                                         "note": "set_deferred_collection(True, 4): the same per-step calls; the sweeps of four consecutive steps are queued as ONE pass (one mask "
                                                 "launch, one tail) when the fourth is advanced, up to three passes in flight; same records, visible up to 3 x 4 steps later; sync() inside "
                                                 "the timed series; best of 3"}
    # host-fed: numpy arrays in pageable memory, like an ndarray<double> of the reference
    h = min(host_steps, nt_run)
    host_snaps = [dev_snaps[t].cpu().numpy() for t in range(h)]
    bytes_each = host_snaps[0].nbytes
    del dev_snaps
    torch.cuda.empty_cache()
    # the box's own rate: one pinned buffer, hipMemcpyAsync into a device buffer
    pin = torch.empty(host_snaps[0].shape, dtype=torch.float64, pin_memory=True)
    pin.copy_(torch.from_numpy(host_snaps[0]))
    dst = torch.empty(host_snaps[0].shape, dtype=torch.float64, device=dev)
    rates = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
        rates.append(bytes_each / (time.perf_counter() - t0) / 1e9)
    pinned_rate = max(rates[1:])
    del pin, dst
    best = None
    for rep in range(2):
        # a FRESH array per timestep, as the reference's callers make them; made before the timed series and released after it (the release
        # of 1 GiB of small pages is 40 ms of the caller's time that no boundary can change)
        fresh = [a.copy() for a in host_snaps]
        tr = make()
        dt = drive(tr, fresh)
        tr.sync() if hasattr(tr, "sync") else None
        torch.cuda.synchronize()
        tr.close()
        del fresh
        best = dt if best is None or dt < best else best
    out["host_fed"] = {"timesteps": h, "bytes_per_step": bytes_each, "ms_per_step": best / h * 1e3, "GB/s": bytes_each * h / best / 1e9,
                       "pinned_hipMemcpyAsync_GB/s": pinned_rate, "frac_of_pinned": bytes_each * h / best / 1e9 / pinned_rate,
                       "note": "pageable host arrays (numpy, 1 GiB each at 512^3, a fresh one per timestep), staged by the push through pinned pieces (upload.cpp); the sweep of every step included"}
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process touches neither torch nor the GPU; it starts N
    fresh child processes -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run sets
    them -- waits for them under a time limit, and ends with rank 0's JSON line and exit code 0, or with the failing ranks' stderr and a
    non-zero code (the ranks still running are ended by their exact PIDs: a rank that died leaves the others waiting in a collective).
    The reference's several-rank entry needs no wrapper either (include/ftk/filters/regular_tracker.hh:127-149; the gather at
    include/ftk/filters/critical_point_tracker.hh:689)."""
    import socket
    n = args.gpus
    attempt = getattr(args, "_launch_attempt", 0)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="ftkx_bench_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        so_, se_ = open(os.path.join(tmp, f"rank{r}.out"), "wb"), open(os.path.join(tmp, f"rank{r}.err"), "wb")
        procs.append((r, subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=so_, stderr=se_, start_new_session=True), so_, se_))
    deadline = time.monotonic() + args.rank_timeout
    failed, timed_out = [], False
    while True:
        codes = [p.poll() for _, p, _, _ in procs]
        failed = [r for (r, _, _, _), c in zip(procs, codes) if c not in (None, 0)]
        if failed or all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            timed_out = True
            break
        time.sleep(0.05)
    if failed or timed_out:
        time.sleep(0.5)                                      # (ranks that were about to fail the same way get to say so)
        for r, p, _, _ in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)                     # (its own session: exactly this rank and what it started)
                except ProcessLookupError:
                    pass
        for r, p, _, _ in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except ProcessLookupError:
                    pass
                p.wait()
    for _, _, so_, se_ in procs:
        so_.close(); se_.close()

    def tail(r, what, nbytes=6000):
        with open(os.path.join(tmp, f"rank{r}.{what}"), "rb") as f:
            return f.read()[-nbytes:].decode(errors="replace")
    if failed and not timed_out and attempt == 0 and any("address already in use" in tail(r, "err").lower() or "eaddrinuse" in tail(r, "err").lower() for r, _, _, _ in procs):
        # the port was free when it was picked and taken when rank 0's store bound it (another launch on this box): once more, with a fresh one
        print("bench.py --gpus %d: rendezvous port %d was taken meanwhile; launching the ranks once more" % (n, port), file=sys.stderr)
        args._launch_attempt = 1
        return launch_ranks(args, argv)
    if failed or timed_out:
        codes = {r: p.returncode for r, p, _, _ in procs}
        bad = failed if failed else [r for r, _, _, _ in procs]
        print("bench.py --gpus %d: %s; exit codes by rank %s" % (n, ("rank(s) %s failed" % failed) if failed else ("no result within %d s (--rank-timeout)" % args.rank_timeout), codes), file=sys.stderr)
        for r in bad[:4]:
            print("---- rank %d stderr (tail) ----\n%s" % (r, tail(r, "err")), file=sys.stderr)
        code = next((codes[r] for r in failed if codes[r] and codes[r] > 0), 1)
        sys.exit(code if 0 < code < 256 else 1)
    sys.stderr.write(tail(0, "err", 2000))
    line = [ln for ln in tail(0, "out", 1 << 24).splitlines() if ln.startswith("{")]
    if not line:
        print("bench.py --gpus %d: every rank ended with code 0 but rank 0 printed no JSON line" % n, file=sys.stderr)
        sys.exit(1)
    print(line[-1])
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=os.environ.get("FTKX_BENCH_CONFIG", "c4"), choices=sorted(CONFIGS))
    ap.add_argument("--exact-only", action="store_true", help="disable the sign cull (every simplex takes the integer test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: dry run of the N>1 path (host-staged halo)")
    ap.add_argument("--single-device", action="store_true", help="all ranks on cuda:0 (dry run of the N>1 logic on a 1-GPU box)")
    ap.add_argument("--halo-in-loop", action="store_true", help="N > 1: re-send the slab-boundary slice inside every timed pass")
    ap.add_argument("--compact-halo", dest="compact_halo", action="store_true", default=True,
                    help="N > 1 (default): inside every timed pass, exchange sign masks + patches around the surviving cells instead of the boundary slice")
    ap.add_argument("--full-halo", dest="compact_halo", action="store_false", help="N > 1: the boundary slice itself (before the timed region, or with --halo-in-loop inside it)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default) = the configuration's own series cut into N slabs (c4 over 8 GPUs is BASELINE.json's literal 512^3 x 32 case); weak = one slab of the configuration's length per rank (series of nt*N timesteps)")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: skip the few weak-scaling passes reported as `other`")
    ap.add_argument("--force-dist", action="store_true", help="take the several-rank code paths (process group, all_gather of the reductions, halo protocol, record gather) even with one rank")
    ap.add_argument("--timesteps", type=int, default=0, help="override the length of the series")
    ap.add_argument("--no-cull-ahead", action="store_true", help="experiment: do not announce the sweeps to slices_prepare (the cull then waits for the factors)")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no HIP events around the kernels (what do they cost a pass?); the line then carries no roofline")
    ap.add_argument("--no-pipeline", action="store_true", help="N = 1: one pass after the other (ftkx_sweep_series) instead of passes in flight (ftkx_sweep_series_submit / _complete)")
    ap.add_argument("--host-driven", action="store_true", help="N = 1: the host-driven batch (slices_prepare, factors on the host, enqueue, collect) instead of the device-driven ftkx_sweep_series")
    ap.add_argument("--no-streaming-tracker", action="store_true", help="N = 1: skip the per-timestep tracker measurement (device-resident and host-fed) that follows the timed region")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the few passes of the other BASELINE configurations that follow the timed region")
    ap.add_argument("--dump-merged", default=None, help="rank 0 writes the merged records and the curves traced from them (npz)")
    ap.add_argument("--no-sustained", action="store_true", help="N = 1: skip the >= 1.2 s repetition of the timed loop that follows it")
    ap.add_argument("--rank-timeout", type=int, default=900, help="--gpus N without a launcher: seconds the N child processes get before they are ended")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)      # (tests: this rank exits with code 7 before it joins the process group)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args, sys.argv[1:])                     # (does not return; nothing above or in it touches torch or the GPU)
    if args.fail_rank >= 0 and int(os.environ.get("RANK", "0")) == args.fail_rank:
        print("bench.py: rank %d asked to fail (--fail-rank)" % args.fail_rank, file=sys.stderr)
        sys.exit(7)

    # before anything initialises the HIP runtime: the host driver only supports dmabuf IPC (RCCL / device-tensor sharing across
    # processes fails with the legacy mode)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import ftk_amd
    from ftk_amd import synthetic, tslab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` on its own, or under torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist       # take the several-rank code paths (collectives, halo protocol, record gather)
    if multi:
        if world == 1:                          # --force-dist without a launcher: a process group of one
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    env = (torch, dist, ftk_amd, synthetic, tslab, world, rank, local_rank, dev, multi)
    out = job(args, env)
    if world > 1 and args.scaling == "strong" and not args.no_other_scaling:
        # the other way to scale, for the record: every rank a slab of the configuration's whole length (weak scaling), a few passes
        import copy
        a2 = copy.copy(args)
        a2.scaling, a2.steps, a2.warmup, a2.light = "weak", min(args.steps, 3), 1, True
        o2 = job(a2, env)
        if rank == 0:
            out["other"] = {k: o2[k] for k in ("scaling", "steps", "ms_per_step", "value", "config", "halo_exchange", "roofline_end_to_end")}
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


def job(args, env):
    """one measurement: set-up, warm-up, timed passes, the JSON line's dictionary (rank 0; None elsewhere)"""
    torch, dist, ftk_amd, synthetic, tslab, world, rank, local_rank, dev, multi = env
    light = getattr(args, "light", False)
    nd, nv, case, dims, nt = CONFIGS[args.config]
    if args.timesteps > 0:
        nt = args.timesteps
    elif world > 1 and args.scaling == "weak":
        nt *= world                 # per-GPU work fixed: every rank sweeps a slab as long as the configuration's whole series
    scalar_input = nv == 1
    t0_own, t1_own = tslab.slab_range(nt, world, rank)
    own = list(range(t0_own, t1_own))

    # ---- setup (untimed): generate the slab, push it (upload/adopt + derive V), pre-pass ----
    # a real (non-null) HIP stream shared by torch and the library, so that the events below time the library's kernels
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = ftk_amd.Context(nd, local_rank)
    ctx.set_stream(stream.cuda_stream)
    lo = 2 if scalar_input else 1
    dom = ([lo] * nd, [d - (3 if scalar_input else 2) for d in dims])
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=scalar_input, derive_jacobian=1, tag_mode=ftk_amd.TAG_EXACT64, exact_only=args.exact_only)
    slices = {}
    for t in own:
        slices[t] = synthetic.generate(case, dims, t, nt, torch, dev)
        torch.cuda.synchronize()
        (ctx.push_scalar_slice if scalar_input else ctx.push_slice)(t, slices[t])
    halo_buf = torch.empty_like(slices[own[0]]) if (multi and own) else None
    torch.cuda.synchronize()

    # One pass = the whole sweep from resident input to records on the host, INCLUDING what the reference does at the top of
    # update_timestep (update_vector_field_scaling_factor): ftkx_slices_prepare reads every slice once and yields the sign masks
    # and the per-slice reduction together; the factors follow on the host (N > 1: one all_gather of 2 nt doubles).
    def scope_of(t):
        return ftk_amd.SCOPE_BOTH if (t + 1 < nt) else ftk_amd.SCOPE_ORDINAL

    ann_ts, ann_scopes = np.array(own, dtype=np.int32), np.array([scope_of(t) for t in own], dtype=np.int32)
    own_next_resident = (not own) or t1_own >= nt       # the last slab needs no halo slice: every slice its sweeps read is its own

    def prepare_and_factors():
        # the sweeps that will follow are known before the factors are: announced, their cull is queued right behind the mask kernel
        # and runs while the host waits for the reduction and forms the factors.  N > 1: the next slab's first slice, once resident,
        # is prepared in the same launch (its owner's reduction stands: set_slice_resolution)
        prep = list(own)
        if halo_pushed[0] and not args.compact_halo:
            prep.append(t1_own)
        if not args.compact_halo and not args.no_cull_ahead and (own_next_resident or halo_pushed[0]):
            ctx.sweep_announce(ann_ts, ann_scopes)
        local_rm = ctx.slices_prepare(prep, 0)
        local_res = {t: local_rm[t][0] for t in own}
        local_rm = {t: local_rm[t] for t in own}
        if multi:
            return tslab.global_factors(local_res, nt, local_max={t: v[1] for t, v in local_rm.items()})
        return tslab.factors_from_resolutions([local_res[t] for t in range(nt)]), None, None

    halo_pushed = [False]
    factors, all_res, all_max = prepare_and_factors()      # also loads the code objects (a one-time cost of a few ms)
    torch.cuda.synchronize()

    host_ms = [0.0, 0.0, 0.0]

    def halo():
        # the slab-boundary slice: one RCCL send/recv pair per neighbour over xGMI, into a device buffer the context adopts
        have_halo = tslab.exchange_halo(slices[own[0]], halo_buf, nt)
        if have_halo and not halo_pushed[0]:
            (ctx.push_scalar_slice if scalar_input else ctx.push_slice)(t1_own, halo_buf)
            ctx.set_slice_resolution(t1_own, all_res[t1_own], all_max[t1_own])   # its owner's reduction, from the all_gather
            halo_pushed[0] = True

    halo_info = None
    compact_bytes = [0, 0, 0, 0]
    if multi and args.compact_halo:
        args.halo_in_loop = False
        halo_info = {"compact": True, "in_timed_region": True}
    elif multi:
        # input distribution (untimed region, reported separately): a first exchange warms RCCL up, the second one is timed
        for rep in range(2):
            dist.barrier(); torch.cuda.synchronize()
            th0 = time.perf_counter()
            if own:
                halo()
            torch.cuda.synchronize(); dist.barrier()
            halo_ms = (time.perf_counter() - th0) * 1e3
        hbytes = int(halo_buf.numel() * halo_buf.element_size()) if halo_buf is not None else 0
        halo_info = {"in_timed_region": bool(args.halo_in_loop), "ms": halo_ms, "bytes_per_rank": hbytes,
                     "GB/s_per_link": hbytes / (halo_ms * 1e-3) / 1e9 if halo_ms > 0 else None}

    # N > 1: the device-driven slab pass (ftk_amd/tslab.py: SlabSeries over ftkx_series_dist_*) -- masks + reduction, the all_gather of
    # the ranks' contributions, the compact halo's three neighbour messages, cull, factors, exact test and records queued on the stream,
    # ONE host wait per pass, two passes in flight; --host-driven: round 2's sequence (prepare, host all_gather, enqueue, collect)
    slab = None
    if multi and not args.host_driven and args.compact_halo and not args.exact_only:
        try:
            slab = tslab.SlabSeries(ctx, nt, own, scalar_input, torch, dev, first_slice=slices[own[0]] if own else None)
        except RuntimeError as e:      # (a mesh without summarised masks -- the same on every rank: they share it -- takes the host-driven sequence)
            if rank == 0:
                print("bench.py: %s; the host-driven sequence instead" % e, file=sys.stderr)
            slab = None
    series_paths = {}
    pass_stamps, path_list = [], []      # wall-clock time at which each timed pass's records were on the host; the way each pass went

    def one_pass():
        ctx.invalidate_masks()      # every pass redoes ALL the work of the sweep: masks, reduction, factors, cull, exact test, download
        if slab is not None:
            tp0 = time.perf_counter()
            slab.submit()
            recs, f, _ = slab.complete(copy=False)
            host_ms[1] += (time.perf_counter() - tp0) * 1e3
            if own:
                series_paths[slab.last_path] = series_paths.get(slab.last_path, 0) + 1
                path_list.append(slab.last_path)
            one_pass.factors = f if own else factors
            return recs, ctx.stats()
        if not multi and not args.host_driven:
            # one GPU: the device-driven pass -- masks + reduction, factors (on the device), cull, exact test, records, their order and
            # their way into the pinned host buffer queued at once; the host waits once (ftkx_sweep_series)
            tp0 = time.perf_counter()
            recs, f, _ = ctx.sweep_series(ann_ts, ann_scopes, copy=False)
            host_ms[1] += (time.perf_counter() - tp0) * 1e3
            p = ctx.series_last_path()
            series_paths[p] = series_paths.get(p, 0) + 1
            path_list.append(p)
            one_pass.factors = f
            return recs, ctx.stats()
        if multi and own and args.halo_in_loop:
            halo()
        tp0 = time.perf_counter()
        f, _, mx_all = prepare_and_factors()
        t_masked = None
        if multi and args.compact_halo:
            # compact halo, step 1: the next slab's first slice arrives as sign masks only (they were just built by its owner)
            t_masked, sb, rb = tslab.compact_halo_masks(ctx, own, nt, scalar_input, mask_factor=256, max_abs=mx_all)
            compact_bytes[0] += sb; compact_bytes[1] += rb
        te0 = time.perf_counter()
        if own:     # one call for the slab's sweeps (per-sweep calls through ctypes cost a hit-dense 2D pass 5 %)
            ctx.sweep_enqueue_many(ann_ts, ann_scopes, [f[t] for t in own])
        if multi and args.compact_halo:
            # step 2: cull, then the input values around the boundary step's surviving cells from the slice's owner -- or, where
            # that would be more bytes than the slice (hit-dense data on small slices), the slice itself after all
            def push_full(t, buf):
                ctx.sweep_cancel()
                (ctx.push_scalar_slice if scalar_input else ctx.push_slice)(t, buf)
            ncell, sb, rb = tslab.compact_halo_patches(ctx, own, nt, t_masked, first_slice=slices[own[0]] if own else None,
                                                       halo_buffer=halo_buf, push_full=push_full)
            compact_bytes[0] += sb; compact_bytes[1] += rb; compact_bytes[2] += max(ncell, 0); compact_bytes[3] += 1 if ncell < 0 else 0
            if ncell < 0:
                ctx.sweep_enqueue_many(ann_ts, ann_scopes, [f[t] for t in own])
        te1 = time.perf_counter()
        # one cull / exact launch for the whole slab (plus the masks of the halo slice, N > 1), then the hit download into the
        # library's pinned host buffer
        recs = ctx.sweep_collect(copy=False)
        host_ms[2] += (te0 - tp0) * 1e3; host_ms[0] += (te1 - te0) * 1e3; host_ms[1] += (time.perf_counter() - te1) * 1e3
        st = ctx.stats()
        return recs, st

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    pipelined = (not multi or slab is not None) and (not args.host_driven) and (not args.no_pipeline) and (not args.exact_only)

    def passes(k):
        """k passes.  One GPU, device-driven: up to IN_FLIGHT passes in flight (ftkx_sweep_series_submit / _complete) -- the host prepares and queues
        pass i + 1 while the device works on pass i, and the records of pass i cross PCIe on a copy engine while the mask kernel of pass
        i + 1 runs; every pass still does ALL the work and hands its records to the host.  Otherwise: one pass after the other."""
        if not pipelined or k < 2:
            for _ in range(k):
                out = one_pass()
                pass_stamps.append(time.perf_counter())
            return out
        tp0 = time.perf_counter()

        def submit():
            ctx.invalidate_masks()
            slab.submit() if slab is not None else ctx.sweep_series_submit(ann_ts, ann_scopes)
        in_flight = IN_FLIGHT if slab is None else 2          # (slab passes: two)
        submitted = done = 0
        while done < k:
            while submitted < k and submitted - done < in_flight:
                submit()
                submitted += 1
            recs, f, _ = slab.complete(copy=False) if slab is not None else ctx.sweep_series_complete(copy=False)
            done += 1
            pass_stamps.append(time.perf_counter())
            p = slab.last_path if slab is not None else ctx.series_last_path()
            if own:
                series_paths[p] = series_paths.get(p, 0) + 1
                path_list.append(p)
        host_ms[1] += (time.perf_counter() - tp0) * 1e3
        one_pass.factors = f if own else factors
        return recs, ctx.stats()

    if args.warmup or pipelined:
        # (passes in flight, one queued KNOWING how many records the data gives, the library's self-check of the split pass -- a few passes in
        # order, a few split, then its decision: ~22 passes -- and enough of them that the
        # split pass -- taken from the third pass on -- has its stream, its events and the second set of mask arrays it swaps in for the slices
        # the pass before still reads (512^3 x 32: 4.3 GB of hipMalloc, 48 ms once): both sets of buffers, the device-side record buffers, the
        # copy and tail streams exist before the clock starts -- also when --warmup 0 is asked for)
        recs, st = passes(max(args.warmup, 26) if pipelined else args.warmup)
    # HIP events on the stream the kernels run on: around the dominant (mask) kernel only inside the timed region -- a pair of events costs
    # the stream ~10 us of idle time, which a 0.4 ms pass notices --, around every kernel family in a few extra passes afterwards
    ctx.set_profiling(0 if args.no_kernel_events else (1 if args.exact_only else 2))     # (--exact-only: the dominant kernel is the tile kernel)
    host_ms[0] = host_ms[1] = host_ms[2] = 0.0
    compact_bytes[0] = compact_bytes[1] = compact_bytes[2] = compact_bytes[3] = 0
    if slab is not None:
        slab.bytes_sent = slab.bytes_received = slab.fallbacks = 0
    barrier()
    series_paths.clear()
    del pass_stamps[:], path_list[:]
    tt0 = time.perf_counter()
    recs, st = passes(args.steps)
    barrier()
    elapsed = time.perf_counter() - tt0
    per_pass_ms = np.diff(np.array([tt0] + pass_stamps[:args.steps])) * 1e3      # (pipelined: the time between consecutive completions)
    timed_paths = list(path_list)
    own_elapsed = elapsed
    if multi:
        cdev = dev if args.backend == "nccl" else "cpu"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        agg = torch.tensor([float(len(recs)), float(st["simplices_tested"]), float(st["cells_survived"])], dtype=torch.float64, device=cdev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        n_hits, n_tested, n_cells = (int(v) for v in agg.tolist())
    else:
        n_hits, n_tested, n_cells = len(recs), st["simplices_tested"], st["cells_survived"]

    total_simplices = tslab.count_simplices(nd, dims, nt, scalar_input)
    torch.cuda.synchronize()
    ktimes = ctx.kernel_times()
    # the same loop once more, for at least a second: what the K timed steps give, held over a run long enough for an outside sampler of the
    # GPU's activity to see it (a 0.1 s timed region falls between the samples)
    sustained = None
    if not multi and not light and not args.no_sustained:
        ks = max(args.steps, int(np.ceil(1.2 / max(elapsed / args.steps, 1e-6))))
        ks = min(ks, 20000)
        n_before, saved_paths, saved_host = len(path_list), dict(series_paths), list(host_ms)
        ctx.set_profiling(0)             # (no events: nothing stands between the kernels -- the timed region keeps its pair around the mask kernel, as the contract asks)
        barrier()
        ts0 = time.perf_counter()
        passes(ks)
        barrier()
        es = time.perf_counter() - ts0
        series_paths.clear(); series_paths.update(saved_paths)
        host_ms[0], host_ms[1], host_ms[2] = saved_host
        sustained = {"steps": ks, "seconds": es, "ms_per_step": es / ks * 1e3, "value": tslab.count_simplices(nd, dims, nt, scalar_input) * ks / es,
                     "paths": sorted(set(str(p) for p in path_list[n_before:])), "note": "the timed loop again, untimed for `value`: the same passes for >= 1.2 s"}
        del path_list[n_before:]
    # who took part (several ranks): what the process group says its size is, every rank's device and its own wall time -- so that a line
    # claiming N GPUs shows N distinct devices that each did a share of the work
    ranks_info = None
    if multi:
        mine = {"rank": rank, "device": int(dev.index or 0), "pci_bus_id": pci_bus_id(torch, dev), "timesteps": len(own), "ms_per_step": own_elapsed / args.steps * 1e3,
                "slab_fallbacks": int(slab.fallbacks) if slab is not None else None, "pid": os.getpid()}
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        try:
            rv = torch.cuda.nccl.version() if args.backend == "nccl" else None
            rv = ".".join(str(v) for v in rv) if isinstance(rv, (tuple, list)) else rv
        except Exception:   # noqa: BLE001
            rv = None
        ranks_info = {"ranks_seen": int(dist.get_world_size()), "backend": dist.get_backend(), "rccl_version": rv,
                      "slab_host": ("C++ (include/ftkx_slab.h) over " + slab.transport) if slab is not None else "host-driven sequence (Python)",
                      "distinct_devices": len(set(e["pci_bus_id"] for e in everyone)), "per_rank": everyone}
    latency_ms = None
    if pipelined:
        # the same pass on its own (ftkx_sweep_series, nothing else in flight): what one call takes from its first launch to the records
        saved = list(host_ms)
        ctx.set_profiling(0)
        barrier()
        tl0 = time.perf_counter()
        for _ in range(3):
            one_pass()
        barrier()
        latency_ms = (time.perf_counter() - tl0) / 3 * 1e3
        host_ms[0], host_ms[1], host_ms[2] = saved
    ktimes_all, k_all = None, 3
    if not args.no_kernel_events and not (multi and args.compact_halo):
        ctx.set_profiling(1)
        for _ in range(k_all):
            one_pass()
        barrier()
        ktimes_all = ctx.kernel_times()
        ctx.set_profiling(0)

    if slab is not None:
        npass = args.steps + (3 if pipelined else 0)      # (the latency passes behind the timed region count as well)
        compact_bytes[0], compact_bytes[1], compact_bytes[3] = slab.bytes_sent * args.steps / npass, slab.bytes_received * args.steps / npass, slab.fallbacks
        compact_bytes[2] = max(getattr(slab, "last_asked", 0), 0) * args.steps
        halo_info["protocol"] = "queued on the stream between the stages of the slab pass (ftkx_series_dist_*): all_gather of 4 doubles per rank, masks, request, reply"
    if multi and args.compact_halo:
        halo_info.update({"bytes_sent_per_pass_this_rank": compact_bytes[0] / args.steps, "bytes_received_per_pass_this_rank": compact_bytes[1] / args.steps,
                          "cells_requested_per_pass_this_rank": compact_bytes[2] / args.steps,
                          "passes_that_fell_back_to_the_whole_slice": compact_bytes[3],
                          "full_slice_bytes": int(np.prod(dims)) * 8 * (1 if scalar_input else nd)})
    # N > 1: the other convention, for the record -- the slab-boundary slice re-sent inside every pass (same barriers, max over ranks)
    other = None
    if multi and not args.compact_halo and not light:
        k2 = min(args.steps, 3)
        flip = not args.halo_in_loop
        args.halo_in_loop = flip
        recs2, _ = one_pass()                                   # untimed: first pass of the other convention
        barrier()
        t20 = time.perf_counter()
        for _ in range(k2):
            recs2, _ = one_pass()
        barrier()
        e2 = torch.tensor([time.perf_counter() - t20], dtype=torch.float64, device=cdev)
        dist.all_reduce(e2, op=dist.ReduceOp.MAX)
        args.halo_in_loop = not flip
        other = {"halo_in_timed_region": flip, "steps": k2, "ms_per_step": float(e2.item()) / k2 * 1e3,
                 "value": total_simplices * k2 / float(e2.item())}

    # ---- after the timed region: merge the slabs' hit buffers on rank 0 and run pass 2 on the merged set ("host merge of
    # hit buffers into the union_find / trace stage"; critical_point_tracker.hh:689-717).  Curves cross slab boundaries.
    tm0 = time.perf_counter()
    merged = tslab.gather_records(np.array(recs), 0) if multi else np.array(recs)
    tm1 = time.perf_counter()
    pass2 = None
    if rank == 0:
        ftk_amd.pass2(nd, dom, merged, ctx)               # (first call: the worker threads start, the candidate tables go up)
        curves, loop, nspecial, trajs, ms_trace, ms_post = ftk_amd.pass2(nd, dom, merged, ctx)
        _c, _l, _n, _t, ms_trace_host, _p = ftk_amd.pass2(nd, dom, merged)
        pass2 = {"records": int(len(merged)), "curves": len(curves), "branching_points_dropped": int(nspecial), "trajectories_after_post_process": len(trajs),
                 "gather_ms": (tm1 - tm0) * 1e3, "trace_ms": ms_trace, "trace_ms_host_only": ms_trace_host, "post_process_ms": ms_post,
                 "end_to_end_with_pass2_ms": elapsed / args.steps * 1e3 + ms_trace + ms_post,
                 "note": "untimed region: merge of the ranks' records on rank 0, ftkx_trace_curves_ctx (neighbour search + component labelling on the GPU, seeds and walks on host threads) and ftkx_post_process_curves (host threads) on the merged set; the C calls timed by themselves"}
        if args.dump_merged:
            np.savez(args.dump_merged, records=merged, curve_offsets=np.cumsum([0] + [len(c) for c in curves]),
                     curve_indices=np.concatenate(curves) if curves else np.zeros(0, dtype=np.int64), curve_loop=np.asarray(loop))
        n_hits = len(merged)

    # the result itself against what the configuration must give (cheap, size-independent): the device-driven form was what ran, tags
    # ascending and unique, the expected count / type histogram (c1, c2, c5) or the single extremum on x0 + dir * t (c3, c4)
    check = {"hits": n_hits}
    if rank == 0 and not args.exact_only:
        if (not multi or slab is not None) and not args.host_driven:
            # ((1, 64): the fused tail declined late -- few coarse cells with more records than it orders -- and the chain took the pass: device-driven
            # all the same; only the small test configurations see it)
            want_paths = [(2, 32), (5, 0)] if case == "moving_extremum_3d" else ([(1, 0), (2, 32), (4, 544)] if args.config == "c1" else ([(1, 0), (1, 64), (4, 544)] if args.config.startswith("small") else [(1, 0), (5, 0)]))
            if multi:
                want_paths = want_paths + [(4, 544)]      # (a rank's slab may be small enough for the one-launch pass where it sweeps on its own: the whole-slice recovery)
            check = check_records(args.config if args.timesteps == 0 else "", case, dims, nt, merged, timed_paths, want_paths)
        elif case == "moving_extremum_3d" and len(merged):
            check = check_records("", case, dims, nt, merged, [], [])
        if pass2:
            check["curves"] = pass2["curves"]

    out = None
    if rank == 0 and args.no_kernel_events:
        out = {"experiment": "no kernel events", "config": args.config, "ms_per_step": elapsed / args.steps * 1e3,
               "wall_breakdown_ms_per_pass": {"prepare": host_ms[2] / args.steps, "enqueue": host_ms[0] / args.steps, "collect": host_ms[1] / args.steps}}
    elif rank == 0:
        n_vertex = int(np.prod(dims))
        c = 1 if scalar_input else nd
        # dominant kernel = the one with the most device time; one launch of it covers `len(own)` slices (batched launch)
        domk = max(ktimes, key=lambda k: ktimes[k][0])
        dom_ms, dom_n = ktimes[domk]
        if dom_n == 0 and ktimes_all:          # (a pass without a mask kernel: take the figure of the extra passes with events everywhere)
            domk = max(ktimes_all, key=lambda k: ktimes_all[k][0])
            dom_ms, dom_n = ktimes_all[domk][0] * args.steps / k_all, ktimes_all[domk][1] * args.steps / k_all
        avg_ms = dom_ms / max(1, dom_n)
        slices_per_launch = len(own) * args.steps / max(1, dom_n)
        alg_bytes_launch = (8.0 * c * n_vertex) * slices_per_launch + 72.0 * len(recs) * args.steps / max(1, dom_n)
        achieved = alg_bytes_launch / (avg_ms * 1e-3) / 1e9
        kt_break = ktimes_all if ktimes_all else ktimes
        n_break = k_all if ktimes_all else args.steps
        all_ms = sum(v[0] for v in kt_break.values()) / n_break
        alg_bytes_pass = 8.0 * c * n_vertex * len(own) + 72.0 * len(recs)
        # the profiler's name of the dominant kernel (the mask family has several instantiations; the library says which one ran)
        kernel_symbol = "ftkx::%s<%d>" % (domk, nd)
        if domk == "mask_kernel":
            kernel_symbol = (ctx._L.ftkx_last_mask_kernel() or b"").decode() or kernel_symbol
        # HBM bytes per launch from the committed PMC passes (profiles/traffic.json, written by tools/summarize_profiles.py): only
        # for the very kernel instantiation that ran here -- a profile of another kernel generation is not this kernel's traffic
        traffic, traffic_source = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                ent = json.load(open(tj)).get(args.config, {})
                if ent.get("kernel") == kernel_symbol:
                    # (not measured in this run: a PMC pass cannot run inside it -- the figure is the committed profile's, named with the
                    # commit whose kernel sources it was taken from)
                    traffic, traffic_source = ent.get("hbm_bytes_per_launch"), "from %s (kernel sources at %s)" % (ent.get("source"), ent.get("kernel_sources_at", "round 2"))
                else:
                    traffic_source = "profiles/traffic.json holds %s, not the kernel that ran: no traffic figure" % ent.get("kernel")
            except Exception:   # noqa: BLE001
                traffic = None
        out = {
            "metric": "space-time simplices/sec", "value": total_simplices * args.steps / elapsed, "unit": "simplices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # what ran before the clock: a pipelined pass allocates over its first passes and the split pass measures itself (ten passes and the
            # ones it discards) -- all of that is warm-up, whatever --warmup said
            "warmup_effective": (max(args.warmup, 26) if pipelined else args.warmup),
            "ms_per_step": elapsed / args.steps * 1e3,
            **spread(per_pass_ms),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"{case} {'x'.join(str(d) for d in dims)}x{nt} ({args.config}), "
                                   f"{'scalar' if scalar_input else 'vector'} input, t-slab partition over {world} GPU(s): {len(own)} timesteps on rank 0",
                       "simplices_per_step": total_simplices, "exact_only": bool(args.exact_only),
                       "nbits": int(np.log2(max(int(v) for v in getattr(one_pass, "factors", factors)))), "cull": bool(st["cull_enabled"]),
                       "pass": ((("device-driven slab pass (ftkx_series_dist_* with the collectives queued between its stages)" + (", two passes in flight" if pipelined else "")) if slab is not None else
                                 "device-driven, up to %d passes in flight (ftkx_sweep_series_submit / _complete)" % IN_FLIGHT if pipelined else "device-driven (ftkx_sweep_series)") +
                                ": paths taken {(path, status): passes} = %s" % {str(k): v for k, v in series_paths.items()}) if series_paths
                               else "host-driven batch (slices_prepare, host factors, enqueue, collect)",
                       "split_pass": (ctx.series_split_decision() if slab is None and series_paths else None),
                       "input_resident": "the field the tracker API is given (S, or V for vector input) in HBM; gradient/Jacobian evaluated in flight"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_ratio": (traffic / alg_bytes_launch) if traffic else None,     # HBM bytes moved / algorithmic bytes: > 1 = re-reads
                         "streaming_ceiling": STREAM_CEILING_GBS, "frac_of_streaming_ceiling": achieved / STREAM_CEILING_GBS, "kernel": kernel_symbol, "kernel_family": "ftkx::%s<%d>" % (domk, nd), "avg_launch_ms": avg_ms, "launches_timed": int(dom_n),
                         "algorithmic_bytes_per_launch": alg_bytes_launch, "slices_per_launch": slices_per_launch,
                         "all_kernels_ms_per_pass": all_ms, "achieved_all_kernels": alg_bytes_pass / (all_ms * 1e-3) / 1e9,
                         "kernel_ms_per_pass": {k: v[0] / n_break for k, v in kt_break.items()},
                         "kernel_ms_note": "avg_launch_ms: events around the mask kernel only, inside the timed region; kernel_ms_per_pass: %d extra passes with events around every kernel family (cull = coarse cull + factors; exact = everything behind them)" % n_break},
            # the factor pre-pass is inside the timed region now (fused into the mask kernel): nothing of the sweep is left outside
            "prepass_ms": 0.0,
            "end_to_end_ms": elapsed / args.steps * 1e3,
            "single_pass_latency_ms": latency_ms,
            "roofline_end_to_end": {"achieved": alg_bytes_pass * world / (elapsed / args.steps) / 1e9 / world, "frac": alg_bytes_pass / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                    "note": "algorithmic bytes of this rank's pass / wall time of the pass (prepare + factors + cull + exact + sort + download)"},
            "sustained": sustained,
            "halo_exchange": halo_info,
            "ranks": ranks_info,
            "other_halo_convention": other,
            "pass2": pass2,
            "wall_breakdown_ms_per_pass": {"prepare_masks_and_reduction": host_ms[2] / args.steps, "enqueue_calls": host_ms[0] / args.steps,
                                           "collect_launch_sync_sort_download": host_ms[1] / args.steps},
            "stats": {"simplices_tested_exactly": n_tested, "cells_survived_cull": n_cells},
            "check": check,
            # the C library of this box: the one corner of the record path that depends on it (the class of a 3D record with a near-singular
            # Hessian, classified on the host with pow / acos / cos) is pinned by tests/test_gpu_libm.py against fixtures made with glibc 2.35
            "host_libc": glibc_version(),
        }
        if not multi and not light and not args.no_other_configs and not args.exact_only:
            # every BASELINE configuration in the driver's line: the headline fields stay on --config (c4); the others get a few passes each
            for r in list(slices):
                ctx.drop_slice(r)
            slices.clear(); halo_buf = None
            torch.cuda.empty_cache()
            others = {}
            for name in ("c1", "c2", "c3", "c4", "c5"):
                if name == args.config:
                    others[name] = {"workload": out["config"]["workload"], "steps": args.steps, "ms_per_step": out["ms_per_step"], "value": out["value"],
                                    "kernel": kernel_symbol, "kernel_avg_launch_ms": avg_ms, "frac": achieved / HBM_PEAK_GBS,
                                    "frac_of_streaming_ceiling": achieved / STREAM_CEILING_GBS,
                                    "end_to_end_frac": out["roofline_end_to_end"]["frac"], "hits": n_hits, "check": check, "headline": True, **spread(per_pass_ms)}
                    continue
                try:
                    others[name] = side_config(name, torch, dev, ftk_amd, synthetic, tslab, steps=3 if name == "c4" else 48, warmup=6)
                except Exception as e:   # noqa: BLE001
                    others[name] = {"error": repr(e)}
            for name in ("c3_exact_only", "c3o"):
                try:
                    others[name] = integer_config(name, torch, dev, ftk_amd, synthetic, tslab)
                except Exception as e:   # noqa: BLE001
                    others[name] = {"error": repr(e)}
            out["configs"] = others
        if not multi and not light and not args.no_streaming_tracker and not args.exact_only and scalar_input:
            for r in list(slices):
                ctx.drop_slice(r)
            slices.clear(); halo_buf = None
            torch.cuda.empty_cache()
            try:
                out["streaming_tracker"] = streaming_tracker(nd, case, dims, min(2 * nt, 64), torch, dev, ftk_amd, synthetic)
            except Exception as e:   # noqa: BLE001
                out["streaming_tracker"] = {"error": repr(e)}
        if not multi and not light and not args.no_streaming_tracker and not args.exact_only:
            try:
                out["boundary_call"] = boundary_call(torch, dev, ftk_amd, synthetic)
            except Exception as e:   # noqa: BLE001
                out["boundary_call"] = {"error": repr(e)}
            c3 = (out.get("configs") or {}).get("c3") or {}
            out["patched_reference_tracker"] = patched_reference_tracker(synthetic, expect_records=c3.get("hits"))
        if not multi and not light and not args.no_cpu_baseline:
            base, port = cpu_baseline(nd, case)
            out["cpu_baseline"] = base
            if port:
                out["cpu_port"] = port
            # the other dimension's baseline beside C2 / C5 (C3 / C4) when the headline is 3D (2D): the 2D configurations have their own reference
            try:
                other_nd, other_case = (2, "woven") if nd == 3 else (3, "moving_extremum_3d")
                base2, _port2 = cpu_baseline(other_nd, other_case)
                out["cpu_baseline_%dd" % other_nd] = base2
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline_%dd" % (2 if nd == 3 else 3)] = {"error": repr(e)}
    if slab is not None:
        slab.close()                 # (the slab's streams and, over RCCL, its communicator: before the context they were made on goes)
    ctx.close()
    slices.clear()
    torch.cuda.empty_cache()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
