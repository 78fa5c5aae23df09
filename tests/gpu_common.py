"""Drives the PRODUCT (ftk_amd -> libftkx.so -> HIP kernels) the way the reference's callers drive its tracker
(python/pyftk.cpp:93-142, filters/json_interface.hh:606-725)."""
import numpy as np

import ftk_amd


def run_tracker(steps, nd, nv, *, robust=True, type_filter=None, exact_only=False, tag_mode=ftk_amd.TAG_REFERENCE, device=False,
                compute_degrees=False, bounds=None, want_curves=False, after=None, rectilinear=None, explicit=None, t0=0, device_ids=None, block=2,
                factor_each_step=True, streaming=False, deferred=False):
    """returns (records, ordinal, timestep, factors[DT], stats_list)"""
    import torch
    T = ftk_amd.CriticalPointTracker2DRegular if nd == 2 else ftk_amd.CriticalPointTracker3DRegular
    tr = T(device_ids=device_ids, block=block) if device_ids else T()
    shp = steps[0].shape[:nd]
    D = [shp[nd - 1 - d] for d in range(nd)]
    if nv == 1:   # json_interface.hh:634-645
        tr.set_scalar_field_source(ftk_amd.SOURCE_GIVEN); tr.set_vector_field_source(ftk_amd.SOURCE_DERIVED)
        tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(True)
        tr.set_domain([2] * nd, [d - 3 for d in D])
    else:         # json_interface.hh:646-656
        tr.set_scalar_field_source(ftk_amd.SOURCE_NONE); tr.set_vector_field_source(ftk_amd.SOURCE_GIVEN)
        tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(False)
        tr.set_domain([1] * nd, [d - 2 for d in D])
    tr.set_array_domain([0] * nd, D)
    tr.set_enable_robust_detection(robust)
    if type_filter is not None:
        tr.set_type_filter(type_filter)
    tr.set_exact_only(exact_only)
    tr.set_enable_computing_degrees(compute_degrees)
    if bounds is not None:
        tr.set_coords_bounds(bounds)
    if rectilinear is not None:
        tr.set_coords_rectilinear(rectilinear)
    if explicit is not None:
        tr.set_coords_explicit(explicit)
    tr.set_tag_mode(tag_mode)
    if streaming:
        tr.set_enable_streaming_trajectories(True)
    tr.initialize()
    if deferred:
        tr.set_deferred_collection(True)
    if t0:
        tr.set_current_timestep(t0)
    DT = len(steps)
    factors = np.zeros(DT, dtype=np.uint64)
    stats = []
    cur = 0
    for k in range(DT):
        a = steps[k]
        if device:
            a = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            torch.cuda.synchronize()
        (tr.push_scalar_field_snapshot if nv == 1 else tr.push_vector_field_snapshot)(a)
        if k != 0:
            tr.advance_timestep()
            if factor_each_step:      # (reading the factor waits for the queued steps of a multi-device tracker)
                factors[cur] = tr.get_vector_field_scaling_factor(); stats.append(tr.get_last_stats())
            cur += 1
        if k == DT - 1:
            tr.update_timestep(); factors[cur] = tr.get_vector_field_scaling_factor(); stats.append(tr.get_last_stats())
    recs, o, ts = tr.get_critical_points()
    if want_curves:
        tr.finalize()
        curves = tr.get_traced_critical_points()
    if after is not None:
        after(tr)      # e.g. finalize / post_process / write_* on the live tracker
    tr.close()
    out = np.zeros(len(recs), dtype=[("tag", "<u8"), ("type", "<u4"), ("ordinal", "<i4"), ("timestep", "<i4"),
                                     ("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8", (3,))])
    for f in ("tag", "type", "x", "t", "scalar"):
        out[f] = recs[f]
    out["ordinal"] = o; out["timestep"] = ts
    if want_curves:
        return out, factors, stats, curves
    return out, factors, stats
