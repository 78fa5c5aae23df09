"""Host arrays -> HBM (ftk_amd/csrc/upload.cpp): a pageable source of 32 MiB and more goes up through the library's copy threads and
pinned pieces, everything else through the runtime's copy.  Which way a slice went must not show in a single record.

Reference boundary: the reference's accelerator entry points take HOST arrays on every call (critical_point_tracker_2d_regular.hh:369-384);
the patched tracker's resident mode pushes one fresh ndarray<double> per timestep (patches/ftk-xl-hip.patch, hip_push_snapshot)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _sweep(gpu, dims, slices, *, scalar, poison):
    """slices: host arrays (numpy) or device tensors, one per timestep -> (records, factors, (staged, direct))"""
    nd = len(dims)
    lo = 2 if scalar else 1
    dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
    ctx = gpu.Context(nd)
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=scalar, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
    push = ctx.push_scalar_slice if scalar else ctx.push_slice
    if poison:
        # the device arrays the real slices will land in have held something else first: a piece that does not arrive leaves THAT behind
        for t in range(len(slices)):
            push(t, np.ascontiguousarray(-3.0 * slices[t] + 0.125))
        for t in range(len(slices)):
            ctx.drop_slice(t)
    for t, a in enumerate(slices):
        push(t, a)
    nt = len(slices)
    ts = list(range(nt))
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in ts]
    recs, f, run = ctx.sweep_series(ts, scopes)
    counts = ctx.upload_counts()
    ctx.close()
    return recs, f, run, counts


@pytest.mark.parametrize("case,dims,scalar", [("woven", (2500, 2100), True), ("double_gyre", (2200, 1000), False), ("woven", (6000, 5600), True)],
                         ids=["scalar_2d", "vector_2d", "scalar_2d_256MiB"])
def test_staged_upload_equals_the_runtimes_copy_equals_device_fed(gpu, monkeypatch, case, dims, scalar):
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    nt = 3
    dev_slices = [synthetic.generate(case, dims, t, nt, torch, dev) for t in range(nt)]
    torch.cuda.synchronize()
    host = [np.ascontiguousarray(a.cpu().numpy()) for a in dev_slices]
    assert host[0].nbytes >= 32 << 20 and host[0].nbytes % (8 << 20) != 0 and host[0].nbytes % (4 << 20) != 0   # (staged, and the last piece ragged; 256 MiB and more: 8 MiB pieces)
    want, fw, rw, cw = _sweep(gpu, dims, dev_slices, scalar=scalar, poison=False)
    assert cw == (0, 0) and len(want) > 500
    monkeypatch.setenv("FTKX_UPLOAD_THREADS", "0")
    direct, fd, rd, cd = _sweep(gpu, dims, host, scalar=scalar, poison=True)
    assert cd[0] == 0 and cd[1] == 2 * nt
    for threads in ("4", "2", "7"):
        monkeypatch.setenv("FTKX_UPLOAD_THREADS", threads)
        staged, fs, rs, cs = _sweep(gpu, dims, host, scalar=scalar, poison=True)
        assert cs == (2 * nt, 0), cs
        assert staged.tobytes() == want.tobytes() and np.array_equal(fs, fw) and rs == rw
    assert direct.tobytes() == want.tobytes() and np.array_equal(fd, fw) and rd == rw


def test_small_and_pinned_sources_take_the_runtimes_copy(gpu, monkeypatch):
    import torch
    from ftk_amd import synthetic
    monkeypatch.delenv("FTKX_UPLOAD_THREADS", raising=False)
    dev = torch.device("cuda", 0)
    # small: 1 MiB slices
    dims = (512, 256)
    small = [synthetic.generate("woven", dims, t, 2, torch, dev).cpu().numpy() for t in range(2)]
    *_, counts = _sweep(gpu, dims, small, scalar=True, poison=False)
    assert counts == (0, 2)
    # pinned: the runtime reads it by DMA as it is
    dims = (2500, 2100)
    a = synthetic.generate("woven", dims, 0, 2, torch, dev).cpu().pin_memory()
    nd = 2
    dom = ([2] * nd, [d - 3 for d in dims])
    ctx = gpu.Context(nd)
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
    ctx.push_scalar_slice(0, a.numpy())
    assert ctx.upload_counts() == (0, 1)
    b = np.ascontiguousarray(a.numpy().copy())          # ... and its pageable copy is staged
    ctx.push_scalar_slice(1, b)
    assert ctx.upload_counts() == (1, 1)
    recs, _, _ = ctx.sweep_series([0, 1], [gpu.SCOPE_BOTH, gpu.SCOPE_ORDINAL])
    ctx.close()
    assert len(recs) > 100


def test_two_contexts_pushing_at_once_share_the_staging_without_mixing_anything(gpu, monkeypatch):
    """the pinned rings are the process's: of two contexts that push at the same moment one is staged, the other takes the runtime's copy
    (upload.cpp: try_lock) -- both must end up with their own data"""
    import threading
    import torch
    from ftk_amd import synthetic
    monkeypatch.delenv("FTKX_UPLOAD_THREADS", raising=False)
    dev = torch.device("cuda", 0)
    dims, nt = (2500, 2100), 4
    series = {}
    for name, case in (("a", "woven"), ("b", "moving_extremum_2d")):
        d = [synthetic.generate(case, dims, t, nt, torch, dev) for t in range(nt)]
        torch.cuda.synchronize()
        series[name] = (d, [np.ascontiguousarray(x.cpu().numpy()) for x in d])
    want = {k: _sweep(gpu, dims, v[0], scalar=True, poison=False) for k, v in series.items()}
    got, errors = {}, []
    go = threading.Barrier(2)

    def work(name):
        try:
            go.wait()
            got[name] = _sweep(gpu, dims, series[name][1], scalar=True, poison=True)
        except Exception as e:      # noqa: BLE001
            errors.append((name, repr(e)))

    for rep in range(3):
        got.clear()
        th = [threading.Thread(target=work, args=(n,)) for n in series]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for n in series:
            recs, f, run, counts = got[n]
            assert recs.tobytes() == want[n][0].tobytes() and np.array_equal(f, want[n][1]) and run == want[n][2], n
            assert sum(counts) == 2 * nt
    # (over three rounds of eight pushes each both ways were taken, with overwhelming likelihood; not asserted: a race is not a guarantee)
