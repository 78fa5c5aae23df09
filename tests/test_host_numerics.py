"""CPU checks of the PRODUCT's per-simplex arithmetic (ftk_amd/csrc/cp_device.hpp, fan_tables.hpp), compiled for the host
by tests/hostcheck, against the oracle.  The same headers are what the HIP kernels execute on the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HERE, "hostcheck", "libhostcheck.so")
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    L = C.CDLL(so)
    L.hc_quantize.restype = C.c_longlong
    L.hc_quantize.argtypes = [C.c_double, C.c_double]
    return L


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _oracle_batch(oracle, fn, X, ids, nd):
    L = oracle.lib()
    f = getattr(L, fn)
    out = np.zeros(len(X), dtype=np.int32)
    step = (nd + 1) * nd
    for i in range(len(X)):
        out[i] = f(C.c_void_p(X.ctypes.data + 8 * step * i), C.c_void_p(ids.ctypes.data + 4 * (nd + 1) * i))
    return out


def _cases(rng, n, nd, mag):
    X = rng.integers(-mag, mag + 1, size=(n, nd + 1, nd), dtype=np.int64)
    ids = np.stack([rng.permutation(1000)[: nd + 1] for _ in range(n)]).astype(np.int32)
    return np.ascontiguousarray(X), np.ascontiguousarray(ids)


@pytest.mark.parametrize("nd", [2, 3])
@pytest.mark.parametrize("mag", [1, 2, 5, 1000, 2 ** 20, 2 ** 28, 2 ** 31 - 1, 2 ** 31, 2 ** 40, 2 ** 62])
def test_integer_predicate_matches_oracle(hc, oracle, nd, mag):
    """mag 1..5: almost every simplex is degenerate (SoS cascade); 2**31 and up: determinants wrap in int64."""
    rng = np.random.default_rng(nd * 1000 + int(np.log2(mag)))
    n = 4000
    X, ids = _cases(rng, n, nd, mag)
    if mag > 2 ** 40:   # also force the exact values 0 and INT64_MIN into play
        X[::7, 0, 0] = 0
        X[::11, 1, 1] = np.iinfo(np.int64).min
    fast = np.zeros(n, dtype=np.int32); sos = np.zeros(n, dtype=np.int32)
    getattr(hc, f"hc_batch_in_simplex{nd}")(n, _ptr(X), _ptr(ids), _ptr(fast), _ptr(sos))
    ref = _oracle_batch(oracle, f"ftko_hook_cp_in_simplex{nd}", X, ids, nd)
    assert np.array_equal(sos, ref), "literal cascade differs from the oracle"
    assert np.array_equal(fast, ref), "cofactor fast path differs from the oracle"
    tri = np.zeros(n, dtype=np.int32)
    hc.hc_batch_in_simplex_try(nd, n, _ptr(X), _ptr(ids), 0, _ptr(tri))
    assert np.array_equal(tri, ref), "id-free fast path + cascade differs from the oracle"
    hc.hc_batch_in_simplex_resolved(nd, n, _ptr(X), _ptr(ids), _ptr(tri))
    assert np.array_equal(tri, ref), "the cascade for the degenerate determinants only differs from the oracle"
    if mag < 2 ** 31:
        hc.hc_batch_in_simplex_try(nd, n, _ptr(X), _ptr(ids), 1, _ptr(tri))
        assert np.array_equal(tri, ref), "id-free 32-bit-operand fast path + cascade differs from the oracle"
    if mag < 2 ** 31:          # every component fits in 32 bits: the cheaper-multiply form of the fast path (what the kernels take then)
        narrow = np.zeros(n, dtype=np.int32)
        getattr(hc, f"hc_batch_in_simplex{nd}_s32")(n, _ptr(X), _ptr(ids), _ptr(narrow))
        assert np.array_equal(narrow, ref), "32-bit-operand fast path differs from the oracle"
    if mag <= 1000:
        assert ref.any() and not ref.all()


def test_predicate_with_truncated_and_negative_ids(hc, oracle):
    rng = np.random.default_rng(7)
    n = 3000
    for nd in (2, 3):
        X, _ = _cases(rng, n, nd, 3)
        ids = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(n, nd + 1)).astype(np.int32)   # wrapped vertex ids
        fast = np.zeros(n, dtype=np.int32); sos = np.zeros(n, dtype=np.int32)
        getattr(hc, f"hc_batch_in_simplex{nd}")(n, _ptr(X), _ptr(ids), _ptr(fast), _ptr(sos))
        ref = _oracle_batch(oracle, f"ftko_hook_cp_in_simplex{nd}", X, ids, nd)
        assert np.array_equal(fast, ref) and np.array_equal(sos, ref)
        res = np.zeros(n, dtype=np.int32)
        hc.hc_batch_in_simplex_resolved(nd, n, _ptr(X), _ptr(ids), _ptr(res))
        assert np.array_equal(res, ref)


def test_fp64_solvers_bit_identical(hc, oracle):
    rng = np.random.default_rng(11)
    L = oracle.lib()
    for _ in range(3000):
        V2 = rng.standard_normal((3, 2)); V3 = rng.standard_normal((4, 3))
        if rng.random() < 0.2:
            V2[2] = V2[1]; V3[3] = V3[0]          # singular systems -> Inf/NaN -> clamp rule
        a = np.zeros(3); b = np.zeros(3)
        ra = hc.hc_solve2(_ptr(V2), _ptr(a)); rb = L.ftko_hook_inverse_lerp2(_ptr(V2), _ptr(b))
        assert bool(ra) == bool(rb) and a.tobytes() == b.tobytes()
        hc.hc_clamp3(_ptr(a)); L.ftko_hook_clamp(3, _ptr(b))
        assert a.tobytes() == b.tobytes()
        a = np.zeros(4); b = np.zeros(4)
        ra = hc.hc_solve3(_ptr(V3), _ptr(a)); rb = L.ftko_hook_inverse_lerp3(_ptr(V3), _ptr(b))
        assert bool(ra) == bool(rb) and a.tobytes() == b.tobytes()
        hc.hc_clamp4(_ptr(a)); L.ftko_hook_clamp(4, _ptr(b))
        assert a.tobytes() == b.tobytes()


def test_classification_matches_oracle(hc, oracle):
    rng = np.random.default_rng(13)
    L = oracle.lib()
    for i in range(4000):
        J2 = rng.standard_normal((2, 2)); J3 = rng.standard_normal((3, 3))
        if i % 3 == 0:
            J2 = (J2 + J2.T) / 2; J3 = (J3 + J3.T) / 2
        if i % 17 == 0:
            J2[:] = 0; J3[:] = 0
        if i % 19 == 0:
            J2[0, 0] = np.nan; J3[1, 1] = np.nan
        for sym in (0, 1):
            assert hc.hc_classify2(_ptr(J2), sym) == L.ftko_hook_type2(_ptr(J2), sym)
            assert hc.hc_classify3(_ptr(J3), sym) == L.ftko_hook_type3(_ptr(J3), sym)


def test_quantize(hc, oracle):
    L = oracle.lib()
    L.ftko_hook_quantize.restype = C.c_longlong
    L.ftko_hook_quantize.argtypes = [C.c_double, C.c_ulonglong]
    for v in (0.0, -0.0, 0.999, -0.999, 1.5, -1.5, 1e-9, 3e4, -7.3e12, 1e300, -1e300, 2.0 ** 42, -(2.0 ** 42), 4.4e12):
        for nbits in (8, 13, 21):
            assert hc.hc_quantize(v, float(1 << nbits)) == L.ftko_hook_quantize(v, 1 << nbits)


def test_compile_time_fan_equals_oracle_fan(hc, oracle):
    for n, nt, no in ((3, 12, 2), (4, 60, 6)):
        verts = np.zeros((nt, n, n), dtype=np.int32); ordinal = np.zeros(nt, dtype=np.int32)
        ot = np.zeros(no, dtype=np.int32); it = np.zeros(nt - no, dtype=np.int32)
        assert hc.hc_fan(n, _ptr(verts), _ptr(ordinal), _ptr(ot), _ptr(it)) == nt
        ov, oo = oracle.unit_simplices(n)
        assert np.array_equal(verts, ov) and np.array_equal(ordinal.astype(bool), oo)
        assert ot.tolist() == [i for i in range(nt) if oo[i]] and it.tolist() == [i for i in range(nt) if not oo[i]]
