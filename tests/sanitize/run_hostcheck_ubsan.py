"""Runs the product's per-simplex arithmetic (tests/hostcheck, built with -fsanitize=undefined) over random inputs from magnitude 1
(all degenerate) to 2^62 (every determinant wraps), NaN / Inf / denormal doubles and extreme quantisation inputs.  The integer
predicates are written in the uint64 ring precisely so that wrapping is defined behaviour; UBSan halts on any signed overflow,
out-of-range double -> integer conversion or bad shift."""
import ctypes as C
import sys

import numpy as np

L = C.CDLL(sys.argv[1])
L.hc_quantize.restype = C.c_longlong
L.hc_quantize.argtypes = [C.c_double, C.c_double]
rng = np.random.default_rng(3)
for nd in (2, 3):
    fn = getattr(L, "hc_batch_in_simplex%d" % nd)
    for mag in (1, 2, 5, 1000, 2 ** 20, 2 ** 31, 2 ** 40, 2 ** 62, 2 ** 63 - 1):
        n = 20000
        X = np.ascontiguousarray(rng.integers(-mag, mag, size=(n, nd + 1, nd), dtype=np.int64, endpoint=True))
        X[::7, 0, 0] = np.iinfo(np.int64).min                      # the value whose negation wraps onto itself
        ids = np.ascontiguousarray(np.stack([rng.permutation(1000)[: nd + 1] for _ in range(n)]).astype(np.int32))
        ids[::11] -= 2 ** 31 - 500                                 # negative / truncated vertex ids
        a = np.zeros(n, dtype=np.int32); b = np.zeros(n, dtype=np.int32)
        fn(n, C.c_void_p(X.ctypes.data), C.c_void_p(ids.ctypes.data), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data))
        assert np.array_equal(a, b), (nd, mag)                     # fast cofactor path == literal SoS cascade
specials = [0.0, -0.0, 1.0, -1.0, 5e-324, 2.2e-308, 1e300, -1e300, 4.4e12, -4.4e12]      # finite: the reference casts these to int64
for v in specials:
    for f in (256.0, 2.0 ** 21):
        if abs(v * f) < 9.2e18:                                    # beyond int64 the reference's cast is itself undefined; the sweep rejects non-finite input
            L.hc_quantize(v, f)
for nd, solve, clamp in ((2, L.hc_solve2, L.hc_clamp3), (3, L.hc_solve3, L.hc_clamp4)):
    for scale in (1e-300, 1.0, 1e300):
        V = np.ascontiguousarray(rng.standard_normal((2000, nd + 1, nd)) * scale)
        V[::5, 0] = V[::5, 1]                                      # singular systems
        V[::9, 0, 0] = np.nan; V[::13, 1, 0] = np.inf
        mu = np.zeros(nd + 1)
        for i in range(len(V)):
            solve(C.c_void_p(V[i].ctypes.data), C.c_void_p(mu.ctypes.data)); clamp(C.c_void_p(mu.ctypes.data))
J = rng.standard_normal((4000, 9)); J[::3] *= 1e200; J[::7, 0] = np.nan
for i in range(len(J)):
    for sym in (0, 1):
        L.hc_classify2(C.c_void_p(J[i].ctypes.data), sym); L.hc_classify3(C.c_void_p(J[i].ctypes.data), sym)
print("ubsan run complete")
