"""The split pass's policy (ftk_amd/csrc/split_policy.hpp) as a state machine driven WITHOUT a GPU: which passes are split under each
setting of FTKX_SERIES_HOOKS split=..., how "auto" measures itself (five passes in order, five split, samples only while the pipeline is
full), what it decides on which medians, how a bad verdict is re-examined, and what a change of the pass's shape does.  series.hip feeds
the same functions (split_decide when a pass is planned, split_sample when one completes); this is the unit test the round-5 review asked
for in place of end-to-end equality as the only safety net of that decision."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GB = 1000000000


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HERE, "hostcheck", "libhostcheck_split.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "hostcheck", "hostcheck.cpp")])
    L = C.CDLL(so)
    L.hc_split_new.restype = C.c_void_p
    L.hc_split_delete.argtypes = [C.c_void_p]
    L.hc_split_decide.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong]
    L.hc_split_sample.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    L.hc_split_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    return L


class Policy:
    def __init__(self, L):
        self.L, self.k = L, C.c_void_p(L.hc_split_new())
        self.clock, self.last_s, self.last_kind = 100.0, 0.0, 0

    def decide(self, mode=1, pipelined=True, dist=False, profiling_ok=True, sparse=True, ntodo=4, mask_bytes=2 * GB, sig=7):
        v = self.L.hc_split_decide(self.k, mode, pipelined, dist, profiling_ok, sparse, ntodo, mask_bytes, sig)
        return bool(v & 1), (v >> 1) & 3, v >> 3          # split, cal_kind, forced

    def complete(self, cal_kind, period, chained=True):
        """a pass completes `period` seconds after the one before it"""
        self.clock += period
        self.L.hc_split_sample(self.k, cal_kind, self.clock, self.last_s, self.last_kind, chained)
        self.last_kind, self.last_s = cal_kind, (self.clock if chained else 0.0)

    def state(self, forced=0):
        a, b, ph, cd = C.c_double(), C.c_double(), C.c_int(), C.c_uint()
        st = self.L.hc_split_state(self.k, forced, C.byref(a), C.byref(b), C.byref(ph), C.byref(cd))
        return st, a.value, b.value, ph.value, cd.value

    def run_until_decided(self, t_order, t_split, limit=64):
        kinds = []
        for _ in range(limit):
            split, kind, _f = self.decide()
            if self.state()[3] >= 2:
                break
            kinds.append((split, kind))
            self.complete(kind, t_split if split else t_order)
        return kinds


def test_forced_settings_and_the_size_rule(hc):
    P = Policy(hc)
    assert P.decide(mode=0) == (False, 0, 2)                                     # never
    assert P.decide(mode=4, mask_bytes=2 * GB) == (True, 0, 1)                   # on: the size rule alone, no measuring
    assert P.decide(mode=4, mask_bytes=GB // 2) == (False, 0, 1)                 # ... which a short mask launch does not meet
    assert P.decide(mode=2, mask_bytes=1000) == (True, 0, 1)                     # tests: whatever the size
    assert P.decide(mode=4, sparse=False, mask_bytes=3 * GB) == (False, 0, 1)    # hit-dense: 4 GB and more
    assert P.decide(mode=4, sparse=False, mask_bytes=4 * GB) == (True, 0, 1)
    for kw in (dict(pipelined=False), dict(dist=True), dict(profiling_ok=False), dict(ntodo=0)):
        assert P.decide(mode=4, **kw)[0] is False, kw                            # a pass on its own, a slab pass, kernel events in the way, no masks to build
    assert P.state(forced=1)[0] == 3 and P.state(forced=2)[0] == 4 and P.state()[0] == 0


def test_auto_measures_five_in_order_then_five_split_and_keeps_the_faster(hc):
    P = Policy(hc)
    kinds = P.run_until_decided(t_order=0.44e-3, t_split=0.40e-3)
    # in order first (2 discarded + the first one has no predecessor + 5 samples), then split (4 discarded + 5)
    first_split = next(i for i, (s, k) in enumerate(kinds) if s)
    assert all(k == 1 and not s for s, k in kinds[:first_split]) and all(k == 2 and s for s, k in kinds[first_split:])
    assert first_split == 1 + 2 + 5 and len(kinds) - first_split == 1 + 4 + 5
    st, mo, ms, phase, _cd = P.state()
    assert st == 1 and phase == 2 and abs(mo - 0.44e-3) < 1e-9 and abs(ms - 0.40e-3) < 1e-9
    assert P.decide() == (True, 0, 0)                                            # from now on: split, no more samples


def test_auto_goes_back_in_order_where_split_is_slower_and_asks_again_later(hc):
    P = Policy(hc)
    P.run_until_decided(t_order=0.44e-3, t_split=0.51e-3)
    st, mo, ms, phase, cd = P.state()
    assert st == 2 and ms > 1.02 * mo
    n = 4096 - cd                                                                # (the loop above has planned one pass under the verdict already)
    while P.state()[3] == 2:
        assert P.decide() == (False, 0, 0)
        n += 1
    assert n == 4096                                                             # the 4096th pass in order drops the verdict ...
    assert P.decide() == (False, 1, 0)                                           # ... and the next passes are measured afresh
    # within 2 %: kept
    Q = Policy(hc)
    Q.run_until_decided(t_order=0.400e-3, t_split=0.407e-3)
    assert Q.state()[0] == 1


def test_samples_count_only_while_the_pipeline_is_full_and_of_one_kind(hc):
    P = Policy(hc)
    for _ in range(30):
        split, kind, _f = P.decide()
        P.complete(kind, 0.44e-3, chained=False)                                 # every pass waited for on its own: no period to sample
    assert P.state()[3] == 0 and P.decide() == (False, 1, 0)
    # a change of the pass's shape starts over
    Q = Policy(hc)
    Q.run_until_decided(t_order=0.44e-3, t_split=0.40e-3)
    assert Q.state()[0] == 1
    assert Q.decide(sig=8) == (False, 1, 0) and Q.state()[3] == 0
