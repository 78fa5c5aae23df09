import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


def pytest_collection_finish(session):
    """the full-size oracle comparisons (tests/test_gpu_zz_fullsize_oracle.py) compute their oracle in the background from here on"""
    if any("test_gpu_zz_fullsize_oracle" in item.nodeid for item in session.items):
        try:
            import test_gpu_zz_fullsize_oracle as Z
            Z.start_background()
        except Exception:   # noqa: BLE001
            pass


def pytest_sessionfinish(session, exitstatus):
    """after a GPU session: which mask kernel took how many launches in this process (ftkx_debug_mask_kernel_launches) -> gpurun_out/"""
    try:
        if "ftk_amd" not in sys.modules:
            return
        import ctypes as C
        import json
        from ftk_amd import _lib
        L = _lib.load()
        n = 7
        cnt = (C.c_ulonglong * n)(); names = (C.c_char_p * n)()
        L.ftkx_debug_mask_kernel_launches(cnt, names, n)
        if not any(cnt):
            return
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r06_mask_kernel_coverage.json"), "w") as f:
            json.dump({"note": "launches per mask kernel in the pytest process (subprocess tests not counted)", "tests_run": session.testscollected,
                       "launches": {names[i].decode(): int(cnt[i]) for i in range(n)}}, f, indent=1)
    except Exception:   # noqa: BLE001
        pass
