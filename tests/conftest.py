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
