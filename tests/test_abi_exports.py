"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports every entry point that
include/ftkx.h, include/ftkx_slab.h and include/ftkx_tracker.hh declare (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ftk_amd import build, _lib
    build.build()          # hipcc cross-compiles for gfx950 without a GPU
    return _lib.load()


def declared_symbols():
    names = set()
    for hdr in ("ftkx.h", "ftkx_slab.h", "ftkx_tracker.hh"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", "", txt)
        for m in re.finditer(r"\b(ftkx_\w+)\s*\(", txt):
            names.add(m.group(1))
    return sorted(names - {"ftkx_cp_aux", "ftkx_cp_ordinal", "ftkx_cp_timestep", "ftkx_error"})   # static inline helpers / C++ type


def test_library_exports_every_declared_symbol(lib):
    from ftk_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert set(_lib.EXPORTS) <= set(syms) | {"ftkx_free"}


def test_record_layout_is_feature_point_lite():
    from ftk_amd import CP_DTYPE
    assert CP_DTYPE.itemsize == 72                      # sizeof(ftk::feature_point_lite_t)
    assert CP_DTYPE.fields["t"][1] == 24 and CP_DTYPE.fields["scalar"][1] == 32
    assert CP_DTYPE.fields["type"][1] == 56 and CP_DTYPE.fields["aux"][1] == 60 and CP_DTYPE.fields["tag"][1] == 64


def test_scaling_factor_rule(lib, oracle):
    from ftk_amd import scaling_factor
    for res in (1.7976931348623157e308, 20.0, 0.25, 0.02, 1.0 / 256, 1.0 / 257, 3e-4, 1e-7, 1e-19, 5e-324):
        assert scaling_factor(res) == oracle.scaling_factor(res)


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ftk_amd
    with pytest.raises(ftk_amd.FtkxError):
        ftk_amd.Context(2)
    with pytest.raises(ftk_amd.FtkxError):
        ftk_amd.CriticalPointTracker3DRegular()


def test_product_does_not_reference_the_oracle():
    """the product path must never route through oracle/ (or any CPU fallback)"""
    for root, _, files in os.walk(os.path.join(ROOT, "ftk_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".hh")):
                txt = open(os.path.join(root, f)).read()
                assert "pyoracle" not in txt and "ftk_oracle" not in txt and "libftk_oracle" not in txt, f
