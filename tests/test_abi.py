"""The C-ABI library loads on a CPU-only box and exports every symbol include/dlwpmi.h declares
(no compute calls here: those need a GPU and are in the gpu-marked tests)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dlwp_benchmark_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return L


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dlwpmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dlwp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    handle = lib.load()
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(handle, n), f"{n} declared in dlwpmi.h but not exported"
        assert n in lib.SIGNATURES, f"{n} has no ctypes signature in lib.py"
    for n in lib.SIGNATURES:
        assert n in names, f"{n} bound in lib.py but not declared in dlwpmi.h"


def test_version_and_error_string(lib):
    handle = lib.load()
    assert handle.dlwp_version() == 100
    # invalid arguments are rejected on the host before anything touches a GPU
    rc = handle.dlwp_pwmlp_fwd(None, None, None, None, None, None, 1, 1, 1, 1, 1, None)
    assert rc == -1
    assert b"NULL" in handle.dlwp_last_error()
    with pytest.raises(lib.DlwpError):
        lib.check(rc)


def test_flat_parameter_layout_is_consistent(lib):
    from dlwp_benchmark_amd.fno_engine import FnoParamLayout
    lay = FnoParamLayout(10, 32, 256, 256, 1, 4, [12, 12])
    assert lay.n_params() == 10 * 256 + 256 + 256 * 32 + 32 + 32 * 256 + 256 + 256 + 1 + 4 * (2 * 32 * 32 * 84 + 32 * 32 + 32)
    spans = sorted((o, o + n) for (o, n, _) in lay.entries.values())
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0           # no overlap
        assert b0 % 4 == 0        # 16-byte aligned tensors
    assert spans[-1][1] <= lay.total


def test_product_has_no_cpu_fallback(lib):
    import torch
    with pytest.raises(lib.DlwpError):
        lib.ptr(torch.zeros(4))  # CPU tensors are refused, never silently computed on
