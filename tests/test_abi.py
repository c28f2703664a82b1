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


def test_tuning_registry_round_trip(monkeypatch):
    """include/dlwpmi.h dlwp_set_tuning / _clear_ / _get_ / _list: the one documented registry behind every dispatch override
    (host logic only: no GPU call).  An override set through the API wins over the environment variable of the same name; without
    one the environment is consulted at the time of the call; unknown names are refused."""
    import ctypes as C
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    knobs = L.tuning_knobs()
    assert len(knobs) >= 25 and "GEMM_NOGLDS" in knobs and "WGRAD_MULTI_WGS" in knobs and all(knobs.values())
    v = C.c_int(-7)
    monkeypatch.delenv("DLWP_GEMM_GLDS_KD", raising=False)
    assert lib.dlwp_clear_tuning(b"GEMM_GLDS_KD") == 0
    assert lib.dlwp_get_tuning(b"GEMM_GLDS_KD", C.byref(v)) == 0                      # unset: the library's own choice
    monkeypatch.setenv("DLWP_GEMM_GLDS_KD", "32")
    assert lib.dlwp_get_tuning(b"GEMM_GLDS_KD", C.byref(v)) == 1 and v.value == 32   # environment, read at the time of the call
    L.set_tuning("DLWP_GEMM_GLDS_KD", 64)                                             # the prefix is optional
    assert lib.dlwp_get_tuning(b"GEMM_GLDS_KD", C.byref(v)) == 1 and v.value == 64   # the API override wins
    L.set_tuning("GEMM_GLDS_KD", None)
    assert lib.dlwp_get_tuning(b"GEMM_GLDS_KD", C.byref(v)) == 1 and v.value == 32
    monkeypatch.setenv("DLWP_GEMM_TRACE", "yes")                                      # a non-numeric value means "on"
    assert lib.dlwp_get_tuning(b"GEMM_TRACE", C.byref(v)) == 1 and v.value == 1
    assert lib.dlwp_set_tuning(b"NO_SUCH_KNOB", 1) != 0 and b"unknown knob" in lib.dlwp_last_error()
    assert lib.dlwp_clear_tuning(None) == 0
