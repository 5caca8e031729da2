"""The C-ABI library loads without a GPU and exports exactly what the header declares."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "katsdpsigproc_hip.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ksp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from katsdpsigproc_amd import _lib, build_native

    if not os.path.exists(_lib.LIB_PATH):
        build_native.build()
    return _lib.load()


def test_header_and_binding_agree():
    from katsdpsigproc_amd import _lib

    assert header_functions() == _lib.declared_symbols()


def test_every_declared_symbol_is_exported(lib):
    for name in header_functions():
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"


def test_abi_version_and_no_device_calls(lib):
    from katsdpsigproc_amd import _lib

    assert lib.ksp_abi_version() == _lib.ABI_VERSION == 5
    count = ctypes.c_int(-1)
    assert lib.ksp_device_count(ctypes.byref(count)) == 0
    assert count.value >= 0  # 0 in the CPU container
    assert lib.ksp_flagger_fused_supported(4096, 13, 4) == 1
    assert lib.ksp_flagger_fused_supported(8192, 13, 4) == 1
    assert lib.ksp_flagger_fused_supported(10240, 13, 4) == 1
    assert lib.ksp_flagger_fused_supported(12289, 13, 4) == 0
    assert lib.ksp_flagger_fused_supported(8192, 5, 4) == 0
    assert lib.ksp_flagger_fused_supported(4096, 5, 4) == 1
    assert lib.ksp_flagger_fused_supported(4096, 25, 4) == 1  # (round 3: widths up to 31)
    assert lib.ksp_flagger_fused_supported(4096, 33, 4) == 0
    assert lib.ksp_flagger_fused_supported(4096, 13, 8) == 1  # (round 3: up to 8 windows)
    assert lib.ksp_flagger_fused_supported(4096, 13, 9) == 0
    assert lib.ksp_flagger_fused_supported(8192, 13, 5) == 0
    assert lib.ksp_flagger_fused_last_path() == 0  # no launch yet on this thread
    before = lib.ksp_flagger_fused_ring_mode(1)
    assert before in (-1, 0, 1) and lib.ksp_flagger_fused_ring_mode(7) == 1  # (7: query only)
    assert lib.ksp_flagger_fused_ring_mode(before) == 1
    assert lib.ksp_flagger_fused_supported(4096, 12, 4) == 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No CPU fallback: a missing library is a RuntimeError, not a silent detour."""
    from katsdpsigproc_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback|not been built"):
        _lib.load(str(tmp_path / "nope.so"))


def test_argument_validation_without_gpu(lib):
    """Launchers validate shapes before touching the device."""
    from katsdpsigproc_amd import _lib

    rc = lib.ksp_madnz_t(0, None, None, None, 16, 4, 16)
    assert rc != 0 and "NULL" in _lib.last_error()
    rc = lib.ksp_threshold_sum(0, None, ctypes.c_void_p(8), ctypes.c_void_p(8),
                               ctypes.c_void_p(8), 16, 4, 16, 11.0,
                               (ctypes.c_float * 9)(), 9, 1, 0)  # fmt: skip
    assert rc != 0 and "n_windows" in _lib.last_error()
