"""ctypes binding of the C-ABI in ``include/katsdpsigproc_hip.h``.

This is the only place where Python meets native code. The shared library
(``_native/libkatsdpsigproc_hip.so``) is built in-tree by ``build_native.py``
(``__graft_entry__.build()`` calls it). There is deliberately NO CPU fallback: if
the library is missing or a call fails, a :class:`RuntimeError` is raised.
"""

import ctypes
import os
import sys
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t
from ctypes import c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_native", "libkatsdpsigproc_hip.so")
ABI_VERSION = 5

_lib = None


class DeviceProps(ctypes.Structure):
    """Mirror of ``ksp_device_props``."""

    _fields_ = [
        ("name", ctypes.c_char * 256),
        ("arch", ctypes.c_char * 64),
        ("compute_units", c_int32),
        ("wavefront_size", c_int32),
        ("max_threads_per_block", c_int32),
        ("lds_bytes_per_block", c_int32),
        ("clock_khz", c_int32),
        ("driver_version", c_int32),
        ("runtime_version", c_int32),
        ("total_memory", c_int64),
    ]


_SIZE3 = c_size_t * 3

# name -> argtypes; every function returns int (0 = success) unless listed in _OTHER
SIGNATURES = {
    "ksp_device_count": [POINTER(c_int)],
    "ksp_device_get_props": [c_int, POINTER(DeviceProps)],
    "ksp_malloc": [c_int, c_size_t, POINTER(c_void_p)],
    "ksp_free": [c_int, c_void_p],
    "ksp_host_alloc": [c_size_t, POINTER(c_void_p)],
    "ksp_host_free": [c_void_p],
    "ksp_stream_create": [c_int, POINTER(c_void_p)],
    "ksp_stream_destroy": [c_int, c_void_p],
    "ksp_stream_synchronize": [c_int, c_void_p],
    "ksp_event_create": [c_int, POINTER(c_void_p)],
    "ksp_event_create_ordering": [c_int, POINTER(c_void_p)],
    "ksp_event_destroy": [c_int, c_void_p],
    "ksp_event_record": [c_int, c_void_p, c_void_p],
    "ksp_event_synchronize": [c_int, c_void_p],
    "ksp_event_elapsed_ms": [c_int, c_void_p, c_void_p, POINTER(c_float)],
    "ksp_stream_wait_event": [c_int, c_void_p, c_void_p],
    "ksp_memcpy_async": [c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p],
    "ksp_memcpy_rect_async": [
        c_int, c_void_p, c_size_t, POINTER(c_size_t), c_void_p, c_size_t, POINTER(c_size_t),
        POINTER(c_size_t), c_int, c_int, c_void_p,
    ],
    "ksp_memset_async": [c_int, c_void_p, c_int, c_size_t, c_void_p],
    "ksp_transpose": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int],
    "ksp_percentile5_float": [
        c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int
    ],
    "ksp_maskedsum_float": [
        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int
    ],
    "ksp_flagger_fused_profile": [c_void_p, c_void_p],
    "ksp_selftest_sqrt12": [c_int, c_void_p, c_void_p, c_int],
    "ksp_selftest_abs": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "ksp_selftest_rank": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int],
    "ksp_selftest_minmax": [c_int, c_void_p, c_void_p, c_void_p, c_int],
    "ksp_selftest_median_non_zero": [c_int, c_void_p, c_void_p, c_void_p, c_int],
    "ksp_background_median_filter": [
        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
        c_int, c_int,
    ],
    "ksp_madnz_t": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int],
    "ksp_madnz": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int],
    "ksp_threshold_simple": [
        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_int
    ],
    "ksp_threshold_sum": [
        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
        POINTER(c_float), c_int, c_int, c_int,
    ],
    "ksp_flagger_fused": [
        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double, POINTER(c_double), c_int,
        c_int, c_void_p,
    ],
    "ksp_flagger_fused_supported": [c_int, c_int, c_int],
    "ksp_flagger_fused_last_path": [],
    "ksp_flagger_fused_ring_mode": [c_int],
    "ksp_rtc_compile": [
        c_int, c_char_p, POINTER(c_char_p), c_int, POINTER(c_void_p), c_char_p, c_size_t
    ],
    "ksp_module_get_function": [c_int, c_void_p, c_char_p, POINTER(c_void_p)],
    "ksp_module_unload": [c_int, c_void_p],
    "ksp_fft_plan_create": [
        c_int, c_int, POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong), ctypes.c_longlong,
        POINTER(ctypes.c_longlong), ctypes.c_longlong, c_int, ctypes.c_longlong,
        POINTER(c_void_p), POINTER(c_size_t),
    ],
    "ksp_fft_plan_destroy": [c_int, c_void_p],
    "ksp_fft_exec": [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int],
    "ksp_launch_function": [
        c_int, c_void_p, c_void_p, POINTER(ctypes.c_uint), POINTER(ctypes.c_uint), ctypes.c_uint,
        POINTER(c_void_p),
    ],
}  # fmt: skip

_OTHER = {
    "ksp_abi_version": ([], c_int),
    "ksp_last_error": ([], c_char_p),
}

#: functions whose int return value is a result, not an error code
_VALUE_RETURN = {"ksp_flagger_fused_supported", "ksp_flagger_fused_last_path",
                 "ksp_flagger_fused_ring_mode"}


def declared_symbols():
    """Every symbol ``include/katsdpsigproc_hip.h`` declares (used by the ABI test)."""
    return sorted(list(SIGNATURES) + list(_OTHER))


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load the native library and attach prototypes. Raises RuntimeError if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or "
            "`python -m katsdpsigproc_amd.build_native`). There is no CPU fallback."
        )
    # torch bundles its own libamdhip64 under a different file name; if torch is
    # going to live in this process it must be loaded first so that both bind to one
    # HIP runtime (same SONAME). See DESIGN.md "One HIP runtime per process".
    if "torch" not in sys.modules and os.environ.get("KATSDPSIGPROC_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing
            pass
    try:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:
        raise RuntimeError(f"cannot load {path}: {exc}") from exc
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (argtypes, restype) in _OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.ksp_abi_version() != ABI_VERSION:
        raise RuntimeError(
            f"ABI mismatch: library {lib.ksp_abi_version()}, python binding {ABI_VERSION}"
        )
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().ksp_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def call(name: str, *args) -> int:
    """Call a C-ABI function; raise RuntimeError on a non-zero error code."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if name in _VALUE_RETURN:
        return rc
    if rc != 0:
        raise RuntimeError(f"{name} failed with HIP error {rc}: {last_error()}")
    return rc


__all__ = [
    "ABI_VERSION", "DeviceProps", "LIB_PATH", "SIGNATURES", "byref", "call", "declared_symbols",
    "last_error", "load",
]  # fmt: skip
