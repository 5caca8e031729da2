"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/rfi_oracle.c).

The classes mirror the call signatures of the reference's host classes
(reference: src/katsdpsigproc/rfi/host.py:118-273) so that parity tests read like
the reference's own tests. Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``katsdpsigproc_amd`` never does.

Parity status: PINNED. ``tests/test_oracle_golden.py`` checks these functions against
golden vectors generated from the real reference (``tests/golden/make_golden.py``).

The transpose / percentile5 / maskedsum primitives have no host class in the
reference; their oracle is plain NumPy exactly as the reference's tests state it
(test/test_transpose.py:59, test/test_percentile.py:79-85, test/test_maskedsum.py:62-67).
"""

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librfi_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few hundred ms). Returns the library path."""
    src = os.path.join(_HERE, "rfi_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        L.oracle_abs_c64.argtypes = [vp, vp, i64]
        L.oracle_background_median_filter.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp]
        L.oracle_noise_est_mad_f64.argtypes = [vp, i32, i32, vp]
        L.oracle_noise_est_mad_f32.argtypes = [vp, i32, i32, vp]
        L.oracle_threshold_simple.argtypes = [vp, i32, vp, i32, i32, i32, f64, i32, vp]
        L.oracle_threshold_sum.argtypes = [vp, i32, vp, i32, i32, i32, f64, i32, f64, i32, vp]
        L.oracle_flagger.argtypes = [
            vp, i32, vp, i32, i32, i32, i32, i32, f64, i32, f64, i32, vp, vp, vp
        ]  # fmt: skip
        L.oracle_percentile5_f32.argtypes = [vp, i32, i32, i32, i32, vp]
        L.oracle_max_threads.restype = i32
        L.oracle_set_threads.argtypes = [i32]
        for name in (
            "oracle_abs_c64",
            "oracle_background_median_filter",
            "oracle_noise_est_mad_f64",
            "oracle_noise_est_mad_f32",
            "oracle_threshold_simple",
            "oracle_threshold_sum",
            "oracle_flagger",
            "oracle_percentile5_f32",
            "oracle_set_threads",
        ):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def set_threads(n: int) -> None:
    lib().oracle_set_threads(int(n))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def abs_c64(vis: np.ndarray) -> np.ndarray:
    """numpy-compatible |z| for complex64 (see rfi_oracle.c:abs_c64)."""
    vis = np.ascontiguousarray(vis, dtype=np.complex64)
    out = np.empty(vis.shape, np.float32)
    lib().oracle_abs_c64(_ptr(vis), _ptr(out), vis.size)
    return out


def _flags_args(flags: Optional[np.ndarray], shape):
    if flags is None:
        return None, 0
    flags = np.ascontiguousarray(flags)
    if flags.dtype != np.uint8:
        flags = (flags != 0).astype(np.uint8)
    if flags.ndim == 1:
        assert flags.shape == (shape[0],)
        return flags, 1
    assert flags.shape == tuple(shape)
    return flags, 2


class BackgroundMedianFilterHost:
    """Oracle for host.BackgroundMedianFilterHost (host.py:118-151)."""

    def __init__(self, width: int, amplitudes: bool = False) -> None:
        if width % 2 != 1 or width > 255:
            raise ValueError("width must be odd and <= 255")
        self.width = width
        self.amplitudes = amplitudes

    def __call__(self, vis: np.ndarray, flags: Optional[np.ndarray] = None) -> np.ndarray:
        dtype = np.float32 if self.amplitudes else np.complex64
        vis = np.ascontiguousarray(vis, dtype=dtype)
        channels, baselines = vis.shape
        fl, mode = _flags_args(flags, vis.shape)
        out = np.empty(vis.shape, np.float64)
        lib().oracle_background_median_filter(
            _ptr(vis), int(self.amplitudes), _ptr(fl), mode, self.width, channels, baselines,
            _ptr(out)
        )  # fmt: skip
        return out


class NoiseEstMADHost:
    """Oracle for host.NoiseEstMADHost (host.py:154-163); dtype-aware like numpy."""

    def __call__(self, deviations: np.ndarray) -> np.ndarray:
        channels, baselines = deviations.shape
        out = np.empty(baselines, np.float64)
        if deviations.dtype == np.float32:
            d = np.ascontiguousarray(deviations)
            lib().oracle_noise_est_mad_f32(_ptr(d), channels, baselines, _ptr(out))
        else:
            d = np.ascontiguousarray(deviations, dtype=np.float64)
            lib().oracle_noise_est_mad_f64(_ptr(d), channels, baselines, _ptr(out))
        return out


def _dev_noise(deviations: np.ndarray, noise: np.ndarray):
    if deviations.dtype == np.float32:
        d, d32 = np.ascontiguousarray(deviations), 1
    else:
        d, d32 = np.ascontiguousarray(deviations, dtype=np.float64), 0
    if noise.dtype == np.float32:
        n, n32 = np.ascontiguousarray(noise), 1
    else:
        n, n32 = np.ascontiguousarray(noise, dtype=np.float64), 0
    assert n.shape == (d.shape[1],)
    return d, d32, n, n32


class ThresholdSimpleHost:
    """Oracle for host.ThresholdSimpleHost (host.py:166-183)."""

    def __init__(self, n_sigma: float, flag_value: int = 1) -> None:
        self.n_sigma = n_sigma
        self.flag_value = flag_value

    def __call__(self, deviations: np.ndarray, noise: np.ndarray) -> np.ndarray:
        d, d32, n, n32 = _dev_noise(deviations, noise)
        flags = np.empty(d.shape, np.uint8)
        lib().oracle_threshold_simple(
            _ptr(d), d32, _ptr(n), n32, d.shape[0], d.shape[1], float(self.n_sigma),
            int(self.flag_value), _ptr(flags)
        )  # fmt: skip
        return flags


class ThresholdSumHost:
    """Oracle for host.ThresholdSumHost (host.py:186-254)."""

    def __init__(
        self,
        n_sigma: float,
        n_windows: int = 4,
        threshold_falloff: float = 1.2,
        flag_value: int = 1,
    ) -> None:
        self.n_sigma = n_sigma
        self.n_windows = n_windows
        self.threshold_falloff = threshold_falloff
        self.flag_value = flag_value

    def __call__(self, deviations: np.ndarray, noise: np.ndarray) -> np.ndarray:
        d, d32, n, n32 = _dev_noise(deviations, noise)
        flags = np.empty(d.shape, np.uint8)
        lib().oracle_threshold_sum(
            _ptr(d), d32, _ptr(n), n32, d.shape[0], d.shape[1], float(self.n_sigma),
            int(self.n_windows), float(self.threshold_falloff), int(self.flag_value), _ptr(flags)
        )  # fmt: skip
        return flags


class FlaggerHost:
    """Oracle for host.FlaggerHost (host.py:257-273): chains the three stages."""

    def __init__(self, background, noise_est, threshold) -> None:
        self.background = background
        self.noise_est = noise_est
        self.threshold = threshold

    def __call__(self, vis: np.ndarray, input_flags: Optional[np.ndarray] = None) -> np.ndarray:
        deviations = self.background(vis, input_flags)
        noise = self.noise_est(deviations)
        return self.threshold(deviations, noise)


def flagger_full(
    vis: np.ndarray,
    input_flags: Optional[np.ndarray] = None,
    *,
    width: int = 13,
    amplitudes: bool = False,
    threshold: str = "sum",
    n_sigma: float = 11.0,
    n_windows: int = 4,
    threshold_falloff: float = 1.2,
    flag_value: int = 1,
    want_deviations: bool = False,
):
    """One-call FlaggerHost (median filter + MAD + Sum/Simple threshold) in C.

    Returns (flags, noise[, deviations]); deviations/noise are float64 like the host's.
    """
    dtype = np.float32 if amplitudes else np.complex64
    vis = np.ascontiguousarray(vis, dtype=dtype)
    channels, baselines = vis.shape
    fl, mode = _flags_args(input_flags, vis.shape)
    flags = np.empty(vis.shape, np.uint8)
    noise = np.empty(baselines, np.float64)
    dev = np.empty(vis.shape, np.float64) if want_deviations else None
    lib().oracle_flagger(
        _ptr(vis), int(amplitudes), _ptr(fl), mode, width, channels, baselines,
        1 if threshold == "sum" else 0, float(n_sigma), int(n_windows), float(threshold_falloff),
        int(flag_value), _ptr(flags), _ptr(dev), _ptr(noise)
    )  # fmt: skip
    if want_deviations:
        return flags, noise, dev
    return flags, noise


def percentile5(src: np.ndarray, column_range=None) -> np.ndarray:
    """[min, max, 25 %, 75 %, 50 %] ("lower") of |src| per row -> float32 [5][rows].

    Restates the expectation in the reference's test (test/test_percentile.py:79-85).
    """
    if np.iscomplexobj(src):
        amp = abs_c64(src)
    else:
        amp = np.ascontiguousarray(src, dtype=np.float32)
    rows, cols = amp.shape
    if column_range is None:
        column_range = (0, cols)
    out = np.empty((5, rows), np.float32)
    lib().oracle_percentile5_f32(
        _ptr(amp), rows, cols, column_range[0], column_range[1] - column_range[0], _ptr(out)
    )
    return out


def maskedsum(src: np.ndarray, mask: np.ndarray, use_amplitudes: bool = False) -> np.ndarray:
    """Σ_row mask[row]·src[row, col] (test/test_maskedsum.py:62-67), float64 accumulation.

    The device sums sequentially in float32 with fma (maskedsum.mako:57-67); the
    reference test allows rtol=1e-6, so the oracle accumulates in float64.
    """
    if use_amplitudes:
        x = abs_c64(src).astype(np.float64)
        return np.sum(x * mask.astype(np.float64)[:, None], axis=0)
    x = src.astype(np.complex128)
    return np.sum(x * mask.astype(np.float64)[:, None], axis=0)


def generate_data(channels: int, baselines: int, seed: int = 1) -> np.ndarray:
    """Synthetic visibilities exactly as scripts/rfiflagtest.py:35-44 (seed 1)."""
    rs = np.random.RandomState(seed=seed)
    out = np.empty((channels, baselines), np.complex64)
    for i in range(channels):
        real = rs.standard_normal(size=baselines).astype(np.float32)
        imag = rs.standard_normal(size=baselines).astype(np.float32)
        out[i] = real + 1j * imag
    return out


def inject_rfi(vis: np.ndarray, seed: int = 3, fraction: float = 1.0 / 16.0):
    """Add spikes as test/rfi/test_flagger.py:42-50 does (amplitude U(50,70), random phase).

    Returns (vis_with_rfi complex64, spikes bool). Works in row blocks to bound memory.
    """
    rs = np.random.RandomState(seed=seed)
    out = np.array(vis, dtype=np.complex64, copy=True)
    spikes = np.zeros(vis.shape, np.bool_)
    for i in range(vis.shape[0]):
        s = rs.random_sample(vis.shape[1]) < fraction
        amp = rs.random_sample(vis.shape[1]) * 20.0 + 50.0
        phase = rs.random_sample(vis.shape[1]) * (2j * np.pi)
        out[i] = (out[i].astype(np.complex128) + s * (amp * np.exp(phase))).astype(np.complex64)
        spikes[i] = s
    return out, spikes
