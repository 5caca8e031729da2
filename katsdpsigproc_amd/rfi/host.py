"""Host-side (NumPy) interface of the RFI flaggers.

The abstract classes are the call signatures that the ``*HostFromDevice`` adapters in
:mod:`katsdpsigproc_amd.rfi.device` present, and the concrete classes give the package
the same host API as the reference (reference: src/katsdpsigproc/rfi/host.py:28-273).
They are whole-array NumPy formulations (no pandas, no per-baseline Python loops) with
the same numerics, dtype for dtype, as the reference classes; ``tests/test_host.py``
checks them against golden vectors produced by the reference.

Nothing in the device path calls this module: device operations run HIP kernels only
and raise if the native library is missing.
"""

import warnings
from abc import ABC, abstractmethod
from typing import Optional

import numpy as np

from . import MAD_NORMAL


class AbstractBackgroundHost(ABC):
    @abstractmethod
    def __init__(self, width: int, amplitudes: bool = False) -> None: ...

    @abstractmethod
    def __call__(self, vis: np.ndarray, flags: Optional[np.ndarray] = None) -> np.ndarray:
        """Deviation of each amplitude from a smooth background.

        `vis` is channels x baselines (complex, or amplitudes if constructed with
        ``amplitudes=True``); `flags` (optional, per channel or full shape) marks samples
        that must not influence the background. Returns float deviations, 0 where
        flagged.
        """


class AbstractNoiseEstHost(ABC):
    @abstractmethod
    def __call__(self, deviations: np.ndarray) -> np.ndarray:
        """Per-baseline noise (standard deviation) estimate from channels x baselines deviations."""


class AbstractThresholdHost(ABC):
    @abstractmethod
    def __init__(self, n_sigma: float) -> None: ...

    @abstractmethod
    def __call__(self, deviations: np.ndarray, noise: np.ndarray) -> np.ndarray:
        """uint8 flags (flag value or 0) with the shape of `deviations`."""


class AbstractFlaggerHost(ABC):
    @abstractmethod
    def __call__(self, vis: np.ndarray, input_flags: Optional[np.ndarray] = None) -> np.ndarray:
        """uint8 flags for channels x baselines visibilities.

        `input_flags` only steer the background; they are not copied to the output and a
        sample flagged on input is never flagged on output.
        """


class BackgroundMedianFilterHost(AbstractBackgroundHost):
    """Amplitude minus its centred sliding median along channels.

    The window is clipped at the band edges and skips flagged samples; an even number of
    valid samples gives the mean of the middle two. Amplitudes are float32, the median
    and the result float64 (as the reference, rfi/host.py:133-151).
    """

    #: baselines processed per block, to bound the size of the window tensor
    _BLOCK = 256

    def __init__(self, width: int, amplitudes: bool = False) -> None:
        if width % 2 != 1:
            raise ValueError("width must be odd")
        self.width = width
        self.amplitudes = amplitudes

    def __call__(self, vis: np.ndarray, flags: Optional[np.ndarray] = None) -> np.ndarray:
        vis = np.asarray(vis)
        amp = vis if self.amplitudes else np.abs(vis)
        amp = amp.astype(np.float64)  # exact; the median is taken in float64
        channels, baselines = amp.shape
        if flags is not None:
            mask = np.asarray(flags).astype(np.bool_)
            if mask.ndim == 1:
                mask = mask[:, np.newaxis]
            amp = np.where(np.broadcast_to(mask, amp.shape), np.nan, amp)
        half = self.width // 2
        out = np.empty((channels, baselines), np.float64)
        pad = np.full((half, 1), np.nan)
        for start in range(0, baselines, self._BLOCK):
            block = amp[:, start : start + self._BLOCK]
            padded = np.concatenate(
                [np.broadcast_to(pad, (half, block.shape[1])), block,
                 np.broadcast_to(pad, (half, block.shape[1]))]
            )  # fmt: skip
            windows = np.lib.stride_tricks.sliding_window_view(padded, self.width, axis=0)
            ordered = np.sort(windows, axis=-1)  # NaN sorts last
            count = np.sum(~np.isnan(windows), axis=-1)
            lo = np.take_along_axis(ordered, np.maximum(count - 1, 0)[..., None] // 2, -1)[..., 0]
            hi = np.take_along_axis(ordered, (count // 2)[..., None], -1)[..., 0]
            median = (lo + hi) / 2.0
            dev = block - median
            out[:, start : start + self._BLOCK] = np.where(np.isnan(dev), 0.0, dev)
        return out


class NoiseEstMADHost(AbstractNoiseEstHost):
    """``1.4826 * median(|d| : d != 0)`` per baseline.

    The median keeps the dtype of `deviations` (float32 in, float32 median -- even counts
    average in float32), the scale is applied in float64 (reference rfi/host.py:157-163).
    A baseline with no non-zero deviation gives NaN.
    """

    def __call__(self, deviations: np.ndarray) -> np.ndarray:
        mag = np.abs(np.asarray(deviations))
        ordered = np.sort(np.where(mag > 0, mag, np.inf), axis=0)  # zeros pushed to the end
        count = np.sum(mag > 0, axis=0)
        cols = np.arange(mag.shape[1])
        lo = ordered[np.maximum(count - 1, 0) // 2, cols]
        hi = ordered[np.minimum(count // 2, mag.shape[0] - 1), cols]
        with np.errstate(invalid="ignore"):
            median = (lo + hi) / mag.dtype.type(2)
        median = np.where(count > 0, median, np.nan)
        if np.any(count == 0):
            warnings.warn("baseline with no non-zero deviations", RuntimeWarning)
        return median.astype(np.float64) * MAD_NORMAL


class ThresholdSimpleHost(AbstractThresholdHost):
    """Flag samples whose deviation exceeds ``n_sigma * noise`` of their baseline."""

    def __init__(self, n_sigma: float, flag_value: int = 1) -> None:
        self.n_sigma = n_sigma
        self.flag_value = flag_value

    def __call__(self, deviations: np.ndarray, noise: np.ndarray) -> np.ndarray:
        limit = self.n_sigma * np.asarray(noise)  # keeps noise's dtype (NEP 50)
        return (np.asarray(deviations) > limit).astype(np.uint8) * np.uint8(self.flag_value)


class ThresholdSumHost(AbstractThresholdHost):
    """Offringa SumThreshold along channels with windows 1, 2, 4, ... (rfi/host.py:186-254).

    For window ``w = 2**k`` the per-sample threshold is
    ``float32(n_sigma * noise * falloff**-k)``; samples flagged by earlier windows are
    replaced by that threshold; every full window whose float64 sum exceeds
    ``float32(threshold * w)`` flags all its samples. All baselines are processed
    together.
    """

    def __init__(self, n_sigma: float, n_windows: int = 4, threshold_falloff: float = 1.2,
                 flag_value: int = 1) -> None:  # fmt: skip
        self.n_sigma = n_sigma
        self.windows = [2**i for i in range(n_windows)]
        self.threshold_scales = [pow(threshold_falloff, -i) for i in range(n_windows)]
        self.flag_value = flag_value

    def __call__(self, deviations: np.ndarray, noise: np.ndarray) -> np.ndarray:
        work = np.array(deviations, dtype=np.float64)  # exact copy of float32 input
        channels = work.shape[0]
        threshold1 = self.n_sigma * np.asarray(noise)  # float32 stays float32
        flagged = np.zeros(work.shape, np.bool_)
        for window, scale in zip(self.windows, self.threshold_scales):
            threshold = (threshold1 * scale).astype(np.float32)
            work = np.where(flagged, threshold[np.newaxis, :], work)
            n_sums = channels - window + 1
            if n_sums <= 0:
                continue
            sums = work[:n_sums].copy()
            for offset in range(1, window):  # same left-to-right order as numpy.convolve
                sums += work[offset : offset + n_sums]
            with np.errstate(invalid="ignore"):
                hit = sums > (threshold * np.float32(window))[np.newaxis, :]
            for offset in range(window):
                flagged[offset : offset + n_sums] |= hit
        return flagged.astype(np.uint8) * np.uint8(self.flag_value)


class FlaggerHost(AbstractFlaggerHost):
    """background -> noise estimate -> threshold (reference rfi/host.py:257-273)."""

    def __init__(self, background: AbstractBackgroundHost, noise_est: AbstractNoiseEstHost,
                 threshold: AbstractThresholdHost) -> None:  # fmt: skip
        self.background = background
        self.noise_est = noise_est
        self.threshold = threshold

    def __call__(self, vis: np.ndarray, input_flags: Optional[np.ndarray] = None) -> np.ndarray:
        deviations = self.background(vis, input_flags)
        return self.threshold(deviations, self.noise_est(deviations))
