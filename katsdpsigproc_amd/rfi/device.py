"""RFI flagging on MI355X behind the reference's template/operation API.

Class names, constructor and ``instantiate`` signatures, slot names and error
behaviour follow the reference (reference: src/katsdpsigproc/rfi/device.py). What is
underneath is different: every operation launches an ahead-of-time compiled HIP kernel
for gfx950 through the C-ABI (``include/katsdpsigproc_hip.h``), and
:class:`FlaggerDeviceTemplate` can replace the five-kernel sequence by one fused
single-pass kernel (:class:`FusedFlaggerDevice`) that is bit-identical to
:class:`katsdpsigproc_amd.rfi.host.FlaggerHost`.

As in the reference, noise estimators and thresholders may work on transposed
(baseline-major) data, advertised by their ``transposed`` attribute; the flagger
inserts transposes where needed.
"""

import ctypes
import enum
from abc import ABC, abstractmethod
from typing import Any, List, Mapping, Optional, Tuple, Type, Union

import numpy as np

from .. import accel, transpose, tune
from ..abc import AbstractCommandQueue, AbstractContext
from ..accel import AbstractAllocator
from . import host

_THRESHOLD_SUM_DEFAULT_THRESHOLD_FALLOFF = 1.2


class BackgroundFlags(enum.Enum):
    """How input flags are supplied to a backgrounder (reference rfi/device.py:40-46)."""

    NONE = 0
    CHANNEL = 1
    FULL = 2

    def __bool__(self) -> bool:
        return self is not BackgroundFlags.NONE


# ----------------------------------------------------------------- abstract interfaces
class AbstractBackgroundDevice(accel.Operation):
    pass


class AbstractBackgroundDeviceTemplate(ABC):
    use_flags: BackgroundFlags
    context: AbstractContext
    host_class: Type[host.AbstractBackgroundHost]

    @abstractmethod
    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    allocator: Optional[AbstractAllocator] = None) -> AbstractBackgroundDevice:  # fmt: skip
        """Create an instance."""


class AbstractNoiseEstDevice(accel.Operation):
    transposed: bool


class AbstractNoiseEstDeviceTemplate(ABC):
    transposed: bool
    context: AbstractContext
    host_class: Type[host.AbstractNoiseEstHost]

    @abstractmethod
    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    allocator: Optional[AbstractAllocator] = None) -> AbstractNoiseEstDevice:  # fmt: skip
        """Create an instance."""


class AbstractThresholdDevice(accel.Operation):
    transposed: bool


class AbstractThresholdDeviceTemplate(ABC):
    transposed: bool
    context: AbstractContext
    host_class: Type[host.AbstractThresholdHost]

    @abstractmethod
    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    n_sigma: float, *,
                    allocator: Optional[AbstractAllocator] = None) -> AbstractThresholdDevice:  # fmt: skip
        """Create an instance (concrete classes may add parameters before `allocator`)."""


# ------------------------------------------------------------------------- background
class BackgroundHostFromDevice(host.AbstractBackgroundHost):
    """Present a device backgrounder template through the host call signature.

    Instantiates, uploads, runs and downloads on every call (reference
    rfi/device.py:113-138). ``TypeError`` if flags are given to a template built without
    them, or omitted from one built with them.
    """

    def __init__(self, template: AbstractBackgroundDeviceTemplate,
                 command_queue: AbstractCommandQueue) -> None:  # fmt: skip
        self.template = template
        self.command_queue = command_queue

    def __call__(self, vis: np.ndarray, flags: Optional[np.ndarray] = None) -> np.ndarray:
        if flags is not None and not self.template.use_flags:
            raise TypeError("flags were provided but not included in the template")
        if flags is None and self.template.use_flags:
            raise TypeError("flags were expected but not provided")
        channels, baselines = vis.shape
        fn = self.template.instantiate(self.command_queue, channels, baselines)
        fn.ensure_all_bound()
        fn.buffer("vis").set(self.command_queue, vis)
        if flags is not None:
            fn.buffer("flags").set(self.command_queue, flags)
        fn()
        return fn.buffer("deviations").get(self.command_queue)


class BackgroundMedianFilterDeviceTemplate(AbstractBackgroundDeviceTemplate):
    """Median-filter backgrounder (reference rfi/device.py:141-262).

    Unlike the reference kernel, which uses ``hypot`` and float32 throughout and so "may
    give slightly different results", this one reproduces the host class: numpy's
    complex64 ``abs``, float64 median and subtraction, result rounded once to float32.

    Parameters
    ----------
    context
        Context whose device will run the kernel
    width
        Window width in channels: odd, 3 to 31
    is_amplitude
        Inputs are float32 amplitudes rather than complex64 visibilities
    use_flags
        :class:`BackgroundFlags`, or ``True``/``False`` for CHANNEL/NONE
    tuning
        ``csplit``: into how many channel segments a baseline is cut (each walked by its
        own wavefront, re-reading ``width - 1`` channels of halo); 0 lets the launcher
        choose. ``wgs`` of the reference has no counterpart: a wavefront always covers
        64 adjacent baselines (512-byte coalesced rows). Default: autotuned and cached
        (:mod:`katsdpsigproc_amd.tune`).
    """

    host_class = host.BackgroundMedianFilterHost
    autotune_version = 5
    SUPPORTED_WIDTHS = tuple(range(3, 32, 2))

    def __init__(self, context: AbstractContext, width: int, is_amplitude: bool = False,
                 use_flags: Union[BackgroundFlags, bool] = BackgroundFlags.NONE,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        if use_flags is True:
            use_flags = BackgroundFlags.CHANNEL
        elif use_flags is False:
            use_flags = BackgroundFlags.NONE
        if not isinstance(use_flags, BackgroundFlags):
            raise TypeError("use_flags must be an instance of BackgroundFlags or bool")
        if width not in self.SUPPORTED_WIDTHS:
            raise ValueError(f"width {width} is not one of {self.SUPPORTED_WIDTHS}")
        self.context = context
        self.width = width
        self.is_amplitude = is_amplitude
        self.use_flags = use_flags
        # resolved on first use: a template that only ever feeds the fused flagger
        # never launches this kernel and should not spend a second tuning it
        self._tuning = dict(tuning) if tuning is not None else None
        self.kernel = context.native_kernel("ksp_background_median_filter")

    @property
    def tuning(self) -> Mapping[str, Any]:
        if self._tuning is None:
            self._tuning = dict(
                self.autotune(self.context, self.width, self.is_amplitude, self.use_flags)
            )
        return self._tuning

    @classmethod
    @tune.autotuner(test={"wgs": 64, "csplit": 4})
    def autotune(cls, context, width: int, is_amplitude: bool,
                 use_flags: BackgroundFlags) -> Mapping[str, Any]:  # fmt: skip
        """Search ``csplit`` on the reference's tuning shape, 4096 channels x 8192
        baselines (reference rfi/device.py:215-252)."""
        queue = context.create_tuning_command_queue()
        channels, baselines = 4096, 8192
        shape = (channels, baselines)
        vis = accel.DeviceArray(context, shape, np.float32 if is_amplitude else np.complex64)
        deviations = accel.DeviceArray(context, shape, np.float32)
        rs = np.random.RandomState(seed=1)
        if is_amplitude:
            vis.set(queue, np.abs(rs.standard_normal(shape)).astype(np.float32))
        else:
            block = rs.standard_normal(shape + (2,)).astype(np.float32).view(np.complex64)[..., 0]
            vis.set(queue, block)
        flags = None
        if use_flags == BackgroundFlags.CHANNEL:
            flags = accel.DeviceArray(context, (channels,), np.uint8)
            flags.set(queue, (rs.random_sample(channels) < 1 / 16).astype(np.uint8))
        elif use_flags == BackgroundFlags.FULL:
            flags = accel.DeviceArray(context, shape, np.uint8)
            flags.set(queue, (rs.random_sample(shape) < 1 / 16).astype(np.uint8))

        def generate(csplit: int):
            fn = cls(context, width, is_amplitude, use_flags, tuning={"wgs": 64, "csplit": csplit})
            op = fn.instantiate(queue, channels, baselines)
            op.bind(vis=vis, deviations=deviations)
            if flags is not None:
                op.bind(flags=flags)
            return tune.make_measure(queue, op)

        # building a candidate compiles nothing here, so one thread will do
        best = tune.autotune(generate, threads=1, csplit=[0, 8, 16, 32, 64, 128])
        return {"wgs": 64, "csplit": int(best["csplit"])}

    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    allocator: Optional[AbstractAllocator] = None) -> "BackgroundMedianFilterDevice":  # fmt: skip
        return BackgroundMedianFilterDevice(self, command_queue, channels, baselines, allocator)


class BackgroundMedianFilterDevice(AbstractBackgroundDevice):
    """Concrete :class:`BackgroundMedianFilterDeviceTemplate`.

    .. rubric:: Slots

    **vis** : channels x baselines, float32 or complex64
    **flags** : channels x baselines or channels, uint8 (only with ``use_flags``)
    **deviations** : channels x baselines, float32
    """

    def __init__(self, template: BackgroundMedianFilterDeviceTemplate,
                 command_queue: AbstractCommandQueue, channels: int, baselines: int,
                 allocator: Optional[AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.kernel
        self.channels = channels
        self.baselines = baselines
        vis_type = np.float32 if template.is_amplitude else np.complex64
        # one Dimension object shared by vis, deviations and full flags: equal strides,
        # as in the reference (rfi/device.py:303-307)
        dims = (channels, accel.Dimension(baselines))
        self.slots["vis"] = accel.IOSlot(dims, vis_type)
        self.slots["deviations"] = accel.IOSlot(dims, np.float32)
        if template.use_flags == BackgroundFlags.FULL:
            self.slots["flags"] = accel.IOSlot(dims, np.uint8)
        elif template.use_flags == BackgroundFlags.CHANNEL:
            self.slots["flags"] = accel.IOSlot((channels,), np.uint8)

    def _run(self) -> None:
        vis = self.buffer("vis")
        deviations = self.buffer("deviations")
        mode = self.template.use_flags
        flags = self.buffer("flags") if mode else None
        flags_stride = flags.padded_shape[1] if mode == BackgroundFlags.FULL else 0
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                vis.buffer,
                deviations.buffer,
                flags.buffer if flags is not None else None,
                np.int32(self.channels),
                np.int32(self.baselines),
                np.int32(vis.padded_shape[1]),
                np.int32(flags_stride),
                np.int32(self.template.width),
                np.int32(self.template.is_amplitude),
                np.int32(mode.value),
                np.int32(self.template.tuning.get("csplit", 0)),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "width": self.template.width,
            "use_flags": self.template.use_flags.name,
            "csplit": self.template.tuning.get("csplit", 0),
            "channels": self.channels,
            "baselines": self.baselines,
        }


# ------------------------------------------------------------------------------ noise
class NoiseEstHostFromDevice(host.AbstractNoiseEstHost):
    """Present a device noise-estimator template through the host call signature."""

    def __init__(self, template: AbstractNoiseEstDeviceTemplate,
                 command_queue: AbstractCommandQueue) -> None:  # fmt: skip
        self.template = template
        self.command_queue = command_queue

    def __call__(self, deviations: np.ndarray) -> np.ndarray:
        channels, baselines = deviations.shape
        if self.template.transposed:
            deviations = deviations.T
        fn = self.template.instantiate(self.command_queue, channels, baselines)
        fn.ensure_all_bound()
        fn.buffer("deviations").set(self.command_queue, deviations)
        fn()
        return fn.buffer("noise").get(self.command_queue)


class NoiseEstMADDeviceTemplate(AbstractNoiseEstDeviceTemplate):
    """Median of non-zero absolute deviations on channel-major data
    (reference rfi/device.py:363-409). :class:`NoiseEstMADTDeviceTemplate` is faster, as
    the reference also notes (rfi/device.py:366).

    `tuning`: ``method`` -- 0: the direct kernel (strips of 8 baselines through LDS up to
    4096 channels; a column re-read per search pass beyond, as the reference does); 1:
    transpose into a temporary ``deviations_t`` slot, then the baseline-major kernel
    (one more pass over the data, but both passes are coalesced). Default: autotuned and
    cached. ``wgsx``/``wgsy`` of the reference have no counterpart.
    """

    host_class = host.NoiseEstMADHost
    transposed = False
    autotune_version = 2

    def __init__(self, context: AbstractContext,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.context = context
        self._tuning = dict(tuning) if tuning is not None else None  # resolved on first use
        self.kernel = context.native_kernel("ksp_madnz")
        self.kernel_t = context.native_kernel("ksp_madnz_t")
        self.kernel_transpose = context.native_kernel("ksp_transpose")

    @property
    def tuning(self) -> Mapping[str, Any]:
        if self._tuning is None:
            self._tuning = dict(self.autotune(self.context))
        return self._tuning

    @classmethod
    @tune.autotuner(test={"method": 0})
    def autotune(cls, context: AbstractContext) -> Mapping[str, Any]:
        queue = context.create_tuning_command_queue()
        channels, baselines = 4096, 8192
        rs = np.random.RandomState(seed=1)
        data = rs.standard_normal((channels, baselines)).astype(np.float32)

        def generate(method: int):
            op = cls(context, tuning={"method": method}).instantiate(queue, channels, baselines)
            op.ensure_all_bound()
            op.buffer("deviations").set(queue, data)
            return tune.make_measure(queue, op)

        return {"method": int(tune.autotune(generate, threads=1, method=[0, 1])["method"])}

    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    allocator: Optional[AbstractAllocator] = None) -> "NoiseEstMADDevice":  # fmt: skip
        return NoiseEstMADDevice(self, command_queue, channels, baselines, allocator)


class NoiseEstMADDevice(AbstractNoiseEstDevice):
    """Concrete :class:`NoiseEstMADDeviceTemplate`.

    .. rubric:: Slots

    **deviations** : channels x baselines, float32
    **noise** : baselines, float32

    .. rubric:: Temporary slots

    **deviations_t** : baselines x channels, float32 (only with ``method`` 1)
    """

    transposed = False

    def __init__(self, template: NoiseEstMADDeviceTemplate, command_queue: AbstractCommandQueue,
                 channels: int, baselines: int,
                 allocator: Optional[AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.kernel
        self.channels = channels
        self.baselines = baselines
        # the baseline-major kernel holds a row in one workgroup's registers
        self.method = int(template.tuning.get("method", 0))
        if channels > NoiseEstMADTDeviceTemplate.MAX_CHANNELS_SUPPORTED:
            self.method = 0
        baselines_dim = accel.Dimension(baselines)
        self.slots["noise"] = accel.IOSlot((baselines_dim,), np.float32)
        self.slots["deviations"] = accel.IOSlot((channels, baselines_dim), np.float32)
        if self.method == 1:
            self.slots["deviations_t"] = accel.IOSlot((baselines, accel.Dimension(channels)), np.float32)

    def _run(self) -> None:
        deviations = self.buffer("deviations")
        noise = self.buffer("noise")
        if self.method == 1:
            dev_t = self.buffer("deviations_t")
            self.command_queue.enqueue_kernel(
                self.template.kernel_transpose,
                [
                    dev_t.buffer,
                    deviations.buffer,
                    np.int32(self.channels),
                    np.int32(self.baselines),
                    np.int32(dev_t.padded_shape[1]),
                    np.int32(deviations.padded_shape[1]),
                    np.int32(4),
                ],
            )
            self.command_queue.enqueue_kernel(
                self.template.kernel_t,
                [
                    dev_t.buffer,
                    noise.buffer,
                    np.int32(self.channels),
                    np.int32(self.baselines),
                    np.int32(dev_t.padded_shape[1]),
                ],
            )
            return
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                deviations.buffer,
                noise.buffer,
                np.int32(self.channels),
                np.int32(self.baselines),
                np.int32(deviations.padded_shape[1]),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {"channels": self.channels, "baselines": self.baselines, "method": self.method}


class NoiseEstMADTDeviceTemplate(AbstractNoiseEstDeviceTemplate):
    """Median of non-zero absolute deviations on baseline-major data
    (reference rfi/device.py:475-549).

    Parameters
    ----------
    context
        Context whose device will run the kernel
    max_channels
        Upper bound on channels per instance (at most 16384: a baseline is held in the
        registers of one 256-work-item workgroup)
    tuning
        The kernel's geometry is fixed (a wavefront or a 256-thread workgroup per baseline):
        ``wgsx`` of the reference is accepted without effect, any other key is a ``ValueError``
        (:func:`.tune.fixed_geometry`).
    """

    host_class = host.NoiseEstMADHost
    transposed = True
    MAX_CHANNELS_SUPPORTED = 256 * 64
    TUNING_KEYS = ("wgsx",)

    def __init__(self, context: AbstractContext, max_channels: int,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        if max_channels > self.MAX_CHANNELS_SUPPORTED:
            raise ValueError(f"max_channels exceeds {self.MAX_CHANNELS_SUPPORTED}")
        self.context = context
        self.max_channels = max_channels
        self.tuning = tune.fixed_geometry("NoiseEstMADTDeviceTemplate", tuning, self.TUNING_KEYS)
        self.kernel = context.native_kernel("ksp_madnz_t")

    @classmethod
    def autotune(cls, context: AbstractContext, max_channels: int) -> Mapping[str, Any]:
        """Nothing to search (reference rfi/device.py:523 times wgsx)."""
        return {}

    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    allocator: Optional[AbstractAllocator] = None) -> "NoiseEstMADTDevice":  # fmt: skip
        return NoiseEstMADTDevice(self, command_queue, channels, baselines, allocator)


class NoiseEstMADTDevice(AbstractNoiseEstDevice):
    """Concrete :class:`NoiseEstMADTDeviceTemplate` (``ValueError`` if `channels` exceeds
    the template's ``max_channels``).

    .. rubric:: Slots

    **deviations** : baselines x channels, float32
    **noise** : baselines, float32
    """

    transposed = True

    def __init__(self, template: NoiseEstMADTDeviceTemplate, command_queue: AbstractCommandQueue,
                 channels: int, baselines: int,
                 allocator: Optional[AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        if channels > template.max_channels:
            raise ValueError("channels exceeds max_channels")
        self.template = template
        self.kernel = template.kernel
        self.channels = channels
        self.baselines = baselines
        self.slots["noise"] = accel.IOSlot((baselines,), np.float32)
        self.slots["deviations"] = accel.IOSlot((baselines, channels), np.float32)

    def _run(self) -> None:
        deviations = self.buffer("deviations")
        noise = self.buffer("noise")
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                deviations.buffer,
                noise.buffer,
                np.int32(self.channels),
                np.int32(self.baselines),
                np.int32(deviations.padded_shape[1]),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "max_channels": self.template.max_channels,
            "baselines": self.baselines,
            "channels": self.channels,
        }


# -------------------------------------------------------------------------- threshold
class ThresholdHostFromDevice(host.AbstractThresholdHost):
    """Present a device thresholder template through the host call signature; extra
    positional/keyword arguments are forwarded to ``instantiate``."""

    def __init__(self, template: AbstractThresholdDeviceTemplate,
                 command_queue: AbstractCommandQueue, *args, **kwargs) -> None:  # fmt: skip
        self.template = template
        self.command_queue = command_queue
        self.args = args
        self.kwargs = kwargs

    def __call__(self, deviations: np.ndarray, noise: np.ndarray) -> np.ndarray:
        channels, baselines = deviations.shape
        transposed = self.template.transposed
        if transposed:
            deviations = deviations.T
        fn = self.template.instantiate(
            self.command_queue, channels, baselines, *self.args, **self.kwargs
        )
        fn.ensure_all_bound()
        fn.buffer("deviations").set(self.command_queue, deviations)
        fn.buffer("noise").set(self.command_queue, noise)
        fn()
        flags = fn.buffer("flags").get(self.command_queue)
        return flags.T if transposed else flags


class ThresholdSimpleDeviceTemplate(AbstractThresholdDeviceTemplate):
    """Independent per-sample threshold, either memory order
    (reference rfi/device.py:654-720). The kernel's geometry is fixed: ``wgsx``/``wgsy`` of the
    reference are accepted in `tuning` without effect, any other key is a ``ValueError``
    (:func:`.tune.fixed_geometry`)."""

    host_class = host.ThresholdSimpleHost
    TUNING_KEYS = ("wgsx", "wgsy")

    def __init__(self, context: AbstractContext, transposed: bool, flag_value: int = 1,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.context = context
        self.transposed = transposed
        self.flag_value = flag_value
        self.tuning = tune.fixed_geometry("ThresholdSimpleDeviceTemplate", tuning, self.TUNING_KEYS)
        self.kernel = context.native_kernel("ksp_threshold_simple")

    @classmethod
    def autotune(cls, context: AbstractContext) -> Mapping[str, Any]:
        """Nothing to search (reference rfi/device.py:707 times wgsx/wgsy)."""
        return {}

    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    n_sigma: float,
                    allocator: Optional[AbstractAllocator] = None) -> "ThresholdSimpleDevice":  # fmt: skip
        return ThresholdSimpleDevice(self, command_queue, channels, baselines, n_sigma, allocator)


class ThresholdSimpleDevice(AbstractThresholdDevice):
    """Concrete :class:`ThresholdSimpleDeviceTemplate`.

    .. rubric:: Slots

    **deviations** : channels x baselines (or transposed), float32
    **noise** : baselines, float32
    **flags** : channels x baselines (or transposed), uint8
    """

    def __init__(self, template: ThresholdSimpleDeviceTemplate,
                 command_queue: AbstractCommandQueue, channels: int, baselines: int,
                 n_sigma: float, allocator: Optional[AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.kernel
        self.n_sigma = n_sigma
        self.channels = channels
        self.baselines = baselines
        self.transposed = template.transposed
        shape = (baselines, channels) if self.transposed else (channels, baselines)
        dims = (accel.Dimension(shape[0]), accel.Dimension(shape[1]))
        noise_dim = dims[0] if self.transposed else dims[1]
        self.slots["deviations"] = accel.IOSlot(dims, np.float32)
        self.slots["noise"] = accel.IOSlot((noise_dim,), np.float32)
        self.slots["flags"] = accel.IOSlot(dims, np.uint8)

    def _run(self) -> None:
        deviations = self.buffer("deviations")
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                deviations.buffer,
                self.buffer("noise").buffer,
                self.buffer("flags").buffer,
                np.int32(deviations.shape[0]),
                np.int32(deviations.shape[1]),
                np.int32(deviations.padded_shape[1]),
                np.float32(self.n_sigma),
                np.int32(self.template.flag_value),
                np.int32(self.transposed),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "n_sigma": self.n_sigma,
            "flag_value": self.template.flag_value,
            "transposed": self.transposed,
            "channels": self.channels,
            "baselines": self.baselines,
        }


class ThresholdSumDeviceTemplate(AbstractThresholdDeviceTemplate):
    """SumThreshold on baseline-major data (reference rfi/device.py:812-907).

    Follows :class:`host.ThresholdSumHost` exactly (float32 threshold chain, float64
    window sums over full windows only), where the reference kernel uses float32 sums
    and zero-pads the band edges.

    Parameters
    ----------
    context
        Context whose device will run the kernel
    n_windows
        Number of window sizes 1, 2, 4, ... (1 to 8, i.e. windows of up to 128
        channels; the reference sets no limit but its halo, ``2**n - n - 1`` channels
        on each side of a chunk, stops being practical about there)
    flag_value
        Value stored for flagged samples
    tuning
        ``vt``: channels per thread of the 256-thread workgroup (8, 16 or 32, i.e. chunks
        of 2048, 4096 or 8192 channels with a ``2**n_windows - n_windows - 1`` channel
        halo between chunks); 0 lets the launcher choose. ``wgs`` of the reference is
        fixed at 256. Default: autotuned and cached (:mod:`katsdpsigproc_amd.tune`).
    """

    host_class = host.ThresholdSumHost
    transposed = True
    autotune_version = 1

    def __init__(self, context: AbstractContext, n_windows: int = 4, flag_value: int = 1,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        if not 1 <= n_windows <= 8:
            raise ValueError("n_windows must be between 1 and 8")
        self.context = context
        self.n_windows = n_windows
        self.flag_value = flag_value
        self._tuning = dict(tuning) if tuning is not None else None  # resolved on first use
        self.kernel = context.native_kernel("ksp_threshold_sum")

    @property
    def tuning(self) -> Mapping[str, Any]:
        if self._tuning is None:
            self._tuning = dict(self.autotune(self.context, self.n_windows))
        return self._tuning

    @classmethod
    @tune.autotuner(test={"wgs": 256, "vt": 8})
    def autotune(cls, context: AbstractContext, n_windows: int) -> Mapping[str, Any]:
        """Search ``vt`` on the reference's tuning shape, 4096 channels x 8192 baselines
        with some interference to threshold (reference rfi/device.py:890-907)."""
        queue = context.create_tuning_command_queue()
        channels, baselines = 4096, 8192
        shape = (baselines, channels)
        rs = np.random.RandomState(seed=1)
        dev = rs.standard_normal(shape).astype(np.float32)
        dev[rs.random_sample(shape) < 0.01] += 100.0
        deviations = accel.DeviceArray(context, shape, np.float32)
        deviations.set(queue, dev)
        noise = accel.DeviceArray(context, (baselines,), np.float32)
        noise.set(queue, np.ones(baselines, np.float32))
        flags = accel.DeviceArray(context, shape, np.uint8)

        def generate(vt: int):
            fn = cls(context, n_windows, tuning={"wgs": 256, "vt": vt})
            op = fn.instantiate(queue, channels, baselines, 11.0)
            op.bind(deviations=deviations, noise=noise, flags=flags)
            return tune.make_measure(queue, op)

        best = tune.autotune(generate, threads=1, vt=[8, 16, 32])
        return {"wgs": 256, "vt": int(best["vt"])}

    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    n_sigma: float,
                    threshold_falloff: float = _THRESHOLD_SUM_DEFAULT_THRESHOLD_FALLOFF,
                    allocator: Optional[AbstractAllocator] = None) -> "ThresholdSumDevice":  # fmt: skip
        return ThresholdSumDevice(
            self, command_queue, channels, baselines, n_sigma, threshold_falloff, allocator
        )


class ThresholdSumDevice(AbstractThresholdDevice):
    """Concrete :class:`ThresholdSumDeviceTemplate`.

    .. rubric:: Slots

    **deviations** : baselines x channels, float32
    **noise** : baselines, float32
    **flags** : baselines x channels, uint8
    """

    host_class = host.ThresholdSumHost
    transposed = True
    DEFAULT_THRESHOLD_FALLOFF = _THRESHOLD_SUM_DEFAULT_THRESHOLD_FALLOFF

    def __init__(self, template: ThresholdSumDeviceTemplate, command_queue: AbstractCommandQueue,
                 channels: int, baselines: int, n_sigma: float,
                 threshold_falloff: float = DEFAULT_THRESHOLD_FALLOFF,
                 allocator: Optional[AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.kernel
        self.channels = channels
        self.baselines = baselines
        self.n_sigma = n_sigma
        self.threshold_falloff = threshold_falloff
        # python-float scales rounded to float32, as numpy does when the host class
        # multiplies a float32 threshold by them (rfi/host.py:215,235)
        self.scales = (ctypes_float_array(
            [np.float32(pow(threshold_falloff, -i)) for i in range(template.n_windows)]
        ))
        # deviations and flags share the channel Dimension, hence the stride
        dims = (baselines, accel.Dimension(channels))
        self.slots["deviations"] = accel.IOSlot(dims, np.float32)
        self.slots["noise"] = accel.IOSlot((baselines,), np.float32)
        self.slots["flags"] = accel.IOSlot(dims, np.uint8)

    def _run(self) -> None:
        deviations = self.buffer("deviations")
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                deviations.buffer,
                self.buffer("noise").buffer,
                self.buffer("flags").buffer,
                np.int32(self.channels),
                np.int32(self.baselines),
                np.int32(deviations.padded_shape[1]),
                np.float32(self.n_sigma),
                self.scales,
                np.int32(self.template.n_windows),
                np.int32(self.template.flag_value),
                np.int32(self.template.tuning.get("vt", 0)),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "n_sigma": self.n_sigma,
            "threshold_falloff": self.threshold_falloff,
            "vt": self.template.tuning.get("vt", 0),
            "flag_value": self.template.flag_value,
            "channels": self.channels,
            "baselines": self.baselines,
        }


def ctypes_float_array(values):
    import ctypes

    return (ctypes.c_float * len(values))(*[float(v) for v in values])


def ctypes_double_array(values):
    import ctypes

    return (ctypes.c_double * len(values))(*[float(v) for v in values])


# ---------------------------------------------------------------------------- flagger
class FlaggerDeviceTemplate:
    """Backgrounder + noise estimator + thresholder (reference rfi/device.py:998-1059).

    Parameters
    ----------
    background, noise_est, threshold
        Templates of the three stages (all on one context)
    fused
        ``None`` (default): use the single-pass fused kernel whenever the combination
        and the instantiated shape allow it, otherwise the kernel-per-stage sequence.
        ``True``: require the fused kernel (``ValueError`` at instantiation if it
        cannot be used). ``False``: always build the reference-shaped sequence.
    keep_deviations
        Fused path only: also provide and write the ``deviations`` slot (float32, 4 more
        bytes per sample of HBM traffic). Off by default: in the reference ``deviations``
        is a *temporary* slot of the flagger (reference rfi/device.py:1081-1091), which a
        single-pass kernel has no need to materialise. The sequence always has it.
    tuning
        Fused path only: ``{"vis_pad": n}``, the number of elements by which the rows of the
        ``vis`` slot are padded beyond `baselines` (the kernel reads 32-byte pieces of 4096
        rows at a time, and how fast the memory system serves that depends on the row
        stride: at 32768 baselines 2 % faster with 512, 18 % slower with 256). By default
        it is autotuned per shape when the operation is instantiated.
    """

    autotune_version = 1
    _VIS_PADS = (0, 16, 32, 512, 2048)

    def __init__(self, background: AbstractBackgroundDeviceTemplate,
                 noise_est: AbstractNoiseEstDeviceTemplate,
                 threshold: AbstractThresholdDeviceTemplate,
                 fused: Optional[bool] = None, keep_deviations: bool = False,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.background = background
        self.noise_est = noise_est
        self.threshold = threshold
        self.fused = fused
        self.keep_deviations = keep_deviations
        self._fused_tuning = dict(tuning) if tuning is not None else None
        context = background.context
        assert noise_est.context is context
        assert threshold.context is context
        self.context = context
        if noise_est.transposed or threshold.transposed:
            self.transpose_deviations: Optional[transpose.TransposeTemplate] = (
                transpose.TransposeTemplate(context, np.float32, "float")
            )
        else:
            self.transpose_deviations = None
        if threshold.transposed:
            self.transpose_flags: Optional[transpose.TransposeTemplate] = (
                transpose.TransposeTemplate(context, np.uint8, "unsigned char")
            )
        else:
            self.transpose_flags = None
        self._fused_kernel = context.native_kernel("ksp_flagger_fused")

    def fusable(self, channels: int) -> bool:
        """Can the fused kernel run this combination of stages at `channels`?"""
        from .. import _lib

        bg, th = self.background, self.threshold
        if not isinstance(bg, BackgroundMedianFilterDeviceTemplate):
            return False
        if not isinstance(self.noise_est, (NoiseEstMADDeviceTemplate, NoiseEstMADTDeviceTemplate)):
            return False
        if isinstance(self.noise_est, NoiseEstMADTDeviceTemplate):
            if channels > self.noise_est.max_channels:
                return False  # keep the error behaviour of the sequence
        if isinstance(th, ThresholdSumDeviceTemplate):
            n_windows = th.n_windows
        elif isinstance(th, ThresholdSimpleDeviceTemplate):
            n_windows = 1
        else:
            return False
        return bool(_lib.call("ksp_flagger_fused_supported", channels, bg.width, n_windows))

    @classmethod
    @tune.autotuner(test={"vis_pad": 0})
    def autotune(cls, context: AbstractContext, channels: int, baselines: int, width: int,
                 is_amplitude: bool, use_flags: BackgroundFlags, sum_threshold: bool,
                 n_windows: int) -> Mapping[str, Any]:  # fmt: skip
        """Row padding of ``vis`` for the fused kernel at exactly this shape: the candidates
        are timed on standard-normal data (no interference, 11 sigma)."""
        if baselines < 2048:
            return {"vis_pad": 0}  # a handful of workgroups per CU: nothing to tune
        queue = context.create_tuning_command_queue()
        rs = np.random.RandomState(seed=1)
        tile = min(baselines, 1024)
        if is_amplitude:
            block = np.abs(rs.standard_normal((channels, tile))).astype(np.float32)
        else:
            block = rs.standard_normal((channels, tile, 2)).astype(np.float32).view(np.complex64)[..., 0]
        data = np.tile(block, (1, -(-baselines // tile)))[:, :baselines]
        if use_flags == BackgroundFlags.CHANNEL:
            mask = (rs.random_sample(channels) < 1 / 16).astype(np.uint8)
        elif use_flags == BackgroundFlags.FULL:
            mask = (rs.random_sample((channels, baselines)) < 1 / 16).astype(np.uint8)
        bg = BackgroundMedianFilterDeviceTemplate(context, width, is_amplitude, use_flags,
                                                  tuning={"wgs": 64, "csplit": 0})  # fmt: skip
        ne = NoiseEstMADTDeviceTemplate(context, max(channels, 1), tuning={"wgsx": 256})
        if sum_threshold:
            th: AbstractThresholdDeviceTemplate = ThresholdSumDeviceTemplate(
                context, n_windows, tuning={"wgsx": 256, "vt": 0})  # fmt: skip
        else:
            th = ThresholdSimpleDeviceTemplate(context, False, tuning={"wgsx": 64, "wgsy": 4})

        def generate(vis_pad: int):
            template = cls(bg, ne, th, fused=True, tuning={"vis_pad": vis_pad})
            op = template.instantiate(queue, channels, baselines, threshold_args={"n_sigma": 11.0})
            op.ensure_all_bound()
            op.buffer("vis").set(queue, data)
            if use_flags != BackgroundFlags.NONE:
                op.buffer("input_flags").set(queue, mask)
            for _ in range(60):  # every candidate starts at sustained clocks
                op()
            queue.finish()
            return tune.make_measure(queue, op)

        best = tune.autotune(generate, threads=1, vis_pad=list(cls._VIS_PADS))
        return {"vis_pad": int(best["vis_pad"])}

    def instantiate(self, command_queue: AbstractCommandQueue, channels: int, baselines: int,
                    background_args: Mapping[str, Any] = {},
                    noise_est_args: Mapping[str, Any] = {},
                    threshold_args: Mapping[str, Any] = {},
                    allocator: Optional[AbstractAllocator] = None) -> accel.Operation:  # fmt: skip
        """Create a :class:`FusedFlaggerDevice` or a :class:`FlaggerDevice`."""
        use_fused = self.fused
        if use_fused is None:
            use_fused = (
                self.fusable(channels) and not background_args and not noise_est_args
            )
        elif use_fused and not self.fusable(channels):
            raise ValueError("this combination of stages/shape cannot use the fused kernel")
        if use_fused:
            if self._fused_tuning is not None:
                vis_pad = int(self._fused_tuning.get("vis_pad", 0))
            else:
                th = self.threshold
                vis_pad = int(self.autotune(
                    self.context, channels, baselines, self.background.width,
                    self.background.is_amplitude, self.background.use_flags,
                    isinstance(th, ThresholdSumDeviceTemplate), getattr(th, "n_windows", 1),
                )["vis_pad"])  # fmt: skip
            return FusedFlaggerDevice(
                self, command_queue, channels, baselines, threshold_args, allocator, vis_pad
            )
        return FlaggerDevice(
            self, command_queue, channels, baselines, background_args, noise_est_args,
            threshold_args, allocator,
        )  # fmt: skip


class FlaggerDevice(accel.OperationSequence):
    """The kernel-per-stage flagger, wired exactly as the reference wires it
    (reference rfi/device.py:1062-1166).

    .. rubric:: Slots

    **vis** : channels x baselines, float32 or complex64
    **noise** : baselines, float32
    **flags** : channels x baselines, uint8
    **input_flags** : channels x baselines or channels, uint8 -- only if the backgrounder
        uses flags; they steer the background only, are not copied to the output, and a
        sample flagged here is never flagged as RFI.

    .. rubric:: Temporary slots

    **deviations** : channels x baselines, float32
    **deviations_t** : baselines x channels, float32 (if any stage is transposed)
    **flags_t** : baselines x channels, uint8 (if the thresholder is transposed)
    """

    def __init__(self, template: FlaggerDeviceTemplate, command_queue: AbstractCommandQueue,
                 channels: int, baselines: int, background_args: Mapping[str, Any] = {},
                 noise_est_args: Mapping[str, Any] = {}, threshold_args: Mapping[str, Any] = {},
                 allocator: Optional[AbstractAllocator] = None) -> None:  # fmt: skip
        self.template = template
        self.channels = channels
        self.baselines = baselines
        self.background = template.background.instantiate(
            command_queue, channels, baselines, allocator=allocator, **background_args
        )
        self.noise_est = template.noise_est.instantiate(
            command_queue, channels, baselines, allocator=allocator, **noise_est_args
        )
        self.threshold = template.threshold.instantiate(
            command_queue, channels, baselines, allocator=allocator, **threshold_args
        )
        stages: List[Tuple[str, accel.Operation]] = [("background", self.background)]
        if template.transpose_deviations is not None:
            self.transpose_deviations = template.transpose_deviations.instantiate(
                command_queue, (channels, baselines)
            )
            stages.append(("transpose_deviations", self.transpose_deviations))
        stages.append(("noise_est", self.noise_est))
        stages.append(("threshold", self.threshold))
        if template.transpose_flags is not None:
            self.transpose_flags = template.transpose_flags.instantiate(
                command_queue, (baselines, channels)
            )
            stages.append(("transpose_flags", self.transpose_flags))

        # which layout each consumer reads decides which compound it joins
        dev_of_noise = "deviations_t" if self.noise_est.transposed else "deviations"
        dev_of_threshold = "deviations_t" if self.threshold.transposed else "deviations"
        flags_of_threshold = "flags_t" if self.threshold.transposed else "flags"
        compounds = {
            "vis": ["background:vis"],
            "input_flags": ["background:flags"],
            "deviations": ["background:deviations", "transpose_deviations:src"],
            "deviations_t": ["transpose_deviations:dest"],
            "noise": ["noise_est:noise", "threshold:noise"],
            "flags_t": ["transpose_flags:src"],
            "flags": ["transpose_flags:dest"],
        }
        compounds[dev_of_noise].append("noise_est:deviations")
        compounds[dev_of_threshold].append("threshold:deviations")
        compounds[flags_of_threshold].append("threshold:flags")
        super().__init__(command_queue, stages, compounds, allocator=allocator)


class FusedFlaggerDevice(accel.Operation):
    """Single-pass flagger: one HIP kernel from visibilities to flags.

    Presents the slots of :class:`FlaggerDevice` that survive fusion. Results are
    bit-identical to ``host.FlaggerHost(BackgroundMedianFilterHost, NoiseEstMADHost,
    ThresholdSumHost | ThresholdSimpleHost)`` because everything after the float32
    amplitude is computed in float64, as the host classes do.

    .. rubric:: Slots

    **vis** : channels x baselines, complex64 (float32 if the backgrounder takes amplitudes)
    **input_flags** : channels or channels x baselines, uint8 (only with ``use_flags``)
    **noise** : baselines, float32 (float64 estimate rounded once)
    **flags** : channels x baselines, uint8
    .. rubric:: Optional slots (the reference's temporaries, rfi/device.py:1081-1150)

    **deviations** : channels x baselines, float32
    **deviations_t** : baselines x channels, float32
    **flags_t** : baselines x channels, uint8

    The fused kernel needs none of them, so :meth:`ensure_all_bound` leaves them alone
    (``deviations`` excepted if the template was built with ``keep_deviations=True``). They
    exist all the same: binding one, or asking for it with :meth:`buffer` (which allocates
    it), makes every later call fill it -- ``deviations`` by the kernel itself (13 instead of
    9 bytes per sample, and the 4-baseline kernel instead of the persistent one), the
    transposed ones by a transpose behind it. A buffer materialised after a call holds
    nothing until the next call.
    """

    _OPTIONAL = ("deviations", "deviations_t", "flags_t")

    def __init__(self, template: FlaggerDeviceTemplate, command_queue: AbstractCommandQueue,
                 channels: int, baselines: int, threshold_args: Mapping[str, Any] = {},
                 allocator: Optional[AbstractAllocator] = None, vis_pad: int = 0) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        if vis_pad < 0 or vis_pad % 2:
            raise ValueError("vis_pad must be even and not negative")
        self.template = template
        self.kernel = template._fused_kernel
        self.channels = channels
        self.baselines = baselines
        bg = template.background
        th = template.threshold
        args = dict(threshold_args)
        if "n_sigma" not in args:
            raise TypeError("threshold_args must provide n_sigma")
        self.n_sigma = float(args.pop("n_sigma"))
        if isinstance(th, ThresholdSumDeviceTemplate):
            falloff = float(args.pop("threshold_falloff", _THRESHOLD_SUM_DEFAULT_THRESHOLD_FALLOFF))
            self.threshold_kind = 1
            self.n_windows = th.n_windows
            self.scales = ctypes_double_array([pow(falloff, -i) for i in range(th.n_windows)])
        else:
            self.threshold_kind = 0
            self.n_windows = 1
            self.scales = ctypes_double_array([1.0])
        if args:
            raise TypeError(f"unexpected threshold arguments {sorted(args)}")
        vis_type = np.float32 if bg.is_amplitude else np.complex64
        # 16-byte loads of baseline pairs want an even row stride; `vis_pad` more elements
        # per row where that stride suits the memory system better (template's `tuning`)
        self.vis_pad = vis_pad
        self.slots["vis"] = accel.IOSlot(
            (channels, accel.Dimension(baselines, min_padded_size=baselines + vis_pad, alignment=2)),
            vis_type,
        )
        if bg.use_flags == BackgroundFlags.FULL:
            self.slots["input_flags"] = accel.IOSlot((channels, accel.Dimension(baselines)), np.uint8)
        elif bg.use_flags == BackgroundFlags.CHANNEL:
            self.slots["input_flags"] = accel.IOSlot((channels,), np.uint8)
        self.slots["noise"] = accel.IOSlot((baselines,), np.float32)
        # flags and deviations are shared with the source slots of two transposes that only
        # run when somebody has bound their destinations (the reference's flags_t / deviations_t)
        context = command_queue.context
        self._transpose_dev = transpose.TransposeTemplate(context, np.float32, "float").instantiate(
            command_queue, (channels, baselines), allocator=allocator)
        self._transpose_flags = transpose.TransposeTemplate(
            context, np.uint8, "unsigned char").instantiate(
            command_queue, (channels, baselines), allocator=allocator)
        self.slots["flags"] = accel.CompoundIOSlot([
            accel.IOSlot((channels, accel.Dimension(baselines)), np.uint8),
            self._transpose_flags.slots["src"]])
        self.slots["deviations"] = accel.CompoundIOSlot([
            accel.IOSlot((channels, accel.Dimension(baselines)), np.float32),
            self._transpose_dev.slots["src"]])
        self.slots["deviations_t"] = self._transpose_dev.slots["dest"]
        self.slots["flags_t"] = self._transpose_flags.slots["dest"]
        self._mandatory = [name for name in self.slots
                           if name not in self._OPTIONAL
                           or (name == "deviations" and template.keep_deviations)]
        # scheduling counters of the kernel (64 bytes, zeroed once; not a slot: nothing a
        # caller could usefully bind)
        self._workspace = accel.DeviceArray(command_queue.context, (16,), np.uint32)
        self._workspace.zero(command_queue)
        self._armed = None

    def ensure_all_bound(self) -> None:
        for name in self._mandatory:
            self.ensure_bound(name)

    def buffer(self, name: str) -> accel.DeviceArray:
        if name in self._OPTIONAL and not self.slots[name].is_bound():
            self.ensure_bound(name)  # materialise: filled by every call from now on
        return super().buffer(name)

    def required_bytes(self) -> int:
        return sum(slot.required_bytes() for name, slot in self.slots.items()
                   if name in self._mandatory or slot.is_bound())

    def _run(self) -> None:
        bg = self.template.background
        vis = self.buffer("vis")
        flags = self.buffer("flags")
        in_flags = self.buffer("input_flags") if bg.use_flags else None
        want_dev_t = self.slots["deviations_t"].is_bound()
        if want_dev_t:
            self.ensure_bound("deviations")
        dev = super().buffer("deviations") if self.slots["deviations"].is_bound() else None
        in_flags_stride = in_flags.padded_shape[1] if bg.use_flags == BackgroundFlags.FULL else 0
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                vis.buffer,
                in_flags.buffer if in_flags is not None else None,
                flags.buffer,
                dev.buffer if dev is not None else None,
                self.buffer("noise").buffer,
                np.int32(self.channels),
                np.int32(self.baselines),
                np.int32(vis.padded_shape[1]),
                np.int32(in_flags_stride),
                np.int32(flags.padded_shape[1]),
                np.int32(dev.padded_shape[1] if dev is not None else 0),
                np.int32(bg.width),
                np.int32(bg.is_amplitude),
                np.int32(bg.use_flags.value),
                np.int32(self.threshold_kind),
                float(self.n_sigma),
                self.scales,
                np.int32(self.n_windows),
                np.int32(self.template.threshold.flag_value),
                self._workspace.buffer,
            ],
        )
        self._armed = None  # (events of profile_next_run are recorded now: ours to drop)
        if want_dev_t:
            self._transpose_dev()
        if self.slots["flags_t"].is_bound():
            self._transpose_flags()

    def profile_next_run(self):
        """Arm two events around the flagger kernel of the next call (excluding the
        zero-fill of `flags` that precedes it); returns (start, stop). After the queue
        has finished, ``stop.time_since(start)`` is that kernel's duration."""
        from .. import _lib

        queue = self.command_queue
        start, stop = queue.create_event(), queue.create_event()
        _lib.call("ksp_flagger_fused_profile", ctypes.c_void_p(start.handle),
                  ctypes.c_void_p(stop.handle))  # fmt: skip
        self._armed = (start, stop)  # alive until the call, whatever the caller does with them
        return start, stop

    def parameters(self) -> Mapping[str, Any]:
        return {
            "fused": True,
            "vis_pad": self.vis_pad,
            "keep_deviations": self.slots["deviations"].is_bound(),
            "width": self.template.background.width,
            "n_sigma": self.n_sigma,
            "n_windows": self.n_windows,
            "channels": self.channels,
            "baselines": self.baselines,
        }


class FlaggerHostFromDevice(host.AbstractFlaggerHost):
    """Make a :class:`FlaggerDeviceTemplate` callable like ``host.FlaggerHost``
    (reference rfi/device.py:1169-1222); allocates on every call."""

    def __init__(self, template: FlaggerDeviceTemplate, command_queue: AbstractCommandQueue,
                 background_args: Mapping[str, Any] = {}, noise_est_args: Mapping[str, Any] = {},
                 threshold_args: Mapping[str, Any] = {}) -> None:  # fmt: skip
        self.template = template
        self.command_queue = command_queue
        self.background_args = dict(background_args)
        self.noise_est_args = dict(noise_est_args)
        self.threshold_args = dict(threshold_args)

    def __call__(self, vis: np.ndarray, input_flags: Optional[np.ndarray] = None) -> np.ndarray:
        if input_flags is not None and not self.template.background.use_flags:
            raise TypeError("channel flags were provided but not included in the template")
        if input_flags is None and self.template.background.use_flags:
            raise TypeError("channel flags were expected but not provided")
        channels, baselines = vis.shape
        fn = self.template.instantiate(
            self.command_queue, channels, baselines, self.background_args, self.noise_est_args,
            self.threshold_args,
        )  # fmt: skip
        fn.ensure_all_bound()
        fn.buffer("vis").set(self.command_queue, vis)
        if input_flags is not None:
            fn.buffer("input_flags").set(self.command_queue, input_flags)
        fn()
        return fn.buffer("flags").get(self.command_queue)
