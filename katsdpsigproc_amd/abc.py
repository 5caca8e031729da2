"""Abstract interfaces of a compute backend.

Same seam as the reference (reference: src/katsdpsigproc/abc.py:43-465): operations are
written against these classes, and :mod:`katsdpsigproc_amd.hip` is the MI355X
implementation. Only the parts the RFI path needs are abstract here; the docstrings say
where behaviour differs from a run-time-compiling backend.
"""

from abc import ABC, abstractmethod
from typing import Any, List, Optional, Sequence, Tuple


class AbstractProgram(ABC):
    """A collection of kernels (reference abc.py:43-54)."""

    @abstractmethod
    def get_kernel(self, name: str) -> "AbstractKernel":
        """Return a new kernel object for the entry point `name`."""


class AbstractKernel(ABC):
    """Something :meth:`AbstractCommandQueue.enqueue_kernel` can launch (abc.py:57-68)."""


class AbstractEvent(ABC):
    """A marker in a command queue (CUDA-event semantics, reference abc.py:71-95)."""

    @abstractmethod
    def wait(self) -> None:
        """Block until the marker has been reached."""

    @abstractmethod
    def time_since(self, prior_event: "AbstractEvent") -> float:
        """Seconds from `prior_event` to this event; waits for both first."""

    @abstractmethod
    def time_till(self, next_event: "AbstractEvent") -> float:
        """Seconds from this event to `next_event`; waits for both first."""


class AbstractDevice(ABC):
    """A compute device (reference abc.py:98-157)."""

    @abstractmethod
    def make_context(self) -> "AbstractContext":
        """Create a context on this device."""

    @property
    @abstractmethod
    def name(self) -> str: ...

    @property
    @abstractmethod
    def platform_name(self) -> str: ...

    @property
    @abstractmethod
    def driver_version(self) -> str: ...

    @property
    @abstractmethod
    def is_cuda(self) -> bool: ...

    @property
    @abstractmethod
    def is_gpu(self) -> bool: ...

    @property
    @abstractmethod
    def is_accelerator(self) -> bool: ...

    @property
    @abstractmethod
    def is_cpu(self) -> bool: ...

    @property
    @abstractmethod
    def simd_group_size(self) -> int:
        """Work-items that run in lock step (64 on gfx950); a tuning hint only."""

    @classmethod
    @abstractmethod
    def get_devices(cls) -> Sequence["AbstractDevice"]:
        """All devices of this backend."""

    @classmethod
    @abstractmethod
    def get_devices_by_platform(cls) -> Sequence[Sequence["AbstractDevice"]]:
        """All devices, one sub-list per platform."""


class AbstractContext(ABC):
    """Owner of memory and queues on one device (reference abc.py:160-245)."""

    @property
    @abstractmethod
    def device(self) -> AbstractDevice: ...

    @abstractmethod
    def compile(self, source: str, extra_flags: Optional[List[str]] = None) -> AbstractProgram:
        """Build a program from source text."""

    @abstractmethod
    def allocate_raw(self, n_bytes: int) -> Any:
        """Untyped device storage."""

    @abstractmethod
    def allocate(self, shape: Tuple[int, ...], dtype: Any, raw: Any = None) -> Any:
        """Typed device buffer, optionally on top of `raw` storage."""

    @abstractmethod
    def allocate_pinned(self, shape: Tuple[int, ...], dtype: Any) -> Any:
        """Page-locked host array suited to fast transfers."""

    @abstractmethod
    def allocate_svm_raw(self, n_bytes: int) -> Any: ...

    @abstractmethod
    def allocate_svm(self, shape: Tuple[int, ...], dtype: Any, raw: Any = None) -> Any: ...

    @abstractmethod
    def create_command_queue(self, profile: bool = False) -> "AbstractCommandQueue": ...

    @abstractmethod
    def create_tuning_command_queue(self) -> "AbstractTuningCommandQueue": ...

    @abstractmethod
    def __enter__(self): ...

    @abstractmethod
    def __exit__(self, exc_type, exc_val, exc_tb): ...


class AbstractCommandQueue(ABC):
    """In-order asynchronous work queue (reference abc.py:248-448)."""

    context: AbstractContext

    @abstractmethod
    def enqueue_read_buffer(self, buffer: Any, data: Any, blocking: bool = True) -> None:
        """Whole-buffer device-to-host copy."""

    @abstractmethod
    def enqueue_write_buffer(self, buffer: Any, data: Any, blocking: bool = True) -> None:
        """Whole-buffer host-to-device copy."""

    @abstractmethod
    def enqueue_copy_buffer_rect(
        self, src_buffer, dest_buffer, src_origin, dest_origin, shape, src_strides, dest_strides
    ) -> None:
        """Device-to-device copy of a <=3-D byte region (shape[0] is a byte count)."""

    @abstractmethod
    def enqueue_read_buffer_rect(
        self, buffer, data, buffer_origin, data_origin, shape, buffer_strides, data_strides,
        blocking: bool = True,
    ) -> None:  # fmt: skip
        """Device-to-host copy of a <=3-D byte region."""

    @abstractmethod
    def enqueue_write_buffer_rect(
        self, buffer, data, buffer_origin, data_origin, shape, buffer_strides, data_strides,
        blocking: bool = True,
    ) -> None:  # fmt: skip
        """Host-to-device copy of a <=3-D byte region."""

    @abstractmethod
    def enqueue_zero_buffer(self, buffer: Any) -> None:
        """Fill a buffer with zero bytes."""

    @abstractmethod
    def enqueue_kernel(
        self,
        kernel: AbstractKernel,
        args: Sequence[Any],
        global_size: Optional[Tuple[int, ...]] = None,
        local_size: Optional[Tuple[int, ...]] = None,
    ) -> None:
        """Launch `kernel` with `args`.

        The ahead-of-time HIP kernels choose their own launch geometry, so for them
        `global_size`/`local_size` are optional and ignored (reference abc.py:406-432
        requires them because its kernels are generic compiled source).
        """

    @abstractmethod
    def enqueue_marker(self) -> AbstractEvent:
        """Record an event at this point of the queue."""

    @abstractmethod
    def enqueue_wait_for_events(self, events: Sequence[AbstractEvent]) -> None:
        """Make later work in this queue wait for `events`."""

    @abstractmethod
    def flush(self) -> None:
        """Start queued work without waiting for it."""

    @abstractmethod
    def finish(self) -> None:
        """Block until all queued work is complete."""


class AbstractTuningCommandQueue(AbstractCommandQueue):
    """Queue that can time what is enqueued between two calls (reference abc.py:451-465)."""

    @abstractmethod
    def start_tuning(self) -> None: ...

    @abstractmethod
    def stop_tuning(self) -> float:
        """Seconds of device time since :meth:`start_tuning`."""
