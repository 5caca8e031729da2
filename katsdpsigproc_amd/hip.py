"""HIP backend for MI355X: devices, contexts, queues, events and native kernels.

Implements the interfaces of :mod:`katsdpsigproc_amd.abc` on top of the C-ABI library
(``include/katsdpsigproc_hip.h``) through ctypes. It plays the role the PyCUDA backend
plays in the reference (reference: src/katsdpsigproc/cuda.py:54-502) and follows its
observable behaviour: blocking copies are an asynchronous copy plus a stream
synchronise (cuda.py:263-281), events are blocking-sync (cuda.py:463), queues are
in-order streams, and every call selects the context's device first (the C side calls
``hipSetDevice``), which is what allows one context per GPU in one process.
"""

import ctypes
import weakref
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .abc import (
    AbstractCommandQueue,
    AbstractContext,
    AbstractDevice,
    AbstractEvent,
    AbstractKernel,
    AbstractProgram,
    AbstractTuningCommandQueue,
)

_H2D, _D2H, _D2D = 0, 1, 2


def _size3(values: Sequence[int]):
    padded = list(values) + [1] * (3 - len(values))
    return (ctypes.c_size_t * 3)(*padded)


class RawBuffer:
    """Untyped device allocation; freed when the last reference goes away."""

    def __init__(self, device_index: int, n_bytes: int) -> None:
        ptr = ctypes.c_void_p()
        _lib.call("ksp_malloc", device_index, n_bytes, ctypes.byref(ptr))
        self.ptr = ptr.value or 0
        self.nbytes = int(n_bytes)
        self.device_index = device_index
        self._finalizer = weakref.finalize(self, _free_device, device_index, self.ptr)


def _free_device(device_index: int, ptr: int) -> None:
    try:
        _lib.call("ksp_free", device_index, ctypes.c_void_p(ptr))
    except Exception:  # interpreter shutdown, device already gone
        pass


def _free_host(ptr: int) -> None:
    try:
        _lib.call("ksp_host_free", ctypes.c_void_p(ptr))
    except Exception:
        pass


class Buffer:
    """Typed view of device memory: what ``DeviceArray.buffer`` holds.

    ``ptr`` is the raw device address handed to kernels. The object also exposes
    ``__cuda_array_interface__`` so that it can be wrapped without a copy by anything
    that understands it (used for the RCCL broadcast of the channel mask).
    """

    def __init__(self, raw: RawBuffer, shape: Tuple[int, ...], dtype) -> None:
        self.raw = raw
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if self.nbytes > raw.nbytes:
            raise ValueError("raw storage is smaller than the requested array")
        self.ptr = raw.ptr

    @property
    def __cuda_array_interface__(self):
        return {
            "shape": self.shape,
            "typestr": self.dtype.str,
            "data": (self.ptr, False),
            "version": 2,
        }


class NativeKernel(AbstractKernel):
    """Handle to one ``ksp_*`` launcher of the C-ABI library."""

    def __init__(self, program: "NativeProgram", name: str) -> None:
        self.program = program
        self.name = name
        self._fn = getattr(_lib.load(), name)  # AttributeError if the symbol is absent


class NativeProgram(AbstractProgram):
    """The ahead-of-time compiled kernel library, presented as a program."""

    def get_kernel(self, name: str) -> NativeKernel:
        return NativeKernel(self, name)


def _unload_module(device_index: int, handle: int) -> None:
    try:
        _lib.call("ksp_module_unload", device_index, ctypes.c_void_p(handle))
    except Exception:
        pass


class CompileError(RuntimeError):
    """hiprtc rejected the source; ``log`` holds the compiler's messages."""

    def __init__(self, message: str, log: str, source: str) -> None:
        super().__init__(message + "\n" + log)
        self.log = log
        self.source = source


class RtcKernel(AbstractKernel):
    """One ``extern "C" __global__`` function of a run-time compiled program."""

    def __init__(self, program: "RtcProgram", name: str) -> None:
        handle = ctypes.c_void_p()
        _lib.call("ksp_module_get_function", program.device_index,
                  ctypes.c_void_p(program.handle), name.encode(), ctypes.byref(handle))  # fmt: skip
        self.program = program  # keeps the module loaded
        self.name = name
        self.handle = handle.value


class RtcProgram(AbstractProgram):
    """HIP source compiled for this device by hiprtc (reference cuda.py:182-187 hands it
    to nvcc through ``pycuda.compiler.SourceModule``)."""

    LOG_BYTES = 1 << 16

    def __init__(self, device_index: int, source: str, flags: Sequence[str]) -> None:
        options = [f.encode() for f in flags]
        array = (ctypes.c_char_p * max(1, len(options)))(*options)
        handle = ctypes.c_void_p()
        log = ctypes.create_string_buffer(self.LOG_BYTES)
        try:
            _lib.call("ksp_rtc_compile", device_index, source.encode(), array, len(options),
                      ctypes.byref(handle), log, self.LOG_BYTES)  # fmt: skip
        except RuntimeError as exc:
            raise CompileError(str(exc), log.value.decode("utf-8", "replace"), source) from None
        self.device_index = device_index
        self.handle = handle.value
        self.log = log.value.decode("utf-8", "replace")
        self._finalizer = weakref.finalize(self, _unload_module, device_index, self.handle)

    def get_kernel(self, name: str) -> RtcKernel:
        return RtcKernel(self, name)


class Event(AbstractEvent):
    """A recorded ``hipEvent_t`` (blocking-sync)."""

    def __init__(self, device_index: int, handle: int) -> None:
        self.device_index = device_index
        self.handle = handle
        self._finalizer = weakref.finalize(self, _destroy_event, device_index, handle)

    def wait(self) -> None:
        _lib.call("ksp_event_synchronize", self.device_index, ctypes.c_void_p(self.handle))

    def time_since(self, prior_event: "Event") -> float:
        prior_event.wait()
        self.wait()
        ms = ctypes.c_float()
        _lib.call(
            "ksp_event_elapsed_ms", self.device_index, ctypes.c_void_p(prior_event.handle),
            ctypes.c_void_p(self.handle), ctypes.byref(ms)
        )  # fmt: skip
        return ms.value * 1e-3

    def time_till(self, next_event: "Event") -> float:
        return next_event.time_since(self)


def _destroy_event(device_index: int, handle: int) -> None:
    try:
        _lib.call("ksp_event_destroy", device_index, ctypes.c_void_p(handle))
    except Exception:
        pass


class Device(AbstractDevice):
    """One HIP device."""

    def __init__(self, index: int) -> None:
        self.index = index
        props = _lib.DeviceProps()
        _lib.call("ksp_device_get_props", index, ctypes.byref(props))
        self._props = props

    def make_context(self) -> "Context":
        return Context(self)

    @property
    def name(self) -> str:
        return self._props.name.decode("utf-8", "replace")

    @property
    def arch(self) -> str:
        """``gcnArchName`` (e.g. ``gfx950:sramecc+:xnack-``)."""
        return self._props.arch.decode("utf-8", "replace")

    @property
    def compute_units(self) -> int:
        return int(self._props.compute_units)

    @property
    def total_memory(self) -> int:
        return int(self._props.total_memory)

    @property
    def platform_name(self) -> str:
        return "HIP"

    @property
    def driver_version(self) -> str:
        return f"HIP {self._props.driver_version} (runtime {self._props.runtime_version})"

    @property
    def is_cuda(self) -> bool:
        return False

    @property
    def is_gpu(self) -> bool:
        return True

    @property
    def is_accelerator(self) -> bool:
        return False

    @property
    def is_cpu(self) -> bool:
        return False

    @property
    def simd_group_size(self) -> int:
        return int(self._props.wavefront_size)

    @classmethod
    def get_devices(cls) -> List["Device"]:
        count = ctypes.c_int()
        _lib.call("ksp_device_count", ctypes.byref(count))
        return [cls(i) for i in range(count.value)]

    @classmethod
    def get_devices_by_platform(cls) -> List[List["Device"]]:
        return [cls.get_devices()]


class Context(AbstractContext):
    """Memory and queues of one device. HIP has no explicit context object to push, so
    entering/leaving is a no-op; the device is selected per call."""

    def __init__(self, device: Device) -> None:
        self._device = device
        self.program = NativeProgram()

    @property
    def device(self) -> Device:
        return self._device

    def native_kernel(self, name: str) -> NativeKernel:
        """Kernel handle for the C-ABI launcher `name` (e.g. ``"ksp_transpose"``)."""
        return self.program.get_kernel(name)

    def compile(self, source: str, extra_flags: Optional[List[str]] = None) -> AbstractProgram:
        """Compile HIP source for this device (hiprtc) -- for kernels that are not part
        of the ahead-of-time library: the templated operations (fill, hreduce) and
        user kernels. Raises :class:`CompileError` with the compiler's log."""
        flags = ["-O3", "-std=c++17"] + list(extra_flags or [])
        return RtcProgram(self._device.index, source, flags)

    def allocate_raw(self, n_bytes: int) -> RawBuffer:
        return RawBuffer(self._device.index, int(n_bytes))

    def allocate(self, shape: Tuple[int, ...], dtype, raw: Optional[RawBuffer] = None) -> Buffer:
        n_bytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if raw is None:
            raw = self.allocate_raw(n_bytes)
        return Buffer(raw, shape, dtype)

    def allocate_pinned(self, shape: Tuple[int, ...], dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        n_bytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr = ctypes.c_void_p()
        _lib.call("ksp_host_alloc", max(n_bytes, 1), ctypes.byref(ptr))
        storage = (ctypes.c_char * max(n_bytes, 1)).from_address(ptr.value)
        # the ctypes array is the numpy base object; tie the allocation's life to it
        storage._finalizer = weakref.finalize(storage, _free_host, ptr.value)
        return np.frombuffer(storage, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    def allocate_svm_raw(self, n_bytes: int):
        raise NotImplementedError("shared virtual memory is outside the RFI path")

    def allocate_svm(self, shape, dtype, raw=None):
        raise NotImplementedError("shared virtual memory is outside the RFI path")

    def create_command_queue(self, profile: bool = False) -> "CommandQueue":
        return CommandQueue(self, profile=profile)

    def create_tuning_command_queue(self) -> "TuningCommandQueue":
        return TuningCommandQueue(self)

    def __enter__(self) -> "Context":
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        return None


def _destroy_stream(device_index: int, handle: int) -> None:
    try:
        _lib.call("ksp_stream_destroy", device_index, ctypes.c_void_p(handle))
    except Exception:
        pass


class CommandQueue(AbstractCommandQueue):
    """An in-order HIP stream."""

    def __init__(self, context: Context, profile: bool = False, stream: Optional[int] = None):
        self.context = context
        self.profile = profile
        self._dev = context.device.index
        if stream is None:
            handle = ctypes.c_void_p()
            _lib.call("ksp_stream_create", self._dev, ctypes.byref(handle))
            self.stream = handle.value or 0
            self._finalizer = weakref.finalize(self, _destroy_stream, self._dev, self.stream)
        else:
            self.stream = int(stream)  # borrowed (e.g. torch's current stream)
        self._keepalive: List[Any] = []

    @property
    def _s(self):
        return ctypes.c_void_p(self.stream)

    # -- whole-buffer copies
    @staticmethod
    def _host_ptr(data: np.ndarray, n_bytes: int):
        if not isinstance(data, np.ndarray) or not data.flags.c_contiguous:
            raise ValueError("host data must be a C-contiguous numpy array")
        if data.nbytes != n_bytes:
            raise ValueError("host array and device buffer differ in size")
        return ctypes.c_void_p(data.ctypes.data)

    def enqueue_read_buffer(self, buffer: Buffer, data: np.ndarray, blocking: bool = True) -> None:
        _lib.call(
            "ksp_memcpy_async", self._dev, self._host_ptr(data, buffer.nbytes),
            ctypes.c_void_p(buffer.ptr), buffer.nbytes, _D2H, self._s
        )  # fmt: skip
        self._after_copy(data, blocking)

    def enqueue_write_buffer(self, buffer: Buffer, data: np.ndarray, blocking: bool = True) -> None:
        _lib.call(
            "ksp_memcpy_async", self._dev, ctypes.c_void_p(buffer.ptr),
            self._host_ptr(data, buffer.nbytes), buffer.nbytes, _H2D, self._s
        )  # fmt: skip
        self._after_copy(data, blocking)

    def _after_copy(self, data: Any, blocking: bool) -> None:
        if blocking:
            self.finish()
        else:
            self._keepalive.append(data)  # keep the host memory alive until finish()

    # -- rectangular copies (byte units; reference abc.py:291-400)
    def _rect(self, dst_ptr, dst_origin, dst_strides, src_ptr, src_origin, src_strides, shape, kind):
        ndim = len(shape)
        if not 1 <= ndim <= 3:
            raise ValueError("rect copies support 1 to 3 dimensions")
        _lib.call(
            "ksp_memcpy_rect_async", self._dev, ctypes.c_void_p(dst_ptr), int(dst_origin),
            _size3(dst_strides), ctypes.c_void_p(src_ptr), int(src_origin), _size3(src_strides),
            _size3(shape), ndim, kind, self._s
        )  # fmt: skip

    def enqueue_copy_buffer_rect(
        self, src_buffer, dest_buffer, src_origin, dest_origin, shape, src_strides, dest_strides
    ) -> None:
        self._rect(dest_buffer.ptr, dest_origin, dest_strides, src_buffer.ptr, src_origin,
                   src_strides, shape, _D2D)  # fmt: skip

    def enqueue_read_buffer_rect(
        self, buffer, data, buffer_origin, data_origin, shape, buffer_strides, data_strides,
        blocking: bool = True,
    ) -> None:  # fmt: skip
        self._rect(data.ctypes.data, data_origin, data_strides, buffer.ptr, buffer_origin,
                   buffer_strides, shape, _D2H)  # fmt: skip
        self._after_copy(data, blocking)

    def enqueue_write_buffer_rect(
        self, buffer, data, buffer_origin, data_origin, shape, buffer_strides, data_strides,
        blocking: bool = True,
    ) -> None:  # fmt: skip
        self._rect(buffer.ptr, buffer_origin, buffer_strides, data.ctypes.data, data_origin,
                   data_strides, shape, _H2D)  # fmt: skip
        self._after_copy(data, blocking)

    def enqueue_zero_buffer(self, buffer: Buffer) -> None:
        _lib.call("ksp_memset_async", self._dev, ctypes.c_void_p(buffer.ptr), 0, buffer.nbytes,
                  self._s)  # fmt: skip

    # -- kernels
    def enqueue_kernel(self, kernel, args, global_size=None, local_size=None) -> None:
        """Launch a kernel.

        :class:`NativeKernel` (a launcher of the ahead-of-time library): `args` are the
        launcher's arguments after ``(device, stream)``; the launch geometry is fixed
        inside the launcher, so `global_size`/`local_size` are ignored.

        :class:`RtcKernel` (run-time compiled): `args` are the kernel's arguments in
        order -- :class:`Buffer` objects (their address is passed), ``None`` (a null
        pointer) or numpy scalars (passed by value with exactly their dtype) -- and
        `global_size`/`local_size` are the work sizes in *threads* per dimension, with
        each global size a multiple of the local one (reference abc.py:406-432).
        """
        if isinstance(kernel, RtcKernel):
            self._enqueue_rtc(kernel, args, global_size, local_size)
            return
        if not isinstance(kernel, NativeKernel):
            raise TypeError("not a kernel of the HIP backend")
        converted = []
        for arg in args:
            if isinstance(arg, Buffer):
                converted.append(ctypes.c_void_p(arg.ptr))
            elif arg is None:
                converted.append(ctypes.c_void_p(0))
            elif isinstance(arg, (np.integer, np.bool_)):
                converted.append(int(arg))
            elif isinstance(arg, np.floating):
                converted.append(float(arg))
            else:
                converted.append(arg)
        _lib.call(kernel.name, self._dev, self._s, *converted)

    _CTYPES = {
        np.dtype(np.int8): ctypes.c_int8, np.dtype(np.uint8): ctypes.c_uint8,
        np.dtype(np.int16): ctypes.c_int16, np.dtype(np.uint16): ctypes.c_uint16,
        np.dtype(np.int32): ctypes.c_int32, np.dtype(np.uint32): ctypes.c_uint32,
        np.dtype(np.int64): ctypes.c_int64, np.dtype(np.uint64): ctypes.c_uint64,
        np.dtype(np.float32): ctypes.c_float, np.dtype(np.float64): ctypes.c_double,
        np.dtype(np.bool_): ctypes.c_bool,
    }  # fmt: skip

    def _enqueue_rtc(self, kernel: RtcKernel, args, global_size, local_size) -> None:
        if global_size is None or local_size is None:
            raise ValueError("global_size and local_size are required for compiled kernels")
        if len(global_size) != len(local_size) or not 1 <= len(global_size) <= 3:
            raise ValueError("global_size and local_size must have 1 to 3 dimensions each")
        grid, block = [], []
        for g, l in zip(global_size, local_size):
            if l <= 0 or g % l != 0:
                raise ValueError("global size is not a multiple of the local size")
            grid.append(g // l)
            block.append(l)
        if any(n == 0 for n in grid):
            return  # nothing to do
        grid += [1] * (3 - len(grid))
        block += [1] * (3 - len(block))
        values = []
        for arg in args:
            if isinstance(arg, Buffer):
                values.append(ctypes.c_void_p(arg.ptr))
            elif isinstance(arg, RawBuffer):
                values.append(ctypes.c_void_p(arg.ptr))
            elif arg is None:
                values.append(ctypes.c_void_p(0))
            elif isinstance(arg, np.generic):
                if arg.dtype == np.complex64:
                    values.append((ctypes.c_float * 2)(arg.real, arg.imag))
                elif arg.dtype in self._CTYPES:
                    values.append(self._CTYPES[arg.dtype](arg.item()))
                else:
                    raise TypeError(f"cannot pass a {arg.dtype} scalar to a kernel")
            else:
                raise TypeError(
                    f"kernel argument {arg!r} is neither a device buffer nor a numpy scalar "
                    "(plain Python numbers have no definite C type)"
                )
        params = (ctypes.c_void_p * max(1, len(values)))(
            *[ctypes.cast(ctypes.pointer(v), ctypes.c_void_p) for v in values]
        )
        _lib.call("ksp_launch_function", self._dev, self._s, ctypes.c_void_p(kernel.handle),
                  (ctypes.c_uint * 3)(*grid), (ctypes.c_uint * 3)(*block), 0, params)  # fmt: skip

    # -- synchronisation
    def enqueue_marker(self, ordering_only: bool = False) -> Event:
        """Record an event behind the work enqueued so far. ``ordering_only`` events serve
        ``enqueue_wait_for_events`` of other queues of this device and nothing else (no
        ``time_since``, no hand-over to the host): recording them costs the stream no
        cache write-back."""
        handle = ctypes.c_void_p()
        _lib.call("ksp_event_create_ordering" if ordering_only else "ksp_event_create",
                  self._dev, ctypes.byref(handle))  # fmt: skip
        event = Event(self._dev, handle.value)
        _lib.call("ksp_event_record", self._dev, ctypes.c_void_p(event.handle), self._s)
        return event

    def create_event(self) -> Event:
        """An event that has not been recorded yet (for kernel-level profiling hooks)."""
        handle = ctypes.c_void_p()
        _lib.call("ksp_event_create", self._dev, ctypes.byref(handle))
        return Event(self._dev, handle.value)

    def enqueue_wait_for_events(self, events: Sequence[Event]) -> None:
        for event in events:
            _lib.call("ksp_stream_wait_event", self._dev, self._s, ctypes.c_void_p(event.handle))

    def flush(self) -> None:
        return None  # HIP submits work eagerly

    def finish(self) -> None:
        _lib.call("ksp_stream_synchronize", self._dev, self._s)
        self._keepalive.clear()

    def release_host_references(self) -> None:
        """Forget the host arrays of asynchronous copies enqueued so far. For callers that
        never :meth:`finish` a queue but know, from events they waited for, that those
        copies are complete (a staging pipeline that owns its pinned buffers anyway)."""
        self._keepalive.clear()


class TuningCommandQueue(CommandQueue, AbstractTuningCommandQueue):
    """Times everything enqueued between :meth:`start_tuning` and :meth:`stop_tuning`."""

    def __init__(self, context: Context) -> None:
        super().__init__(context, profile=True)
        self._start: Optional[Event] = None

    def start_tuning(self) -> None:
        self._start = self.enqueue_marker()

    def stop_tuning(self) -> float:
        end = self.enqueue_marker()
        self.finish()
        assert self._start is not None
        elapsed = end.time_since(self._start)
        self._start = None
        return elapsed
