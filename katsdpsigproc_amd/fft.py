"""Batched FFTs as operations, over hipFFT.

Counterpart of the reference's ``fft`` module (reference: src/katsdpsigproc/fft.py:205-422,
which drives cuFFT): the template bakes in the data shapes, because the library plans
for them; real-to-complex, complex-to-real and complex-to-complex transforms in single
or double precision over the last N dimensions, earlier dimensions being batches. The
transform is unnormalised: forward then inverse scales by the number of elements.
"""

import ctypes
import enum
import threading
import weakref
from typing import Any, Dict, Mapping, Optional, Tuple

import numpy as np

from . import _lib, accel, hip
from .abc import AbstractCommandQueue, AbstractContext


class FftMode(enum.Enum):
    FORWARD = 0
    INVERSE = 1


_TYPES = {
    (np.dtype(np.float32), np.dtype(np.complex64)): 0x2A,   # R2C
    (np.dtype(np.complex64), np.dtype(np.float32)): 0x2C,   # C2R
    (np.dtype(np.complex64), np.dtype(np.complex64)): 0x29,  # C2C
    (np.dtype(np.float64), np.dtype(np.complex128)): 0x6A,  # D2Z
    (np.dtype(np.complex128), np.dtype(np.float64)): 0x6C,  # Z2D
    (np.dtype(np.complex128), np.dtype(np.complex128)): 0x69,  # Z2Z
}  # fmt: skip


def _destroy_plan(device_index: int, plan: int) -> None:
    try:
        _lib.call("ksp_fft_plan_destroy", device_index, ctypes.c_void_p(plan))
    except Exception:
        pass


class FftTemplate:
    r"""
    Parameters
    ----------
    context
        Context (HIP backend) for the operation
    N
        Number of trailing dimensions to transform (1 to 3)
    shape
        Shape of the data, N or more dimensions; for real transforms the shape of the
        real side
    dtype_src, dtype_dest
        float32/complex64 or float64/complex128 in one of the combinations real->complex
        (forward only), complex->real (inverse only), complex->complex (either)
    padded_shape_src, padded_shape_dest
        Padded shapes of input and output; batch dimensions must not be padded, and the
        last dimension of the complex side of a real transform needs only
        :math:`\lfloor L/2 \rfloor + 1` elements
    tuning
        Unused (the library plans for itself), as in the reference
    """

    def __init__(self, context: AbstractContext, N: int, shape: Tuple[int, ...], dtype_src,
                 dtype_dest, padded_shape_src: Tuple[int, ...],
                 padded_shape_dest: Tuple[int, ...],
                 tuning: Optional[Dict[str, Any]] = None) -> None:  # fmt: skip
        if not isinstance(context, hip.Context):
            raise TypeError("Only HIP contexts are supported")
        shape = tuple(shape)
        padded_shape_src = tuple(padded_shape_src)
        padded_shape_dest = tuple(padded_shape_dest)
        if not 1 <= N <= min(3, len(shape)):
            raise ValueError("N must be between 1 and 3 and at most the number of dimensions")
        if len(padded_shape_src) != len(shape):
            raise ValueError("padded_shape_src and shape must have same length")
        if len(padded_shape_dest) != len(shape):
            raise ValueError("padded_shape_dest and shape must have same length")
        if padded_shape_src[:-N] != shape[:-N]:
            raise ValueError("Source must not be padded on batch dimensions")
        if padded_shape_dest[:-N] != shape[:-N]:
            raise ValueError("Destination must not be padded on batch dimensions")
        key = (np.dtype(dtype_src), np.dtype(dtype_dest))
        if key not in _TYPES:
            raise ValueError("Invalid combination of dtypes")
        self.context = context
        self.N = N
        self.shape = shape
        self.dtype_src, self.dtype_dest = key
        self.padded_shape_src = padded_shape_src
        self.padded_shape_dest = padded_shape_dest
        self._fft_type = _TYPES[key]
        arr = ctypes.c_longlong * N
        plan = ctypes.c_void_p()
        work_size = ctypes.c_size_t()
        _lib.call(
            "ksp_fft_plan_create", context.device.index, N, arr(*shape[-N:]),
            arr(*padded_shape_src[-N:]), int(np.prod(padded_shape_src[-N:])),
            arr(*padded_shape_dest[-N:]), int(np.prod(padded_shape_dest[-N:])),
            self._fft_type, int(np.prod(shape[:-N], dtype=np.int64)),
            ctypes.byref(plan), ctypes.byref(work_size),
        )  # fmt: skip
        self._plan = plan.value
        self._work_size = int(work_size.value)
        self._finalizer = weakref.finalize(self, _destroy_plan, context.device.index, self._plan)
        # stream and work area belong to the plan, not to one execution: serialise them
        self._lock = threading.RLock()

    def instantiate(self, command_queue: AbstractCommandQueue, mode: FftMode,
                    allocator: Optional[accel.AbstractAllocator] = None) -> "Fft":  # fmt: skip
        return Fft(self, command_queue, mode, allocator)


class Fft(accel.Operation):
    """Forward or inverse transform of a :class:`FftTemplate`.

    .. rubric:: Slots

    **src**, **dest** : input and output with exactly the template's padded shapes
    **work_area** : scratch bytes for the library (absent if it needs none); exposed so
        that it can be aliased with other scratch space
    """

    def __init__(self, template: FftTemplate, command_queue: AbstractCommandQueue, mode: FftMode,
                 allocator: Optional[accel.AbstractAllocator] = None) -> None:  # fmt: skip
        if not isinstance(command_queue, hip.CommandQueue):
            raise TypeError("Only HIP command queues are supported")
        super().__init__(command_queue, allocator)
        self.template = template
        src_shape = list(template.shape)
        dest_shape = list(template.shape)
        if template.dtype_src.kind != "c":
            if mode != FftMode.FORWARD:
                raise ValueError("R2C transform must use FftMode.FORWARD")
            dest_shape[-1] = template.shape[-1] // 2 + 1
        if template.dtype_dest.kind != "c":
            if mode != FftMode.INVERSE:
                raise ValueError("C2R transform must use FftMode.INVERSE")
            src_shape[-1] = template.shape[-1] // 2 + 1
        self.slots["src"] = accel.IOSlot(
            tuple(accel.Dimension(n, min_padded_size=p, exact=True)
                  for n, p in zip(src_shape, template.padded_shape_src)),
            template.dtype_src,
        )  # fmt: skip
        self.slots["dest"] = accel.IOSlot(
            tuple(accel.Dimension(n, min_padded_size=p, exact=True)
                  for n, p in zip(dest_shape, template.padded_shape_dest)),
            template.dtype_dest,
        )  # fmt: skip
        if template._work_size > 0:
            self.slots["work_area"] = accel.IOSlot((template._work_size,), np.uint8)
        self.mode = mode

    def _run(self) -> None:
        src = self.buffer("src")
        dest = self.buffer("dest")
        work = self.buffer("work_area") if "work_area" in self.slots else None
        queue = self.command_queue
        with self.template._lock:
            _lib.call(
                "ksp_fft_exec", queue.context.device.index, ctypes.c_void_p(queue.stream),
                ctypes.c_void_p(self.template._plan), self.template._fft_type,
                ctypes.c_void_p(src.buffer.ptr), ctypes.c_void_p(dest.buffer.ptr),
                ctypes.c_void_p(work.buffer.ptr) if work is not None else None,
                int(self.mode == FftMode.INVERSE),
            )  # fmt: skip

    def parameters(self) -> Mapping[str, Any]:
        return {
            "N": self.template.N,
            "shape": self.template.shape,
            "dtype_src": self.template.dtype_src,
            "dtype_dest": self.template.dtype_dest,
            "padded_shape_src": self.template.padded_shape_src,
            "padded_shape_dest": self.template.padded_shape_dest,
            "mode": self.mode.name,
        }
