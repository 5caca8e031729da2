"""2-D transpose operation (reference: src/katsdpsigproc/transpose.py:39-174)."""

from typing import Any, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from .abc import AbstractCommandQueue, AbstractContext


class TransposeTemplate:
    """Transpose a 2-D array of any 1/2/4/8/16-byte element type.

    Parameters
    ----------
    context
        Context whose device will run the kernel
    dtype
        Element type
    ctype
        C name of the element type. Unused (kernels are not generated from source) but
        accepted so that reference call sites work unchanged (transpose.py:58-64).
    tuning
        The 64x64 LDS tile of the gfx950 kernel is fixed: ``block``/``vtx``/``vty`` of the
        reference are accepted without effect, any other key is a ``ValueError``
        (:func:`.tune.fixed_geometry`).
    """

    TUNING_KEYS = ("block", "vtx", "vty")

    def __init__(self, context: AbstractContext, dtype, ctype: str = "",
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.context = context
        self.dtype = np.dtype(dtype)
        self.ctype = ctype
        if self.dtype.itemsize not in (1, 2, 4, 8, 16):
            raise ValueError(f"unsupported element size {self.dtype.itemsize}")
        self.tuning = tune.fixed_geometry("TransposeTemplate", tuning, self.TUNING_KEYS)
        self.kernel = context.native_kernel("ksp_transpose")

    @classmethod
    def autotune(cls, context: AbstractContext, dtype, ctype: str) -> Mapping[str, Any]:
        """Nothing to search (reference transpose.py:87-108 times block/vtx/vty)."""
        return {}

    def instantiate(self, command_queue: AbstractCommandQueue, shape: Tuple[int, int],
                    allocator: Optional[accel.AbstractAllocator] = None) -> "Transpose":  # fmt: skip
        return Transpose(self, command_queue, shape, allocator)


class Transpose(accel.Operation):
    """Concrete transpose.

    .. rubric:: Slots

    **src** : shape, input
    **dest** : shape reversed, output
    """

    def __init__(self, template: TransposeTemplate, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, int], allocator: Optional[accel.AbstractAllocator] = None):  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.kernel
        self.shape = tuple(shape)
        self.slots["src"] = accel.IOSlot(shape, template.dtype)
        self.slots["dest"] = accel.IOSlot((shape[1], shape[0]), template.dtype)

    def _run(self) -> None:
        src = self.buffer("src")
        dest = self.buffer("dest")
        # argument order of the reference kernel call (transpose.py:152-167)
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                dest.buffer,
                src.buffer,
                np.int32(src.shape[0]),
                np.int32(src.shape[1]),
                np.int32(dest.padded_shape[1]),
                np.int32(src.padded_shape[1]),
                np.int32(self.template.dtype.itemsize),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "dtype": self.template.dtype,
            "ctype": self.template.ctype,
            "shape": self.slots["src"].shape,
        }
