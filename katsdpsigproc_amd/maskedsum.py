"""Masked column sums (reference: src/katsdpsigproc/maskedsum.py:33-162)."""

from typing import Any, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from .abc import AbstractCommandQueue, AbstractContext


class MaskedSumTemplate:
    """``dest[col] = sum_row mask[row] * src[row, col]`` (or of ``abs(src)``).

    Parameters
    ----------
    context
        Context whose device will run the kernel
    use_amplitudes
        Sum amplitudes (float32 result) instead of complex values (complex64 result)
    tuning
        The kernel's geometry is fixed: ``size`` of the reference is accepted without effect,
        any other key is a ``ValueError`` (:func:`.tune.fixed_geometry`).
    """

    TUNING_KEYS = ("size",)

    def __init__(self, context: AbstractContext, use_amplitudes: bool = False,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.context = context
        self.use_amplitudes = use_amplitudes
        self.tuning = tune.fixed_geometry("MaskedSumTemplate", tuning, self.TUNING_KEYS)
        self.kernel = context.native_kernel("ksp_maskedsum_float")

    @classmethod
    def autotune(cls, context: AbstractContext, use_amplitudes: bool) -> Mapping[str, Any]:
        """Nothing to search (reference maskedsum.py:73 times size)."""
        return {}

    def instantiate(self, command_queue: AbstractCommandQueue, shape: Tuple[int, int],
                    allocator: Optional[accel.AbstractAllocator] = None) -> "MaskedSum":  # fmt: skip
        return MaskedSum(self, command_queue, shape, allocator)


class MaskedSum(accel.Operation):
    """Concrete :class:`MaskedSumTemplate`.

    .. rubric:: Slots

    **src** : rows x columns, complex64
    **mask** : rows, float32
    **dest** : columns, complex64 (float32 if ``use_amplitudes``)
    """

    def __init__(self, template: MaskedSumTemplate, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, int], allocator: Optional[accel.AbstractAllocator] = None):  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.kernel
        self.shape = tuple(shape)
        self.slots["src"] = accel.IOSlot((shape[0], accel.Dimension(shape[1])), np.complex64)
        self.slots["mask"] = accel.IOSlot((shape[0],), np.float32)
        self.slots["dest"] = accel.IOSlot(
            (accel.Dimension(shape[1]),),
            np.float32 if template.use_amplitudes else np.complex64,
        )

    def _run(self) -> None:
        src = self.buffer("src")
        mask = self.buffer("mask")
        dest = self.buffer("dest")
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                src.buffer,
                mask.buffer,
                dest.buffer,
                np.int32(src.padded_shape[1]),
                np.int32(src.shape[0]),
                np.int32(src.shape[1]),
                np.int32(self.template.use_amplitudes),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "shape": self.slots["src"].shape,
            "use_amplitudes": self.template.use_amplitudes,
        }
