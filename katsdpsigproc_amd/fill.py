"""Set every element of a device array, padding included, to one value.

Interface of the reference's ``fill`` module (reference: src/katsdpsigproc/fill.py:32-148:
``FillTemplate(context, dtype, ctype, tuning)``, ``instantiate(queue, shape, allocator)``, a
``data`` slot, ``set_value``); the kernel is the run-time compiled ``kernels/fill.hip.in``.
Unlike the reference's one-thread-per-element launch, a bounded number of workgroups walks the
array with a grid stride, and the workgroup size is tuned on the element type it will be
used with. Zeros are better written with :meth:`.DeviceArray.zero`.
"""

import math
from typing import Any, List, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from ._rtc_ops import RuntimeCompiledTemplate
from .abc import AbstractCommandQueue, AbstractContext

#: Bytes written by the tuning launches (well past the last-level cache)
_TUNE_BYTES = 512 << 20
#: Most workgroups one launch starts (a few rounds of the whole chip; the rest is grid stride)
_MAX_GROUPS = 4096


def _workgroup_sizes(context: AbstractContext) -> List[int]:
    """Candidate workgroup sizes: one to eight whole wavefronts."""
    return [context.device.simd_group_size << k for k in range(4)]


class FillTemplate(RuntimeCompiledTemplate):
    """
    Parameters
    ----------
    context
        Context for which the kernel is compiled
    dtype
        numpy type of the elements
    ctype
        The same type as spelled in HIP C++ (``"float"``, ``"unsigned char"``, ``"float2"`` ...)
    tuning
        ``wgs``: threads per workgroup; searched (and cached) when omitted
    """

    SOURCE = "fill.hip.in"
    TUNING_KEYS = ("wgs",)
    autotune_version = 2

    def __init__(self, context: AbstractContext, dtype, ctype: str,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        super().__init__(context, dtype, ctype, tuning, autotune_args=(dtype, ctype))

    def _check_tuning(self, wgs: int) -> None:
        if wgs not in range(1, 1025):
            raise ValueError("wgs must be between 1 and 1024")

    def _substitutions(self):
        return {"ctype": self.ctype}

    @classmethod
    @tune.autotuner(test={"wgs": 128})
    def autotune(cls, context: AbstractContext, dtype, ctype: str) -> Mapping[str, Any]:
        """Time the candidate workgroup sizes on an array of this element type."""
        n = max(1, _TUNE_BYTES // np.dtype(dtype).itemsize)
        scratch = accel.DeviceArray(context, (n,), dtype)
        timer = context.create_tuning_command_queue()

        def candidate(wgs: int):
            op = Fill(cls(context, dtype, ctype, tuning={"wgs": wgs}), timer, (n,))
            op.bind(data=scratch)
            return tune.make_measure(timer, op)

        return tune.autotune(candidate, wgs=_workgroup_sizes(context))


class Fill(accel.Operation):
    """A :class:`FillTemplate` bound to a command queue and a shape.

    .. rubric:: Slots

    **data** : `shape` -- written in full, padding elements as well
    """

    def __init__(self, template: FillTemplate, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, ...],
                 allocator: Optional[accel.AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.slots["data"] = accel.IOSlot(tuple(shape), template.dtype)
        self.kernel = template.program.get_kernel("fill")
        self.set_value(0)

    def set_value(self, value: Any) -> None:
        """Choose the value written by later calls (converted to the element type)."""
        self.value = np.asarray(value).astype(self.template.dtype)[()]

    def _run(self) -> None:
        target = self.buffer("data")
        count = math.prod(int(n) for n in target.padded_shape)
        if count >= 1 << 32:
            raise ValueError("array too large for the 32-bit element count of the fill kernel")
        wgs = self.template.wgs
        threads = wgs * min(accel.divup(count, wgs), _MAX_GROUPS)
        self.command_queue.enqueue_kernel(self.kernel, [target.buffer, np.uint32(count), self.value],
                                          global_size=(threads,), local_size=(wgs,))  # fmt: skip

    def parameters(self) -> Mapping[str, Any]:
        slot = self.slots["data"]
        return dict(dtype=self.template.dtype, ctype=self.template.ctype,
                    shape=slot.shape, value=self.value)  # type: ignore[attr-defined]  # fmt: skip


FillTemplate.OPERATION = Fill  # template.instantiate(command_queue, shape, allocator=None)
