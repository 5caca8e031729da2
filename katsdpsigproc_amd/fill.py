"""Fill a device array (padding included) with a constant.

Counterpart of the reference's ``fill`` module (reference: src/katsdpsigproc/fill.py:32-148);
like there, the kernel is a source template compiled at run time for the element type
(``accel.build`` -> hiprtc). To fill with zeros use :meth:`.DeviceArray.zero`.
"""

from typing import Any, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from .abc import AbstractCommandQueue, AbstractContext


class FillTemplate:
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
        ``wgs``: threads per workgroup (default: autotuned and cached)
    """

    autotune_version = 1

    def __init__(self, context: AbstractContext, dtype, ctype: str,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.context = context
        self.dtype = np.dtype(dtype)
        self.ctype = ctype
        if tuning is None:
            tuning = self.autotune(context, dtype, ctype)
        self.wgs = int(tuning["wgs"])
        self.program = accel.build(context, "fill.hip.in", {"wgs": self.wgs, "ctype": ctype})

    @classmethod
    @tune.autotuner(test={"wgs": 128})
    def autotune(cls, context: AbstractContext, dtype, ctype: str) -> Mapping[str, Any]:
        queue = context.create_tuning_command_queue()
        shape = (1048576,)
        data = accel.DeviceArray(context, shape, dtype=dtype)

        def generate(wgs: int):
            fn = cls(context, dtype, ctype, {"wgs": wgs}).instantiate(queue, shape)
            fn.bind(data=data)
            return tune.make_measure(queue, fn)

        return tune.autotune(generate, wgs=[64, 128, 256, 512])

    def instantiate(self, command_queue: AbstractCommandQueue, shape: Tuple[int, ...],
                    allocator: Optional[accel.AbstractAllocator] = None) -> "Fill":  # fmt: skip
        return Fill(self, command_queue, shape, allocator)


class Fill(accel.Operation):
    """Concrete :class:`FillTemplate`.

    .. rubric:: Slots

    **data** : any shape -- the array to fill; its padding is filled too
    """

    def __init__(self, template: FillTemplate, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, ...],
                 allocator: Optional[accel.AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.program.get_kernel("fill")
        self.shape = tuple(shape)
        self.slots["data"] = accel.IOSlot(self.shape, template.dtype)
        self.value = template.dtype.type()

    def set_value(self, value: Any) -> None:
        self.value = self.template.dtype.type(value)

    def _run(self) -> None:
        data = self.buffer("data")
        elements = int(np.prod(data.padded_shape))
        self.command_queue.enqueue_kernel(
            self.kernel,
            [data.buffer, np.uint32(elements), self.value],
            global_size=(accel.roundup(elements, self.template.wgs),),
            local_size=(self.template.wgs,),
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "dtype": self.template.dtype,
            "ctype": self.template.ctype,
            "shape": self.shape,
            "value": self.value,
        }
