"""Row-wise reduction of a 2-D array with a caller-supplied operator.

Counterpart of the reference's ``reduce`` module (reference: src/katsdpsigproc/reduce.py:22-214,
kernel hreduce.mako:51-84): the operator is a C expression in ``a`` and ``b`` pasted into
a source template that is compiled at run time (``accel.build`` -> hiprtc). Only
commutative, associative operators are supported, as in the reference.
"""

from typing import Any, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from .abc import AbstractCommandQueue, AbstractContext


class HReduceTemplate:
    """
    Parameters
    ----------
    context
        Context for which the kernel is compiled
    dtype, ctype
        Element type as numpy and as HIP C++ spell it
    op
        C expression combining the variables ``a`` and ``b``, e.g. ``"a + b"``, ``"max(a, b)"``
    identity
        C expression for the identity of `op`
    extra_code
        Any C++ code `op` or `identity` need (helper functions, typedefs)
    tuning
        ``wgsx``: threads per data row (a power of two up to 1024); ``wgsy``: rows per
        workgroup; ``wgsx * wgsy`` between 32 and 1024 (default: autotuned and cached)
    """

    autotune_version = 1

    def __init__(self, context: AbstractContext, dtype, ctype: str, op: str, identity: str,
                 extra_code: str = "", tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.context = context
        self.dtype = np.dtype(dtype)
        self.ctype = ctype
        if tuning is None:
            tuning = self.autotune(context, dtype, ctype, op, identity, extra_code)
        self.wgsx = int(tuning["wgsx"])
        self.wgsy = int(tuning["wgsy"])
        if self.wgsx & (self.wgsx - 1) or not 1 <= self.wgsx <= 1024:
            raise ValueError("wgsx must be a power of two between 1 and 1024")
        if not 1 <= self.wgsx * self.wgsy <= 1024:
            raise ValueError("wgsx * wgsy must be at most 1024")
        self.op = op
        self.identity = identity
        self.extra_code = extra_code
        self.program = accel.build(
            context, "hreduce.hip.in",
            {"wgsx": self.wgsx, "wgsy": self.wgsy, "type": ctype, "op": op,
             "identity": identity, "extra_code": extra_code},
        )  # fmt: skip

    @classmethod
    @tune.autotuner(test={"wgsx": 64, "wgsy": 4})
    def autotune(cls, context: AbstractContext, dtype, ctype: str, op: str, identity: str,
                 extra_code: str) -> Mapping[str, Any]:  # fmt: skip
        queue = context.create_tuning_command_queue()
        shape = (2048, 1024)
        src = accel.DeviceArray(context, shape, dtype=dtype)
        dest = accel.DeviceArray(context, (shape[0],), dtype=dtype)

        def generate(wgsx: int, wgsy: int):
            if not 32 <= wgsx * wgsy <= 1024:
                return None
            template = cls(context, dtype, ctype, op, identity, extra_code,
                           {"wgsx": wgsx, "wgsy": wgsy})  # fmt: skip
            fn = template.instantiate(queue, shape)
            fn.bind(src=src, dest=dest)
            return tune.make_measure(queue, fn)

        return tune.autotune(generate, wgsx=[32, 64, 128], wgsy=[1, 2, 4, 8, 16])

    def instantiate(self, command_queue: AbstractCommandQueue, shape: Tuple[int, int],
                    column_range: Optional[Tuple[int, int]] = None,
                    allocator: Optional[accel.AbstractAllocator] = None) -> "HReduce":  # fmt: skip
        return HReduce(self, command_queue, shape, column_range, allocator)


class HReduce(accel.Operation):
    """Concrete :class:`HReduceTemplate`: in every row, the elements of the column range
    are combined with the template's operator.

    .. rubric:: Slots

    **src** : rows x columns -- input (rows padded to a multiple of ``wgsy``)
    **dest** : rows -- one reduced value per row
    """

    def __init__(self, template: HReduceTemplate, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, int], column_range: Optional[Tuple[int, int]] = None,
                 allocator: Optional[accel.AbstractAllocator] = None) -> None:  # fmt: skip
        if len(shape) != 2:
            raise ValueError("shape must be 2-dimensional")
        if column_range is None:
            column_range = (0, shape[1])
        if column_range[0] < 0 or column_range[1] > shape[1]:
            raise ValueError("column range overflows the array")
        if column_range[0] >= column_range[1]:
            raise ValueError("column range is empty")
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.program.get_kernel("hreduce")
        self.shape = tuple(shape)
        self.column_range = tuple(column_range)
        rows = accel.Dimension(shape[0], template.wgsy)
        self.slots["src"] = accel.IOSlot((rows, shape[1]), template.dtype)
        self.slots["dest"] = accel.IOSlot((accel.Dimension(shape[0], template.wgsy),),
                                          template.dtype)  # fmt: skip

    def _run(self) -> None:
        src = self.buffer("src")
        dest = self.buffer("dest")
        rows = accel.roundup(self.shape[0], self.template.wgsy)
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                src.buffer,
                dest.buffer,
                np.int32(self.column_range[0]),
                np.int32(self.column_range[1] - self.column_range[0]),
                np.int32(src.padded_shape[1]),
            ],
            global_size=(self.template.wgsx, rows),
            local_size=(self.template.wgsx, self.template.wgsy),
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "dtype": self.template.dtype,
            "ctype": self.template.ctype,
            "shape": self.shape,
            "column_range": self.column_range,
            "op": self.template.op,
            "identity": self.template.identity,
            "extra_code": self.template.extra_code,
        }
