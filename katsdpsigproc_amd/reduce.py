"""Reduce every row of a 2-D array (or a column range of it) with a caller-supplied operator.

Interface of the reference's ``reduce`` module (reference: src/katsdpsigproc/reduce.py:22-214,
kernel hreduce.mako:51-84): the operator is a C expression in ``a`` and ``b`` and, as there,
must be commutative and associative; it is pasted into ``kernels/hreduce.hip.in`` and compiled
at run time. The kernel combines a row's partial values with wavefront shuffles (one LDS word
per wavefront only when a row's threads span several wavefronts) instead of the reference's
LDS rake, and the workgroup shape is searched over what that kernel supports: ``wgsx`` threads
along a row (a power of two), ``wgsy`` rows per workgroup.
"""

import itertools
from typing import Any, Iterator, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from ._rtc_ops import RuntimeCompiledTemplate
from .abc import AbstractCommandQueue, AbstractContext

_MAX_THREADS = 1024


def _check_geometry(wgsx: int, wgsy: int) -> None:
    """The shapes kernels/hreduce.hip.in is written for."""
    if wgsx < 1 or wgsx > _MAX_THREADS or (wgsx & (wgsx - 1)) != 0:
        raise ValueError(f"wgsx must be a power of two, at most {_MAX_THREADS} (got {wgsx})")
    if wgsy < 1 or wgsx * wgsy > _MAX_THREADS:
        raise ValueError(f"a workgroup has at most {_MAX_THREADS} threads (got {wgsx} x {wgsy})")


def _candidate_geometries(wave: int) -> Iterator[Tuple[int, int]]:
    """(wgsx, wgsy) worth timing: half a wavefront to two wavefronts along a row, workgroups
    of one to sixteen wavefronts."""
    for wgsx, wgsy in itertools.product((wave // 2, wave, 2 * wave), (1, 2, 4, 8, 16)):
        if wave // 2 <= wgsx * wgsy <= 16 * wave and wgsx * wgsy <= _MAX_THREADS:
            yield wgsx, wgsy


class HReduceTemplate(RuntimeCompiledTemplate):
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
        ``wgsx`` and ``wgsy`` (see the module description); searched and cached when omitted
    """

    SOURCE = "hreduce.hip.in"
    TUNING_KEYS = ("wgsx", "wgsy")
    autotune_version = 2

    def __init__(self, context: AbstractContext, dtype, ctype: str, op: str, identity: str,
                 extra_code: str = "", tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        self.op, self.identity, self.extra_code = op, identity, extra_code
        super().__init__(context, dtype, ctype, tuning,
                         autotune_args=(dtype, ctype, op, identity, extra_code))  # fmt: skip

    def _check_tuning(self, wgsx: int, wgsy: int) -> None:
        _check_geometry(wgsx, wgsy)

    def _substitutions(self):
        return dict(type=self.ctype, op=self.op, identity=self.identity, extra_code=self.extra_code)

    @classmethod
    @tune.autotuner(test={"wgsx": 64, "wgsy": 4})
    def autotune(cls, context: AbstractContext, dtype, ctype: str, op: str, identity: str,
                 extra_code: str) -> Mapping[str, Any]:  # fmt: skip
        """Time the candidate workgroup shapes on a 64 MiB array with rows of 4096 elements (a
        reduction's cost per row depends on the shape far more than on the operator)."""
        queue = context.create_tuning_command_queue()
        columns = 4096
        rows = max(256, (64 << 20) // (columns * np.dtype(dtype).itemsize))
        wave = context.device.simd_group_size

        def trial(geometry: Tuple[int, int]):
            wgsx, wgsy = geometry
            template = cls(context, dtype, ctype, op, identity, extra_code,
                           tuning={"wgsx": wgsx, "wgsy": wgsy})  # fmt: skip
            fn = template.instantiate(queue, (rows, columns))
            fn.ensure_all_bound()  # (each shape pads the rows to its own multiple)
            return tune.make_measure(queue, fn)

        best = tune.autotune(trial, geometry=list(_candidate_geometries(wave)))
        wgsx, wgsy = best["geometry"]
        return {"wgsx": wgsx, "wgsy": wgsy}


class HReduce(accel.Operation):
    """A :class:`HReduceTemplate` bound to a command queue, a shape and a column range: in
    every row the elements of the range are combined with the template's operator.

    .. rubric:: Slots

    **src** : rows x columns -- input (rows padded to a multiple of ``wgsy``)
    **dest** : rows -- one reduced value per row (padded likewise)
    """

    def __init__(self, template: HReduceTemplate, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, int], column_range: Optional[Tuple[int, int]] = None,
                 allocator: Optional[accel.AbstractAllocator] = None) -> None:  # fmt: skip
        if len(shape) != 2:
            raise ValueError(f"HReduce works on 2-D arrays, not on shape {tuple(shape)}")
        rows, columns = int(shape[0]), int(shape[1])
        first, last = (0, columns) if column_range is None else map(int, column_range)
        if not 0 <= first < last <= columns:
            raise ValueError(f"columns [{first}, {last}) are not a non-empty range inside "
                             f"[0, {columns})")  # fmt: skip
        super().__init__(command_queue, allocator)
        self.template = template
        self.kernel = template.program.get_kernel("hreduce")
        self.column_range = (first, last)
        # whole workgroups of rows: the kernel has no row bound to test
        self.slots["src"] = accel.IOSlot((accel.Dimension(rows, template.wgsy), columns),
                                         template.dtype)  # fmt: skip
        self.slots["dest"] = accel.IOSlot((accel.Dimension(rows, template.wgsy),), template.dtype)

    def _run(self) -> None:
        src, dest = self.buffer("src"), self.buffer("dest")
        wgsx, wgsy = self.template.wgsx, self.template.wgsy
        first, last = self.column_range
        scalars = [np.int32(first), np.int32(last - first), np.int32(src.padded_shape[1])]
        self.command_queue.enqueue_kernel(
            self.kernel, [src.buffer, dest.buffer] + scalars,
            global_size=(wgsx, accel.roundup(src.shape[0], wgsy)), local_size=(wgsx, wgsy))  # fmt: skip

    def parameters(self) -> Mapping[str, Any]:
        template = self.template
        described = {name: getattr(template, name)
                     for name in ("dtype", "ctype", "op", "identity", "extra_code")}  # fmt: skip
        described["shape"] = self.slots["src"].shape  # type: ignore[attr-defined]
        described["column_range"] = self.column_range
        return described


# template.instantiate(command_queue, shape, column_range=None, allocator=None)
HReduceTemplate.OPERATION = HReduce
