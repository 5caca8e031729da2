"""Per-row five-percentile operation (reference: src/katsdpsigproc/percentile.py:34-217)."""

from typing import Any, Mapping, Optional, Tuple

import numpy as np

from . import accel, tune
from .abc import AbstractCommandQueue, AbstractContext

#: the gfx950 kernel keeps a row in registers: 256 work-items x up to 64 values
MAX_COLUMNS_SUPPORTED = 256 * 64


class Percentile5Template:
    """Percentiles [0, 100, 25, 75, 50] of each row, "lower" element, no interpolation.

    WARNING: assumes all values are positive (as the reference does).

    Parameters
    ----------
    context
        Context whose device will run the kernel
    max_columns
        Upper bound on the number of columns processed per row
    is_amplitude
        True: float32 amplitudes in; False: complex64 in, statistics of ``abs``
    tuning
        The kernels' geometry is fixed (a wavefront or a 256-thread workgroup per row, chosen
        from the number of columns): ``size``/``wgsy`` of the reference are accepted without
        effect, any other key is a ``ValueError`` (:func:`.tune.fixed_geometry`).
    """

    TUNING_KEYS = ("size", "wgsy")

    def __init__(self, context: AbstractContext, max_columns: int, is_amplitude: bool = True,
                 tuning: Optional[Mapping[str, Any]] = None) -> None:  # fmt: skip
        if max_columns > MAX_COLUMNS_SUPPORTED:
            raise ValueError(f"max_columns exceeds {MAX_COLUMNS_SUPPORTED}")
        self.context = context
        self.max_columns = max_columns
        self.is_amplitude = is_amplitude
        self.tuning = tune.fixed_geometry("Percentile5Template", tuning, self.TUNING_KEYS)
        self.kernel = context.native_kernel("ksp_percentile5_float")

    @classmethod
    def autotune(cls, context, max_columns: int, is_amplitude: bool) -> Mapping[str, Any]:
        """Nothing to search (reference percentile.py:88-121 times size/wgsy)."""
        return {}

    def instantiate(self, command_queue: AbstractCommandQueue, shape: Tuple[int, int],
                    column_range: Optional[Tuple[int, int]] = None,
                    allocator: Optional[accel.AbstractAllocator] = None) -> "Percentile5":  # fmt: skip
        return Percentile5(self, command_queue, shape, column_range, allocator)


class Percentile5(accel.Operation):
    """Concrete :class:`Percentile5Template`.

    .. rubric:: Slots

    **src** : rows x columns, float32 or complex64
    **dest** : 5 x rows, float32

    Raises ValueError for an empty column range or one wider than ``max_columns``,
    IndexError for a range outside the array (reference percentile.py:174-180).
    """

    def __init__(self, template: Percentile5Template, command_queue: AbstractCommandQueue,
                 shape: Tuple[int, int], column_range: Optional[Tuple[int, int]],
                 allocator: Optional[accel.AbstractAllocator] = None) -> None:  # fmt: skip
        super().__init__(command_queue, allocator)
        if column_range is None:
            column_range = (0, shape[1])
        if column_range[1] <= column_range[0]:
            raise ValueError("column range is empty")
        if column_range[0] < 0 or column_range[1] > shape[1]:
            raise IndexError("column range is out of range")
        if column_range[1] - column_range[0] > template.max_columns:
            raise ValueError("columns exceeds max_columns")
        self.template = template
        self.kernel = template.kernel
        self.shape = tuple(shape)
        self.column_range = tuple(column_range)
        src_type = np.float32 if template.is_amplitude else np.complex64
        row_dim = accel.Dimension(shape[0])
        col_dim = accel.Dimension(shape[1])
        self.slots["src"] = accel.IOSlot((row_dim, col_dim), src_type)
        self.slots["dest"] = accel.IOSlot((5, row_dim), np.float32)

    def _run(self) -> None:
        src = self.buffer("src")
        dest = self.buffer("dest")
        self.command_queue.enqueue_kernel(
            self.kernel,
            [
                src.buffer,
                dest.buffer,
                np.int32(src.shape[0]),
                np.int32(src.padded_shape[1]),
                np.int32(dest.padded_shape[1]),
                np.int32(self.column_range[0]),
                np.int32(self.column_range[1] - self.column_range[0]),
                np.int32(self.template.is_amplitude),
            ],
        )

    def parameters(self) -> Mapping[str, Any]:
        return {
            "max_columns": self.template.max_columns,
            "is_amplitude": self.template.is_amplitude,
            "shape": self.slots["src"].shape,
            "column_range": self.column_range,
        }
