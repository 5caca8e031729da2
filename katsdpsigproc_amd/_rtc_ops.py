"""What the operation templates built on run-time compiled kernels share (fill, reduce).

A template of this kind is described by class attributes -- the kernel source file, the names
of its tuning parameters -- and two hooks: :meth:`_check_tuning` (validate a tuning mapping)
and :meth:`_substitutions` (the template keys of the source file). The base class resolves the
``tuning`` argument (given, or searched and cached through the subclass's ``autotune``), keeps
the tuning values as attributes, compiles the program, and ``instantiate`` constructs the
operation class named by ``OPERATION``.
"""

from typing import Any, Callable, ClassVar, Dict, Mapping, Optional, Tuple

import numpy as np

from . import accel
from .abc import AbstractCommandQueue, AbstractContext


class RuntimeCompiledTemplate:
    SOURCE: ClassVar[str]  #: file under ``kernels/``
    TUNING_KEYS: ClassVar[Tuple[str, ...]]  #: names of the integer tuning parameters
    OPERATION: ClassVar[Callable[..., accel.Operation]]  #: the operation class (set after it)

    def __init__(self, context: AbstractContext, dtype, ctype: str,
                 tuning: Optional[Mapping[str, Any]], autotune_args: Tuple[Any, ...]) -> None:  # fmt: skip
        self.context = context
        self.dtype = np.dtype(dtype)
        self.ctype = ctype
        if tuning is None:
            tuning = type(self).autotune(context, *autotune_args)  # type: ignore[attr-defined]
        stray = sorted(set(tuning) - set(self.TUNING_KEYS))
        if stray:
            raise ValueError(f"unknown tuning parameters {stray} (known: {list(self.TUNING_KEYS)})")
        values = {key: int(tuning[key]) for key in self.TUNING_KEYS}
        self._check_tuning(**values)
        for key, value in values.items():
            setattr(self, key, value)
        self.program = accel.build(context, self.SOURCE, {**values, **self._substitutions()})

    def instantiate(self, command_queue: AbstractCommandQueue, *args: Any, **kwargs: Any):
        """The operation for `command_queue`; the other arguments are the operation class's."""
        return self.OPERATION(self, command_queue, *args, **kwargs)

    def _check_tuning(self, **values: int) -> None:
        raise NotImplementedError

    def _substitutions(self) -> Dict[str, Any]:
        raise NotImplementedError
