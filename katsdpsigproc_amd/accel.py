"""Arrays, padding rules, slots and operations: the template -> instantiate -> bind -> call API.

This module keeps the contract of the reference's ``accel`` module for everything the
RFI path touches (reference: src/katsdpsigproc/accel.py -- ``HostArray`` 368-456,
``DeviceArray`` 459-924, allocators 1057-1093, ``Dimension`` 1115-1294, slots
1297-1608, ``Operation`` 1611-1756, ``OperationSequence`` 1759-1835), written from that
contract rather than from its code. Differences, all deliberate:

* The hot-path kernels are compiled ahead of time into a C-ABI library and operations
  obtain them from :mod:`katsdpsigproc_amd.hip`; :func:`build` (template rendering +
  run-time compilation through hiprtc, reference accel.py:165-208) serves the templated
  utility operations and user kernels.
* ``SVMArray`` / ``SVMAllocator`` and ``visualize_operation`` are outside the hot path
  and not provided.
"""

import os
from abc import ABC, abstractmethod
from collections import OrderedDict
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from .abc import AbstractCommandQueue, AbstractContext, AbstractDevice


def divup(x: int, y: int) -> int:
    """ceil(x / y) for non-negative integers."""
    return -(-x // y)


def roundup(x: int, y: int) -> int:
    """Smallest multiple of y that is >= x."""
    return divup(x, y) * y


KERNEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels")


def render_template(name, render_kws=None, extra_dirs=None, source=None) -> str:
    """Text of kernel template `name` with `render_kws` filled in.

    Templates are looked up in `extra_dirs`, then in the package's ``kernels/``
    directory. ``*.mako`` files are rendered by Mako when it is installed (the
    reference's format, accel.py:165-208); everything else uses ``${name}`` placeholders
    (:class:`string.Template`), which needs no third-party package. ``simd_group_size``
    (64 on gfx950) is always defined, as in the reference (accel.py:207). With `source`
    the text is taken from there instead of a file.
    """
    import string

    keys = dict(render_kws or {})
    dirs = [os.fspath(d) for d in (extra_dirs or [])] + [KERNEL_DIR]
    if name.endswith(".mako") and source is None:
        try:
            from mako.lookup import TemplateLookup
        except ImportError:
            raise RuntimeError(
                f"{name}: Mako is not installed; use a ${{name}}-style template "
                "(kernels/*.hip.in) or install Mako"
            ) from None
        return TemplateLookup(dirs, strict_undefined=True).get_template(name).render(**keys)
    if source is None:
        for directory in dirs:
            path = os.path.join(directory, name)
            if os.path.exists(path):
                with open(path) as f:
                    source = f.read()
                break
        else:
            raise FileNotFoundError(f"kernel template {name} not found in {dirs}")
    try:
        return string.Template(source).substitute(keys)
    except KeyError as exc:
        raise KeyError(f"{name}: template parameter {exc} was not supplied") from None


def build(context, name, render_kws=None, extra_dirs=None, extra_flags=None, source=None):
    """Render kernel template `name` and compile it for `context` (reference
    accel.py:165-208): returns a program whose ``get_kernel(name)`` gives kernels for
    ``command_queue.enqueue_kernel(kernel, args, global_size, local_size)``.

    The kernels of the RFI hot path do not come this way -- they are compiled ahead of
    time into the native library; this is the route for the templated utility
    operations (:mod:`~katsdpsigproc_amd.fill`, :mod:`~katsdpsigproc_amd.reduce`) and
    for downstream kernels. ``kernels/port.h`` supplies the reference's portability
    vocabulary (``KERNEL``, ``LOCAL_DECL``, ``BARRIER()``, ``get_global_id`` ...,
    port.mako:19-116) so that such kernels keep their spelling.
    """
    keys = dict(render_kws or {})
    keys.setdefault("simd_group_size", context.device.simd_group_size)
    text = render_template(name, keys, extra_dirs, source)
    flags = list(extra_flags or [])
    for directory in list(extra_dirs or []) + [KERNEL_DIR]:
        flags.append("-I" + os.fspath(directory))
    return context.compile(text, flags)


# --------------------------------------------------------------------------- devices
def all_devices() -> List[AbstractDevice]:
    """Every device of the HIP backend."""
    from . import hip

    return list(hip.Device.get_devices())


def candidate_devices(
    device_filter: Optional[Callable[[AbstractDevice], bool]] = None,
) -> Sequence[AbstractDevice]:
    """Devices eligible for :func:`create_some_context`.

    ``KATSDPSIGPROC_DEVICE`` (a device number) narrows the choice to one device, as in
    the reference (accel.py:264-299; ``CUDA_DEVICE``/``PYOPENCL_CTX`` have no meaning
    here). An out-of-range number raises :class:`RuntimeError`.
    """
    devices: List[AbstractDevice] = all_devices()
    if device_filter is not None:
        devices = [d for d in devices if device_filter(d)]
    text = os.environ.get("KATSDPSIGPROC_DEVICE")
    if text is not None:
        try:
            index = int(text)
        except ValueError:
            index = -1
        if index >= 0:
            if index >= len(devices):
                raise RuntimeError("Out-of-range device selected")
            devices = [devices[index]]
    return devices


def create_some_context(
    interactive: bool = True,
    device_filter: Optional[Callable[[AbstractDevice], bool]] = None,
) -> AbstractContext:
    """Make a context on the first suitable device (reference accel.py:302-365).

    `interactive` is accepted for signature compatibility; the choice is never prompted
    for. Raises :class:`RuntimeError` when there is no device.
    """
    devices = candidate_devices(device_filter)
    if not devices:
        raise RuntimeError("No compute devices found")
    return devices[0].make_context()


# ---------------------------------------------------------------------------- arrays
class HostArray(np.ndarray):
    """Host array with optional padding that can be copied to a :class:`DeviceArray` as is.

    The array is the origin-anchored ``[:shape]`` window of a contiguous allocation of
    `padded_shape` elements; only the array returned by the constructor (not views of
    it) is trusted for direct transfers (reference accel.py:368-456).
    """

    def __new__(cls, shape, dtype, padded_shape=None, context: Optional[AbstractContext] = None):
        shape = tuple(shape)
        padded_shape = shape if padded_shape is None else tuple(padded_shape)
        assert len(padded_shape) == len(shape)
        assert all(p >= s for p, s in zip(padded_shape, shape))
        if context is None:
            storage = np.empty(padded_shape, dtype)
        else:
            storage = context.allocate_pinned(padded_shape, dtype)
        window = storage[tuple(slice(0, n) for n in shape)] if shape else storage
        self = window.view(cls)
        self._owner = storage
        self.padded_shape = padded_shape
        return self

    def __array_finalize__(self, obj):
        if obj is None:
            return
        # views and copies are not anchored at the origin of known storage
        self._owner = None
        self.padded_shape = getattr(obj, "padded_shape", None) if isinstance(obj, HostArray) else None

    @classmethod
    def safe(cls, obj) -> bool:
        """True if `obj` can be transferred without an intermediate copy."""
        return getattr(obj, "_owner", None) is not None

    @classmethod
    def padded_view(cls, obj) -> Optional[np.ndarray]:
        """The whole padded allocation behind `obj`, or None if `obj` is not :meth:`safe`."""
        return getattr(obj, "_owner", None)


class DeviceArray:
    """C-ordered, possibly padded array in device memory (reference accel.py:459-924)."""

    def __init__(self, context, shape, dtype, padded_shape=None, raw=None):
        shape = tuple(shape)
        padded_shape = shape if padded_shape is None else tuple(padded_shape)
        assert len(shape) == len(padded_shape)
        assert all(p >= s for p, s in zip(padded_shape, shape))
        self._shape = shape
        self._dtype = np.dtype(dtype)
        self.padded_shape = padded_shape
        self.context = context
        self.buffer = context.allocate(padded_shape, dtype, raw)

    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape

    @property
    def dtype(self) -> np.dtype:
        return self._dtype

    @property
    def ndim(self) -> int:
        return len(self._shape)

    @property
    def strides(self) -> Tuple[int, ...]:
        """Byte strides of the padded layout, numpy style."""
        out = []
        step = self._dtype.itemsize
        for extent in reversed(self.padded_shape):
            out.append(step)
            step *= extent
        return tuple(reversed(out))

    # -- whole-array transfers
    def _copyable(self, ary) -> bool:
        return (
            HostArray.safe(ary)
            and ary.dtype == self.dtype
            and ary.shape == self.shape
            and ary.padded_shape == self.padded_shape
        )

    def empty_like(self) -> HostArray:
        """A pinned :class:`HostArray` with this array's shape and padding."""
        return HostArray(self.shape, self.dtype, self.padded_shape, context=self.context)

    def asarray_like(self, ary) -> HostArray:
        """`ary` itself if it already matches this layout, else a matching copy."""
        assert ary.shape == self.shape
        if self._copyable(ary):
            return ary
        staged = self.empty_like()
        np.copyto(staged, ary, casting="no")
        return staged

    def set(self, command_queue, ary) -> None:
        """Blocking host-to-device copy."""
        ary = self.asarray_like(ary)
        command_queue.enqueue_write_buffer(self.buffer, HostArray.padded_view(ary))

    def set_async(self, command_queue, ary) -> None:
        """Host-to-device copy that may still be in flight on return."""
        ary = self.asarray_like(ary)
        command_queue.enqueue_write_buffer(self.buffer, HostArray.padded_view(ary), blocking=False)

    def get(self, command_queue, ary=None):
        """Blocking device-to-host copy; returns the array actually filled."""
        if ary is None or not self._copyable(ary):
            ary = self.empty_like()
        command_queue.enqueue_read_buffer(self.buffer, HostArray.padded_view(ary))
        return ary

    def get_async(self, command_queue, ary=None):
        """Device-to-host copy that may still be in flight on return."""
        if ary is None or not self._copyable(ary):
            ary = self.empty_like()
        command_queue.enqueue_read_buffer(self.buffer, HostArray.padded_view(ary), blocking=False)
        return ary

    def zero(self, command_queue) -> None:
        """Asynchronous fill with zero bytes."""
        command_queue.enqueue_zero_buffer(self.buffer)

    # -- region transfers
    @staticmethod
    def _canonical_slice(region, shape, strides):
        """Index expression -> (byte origin, region shape, byte strides).

        Supports ints, slices with positive step and ``np.newaxis``; missing trailing
        axes are taken whole (reference accel.py:588-654).
        """
        if not isinstance(region, tuple):
            region = (region,)
        origin = 0
        out_shape: List[int] = []
        out_strides: List[int] = []
        axis = 0
        for item in region:
            if item is np.newaxis:
                out_shape.append(1)
                out_strides.append(0)
                continue
            if not isinstance(item, (slice, int, np.integer)):
                raise TypeError(f"Invalid type in slice: {type(item)}")
            if axis >= len(shape):
                raise IndexError("Too many axes in index expression")
            if isinstance(item, slice):
                start, stop, step = item.indices(shape[axis])
                if step <= 0:
                    raise IndexError("Only positive strides are supported")
                count = (stop - start) // step
                if count <= 0:
                    raise IndexError("Empty slice selection")
                origin += start * strides[axis]
                out_shape.append(count)
                out_strides.append(step * strides[axis])
            else:
                index = int(item)
                if index < 0:
                    index += shape[axis]
                if not 0 <= index < shape[axis]:
                    raise IndexError("Index out of range")
                origin += index * strides[axis]
            axis += 1
        for rest in range(axis, len(shape)):
            out_shape.append(shape[rest])
            out_strides.append(strides[rest])
        return origin, tuple(out_shape), tuple(out_strides)

    @classmethod
    def _region_transfer_params(cls, src, dest, src_region, dest_region):
        """Reduce a region copy to byte origins, a byte-level shape and strides.

        Contiguous axes are merged so that the copy has as few dimensions as possible;
        element 0 of the returned shape is a byte count (reference accel.py:656-725).
        """
        if src.dtype != dest.dtype:
            raise TypeError(f"dtypes do not match ({src.dtype} and {dest.dtype})")
        s_origin, s_shape, s_strides = cls._canonical_slice(src_region, src.shape, src.strides)
        d_origin, d_shape, d_strides = cls._canonical_slice(dest_region, dest.shape, dest.strides)
        if s_shape != d_shape:
            raise ValueError("Source and destination shapes for the copy do not match")
        shape = [src.dtype.itemsize]
        ss = [1]
        ds = [1]
        for axis in reversed(range(len(s_shape))):
            n = s_shape[axis]
            if n == 1:
                continue
            if s_strides[axis] == shape[-1] * ss[-1] and d_strides[axis] == shape[-1] * ds[-1]:
                shape[-1] *= n
            else:
                shape.append(n)
                ss.append(s_strides[axis])
                ds.append(d_strides[axis])
        return s_origin, d_origin, tuple(shape), tuple(ss), tuple(ds)

    @classmethod
    def _transfer_region(cls, func, buf1, buf2, origin1, origin2, shape, strides1, strides2, **kw):
        """Call a <=3-D backend copy, looping over any extra outer axes."""
        if len(shape) <= 3:
            func(buf1, buf2, origin1, origin2, shape, strides1, strides2, **kw)
            return
        for i in range(shape[-1]):
            cls._transfer_region(
                func, buf1, buf2, origin1 + i * strides1[-1], origin2 + i * strides2[-1],
                shape[:-1], strides1[:-1], strides2[:-1], **kw
            )  # fmt: skip

    def copy_region(self, command_queue, dest: "DeviceArray", src_region, dest_region) -> None:
        """Device-to-device copy of a sub-region (see :meth:`_canonical_slice`)."""
        s0, d0, shape, ss, ds = self._region_transfer_params(self, dest, src_region, dest_region)
        self._transfer_region(
            command_queue.enqueue_copy_buffer_rect, self.buffer, dest.buffer, s0, d0, shape, ss, ds
        )

    def get_region(self, command_queue, ary, device_region, ary_region, blocking=True) -> None:
        """Device-to-host copy of a sub-region into a (non-view) :class:`HostArray`."""
        if not HostArray.safe(ary):
            raise ValueError("Target region is not suitable for device-to-host copy")
        d0, a0, shape, dstr, astr = self._region_transfer_params(
            self, ary, device_region, ary_region
        )
        self._transfer_region(
            command_queue.enqueue_read_buffer_rect, self.buffer, HostArray.padded_view(ary),
            d0, a0, shape, dstr, astr, blocking=blocking
        )  # fmt: skip

    def set_region(self, command_queue, ary, device_region, ary_region, blocking=True) -> None:
        """Host-to-device copy of a sub-region; plain arrays are staged through pinned memory."""
        if not HostArray.safe(ary):
            staged = HostArray(ary[ary_region].shape, ary.dtype, context=self.context)
            np.copyto(staged, ary[ary_region], casting="no")
            ary, ary_region = staged, np.s_[()]
        a0, d0, shape, astr, dstr = self._region_transfer_params(
            ary, self, ary_region, device_region
        )
        self._transfer_region(
            command_queue.enqueue_write_buffer_rect, self.buffer, HostArray.padded_view(ary),
            d0, a0, shape, dstr, astr, blocking=blocking
        )  # fmt: skip


# ------------------------------------------------------------------------ allocators
class AbstractAllocator(ABC):
    """Source of device memory for slots (reference accel.py:1057-1074)."""

    context: AbstractContext

    @abstractmethod
    def allocate(self, shape, dtype, padded_shape=None, raw=None) -> DeviceArray: ...

    @abstractmethod
    def allocate_raw(self, n_bytes: int) -> Any: ...


class DeviceAllocator(AbstractAllocator):
    """Allocates :class:`DeviceArray` objects straight from a context."""

    def __init__(self, context: AbstractContext) -> None:
        self.context = context

    def allocate(self, shape, dtype, padded_shape=None, raw=None) -> DeviceArray:
        return DeviceArray(self.context, shape, dtype, padded_shape, raw)

    def allocate_raw(self, n_bytes: int) -> Any:
        return self.context.allocate_raw(n_bytes)


# ------------------------------------------------------------------------ dimensions
class Dimension:
    """Padding / alignment requirement of one axis, shareable between slots.

    Linked dimensions form one equivalence class (union-find) whose requirement is the
    combination of its members' (reference accel.py:1115-1294):

    * ``min_padded_size`` -- at least this many elements are allocated;
    * ``alignment`` (power of two) -- the padded size is a multiple of it;
    * an alignment *hint* from :meth:`add_align_dtype` -- rows of at least
      ``ALIGN_BYTES`` are rounded to a multiple of ``ALIGN_BYTES`` unless `exact`;
    * ``exact`` -- no padding beyond the minimum is acceptable.

    Binding a buffer freezes the class so that strides cannot change underneath it.
    """

    ALIGN_BYTES = 128

    @classmethod
    def _is_power2(cls, value: int) -> bool:
        return value > 0 and (value & (value - 1)) == 0

    def __init__(
        self,
        size: int,
        min_padded_round: Optional[int] = None,
        min_padded_size: Optional[int] = None,
        alignment: int = 1,
        align_dtype=None,
        exact: bool = False,
    ) -> None:
        if min_padded_size is None:
            min_padded_size = size if min_padded_round is None else roundup(size, min_padded_round)
        if not self._is_power2(alignment):
            raise ValueError("alignment is not a power of 2")
        if min_padded_size < size:
            raise ValueError("padded size is less than size")
        self._link: Optional["Dimension"] = None
        self._size = size
        self._req = {
            "min": min_padded_size,
            "align": alignment,
            "hint": alignment,
            "exact": exact,
            "frozen": False,
        }
        if align_dtype is not None:
            self.add_align_dtype(align_dtype)

    def _root(self) -> "Dimension":
        node = self
        while node._link is not None:
            node = node._link
        # path compression
        walk = self
        while walk._link is not None:
            walk._link, walk = node, walk._link
        return node

    @property
    def size(self) -> int:
        return self._root()._size

    @property
    def min_padded_size(self) -> int:
        return self._root()._req["min"]

    @property
    def alignment(self) -> int:
        return self._root()._req["align"]

    @property
    def alignment_hint(self) -> int:
        return self._root()._req["hint"]

    @property
    def exact(self) -> bool:
        return self._root()._req["exact"]

    @property
    def frozen(self) -> bool:
        return self._root()._req["frozen"]

    def required_padded_size(self) -> int:
        """The padded size a conforming buffer must have."""
        req = self._root()._req
        padded = roundup(req["min"], req["align"])
        # Only honour the hint when it cannot blow a tiny axis up to a full line.
        if not req["exact"] and padded >= req["hint"]:
            padded = roundup(padded, req["hint"])
        return padded

    def valid(self, padded_size: int) -> bool:
        """Would `padded_size` satisfy the hard requirements?"""
        req = self._root()._req
        if req["exact"]:
            return padded_size == self.required_padded_size()
        return padded_size >= req["min"] and padded_size % req["align"] == 0

    def add_align_dtype(self, dtype) -> None:
        """Hint that this is the fastest axis of an array of `dtype`."""
        if self.frozen:
            raise ValueError("cannot modify a frozen requirement")
        itemsize = np.dtype(dtype).itemsize
        if self._is_power2(itemsize):
            req = self._root()._req
            req["hint"] = max(req["hint"], self.ALIGN_BYTES // itemsize)

    def link(self, other: "Dimension") -> None:
        """Merge the requirement classes of `self` and `other`."""
        a, b = self._root(), other._root()
        if a._req["frozen"] or b._req["frozen"]:
            raise ValueError("cannot link frozen requirements")
        if a is b:
            return
        if a._size != b._size:
            raise ValueError("sizes are incompatible")
        if a._req["exact"] and not b.valid(a.required_padded_size()):
            raise ValueError("linked requirement is unsatisfiable")
        if b._req["exact"] and not a.valid(b.required_padded_size()):
            raise ValueError("linked requirement is unsatisfiable")
        a._req = {
            "min": max(a._req["min"], b._req["min"]),
            "align": max(a._req["align"], b._req["align"]),
            "hint": max(a._req["hint"], b._req["hint"]),
            "exact": a._req["exact"] or b._req["exact"],
            "frozen": False,
        }
        b._link = a
        b._req = None  # a stale requirement must never be consulted

    def freeze(self) -> None:
        self._root()._req["frozen"] = True


# ----------------------------------------------------------------------------- slots
class IOSlotBase(ABC):
    """A named input/output of an operation; slots form trees sharing storage."""

    def __init__(self) -> None:
        self.is_root = True

    def check_root(self) -> None:
        if not self.is_root:
            raise ValueError("not a root slot")

    @abstractmethod
    def required_bytes(self) -> int: ...

    @abstractmethod
    def is_bound(self) -> bool: ...

    def attachable(self) -> bool:
        """Can this slot become a child of a compound or alias slot?"""
        return self.is_root and not self.is_bound()

    @abstractmethod
    def _allocate(self, allocator, raw=None, *, bind: bool): ...

    def allocate(self, allocator, raw=None, *, bind: bool = True):
        """Allocate (and by default bind) conforming storage.

        With `raw` given, the caller vouches that it is large enough (no check, as in
        the reference, accel.py:1346-1367).
        """
        self.check_root()
        return self._allocate(allocator, raw, bind=bind)

    @abstractmethod
    def _allocate_host(self, context) -> HostArray: ...

    def allocate_host(self, context) -> HostArray:
        """A :class:`HostArray` laid out like this slot's buffer."""
        self.check_root()
        return self._allocate_host(context)


class IOSlot(IOSlotBase):
    """Slot with a dtype and one :class:`Dimension` per axis (reference accel.py:1379-1502)."""

    def __init__(self, dimensions: Tuple[Union[Dimension, int], ...], dtype) -> None:
        super().__init__()
        self.dimensions = tuple(
            d if isinstance(d, Dimension) else Dimension(int(d)) for d in dimensions
        )
        self.shape = tuple(d.size for d in self.dimensions)
        self.dtype = np.dtype(dtype)
        if len(self.dimensions) > 1:
            self.dimensions[-1].add_align_dtype(self.dtype)
        self.buffer: Optional[DeviceArray] = None

    def is_bound(self) -> bool:
        return self.buffer is not None

    def validate(self, buffer: DeviceArray) -> None:
        """Raise unless `buffer` has exactly the dtype, shape and padding required."""
        if buffer.dtype != self.dtype:
            raise TypeError("dtype does not match")
        if len(buffer.shape) != len(self.shape):
            raise ValueError("number of dimensions does not match")
        for size, padded, dim in zip(buffer.shape, buffer.padded_shape, self.dimensions):
            if size != dim.size:
                raise ValueError("size does not match")
            if padded != dim.required_padded_size():
                raise ValueError("padded size does not match")

    def _bind(self, buffer: Optional[DeviceArray]) -> None:
        if buffer is not None:
            self.validate(buffer)
        self.buffer = buffer
        for dim in self.dimensions:
            dim.freeze()

    def bind(self, buffer: Optional[DeviceArray]) -> None:
        """Attach `buffer` (validated) or detach with None; freezes the dimensions."""
        self.check_root()
        self._bind(buffer)

    def required_padded_shape(self) -> Tuple[int, ...]:
        return tuple(d.required_padded_size() for d in self.dimensions)

    def required_bytes(self) -> int:
        return int(np.prod(self.required_padded_shape(), dtype=np.int64)) * self.dtype.itemsize

    def _allocate(self, allocator, raw=None, *, bind: bool = True) -> DeviceArray:
        buffer = allocator.allocate(self.shape, self.dtype, self.required_padded_shape(), raw=raw)
        if bind:
            self._bind(buffer)
        return buffer

    def _allocate_host(self, context) -> HostArray:
        return HostArray(self.shape, self.dtype, self.required_padded_shape(), context=context)


class CompoundIOSlot(IOSlot):
    """One buffer feeding several same-shaped slots; their dimensions are linked."""

    def __init__(self, children: Iterable[IOSlot]) -> None:
        self.children = list(children)
        if not self.children:
            raise ValueError("empty child list")
        first = self.children[0]
        for child in self.children:
            if not child.attachable():
                raise ValueError("child is not attachable")
            if child.shape != first.shape:
                raise ValueError("inconsistent shapes")
            if child.dtype != first.dtype:
                raise TypeError("inconsistent dtypes")
            if any(d.frozen for d in child.dimensions):
                raise ValueError("child has frozen dimensions")
        for child in self.children:
            for mine, theirs in zip(first.dimensions, child.dimensions):
                mine.link(theirs)
        super().__init__(first.dimensions, first.dtype)
        for child in self.children:
            child.is_root = False

    def _bind(self, buffer: Optional[DeviceArray]) -> None:
        super()._bind(buffer)
        for child in self.children:
            child._bind(buffer)


class AliasIOSlot(IOSlotBase):
    """One raw allocation backing several slots that are never live together."""

    def __init__(self, children: Iterable[IOSlotBase]) -> None:
        super().__init__()
        self.children = list(children)
        self.raw: Optional[Any] = None
        if not self.children:
            raise ValueError("empty child list")
        for child in self.children:
            if not child.attachable():
                raise ValueError("child is not attachable")
        for child in self.children:
            child.is_root = False

    def is_bound(self) -> bool:
        return self.raw is not None

    def required_bytes(self) -> int:
        return max(child.required_bytes() for child in self.children)

    def _allocate_host(self, context) -> HostArray:
        return HostArray((self.required_bytes(),), np.uint8, context=context)

    def _allocate(self, allocator, raw=None, *, bind: bool = True):
        if raw is None:
            raw = allocator.allocate_raw(self.required_bytes())
        if bind:
            for child in self.children:
                child._allocate(allocator, raw, bind=True)
            self.raw = raw
        return raw


# ------------------------------------------------------------------------ operations
class Operation(ABC):
    """A device operation with named slots (reference accel.py:1611-1756).

    Subclasses fill :attr:`slots` in their constructor and implement :meth:`_run`.
    """

    def __init__(
        self, command_queue: AbstractCommandQueue, allocator: Optional[AbstractAllocator] = None
    ) -> None:
        if allocator is None:
            allocator = DeviceAllocator(command_queue.context)
        elif allocator.context is not None and allocator.context is not command_queue.context:
            raise ValueError("command_queue and allocator have different contexts")
        self.slots: Dict[str, IOSlotBase] = {}
        self.hidden_slots: Dict[str, IOSlotBase] = {}
        self.command_queue = command_queue
        self.allocator = allocator
        self.is_root = True

    def bind(self, **kwargs) -> None:
        """Bind buffers to slots by name (KeyError for unknown names, TypeError for aliases)."""
        for name, buffer in kwargs.items():
            slot = self.slots[name]
            if not isinstance(slot, IOSlot):
                raise TypeError(f"Slot {slot} is not an IOSlot")
            slot.bind(buffer)

    def ensure_bound(self, name: str) -> None:
        slot = self.slots[name]
        if not slot.is_bound():
            slot.allocate(self.allocator)

    def ensure_all_bound(self) -> None:
        for slot in self.slots.values():
            if not slot.is_bound():
                slot.allocate(self.allocator)

    def buffer(self, name: str) -> DeviceArray:
        """The buffer bound to slot `name` (visible or hidden)."""
        slot = self.slots.get(name)
        if slot is None:
            slot = self.hidden_slots.get(name)
        if slot is None:
            raise KeyError("no slot named " + name)
        if not isinstance(slot, IOSlot):
            raise TypeError("slot " + name + " is an alias slot")
        if slot.buffer is None:
            raise ValueError("slot " + name + " has no buffer bound")
        return slot.buffer

    def required_bytes(self) -> int:
        return sum(slot.required_bytes() for slot in self.slots.values())

    def parameters(self) -> Mapping[str, Any]:
        return {}

    @abstractmethod
    def _run(self) -> Any:
        raise NotImplementedError("abstract base class")

    def __call__(self, **kwargs) -> Any:
        """Bind `kwargs`, allocate whatever is still unbound, and run."""
        self.bind(**kwargs)
        self.ensure_all_bound()
        return self._run()


class OperationSequence(Operation):
    """Runs named child operations in order, sharing buffers between their slots.

    Child slot `slot` of child `op` first appears as ``op:slot``; each entry of
    `compounds` replaces the named slots that exist by one :class:`CompoundIOSlot`, and
    each entry of `aliases` by one :class:`AliasIOSlot` (the originals stay reachable
    through :attr:`hidden_slots`). Names that do not exist are skipped
    (reference accel.py:1759-1835).
    """

    def __init__(
        self,
        command_queue: AbstractCommandQueue,
        operations: Iterable[Tuple[str, Operation]],
        compounds: Optional[Mapping[str, Iterable[str]]] = None,
        aliases: Optional[Mapping[str, Iterable[str]]] = None,
        allocator: Optional[AbstractAllocator] = None,
    ) -> None:
        super().__init__(command_queue, allocator)
        self.operations = OrderedDict(operations)
        for op_name, op in self.operations.items():
            if op.command_queue is not command_queue:
                raise ValueError("child has a different command queue to the parent")
            if not op.is_root:
                raise ValueError("child already has another parent")
            for slot_name, slot in op.slots.items():
                self.slots[op_name + ":" + slot_name] = slot
        for name, members in (compounds or {}).items():
            taken = self._take(members, hide=False)
            if taken:
                if not all(isinstance(s, IOSlot) for s in taken):
                    raise TypeError(f"Children of {name} must all be IOSlots")
                self.slots[name] = CompoundIOSlot(taken)
        for name, members in (aliases or {}).items():
            taken = self._take(members, hide=True)
            if taken:
                self.slots[name] = AliasIOSlot(taken)
        for op in self.operations.values():
            op.is_root = False

    def _take(self, names: Iterable[str], hide: bool) -> List[IOSlotBase]:
        taken = []
        for name in names:
            slot = self.slots.pop(name, None)
            if slot is None:
                continue
            taken.append(slot)
            if hide:
                assert name not in self.hidden_slots
                self.hidden_slots[name] = slot
        return taken

    def _run(self) -> None:
        for op in self.operations.values():
            op()
