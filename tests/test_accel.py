"""CPU tests of the accel layer: padding algebra, slots, operations, host arrays.

The expectations restate the behaviour documented for the reference's accel module
(reference: src/katsdpsigproc/accel.py:1115-1835; its tests in test/test_accel.py:462-829
are the specification these were written from).
"""

import numpy as np
import pytest

from katsdpsigproc_amd import accel
from tests.fakes import FakeBuffer, FakeContext, make_queue


class TestHelpers:
    def test_divup_roundup(self):
        assert accel.divup(10, 5) == 2 and accel.divup(11, 5) == 3 and accel.divup(0, 5) == 0
        assert accel.roundup(10, 5) == 10 and accel.roundup(11, 5) == 15

    def test_build_finds_its_templates(self):
        # (compilation itself is exercised on the GPU in tests/test_build.py)
        with pytest.raises(FileNotFoundError):
            accel.render_template("no_such_kernel.hip.in", {})
        assert "fill" in accel.render_template("fill.hip.in", {"wgs": 64, "ctype": "int"})


class TestDimension:
    def test_power2(self):
        assert accel.Dimension._is_power2(1) and accel.Dimension._is_power2(32)
        assert not accel.Dimension._is_power2(0) and not accel.Dimension._is_power2(12)
        with pytest.raises(ValueError):
            accel.Dimension(10, alignment=3)
        with pytest.raises(ValueError):
            accel.Dimension(10, min_padded_size=9)

    def test_min_padded_round(self):
        assert accel.Dimension(17, min_padded_round=4).min_padded_size == 20
        assert accel.Dimension(20, min_padded_round=5).min_padded_size == 20

    def test_align_dtype_hint(self):
        dim = accel.Dimension(20, alignment=8)
        dim.add_align_dtype(np.complex64)
        assert dim.alignment_hint == 16
        dim.add_align_dtype(np.uint8)
        assert dim.alignment_hint == 128
        dim.add_align_dtype(np.float32)
        assert dim.alignment_hint == 128
        dim.add_align_dtype(np.dtype([("a", np.uint8, (3,))]))  # 3 bytes: ignored
        assert dim.alignment_hint == 128

    def test_valid(self):
        dim = accel.Dimension(17, min_padded_round=8, alignment=4)
        assert dim.valid(24) and dim.valid(28)
        assert not dim.valid(20) and not dim.valid(30)
        exact = accel.Dimension(20, min_padded_size=23, exact=True)
        assert exact.valid(23) and not exact.valid(24) and not exact.valid(20)

    def test_required_padded_size(self):
        assert accel.Dimension(30, 7, alignment=4).required_padded_size() == 36
        assert accel.Dimension(1100, 200, align_dtype=np.float32).required_padded_size() == 1216
        assert accel.Dimension(1100, align_dtype=np.float32, exact=True).required_padded_size() == 1100
        # the hint never inflates an axis shorter than one 128-byte line
        assert accel.Dimension(18, alignment=8, align_dtype=np.uint8).required_padded_size() == 24

    def test_link_combines(self):
        d1 = accel.Dimension(22, min_padded_size=28, alignment=4)
        d2 = accel.Dimension(22, min_padded_size=24, alignment=8, align_dtype=np.int32)
        d3 = accel.Dimension(22, align_dtype=np.uint16)
        d1.link(d2)
        d3.link(d1)
        for d in (d1, d2, d3):
            assert (d.size, d.min_padded_size, d.alignment, d.alignment_hint, d.exact) == (
                22, 28, 8, 64, False)  # fmt: skip
        assert d1.required_padded_size() == d2.required_padded_size() == d3.required_padded_size()
        d1.link(d3)  # already linked: no-op

    def test_link_failures_leave_dimensions_apart(self):
        d1 = accel.Dimension(22, min_padded_size=28)
        with pytest.raises(ValueError):
            d1.link(accel.Dimension(23))
        exact = accel.Dimension(22, exact=True)
        loose = accel.Dimension(22, min_padded_size=28)
        aligned = accel.Dimension(22, alignment=4)
        with pytest.raises(ValueError):
            exact.link(loose)
        with pytest.raises(ValueError):
            exact.link(aligned)
        with pytest.raises(ValueError):
            loose.link(exact)
        assert exact._root() is not loose._root() and exact._root() is not aligned._root()

    def test_freeze(self):
        d1, d2 = accel.Dimension(8), accel.Dimension(8)
        d1.freeze()
        assert d1.frozen and not d2.frozen
        with pytest.raises(ValueError):
            d1.link(d2)
        with pytest.raises(ValueError):
            d1.add_align_dtype(np.float32)


class RecordingAllocator(accel.AbstractAllocator):
    def __init__(self, context):
        self.context = context
        self.calls = []

    def allocate(self, shape, dtype, padded_shape=None, raw=None):
        self.calls.append((shape, np.dtype(dtype), padded_shape, raw))
        return accel.DeviceArray(self.context, shape, dtype, padded_shape, raw)

    def allocate_raw(self, n_bytes):
        self.calls.append(("raw", n_bytes))
        return self.context.allocate_raw(n_bytes)


class TestIOSlot:
    def setup_method(self):
        self.context = FakeContext()
        self.allocator = RecordingAllocator(self.context)

    def slot(self):
        return accel.IOSlot((accel.Dimension(50, min_padded_size=60, alignment=8), 37), np.float32)

    def test_required(self):
        slot = self.slot()
        assert slot.shape == (50, 37)
        assert slot.required_padded_shape() == (64, 64)  # 37 >= 32-element hint: rounded
        assert slot.required_bytes() == 64 * 64 * 4
        assert accel.IOSlot((4, 20), np.float32).required_padded_shape() == (4, 20)  # short rows stay
        wide = accel.IOSlot((4, 100), np.float32)
        assert wide.required_padded_shape() == (4, 128)  # rows rounded to 128 bytes
        assert accel.IOSlot((100,), np.float32).required_padded_shape() == (100,)  # 1-D: no hint

    @pytest.mark.parametrize("bind", [True, False])
    def test_allocate(self, bind):
        slot = self.slot()
        buffer = slot.allocate(self.allocator, bind=bind)
        assert self.allocator.calls == [((50, 37), np.dtype(np.float32), (64, 64), None)]
        assert (slot.buffer is buffer) == bind
        assert slot.is_bound() == bind
        assert all(d.frozen for d in slot.dimensions) == bind

    def test_allocate_raw_passthrough(self):
        slot = self.slot()
        raw = object()
        slot.allocate(self.allocator, raw)
        assert self.allocator.calls[0][3] is raw

    def test_validate(self):
        slot = self.slot()
        ok = accel.DeviceArray(self.context, (50, 37), np.float32, (64, 64))
        slot.validate(ok)
        with pytest.raises(TypeError):
            slot.validate(accel.DeviceArray(self.context, (50, 37), np.int32, (64, 64)))
        with pytest.raises(ValueError):
            slot.validate(accel.DeviceArray(self.context, (50,), np.float32))
        with pytest.raises(ValueError):
            slot.validate(accel.DeviceArray(self.context, (51, 37), np.float32, (64, 64)))
        with pytest.raises(ValueError):  # padding must match exactly
            slot.validate(accel.DeviceArray(self.context, (50, 37), np.float32, (72, 64)))

    def test_bind_none_and_nonroot(self):
        slot = self.slot()
        slot.bind(None)
        assert not slot.is_bound()
        child = self.slot()
        accel.CompoundIOSlot([child])
        with pytest.raises(ValueError):
            child.bind(None)
        with pytest.raises(ValueError):
            child.allocate(self.allocator)

    def test_allocate_host(self):
        host = self.slot().allocate_host(self.context)
        assert isinstance(host, accel.HostArray)
        assert host.shape == (50, 37) and host.padded_shape == (64, 64)


class TestCompoundIOSlot:
    def setup_method(self):
        self.context = FakeContext()
        self.allocator = RecordingAllocator(self.context)
        self.dims1 = (accel.Dimension(13, min_padded_size=17, alignment=1),
                      accel.Dimension(7, min_padded_size=8, alignment=8))  # fmt: skip
        self.dims2 = (accel.Dimension(13, min_padded_size=14, alignment=4),
                      accel.Dimension(7, min_padded_size=10, alignment=4))  # fmt: skip
        self.s1 = accel.IOSlot(self.dims1, np.float32)
        self.s2 = accel.IOSlot(self.dims2, np.float32)

    def test_combined_requirement(self):
        compound = accel.CompoundIOSlot([self.s1, self.s2])
        assert compound.shape == (13, 7)
        assert compound.required_padded_shape() == (20, 16)
        assert self.s1.required_padded_shape() == self.s2.required_padded_shape() == (20, 16)
        assert not self.s1.is_root and not self.s2.is_root and compound.is_root

    def test_bind_propagates(self):
        compound = accel.CompoundIOSlot([self.s1, self.s2])
        buffer = compound.allocate(self.allocator)
        assert self.s1.buffer is buffer and self.s2.buffer is buffer
        compound.bind(None)
        assert self.s1.buffer is None and self.s2.buffer is None

    def test_errors(self):
        with pytest.raises(ValueError):
            accel.CompoundIOSlot([])
        with pytest.raises(ValueError):
            accel.CompoundIOSlot([self.s1, accel.IOSlot((13, 8), np.float32)])
        with pytest.raises(TypeError):
            accel.CompoundIOSlot([self.s1, accel.IOSlot((13, 7), np.int32)])
        bound = accel.IOSlot((13, 7), np.float32)
        bound.allocate(self.allocator)
        with pytest.raises(ValueError):
            accel.CompoundIOSlot([self.s1, bound])
        used = accel.IOSlot((13, 7), np.float32)
        accel.CompoundIOSlot([used])
        with pytest.raises(ValueError):
            accel.CompoundIOSlot([used])


class TestAliasIOSlot:
    def test_alias(self):
        context = FakeContext()
        allocator = RecordingAllocator(context)
        s1 = accel.IOSlot((100,), np.float32)  # 400 bytes
        s2 = accel.IOSlot((64, 3), np.uint8)  # 192 bytes
        alias = accel.AliasIOSlot([s1, s2])
        assert alias.required_bytes() == 400
        assert not alias.is_bound()
        raw = alias.allocate(allocator)
        assert alias.is_bound() and alias.raw is raw
        assert s1.is_bound() and s2.is_bound()
        assert allocator.calls[0] == ("raw", 400)
        assert allocator.calls[1][3] is raw and allocator.calls[2][3] is raw
        host = alias.allocate_host(context)
        assert host.shape == (400,) and host.dtype == np.uint8
        with pytest.raises(ValueError):
            accel.AliasIOSlot([])
        with pytest.raises(ValueError):
            accel.AliasIOSlot([s1])  # already a child


class CountingOp(accel.Operation):
    def __init__(self, queue, names, log, tag):
        super().__init__(queue)
        for name in names:
            self.slots[name] = accel.IOSlot((8, accel.Dimension(5)), np.float32)
        self.log, self.tag = log, tag

    def _run(self):
        self.log.append(self.tag)


class TestOperation:
    def test_bind_and_buffer(self):
        queue = make_queue()
        op = CountingOp(queue, ["a", "b"], [], "x")
        assert op.required_bytes() == 2 * 8 * 5 * 4
        with pytest.raises(KeyError):
            op.bind(c=None)
        with pytest.raises(KeyError, match="no slot named c"):
            op.buffer("c")
        with pytest.raises(ValueError):
            op.buffer("a")  # nothing bound yet
        op.ensure_bound("a")
        assert op.buffer("a").shape == (8, 5)
        assert not op.slots["b"].is_bound()
        op()
        assert op.slots["b"].is_bound() and op.log == ["x"]
        buf = accel.DeviceArray(queue.context, (8, 5), np.float32)
        op(b=buf)
        assert op.buffer("b") is buf

    def test_allocator_context_mismatch(self):
        queue = make_queue()
        with pytest.raises(ValueError):
            CountingOp.__base__.__init__(
                CountingOp.__new__(CountingOp), queue, accel.DeviceAllocator(FakeContext())
            )


class TestOperationSequence:
    def test_wiring(self):
        queue = make_queue()
        log = []
        op1 = CountingOp(queue, ["in", "out"], log, 1)
        op2 = CountingOp(queue, ["in", "out", "tmp"], log, 2)
        seq = accel.OperationSequence(
            queue,
            [("first", op1), ("second", op2)],
            compounds={"mid": ["first:out", "second:in", "missing:slot"], "nothing": ["nope:x"]},
            aliases={"scratch": ["second:tmp", "absent:y"]},
        )
        assert set(seq.slots) == {"first:in", "mid", "second:out", "scratch"}
        assert "nothing" not in seq.slots
        assert seq.hidden_slots["second:tmp"] is op2.slots["tmp"]
        assert not op1.is_root and not op2.is_root
        seq()
        assert log == [1, 2]
        assert op1.buffer("out") is op2.buffer("in") is seq.buffer("mid")
        assert seq.buffer("second:tmp").shape == (8, 5)
        with pytest.raises(TypeError):
            seq.buffer("scratch")
        with pytest.raises(TypeError):
            seq.bind(scratch=None)
        with pytest.raises(ValueError):
            accel.OperationSequence(queue, [("again", op1)])
        other = CountingOp(make_queue(), ["in"], log, 3)
        with pytest.raises(ValueError):
            accel.OperationSequence(queue, [("other", other)])


class TestHostArray:
    def test_safe_and_padding(self):
        ary = accel.HostArray((5, 3), np.int32, (8, 4))
        assert ary.shape == (5, 3) and ary.padded_shape == (8, 4)
        assert accel.HostArray.safe(ary)
        assert accel.HostArray.padded_view(ary).shape == (8, 4)
        assert not accel.HostArray.safe(ary[:3])
        assert not accel.HostArray.safe(ary.copy())
        assert not accel.HostArray.safe(np.zeros((5, 3), np.int32))
        assert accel.HostArray.padded_view(np.zeros(3)) is None
        scalar = accel.HostArray((), np.float32)
        assert scalar.shape == () and accel.HostArray.safe(scalar)


class TestDeviceArray:
    def test_set_get_roundtrip_with_padding(self):
        queue = make_queue()
        dev = accel.DeviceArray(queue.context, (4, 5), np.int32, (6, 8))
        assert dev.strides == (32, 4) and dev.ndim == 2
        data = np.arange(20, dtype=np.int32).reshape(4, 5)
        dev.set(queue, data)  # plain array: staged through a padded HostArray
        out = dev.get(queue)
        assert isinstance(out, accel.HostArray) and out.padded_shape == (6, 8)
        np.testing.assert_array_equal(out, data)
        reuse = dev.empty_like()
        assert dev.get(queue, reuse) is reuse
        assert dev.get(queue, np.zeros((4, 5), np.int32)) is not None
        with pytest.raises(TypeError):
            dev.set(queue, data.astype(np.int64))

    def test_region_params(self):
        queue = make_queue()
        src = accel.DeviceArray(queue.context, (10, 20), np.float32, (12, 24))
        dst = accel.DeviceArray(queue.context, (5, 20), np.float32, (5, 32))
        params = accel.DeviceArray._region_transfer_params(
            src, dst, np.s_[2:7, :], np.s_[:, :])
        assert params == (2 * 96, 0, (80, 5), (1, 96), (1, 128))
        # fully contiguous rows collapse to one dimension
        a = accel.DeviceArray(queue.context, (6, 8), np.int16)
        b = accel.DeviceArray(queue.context, (6, 8), np.int16)
        assert accel.DeviceArray._region_transfer_params(a, b, np.s_[1:4], np.s_[2:5]) == (
            16, 32, (48,), (1,), (1,))  # fmt: skip
        # integer index and newaxis
        origin, shape, strides = accel.DeviceArray._canonical_slice(
            np.s_[3, np.newaxis, 1:7:2], (10, 20), (80, 4))
        assert (origin, shape, strides) == (3 * 80 + 4, (1, 3), (0, 8))
        with pytest.raises(IndexError):
            accel.DeviceArray._canonical_slice(np.s_[10], (10,), (4,))
        with pytest.raises(IndexError):
            accel.DeviceArray._canonical_slice(np.s_[::-1], (10,), (4,))
        with pytest.raises(IndexError):
            accel.DeviceArray._canonical_slice(np.s_[5:5], (10,), (4,))
        with pytest.raises(IndexError):
            accel.DeviceArray._canonical_slice(np.s_[1, 2], (10,), (4,))
        with pytest.raises(TypeError):
            accel.DeviceArray._canonical_slice(np.s_[1.5], (10,), (4,))
        with pytest.raises(ValueError):
            accel.DeviceArray._region_transfer_params(src, dst, np.s_[0:3], np.s_[0:4])
        with pytest.raises(TypeError):
            accel.DeviceArray._region_transfer_params(src, a, np.s_[0:3], np.s_[0:3])
