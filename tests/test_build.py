"""accel.build / Context.compile / enqueue_kernel for run-time compiled kernels, and the
two operations built on them (Fill, HReduce). Rendering is checked without a GPU;
compilation and launches need one (reference test/test_fill.py, test/test_reduce.py,
and the kernel tests that go through accel.build)."""

import numpy as np
import pytest

from katsdpsigproc_amd import accel


# ---------------------------------------------------------------- no GPU: rendering
def test_render_placeholders_and_lookup(tmp_path):
    text = accel.render_template("fill.hip.in", {"wgs": 256, "ctype": "float2"})
    assert "REQD_WORK_GROUP_SIZE(256, 1, 1)" in text and "float2 value" in text
    (tmp_path / "mine.hip.in").write_text("KERNEL void k_${tag}(int *p) { p[0] = ${n}; }")
    text = accel.render_template("mine.hip.in", {"tag": "x", "n": 3}, extra_dirs=[tmp_path])
    assert text == "KERNEL void k_x(int *p) { p[0] = 3; }"
    assert accel.render_template("any", {"v": 1}, source="a${v}b") == "a1b"
    with pytest.raises(KeyError, match="wgs"):
        accel.render_template("fill.hip.in", {"ctype": "float"})
    with pytest.raises(FileNotFoundError):
        accel.render_template("missing.hip.in", {})


def test_build_passes_flags_and_simd_group_size():
    seen = {}

    class Device:
        simd_group_size = 64

    class Context:
        device = Device()

        def compile(self, source, extra_flags=None):
            seen["source"], seen["flags"] = source, list(extra_flags)
            return "program"

    program = accel.build(Context(), "x", {"a": 2}, extra_dirs=["/some/dir"],
                          extra_flags=["-DQ=1"], source="${a} ${simd_group_size}")  # fmt: skip
    assert program == "program" and seen["source"] == "2 64"
    assert seen["flags"][0] == "-DQ=1" and "-I/some/dir" in seen["flags"]
    assert "-I" + accel.KERNEL_DIR in seen["flags"]


def test_mako_templates_need_mako():
    try:
        import mako  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="Mako"):
            accel.render_template("fill.mako", {})


# ------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def context():
    return accel.create_some_context(interactive=False)


@pytest.fixture(scope="module")
def command_queue(context):
    return context.create_command_queue()


SAXPY = """
#include "port.h"
KERNEL void saxpy(GLOBAL float *RESTRICT y, const GLOBAL float *RESTRICT x, float a, int n)
{
    const int i = get_global_id(0);
    if (i < n) y[i] = a * x[i] + y[i] + ${bias};
}
KERNEL void grid2d(GLOBAL int *out, int cols)
{
    LOCAL_DECL int tile[4][8];
    tile[get_local_id(1)][get_local_id(0)] = get_global_id(1) * 1000 + get_global_id(0);
    BARRIER();
    out[get_global_id(1) * cols + get_global_id(0)] = tile[get_local_id(1)][get_local_id(0)];
}
"""


@pytest.mark.gpu
class TestCompile:
    def test_user_kernel(self, context, command_queue):
        program = accel.build(context, "saxpy", {"bias": "1.0f"}, source=SAXPY)
        kernel = program.get_kernel("saxpy")
        n = 1000
        x = np.arange(n, dtype=np.float32)
        y = np.full(n, 2.0, np.float32)
        dx = accel.DeviceArray(context, (n,), np.float32)
        dy = accel.DeviceArray(context, (n,), np.float32)
        dx.set(command_queue, x)
        dy.set(command_queue, y)
        command_queue.enqueue_kernel(kernel, [dy.buffer, dx.buffer, np.float32(0.5), np.int32(n)],
                                     global_size=(1024,), local_size=(256,))  # fmt: skip
        np.testing.assert_array_equal(dy.get(command_queue), 0.5 * x + 2.0 + 1.0)
        # 2-D launch, LDS, barrier: global sizes are in threads, as in the reference
        out = accel.DeviceArray(context, (8, 16), np.int32)
        command_queue.enqueue_kernel(program.get_kernel("grid2d"), [out.buffer, np.int32(16)],
                                     global_size=(16, 8), local_size=(8, 4))  # fmt: skip
        expected = np.arange(8)[:, None] * 1000 + np.arange(16)[None, :]
        np.testing.assert_array_equal(out.get(command_queue), expected)

    def test_errors(self, context, command_queue):
        from katsdpsigproc_amd import hip

        with pytest.raises(hip.CompileError) as info:
            context.compile("this is not HIP")
        assert "error" in info.value.log
        program = accel.build(context, "saxpy", {"bias": "0"}, source=SAXPY)
        with pytest.raises(RuntimeError):
            program.get_kernel("no_such_kernel")
        kernel = program.get_kernel("saxpy")
        buf = accel.DeviceArray(context, (4,), np.float32)
        with pytest.raises(ValueError):  # not a multiple of the local size
            command_queue.enqueue_kernel(kernel, [buf.buffer, buf.buffer, np.float32(1), np.int32(4)],
                                         global_size=(100,), local_size=(64,))  # fmt: skip
        with pytest.raises(TypeError):  # plain Python numbers have no definite C type
            command_queue.enqueue_kernel(kernel, [buf.buffer, buf.buffer, 1.0, 4],
                                         global_size=(64,), local_size=(64,))  # fmt: skip
        with pytest.raises(ValueError):
            command_queue.enqueue_kernel(kernel, [], global_size=None, local_size=None)


@pytest.mark.gpu
class TestFill:
    @pytest.mark.parametrize("shape", [(75,), (75, 63), (3, 5, 7)])
    def test_fill(self, shape, context, command_queue):
        # reference test/test_fill.py: padding is filled too
        from katsdpsigproc_amd import fill

        template = fill.FillTemplate(context, np.uint32, "unsigned int")
        fn = template.instantiate(command_queue, shape)
        fn.ensure_all_bound()
        data = fn.buffer("data")
        data.zero(command_queue)
        fn.set_value(0xDEADBEEF)
        fn()
        raw = np.empty(data.padded_shape, np.uint32)
        command_queue.enqueue_read_buffer(data.buffer, raw)
        assert (raw == 0xDEADBEEF).all()
        assert fn.parameters()["value"] == 0xDEADBEEF

    def test_float_and_complex_layout(self, context, command_queue):
        from katsdpsigproc_amd import fill

        fn = fill.FillTemplate(context, np.float32, "float", tuning={"wgs": 64}).instantiate(
            command_queue, (1000, 33)
        )
        fn.ensure_all_bound()
        fn.set_value(-2.5)
        fn()
        assert (fn.buffer("data").get(command_queue) == np.float32(-2.5)).all()

    @pytest.mark.force_autotune
    def test_autotune(self, context):
        from katsdpsigproc_amd import fill

        assert fill.FillTemplate(context, np.uint8, "unsigned char").wgs in (64, 128, 256, 512)


@pytest.mark.gpu
class TestHReduce:
    @pytest.mark.parametrize("wgsx, wgsy", [(64, 4), (32, 2), (128, 8), (256, 1), (1024, 1), (8, 8)])
    @pytest.mark.parametrize("rows, columns, column_range",
                             [(129, 173, (67, 128)), (7, 1000, None), (64, 5, (0, 5)), (1, 1, None)])  # fmt: skip
    def test_sum(self, wgsx, wgsy, rows, columns, column_range, context, command_queue):
        # reference test/test_reduce.py: integer sums compare exactly
        from katsdpsigproc_amd import reduce

        template = reduce.HReduceTemplate(context, np.uint32, "unsigned int", "a + b", "0",
                                          tuning={"wgsx": wgsx, "wgsy": wgsy})  # fmt: skip
        fn = template.instantiate(command_queue, (rows, columns), column_range)
        fn.ensure_all_bound()
        rs = np.random.RandomState(1)
        src = rs.randint(0, 100000, (rows, columns)).astype(np.uint32)
        fn.buffer("src").set(command_queue, src)
        fn()
        lo, hi = column_range or (0, columns)
        expected = src[:, lo:hi].sum(axis=1, dtype=np.uint32)
        np.testing.assert_array_equal(fn.buffer("dest").get(command_queue), expected)

    def test_custom_operator(self, context, command_queue):
        from katsdpsigproc_amd import reduce

        template = reduce.HReduceTemplate(
            context, np.float32, "float", "pick(a, b)", "-INFINITY",
            extra_code="DEVICE_FN static inline float pick(float a, float b) { return fmaxf(a, b); }",
            tuning={"wgsx": 128, "wgsy": 2},
        )
        fn = template.instantiate(command_queue, (50, 3000))
        fn.ensure_all_bound()
        src = np.random.RandomState(2).standard_normal((50, 3000)).astype(np.float32)
        fn.buffer("src").set(command_queue, src)
        fn()
        np.testing.assert_array_equal(fn.buffer("dest").get(command_queue), src.max(axis=1))
        assert fn.parameters()["op"] == "pick(a, b)"

    def test_bad_arguments(self, context, command_queue):
        from katsdpsigproc_amd import reduce

        template = reduce.HReduceTemplate(context, np.int32, "int", "a + b", "0",
                                          tuning={"wgsx": 64, "wgsy": 4})  # fmt: skip
        with pytest.raises(ValueError):
            template.instantiate(command_queue, (4, 5, 6))
        with pytest.raises(ValueError):
            template.instantiate(command_queue, (4, 5), (2, 2))
        with pytest.raises(ValueError):
            template.instantiate(command_queue, (4, 5), (0, 6))
        with pytest.raises(ValueError):
            reduce.HReduceTemplate(context, np.int32, "int", "a + b", "0",
                                   tuning={"wgsx": 48, "wgsy": 1})  # fmt: skip

    @pytest.mark.force_autotune
    def test_autotune(self, context):
        from katsdpsigproc_amd import reduce

        template = reduce.HReduceTemplate(context, np.float32, "float", "a + b", "0.0f")
        assert template.wgsx in (32, 64, 128) and 32 <= template.wgsx * template.wgsy <= 1024
