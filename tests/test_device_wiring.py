"""CPU tests of the device operation classes with a fake backend: slot names, shapes,
padding, kernel arguments and error behaviour (reference rfi/device.py contracts)."""

import numpy as np
import pytest

from katsdpsigproc_amd import accel, maskedsum, percentile, transpose
from katsdpsigproc_amd.rfi import device
from tests.fakes import FakeContext


@pytest.fixture
def context():
    return FakeContext()


@pytest.fixture
def queue(context):
    return context.create_command_queue()


def templates(context, use_flags=device.BackgroundFlags.NONE, noise_t=True, thr="sum", **kw):
    bg = device.BackgroundMedianFilterDeviceTemplate(context, 13, use_flags=use_flags)
    ne = (device.NoiseEstMADTDeviceTemplate(context, 10240) if noise_t
          else device.NoiseEstMADDeviceTemplate(context))  # fmt: skip
    if thr == "sum":
        th = device.ThresholdSumDeviceTemplate(context)
    else:
        th = device.ThresholdSimpleDeviceTemplate(context, thr == "simple_t")
    return device.FlaggerDeviceTemplate(bg, ne, th, **kw)


def test_background_flags_enum():
    assert not device.BackgroundFlags.NONE
    assert device.BackgroundFlags.CHANNEL and device.BackgroundFlags.FULL
    ctx = FakeContext()
    assert (device.BackgroundMedianFilterDeviceTemplate(ctx, 5, use_flags=True).use_flags
            is device.BackgroundFlags.CHANNEL)  # fmt: skip
    assert (device.BackgroundMedianFilterDeviceTemplate(ctx, 5, use_flags=False).use_flags
            is device.BackgroundFlags.NONE)  # fmt: skip
    with pytest.raises(TypeError):
        device.BackgroundMedianFilterDeviceTemplate(ctx, 5, use_flags=1)
    with pytest.raises(ValueError):
        device.BackgroundMedianFilterDeviceTemplate(ctx, 4)


def test_template_attributes(context):
    from katsdpsigproc_amd.rfi import host

    bg = device.BackgroundMedianFilterDeviceTemplate(context, 13)
    assert bg.host_class is host.BackgroundMedianFilterHost and bg.context is context
    assert device.NoiseEstMADDeviceTemplate(context).transposed is False
    assert device.NoiseEstMADTDeviceTemplate(context, 100).transposed is True
    assert device.ThresholdSumDeviceTemplate(context).transposed is True
    assert device.ThresholdSimpleDeviceTemplate(context, True).transposed is True
    assert device.ThresholdSumDeviceTemplate.host_class is host.ThresholdSumHost
    assert device.BackgroundMedianFilterDeviceTemplate.autotune(context, 13, False,
                                                                device.BackgroundFlags.NONE)  # fmt: skip


def test_background_op(context, queue):
    t = device.BackgroundMedianFilterDeviceTemplate(context, 13, use_flags=device.BackgroundFlags.FULL)
    fn = t.instantiate(queue, 100, 50)
    assert set(fn.slots) == {"vis", "deviations", "flags"}
    assert fn.slots["vis"].dtype == np.complex64 and fn.slots["flags"].shape == (100, 50)
    fn()
    name, args = queue.launches[-1]
    assert name == "ksp_background_median_filter"
    # channels, baselines, stride, flags stride, width, amp, mode, csplit (the stubbed
    # autotuner's test value). vis, deviations and flags share one Dimension; its
    # 128-element hint (uint8) exceeds 50, so no padding
    assert [int(a) for a in args[3:]] == [100, 50, 50, 50, 13, 0, 2, 4]
    assert fn.parameters()["csplit"] == 4
    tuned = device.BackgroundMedianFilterDeviceTemplate(context, 13, tuning={"csplit": 32})
    tuned.instantiate(queue, 100, 50)()
    assert int(queue.launches[-1][1][-1]) == 32
    assert (fn.buffer("vis").padded_shape == fn.buffer("deviations").padded_shape
            == fn.buffer("flags").padded_shape == (100, 50))  # fmt: skip
    wide = t.instantiate(queue, 10, 200)
    assert wide.slots["vis"].required_padded_shape() == (10, 256)
    assert fn.parameters()["use_flags"] == "FULL"
    chan = device.BackgroundMedianFilterDeviceTemplate(context, 5, True, True).instantiate(queue, 10, 4)
    assert chan.slots["flags"].shape == (10,) and chan.slots["vis"].dtype == np.float32


def test_noise_and_threshold_ops(context, queue):
    with pytest.raises(ValueError):
        device.NoiseEstMADTDeviceTemplate(context, 64).instantiate(queue, 65, 4)
    with pytest.raises(ValueError):
        device.NoiseEstMADTDeviceTemplate(context, 1 << 20)
    fn = device.NoiseEstMADTDeviceTemplate(context, 1024).instantiate(queue, 100, 7)
    assert fn.slots["deviations"].shape == (7, 100) and fn.slots["noise"].shape == (7,)
    fn = device.NoiseEstMADDeviceTemplate(context).instantiate(queue, 100, 7)
    assert fn.slots["deviations"].shape == (100, 7)
    fn = device.ThresholdSumDeviceTemplate(context, n_windows=3, flag_value=4).instantiate(
        queue, 100, 7, 9.0, 1.5)
    assert fn.slots["deviations"].shape == fn.slots["flags"].shape == (7, 100)
    assert fn.slots["deviations"].dimensions[1] is fn.slots["flags"].dimensions[1]
    fn()
    name, args = queue.launches[-1]
    assert name == "ksp_threshold_sum"
    assert list(args[7]) == [np.float32(1.0), np.float32(1 / 1.5), np.float32(1.5**-2)]
    assert int(args[8]) == 3 and int(args[9]) == 4 and int(args[10]) == 8  # vt: test value
    with pytest.raises(ValueError):
        device.ThresholdSumDeviceTemplate(context, n_windows=0)
    simple = device.ThresholdSimpleDeviceTemplate(context, True).instantiate(queue, 100, 7, 11.0)
    assert simple.slots["flags"].shape == (7, 100) and simple.slots["noise"].shape == (7,)


def test_flagger_sequence_wiring(context, queue):
    # the four layouts the reference wires (rfi/device.py:1135-1166)
    for noise_t, thr, expect in [
        (False, "simple", {"vis", "deviations", "noise", "flags"}),
        (True, "simple", {"vis", "deviations", "deviations_t", "noise", "flags"}),
        (False, "simple_t", {"vis", "deviations", "deviations_t", "noise", "flags_t", "flags"}),
        (True, "sum", {"vis", "deviations", "deviations_t", "noise", "flags_t", "flags"}),
    ]:
        fn = templates(context, noise_t=noise_t, thr=thr, fused=False).instantiate(
            queue, 64, 24, threshold_args={"n_sigma": 11.0})
        assert isinstance(fn, device.FlaggerDevice)
        assert set(fn.slots) == expect
        assert fn.slots["flags"].shape == (64, 24) and fn.slots["noise"].shape == (24,)
        before = len(queue.launches)
        fn()
        names = [n for n, _ in queue.launches[before:]]
        assert names[0] == "ksp_background_median_filter"
        assert names.count("ksp_transpose") == (noise_t or thr != "simple") + (thr != "simple")
        # one buffer is shared by producer and consumer
        assert fn.background.buffer("deviations") is fn.buffer("deviations")
        assert fn.noise_est.buffer("noise") is fn.threshold.buffer("noise")
    fn = templates(context, device.BackgroundFlags.CHANNEL, fused=False).instantiate(
        queue, 64, 24, threshold_args={"n_sigma": 11.0})
    assert fn.slots["input_flags"].shape == (64,)


def test_fused_selection_and_slots(context, queue):
    # the reference's slot set (rfi/device.py:1081-1150) is there, but its temporaries are
    # optional: not allocated by ensure_all_bound, not computed unless bound or asked for
    temporaries = {"deviations", "deviations_t", "flags_t"}
    fn = templates(context).instantiate(queue, 4096, 64, threshold_args={"n_sigma": 11.0})
    assert isinstance(fn, device.FusedFlaggerDevice)
    assert set(fn.slots) == {"vis", "noise", "flags"} | temporaries
    assert fn.parameters()["fused"] and not fn.parameters()["keep_deviations"]
    fn.ensure_all_bound()
    assert not any(fn.slots[name].is_bound() for name in temporaries)
    lean_bytes = fn.required_bytes()
    assert lean_bytes == sum(fn.slots[n].required_bytes() for n in ("vis", "noise", "flags"))
    before = len(queue.launches)
    fn()
    assert [n for n, _ in queue.launches[before:]] == ["ksp_flagger_fused"]
    name, args = queue.launches[-1]
    assert name == "ksp_flagger_fused" and args[3] is None
    # asking for a temporary materialises it; later calls fill it
    dev_t = fn.buffer("deviations_t")
    assert dev_t.shape == (64, 4096) and fn.slots["deviations_t"].is_bound()
    before = len(queue.launches)
    fn()
    assert [n for n, _ in queue.launches[before:]] == ["ksp_flagger_fused", "ksp_transpose"]
    assert queue.launches[before][1][3] is not None  # the kernel now writes deviations
    assert fn.slots["deviations"].is_bound() and fn.parameters()["keep_deviations"]
    assert fn.required_bytes() > lean_bytes
    fn.bind(flags_t=fn.slots["flags_t"].allocate(fn.allocator, bind=False))
    before = len(queue.launches)
    fn()
    assert [n for n, _ in queue.launches[before:]] == ["ksp_flagger_fused", "ksp_transpose", "ksp_transpose"]
    with pytest.raises(KeyError):
        fn.buffer("no_such_slot")
    fn = templates(context).instantiate(queue, 4096, 64, threshold_args={"n_sigma": 11.0})
    fn()
    name, args = queue.launches[-1]
    assert [int(a) for a in args[5:15]] == [4096, 64, 64, 0, 64, 0, 13, 0, 0, 1]
    fn = templates(context, keep_deviations=True).instantiate(
        queue, 4096, 64, threshold_args={"n_sigma": 11.0})
    assert set(fn.slots) == {"vis", "noise", "flags"} | temporaries
    fn.ensure_all_bound()
    assert fn.slots["deviations"].is_bound() and fn.parameters()["keep_deviations"]
    fn()
    name, args = queue.launches[-1]
    assert name == "ksp_flagger_fused"
    assert [int(a) for a in args[5:15]] == [4096, 64, 64, 0, 64, 64, 13, 0, 0, 1]
    assert args[15] == 11.0 and list(args[16]) == [1.2**-i for i in range(4)]
    lean = templates(context, device.BackgroundFlags.FULL, keep_deviations=False).instantiate(
        queue, 1024, 16, threshold_args={"n_sigma": 11.0, "threshold_falloff": 1.5})
    assert set(lean.slots) == {"vis", "input_flags", "noise", "flags"} | temporaries
    assert list(lean.scales) == [1.5**-i for i in range(4)]
    # row padding of vis: from the template's tuning, into the slot's requirement and the stride
    padded = templates(context, tuning={"vis_pad": 32}).instantiate(
        queue, 4096, 64, threshold_args={"n_sigma": 11.0})
    assert padded.slots["vis"].dimensions[1].required_padded_size() == 96
    assert padded.parameters()["vis_pad"] == 32
    padded.ensure_all_bound()
    padded()
    assert int(queue.launches[-1][1][7]) == 96  # vis_stride
    with pytest.raises(ValueError):
        templates(context, tuning={"vis_pad": 3}).instantiate(queue, 64, 8, threshold_args={"n_sigma": 1})
    # falls back to the sequence when the fused kernel cannot do it
    assert isinstance(templates(context, noise_t=False).instantiate(queue, 16384, 8, threshold_args={"n_sigma": 1}),
                      device.FlaggerDevice)  # fmt: skip
    with pytest.raises(ValueError):
        templates(context, noise_t=False, fused=True).instantiate(queue, 16384, 8, threshold_args={"n_sigma": 1})
    with pytest.raises(TypeError):
        templates(context).instantiate(queue, 64, 8)  # n_sigma missing
    with pytest.raises(TypeError):
        templates(context).instantiate(queue, 64, 8, threshold_args={"n_sigma": 1, "bogus": 2})


def test_host_from_device_type_errors(context, queue):
    vis = np.zeros((64, 8), np.complex64)
    flagger = device.FlaggerHostFromDevice(templates(context), queue, threshold_args={"n_sigma": 11.0})
    with pytest.raises(TypeError):
        flagger(vis, np.zeros(64, np.uint8))
    flagger = device.FlaggerHostFromDevice(
        templates(context, device.BackgroundFlags.CHANNEL), queue, threshold_args={"n_sigma": 11.0})
    with pytest.raises(TypeError):
        flagger(vis)
    bg = device.BackgroundHostFromDevice(
        device.BackgroundMedianFilterDeviceTemplate(context, 5), queue)
    with pytest.raises(TypeError):
        bg(vis, np.zeros(64, np.uint8))


def test_primitive_ops(context, queue):
    fn = transpose.TransposeTemplate(context, np.float32, "float").instantiate(queue, (53, 81))
    assert fn.slots["dest"].shape == (81, 53)
    fn()
    assert [int(a) for a in queue.launches[-1][1][2:]] == [53, 81, 64, 96, 4]
    transpose.TransposeTemplate(context, np.complex128, "double2")  # 16-byte elements are fine
    with pytest.raises(ValueError):
        transpose.TransposeTemplate(context, np.dtype([("a", "u1", 3)]), "uchar3")
    p = percentile.Percentile5Template(context, max_columns=5000)
    with pytest.raises(ValueError):
        p.instantiate(queue, (10, 100), (5, 5))
    with pytest.raises(IndexError):
        p.instantiate(queue, (10, 100), (-1, 5))
    with pytest.raises(ValueError):
        p.instantiate(queue, (10, 6000), (0, 5001))
    fn = p.instantiate(queue, (10, 100), (10, 60))
    assert fn.slots["dest"].shape == (5, 10)
    fn()
    assert [int(a) for a in queue.launches[-1][1][2:]] == [10, 128, 10, 10, 50, 1]
    with pytest.raises(ValueError):
        percentile.Percentile5Template(context, max_columns=100000)
    fn = maskedsum.MaskedSumTemplate(context, True).instantiate(queue, (4096, 30))
    assert fn.slots["dest"].dtype == np.float32 and fn.slots["mask"].shape == (4096,)
    assert maskedsum.MaskedSumTemplate(context).instantiate(queue, (4, 3)).slots["dest"].dtype == np.complex64


def test_accel_build_and_context_errors():
    with pytest.raises(FileNotFoundError):
        accel.render_template("x.hip.in")


def test_fixed_geometry_tuning(context):
    """Operations whose kernel has one geometry: the reference's tuning keys are accepted
    (and change nothing), anything else is an error; nothing pretends to have been tuned."""
    from katsdpsigproc_amd import maskedsum, percentile, transpose
    from katsdpsigproc_amd.rfi import device

    cases = [
        (lambda **kw: transpose.TransposeTemplate(context, np.float32, "float", **kw), {"block": 8, "vtx": 2, "vty": 3}),
        (lambda **kw: percentile.Percentile5Template(context, 4096, **kw), {"size": 64, "wgsy": 4}),
        (lambda **kw: maskedsum.MaskedSumTemplate(context, **kw), {"size": 256}),
        (lambda **kw: device.NoiseEstMADTDeviceTemplate(context, 4096, **kw), {"wgsx": 128}),
        (lambda **kw: device.ThresholdSimpleDeviceTemplate(context, False, **kw), {"wgsx": 32, "wgsy": 4}),
    ]  # fmt: skip
    for make, reference_tuning in cases:
        assert make().tuning == {}
        assert make(tuning=reference_tuning).tuning == reference_tuning
        with pytest.raises(ValueError, match="fixed geometry"):
            make(tuning={"wavefronts": 2})
    assert transpose.TransposeTemplate.autotune(context, np.float32, "float") == {}
