"""GPU parity tests of the flagger: the reference-shaped kernel sequence, the fused
single-pass kernel (bit-identical to FlaggerHost), and the raw C-ABI entry point."""

import contextlib
import ctypes
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


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.uint8)


def digest(a):
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a + 0.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def make_template(context, use_flags="NONE", threshold="sum", noise="MADT", width=13, **kw):
    from katsdpsigproc_amd.rfi import device

    bg = device.BackgroundMedianFilterDeviceTemplate(
        context, width, use_flags=device.BackgroundFlags[use_flags]
    )
    ne = (
        device.NoiseEstMADTDeviceTemplate(context, 10240)
        if noise == "MADT"
        else device.NoiseEstMADDeviceTemplate(context)
    )
    if threshold == "sum":
        th = device.ThresholdSumDeviceTemplate(context)
    elif threshold == "simple_t":
        th = device.ThresholdSimpleDeviceTemplate(context, True)
    else:
        th = device.ThresholdSimpleDeviceTemplate(context, False)
    kw.setdefault("keep_deviations", True)  # the parity tests compare them
    return device.FlaggerDeviceTemplate(bg, ne, th, **kw)


def run_fused(template, command_queue, vis, in_flags=None, **threshold_args):
    from katsdpsigproc_amd.rfi import device

    fn = template.instantiate(
        command_queue, vis.shape[0], vis.shape[1], threshold_args=threshold_args
    )
    assert isinstance(fn, device.FusedFlaggerDevice)
    fn.ensure_all_bound()
    fn.buffer("vis").set(command_queue, vis)
    if in_flags is not None:
        fn.buffer("input_flags").set(command_queue, in_flags)
    fn()
    out = {"flags": fn.buffer("flags").get(command_queue),
           "noise": fn.buffer("noise").get(command_queue)}  # fmt: skip
    if template.keep_deviations:
        out["deviations"] = fn.buffer("deviations").get(command_queue)
    check_ring_path(template, command_queue, vis, in_flags, threshold_args, out)
    return out


@contextlib.contextmanager
def force_ring():
    """Launches of fewer than about 8192 baselines are left to the 4-baseline kernel unless
    told otherwise (ksp_flagger_fused_ring_mode): the tests of the ring kernel on small arrays
    say so."""
    from katsdpsigproc_amd import _lib

    previous = _lib.call("ksp_flagger_fused_ring_mode", 1)
    try:
        yield
    finally:
        _lib.call("ksp_flagger_fused_ring_mode", previous)


def check_ring_path(template, command_queue, vis, in_flags, threshold_args, out):
    """Launches that keep the deviations take the 4-baseline kernel; without them a
    4096-channel launch of complex visibilities without input flags (width 13, at least 8
    baselines) takes the persistent ring kernel. Every parity test of such a shape therefore
    runs it as well: same flags, same noise as the run the test compares with the oracle,
    and ksp_flagger_fused_last_path says the ring kernel did it."""
    from katsdpsigproc_amd import _lib
    from katsdpsigproc_amd.rfi import device

    bg = template.background
    if not (vis.shape[0] == 4096 and vis.shape[1] >= 8 and in_flags is None
            and np.iscomplexobj(vis) and bg.width == 13 and not bg.is_amplitude
            and "deviations" in out):  # fmt: skip
        return
    lean = device.FlaggerDeviceTemplate(bg, template.noise_est, template.threshold, fused=True,
                                        keep_deviations=False, tuning=template._fused_tuning)  # fmt: skip
    fn = lean.instantiate(command_queue, vis.shape[0], vis.shape[1], threshold_args=threshold_args)
    fn.ensure_all_bound()
    fn.buffer("vis").set(command_queue, vis)
    for _ in range(2):  # (twice: the scheduling counters must be left as they were found)
        fn.buffer("flags").set(command_queue, np.full(vis.shape, 255, np.uint8))
        with force_ring():
            fn()
        path = _lib.call("ksp_flagger_fused_last_path")
        assert path & 4, f"expected the ring kernel, last path = {path}"
        assert (path & 1) == (1 if vis.shape[1] % 8 else 0)
        np.testing.assert_array_equal(fn.buffer("flags").get(command_queue), out["flags"])
        np.testing.assert_array_equal(fn.buffer("noise").get(command_queue), out["noise"])


class TestSequence:
    """The kernel-per-stage FlaggerDevice, as the reference tests it."""

    @pytest.mark.parametrize(
        "use_flags, transpose_noise_est, transpose_threshold",
        [("NONE", False, False), ("CHANNEL", True, False), ("FULL", False, True),
         ("NONE", True, True)],
    )  # fmt: skip
    def test_flagger_device(self, use_flags, transpose_noise_est, transpose_threshold, context,
                            command_queue):  # fmt: skip
        # reference test/rfi/test_flagger.py:74-132
        from katsdpsigproc_amd.rfi import device

        vis, spikes, input_flags = inputs.flagger_case()
        template = make_template(
            context, use_flags, "simple_t" if transpose_threshold else "simple",
            "MADT" if transpose_noise_est else "MAD", fused=False,
        )  # fmt: skip
        fn = template.instantiate(command_queue, *vis.shape, threshold_args=dict(n_sigma=11.0))
        assert isinstance(fn, device.FlaggerDevice)
        assert ("deviations_t" in fn.slots) == (transpose_noise_est or transpose_threshold)
        assert ("flags_t" in fn.slots) == transpose_threshold
        flagger = device.FlaggerHostFromDevice(
            template, command_queue, threshold_args=dict(n_sigma=11.0)
        )
        if use_flags == "CHANNEL":
            flags = flagger(vis, input_flags[:, 0])
            bcast = np.broadcast_to(input_flags[:, 0:1], vis.shape)
            np.testing.assert_array_equal(np.where(bcast, 0, spikes), flags)
        elif use_flags == "FULL":
            flags = flagger(vis, input_flags)
            np.testing.assert_array_equal(np.where(input_flags, 0, spikes), flags)
        else:
            np.testing.assert_array_equal(spikes, flagger(vis))

    def test_sequence_sum_matches_staged_oracle(self, context, command_queue, oracle):
        """Stage-by-stage parity: the sequence equals the oracle stages chained through
        float32 intermediates (which is what separate device buffers imply)."""
        vis = inputs.add_rfi(inputs.generate_data(1024, 200, seed=7), seed=8)
        template = make_template(context, fused=False)
        fn = template.instantiate(command_queue, *vis.shape, threshold_args=dict(n_sigma=11.0))
        fn.ensure_all_bound()
        fn.buffer("vis").set(command_queue, vis)
        fn()
        dev32 = oracle.BackgroundMedianFilterHost(13)(vis).astype(np.float32)
        noise32 = oracle.NoiseEstMADHost()(dev32).astype(np.float32)
        flags = oracle.ThresholdSumHost(11.0)(dev32, noise32)
        np.testing.assert_array_equal(dev32, fn.buffer("deviations").get(command_queue))
        np.testing.assert_array_equal(dev32.T, fn.buffer("deviations_t").get(command_queue))
        np.testing.assert_array_equal(noise32, fn.buffer("noise").get(command_queue))
        np.testing.assert_array_equal(flags.T, fn.buffer("flags_t").get(command_queue))
        np.testing.assert_array_equal(flags, fn.buffer("flags").get(command_queue))

    def test_type_errors(self, context, command_queue):
        from katsdpsigproc_amd.rfi import device

        vis, _, input_flags = inputs.flagger_case()
        flagger = device.FlaggerHostFromDevice(
            make_template(context), command_queue, threshold_args=dict(n_sigma=11.0)
        )
        with pytest.raises(TypeError):
            flagger(vis, input_flags)
        flagger = device.FlaggerHostFromDevice(
            make_template(context, "FULL"), command_queue, threshold_args=dict(n_sigma=11.0)
        )
        with pytest.raises(TypeError):
            flagger(vis)


class TestFused:
    """Fused single-pass kernel: bit-identical to FlaggerHost."""

    @pytest.mark.parametrize("threshold", ["simple", "sum"])
    @pytest.mark.parametrize("mode", ["none", "channel", "full"])
    def test_golden_flagger_case(self, threshold, mode, golden, context, command_queue, oracle):
        vis, spikes, in_flags = inputs.flagger_case()
        fl = {"none": None, "channel": in_flags[:, 0], "full": in_flags}[mode]
        template = make_template(context, mode.upper(), threshold)
        out = run_fused(template, command_queue, vis, fl, n_sigma=11.0)
        np.testing.assert_array_equal(
            unpack(golden[f"flagger_{threshold}_{mode}"], vis.shape), out["flags"]
        )
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(
            vis, fl, threshold=threshold, want_deviations=True
        )
        np.testing.assert_array_equal(ref_flags, out["flags"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        if mode == "full":
            np.testing.assert_array_equal(
                golden["flagger_dev_full"].astype(np.float32), out["deviations"]
            )
            np.testing.assert_array_equal(
                golden["flagger_noise_full"].astype(np.float32), out["noise"]
            )

    @pytest.mark.parametrize("tag", ["cfg1", "cfg1rfi"])
    def test_config1_golden(self, tag, golden, context, command_queue):
        """BASELINE.json config 1 (1024 x 2048): digest of flags from the real reference."""
        vis = inputs.config1() if tag == "cfg1" else inputs.config1_rfi()
        out = run_fused(make_template(context), command_queue, vis, n_sigma=11.0)
        assert int(out["flags"].astype(np.int64).sum()) == int(golden[f"{tag}_flags_count"])
        assert digest(out["flags"]) == str(golden[f"{tag}_flags_sha"])
        np.testing.assert_array_equal(golden[f"{tag}_noise"].astype(np.float32), out["noise"])
        np.testing.assert_array_equal(
            golden[f"{tag}_dev_cols"].astype(np.float32), out["deviations"][:, inputs.CFG1_COLS]
        )
        # tolerance of the north star, stated: |dev - host| <= 1e-5 (here it is 0)
        assert np.max(np.abs(out["deviations"][:, inputs.CFG1_COLS] - golden[f"{tag}_dev_cols"])) <= 1e-5

    @pytest.mark.parametrize(
        "channels, baselines",
        [(1, 1), (13, 8), (14, 3), (100, 17), (256, 8), (257, 9), (1000, 37), (1024, 64),
         (1025, 5), (4096, 24), (4095, 11), (4096, 9), (4096, 3), (2048, 13)],
    )  # fmt: skip
    @pytest.mark.parametrize("mode", ["none", "full"])
    def test_ragged_shapes(self, channels, baselines, mode, context, command_queue, oracle):
        """Empty-ish, ragged and maximum sizes; partial strips; band edges."""
        rs = np.random.RandomState(channels * 131 + baselines)
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=5), seed=6, fraction=0.1)
        fl = None
        if mode == "full":
            fl = (rs.random_sample(vis.shape) < 0.15).astype(np.uint8) * 3
            if channels > 40:
                fl[20:40, :] = 1  # a fully flagged block: windows with no valid sample
        template = make_template(context, mode.upper())
        out = run_fused(template, command_queue, vis, fl, n_sigma=11.0)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, fl, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("width", [3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 27, 29, 31])
    @pytest.mark.parametrize("channels, baselines, mode",
                             [(4096, 12, "none"), (4096, 8, "channel"), (1000, 9, "full"),
                              (40, 5, "none")])  # fmt: skip
    def test_other_widths(self, width, channels, baselines, mode, context, command_queue, oracle):
        """Median windows other than the reference script's 13 (reference
        rfi/device.py:141-262 takes any odd width): full band (merging median up to width
        13, sorted window beyond), masked data, ragged shapes, bands shorter than a
        window."""
        rs = np.random.RandomState(width * 1000 + channels)
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=width), seed=width + 1,
                             fraction=0.08)  # fmt: skip
        fl = None
        if mode == "channel":
            fl = inputs.channel_mask(channels, seed=width)
        elif mode == "full":
            fl = (rs.random_sample(vis.shape) < 0.12).astype(np.uint8) * 2
        template = make_template(context, mode.upper(), width=width)
        out = run_fused(template, command_queue, vis, fl, n_sigma=9.0)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, fl, width=width, n_sigma=9.0,
                                                            want_deviations=True)  # fmt: skip
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        assert ref_flags.sum() > 0
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize(
        "channels, baselines, mode",
        [(8192, 9, "none"), (10240, 7, "none"), (8192, 5, "channel"), (10240, 4, "full"),
         (4097, 6, "none"), (5000, 3, "full"), (9088, 5, "none"), (9089, 4, "none"),
         (12288, 4, "none"), (12287, 2, "channel"), (8191, 9, "none")],
    )  # fmt: skip
    def test_long_bands(self, channels, baselines, mode, context, command_queue, oracle):
        """More than 4096 channels (the reference script's 8192- and 10240-channel presets,
        scripts/rfiflagtest.py:190-195): lanes own two or three runs of 64 channels;
        strips of 4 baselines up to 9088 channels, of 3 beyond; whole and partial runs,
        ragged strips, every input-flags mode; weak interference so that SumThreshold's
        wider windows fire across run and group boundaries."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(channels + baselines)
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=3), seed=4, fraction=0.05)
        for b in range(baselines):
            for _ in range(10):
                c = rs.randint(0, channels - 12)
                w = rs.randint(1, 13)
                vis[c : c + w, b] += rs.uniform(2.0, 9.0) * np.exp(2j * np.pi * rs.rand())
        # interference straddling the group boundaries at 4096 and 8192
        vis[4090:4100, 0] += 6.0
        if channels > 8200:
            vis[8188:8197, 1] += 5.0
        fl = None
        if mode == "channel":
            fl = inputs.channel_mask(channels)
        elif mode == "full":
            fl = (rs.random_sample(vis.shape) < 0.1).astype(np.uint8) * 5
        bg = device.BackgroundMedianFilterDeviceTemplate(context, 13, use_flags=device.BackgroundFlags[mode.upper()])
        ne = device.NoiseEstMADTDeviceTemplate(context, 16384)
        th = device.ThresholdSumDeviceTemplate(context)
        template = device.FlaggerDeviceTemplate(bg, ne, th, keep_deviations=True)
        out = run_fused(template, command_queue, vis, fl, n_sigma=7.0)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, fl, n_sigma=7.0, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        assert ref_flags.sum() > 0
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("channels", [8192, 10240])
    def test_long_bands_degenerate(self, channels, context, command_queue, oracle):
        """All-zero and constant baselines, heavy ties (crowded key bins, even and odd
        counts), denormal-sized deviations, NaN and infinite visibilities at 8192 / 10240
        channels, with Simple and Sum thresholds."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(channels)
        baselines = 11
        vis = inputs.generate_data(channels, baselines, seed=51)
        vis[:, 0] = 0
        vis[:, 1] = 3 + 4j
        vis[:, 2] = (rs.randint(0, 4, channels) + 0j).astype(np.complex64)
        vis[:, 3] = (rs.randint(0, 50, channels) * 0.25 + 0j).astype(np.complex64)
        vis[100, 3] = 1000.0
        vis[:, 4] = (rs.randint(0, 3, channels) * 1e-42 + 0j).astype(np.complex64)
        vis[:, 5] = (rs.randint(0, 1000, channels) * 2.0 ** -10 + 0j).astype(np.complex64)
        vis[::2, 6] = vis[1::2, 6]
        vis[5000, 7] = np.nan
        vis[4095, 8] = np.inf
        level = np.float32(1.25)
        vis[:, 9] = level
        dips = 3 * rs.permutation(channels // 3)[:600] + 1
        vis[dips, 9] = (rs.randint(1, 1 << 16, 600) * 2.0 ** -40).astype(np.float32)
        for kind in ("sum", "simple"):
            bg = device.BackgroundMedianFilterDeviceTemplate(context, 13)
            ne = device.NoiseEstMADDeviceTemplate(context)
            th = (device.ThresholdSumDeviceTemplate(context) if kind == "sum"
                  else device.ThresholdSimpleDeviceTemplate(context, False))  # fmt: skip
            template = device.FlaggerDeviceTemplate(bg, ne, th, keep_deviations=True)
            out = run_fused(template, command_queue, vis, None, n_sigma=11.0)
            with np.errstate(all="ignore"):
                ref_flags, ref_noise, ref_dev = oracle.flagger_full(
                    vis, threshold=kind, want_deviations=True)  # fmt: skip
            np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
            np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
            np.testing.assert_array_equal(ref_flags, out["flags"])

    def test_no_deviations_slot(self, context, command_queue, oracle):
        vis = inputs.add_rfi(inputs.generate_data(512, 40, seed=9), seed=10)
        template = make_template(context, keep_deviations=False)
        out = run_fused(template, command_queue, vis, n_sigma=11.0)
        assert "deviations" not in out
        np.testing.assert_array_equal(oracle.flagger_full(vis)[0], out["flags"])

    @pytest.mark.parametrize("channels, baselines", [(4096, 24), (600, 30)])
    def test_optional_slots(self, channels, baselines, context, command_queue, oracle):
        """The reference's temporaries on the default flagger (reference rfi/device.py:1081-1150):
        left alone by ensure_all_bound, materialised by buffer() / bind(), filled by the calls
        that follow -- deviations by the kernel, deviations_t and flags_t by a transpose."""
        from katsdpsigproc_amd import _lib
        from katsdpsigproc_amd.rfi import device

        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=19), seed=20)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, want_deviations=True)
        template = make_template(context, keep_deviations=False)
        fn = template.instantiate(command_queue, channels, baselines, threshold_args={"n_sigma": 11.0})
        assert isinstance(fn, device.FusedFlaggerDevice)
        fn.ensure_all_bound()
        assert not fn.slots["deviations"].is_bound()
        fn.buffer("vis").set(command_queue, vis)
        with force_ring():
            fn()
        if channels == 4096:
            assert _lib.call("ksp_flagger_fused_last_path") & 4  # nothing optional: the ring kernel
        np.testing.assert_array_equal(ref_flags, fn.buffer("flags").get(command_queue))
        flags_t = fn.buffer("flags_t")  # materialised now, filled by the next call
        dev_t = fn.buffer("deviations_t")
        assert flags_t.shape == (baselines, channels) and dev_t.shape == (baselines, channels)
        fn()
        np.testing.assert_array_equal(ref_flags, fn.buffer("flags").get(command_queue))
        np.testing.assert_array_equal(ref_flags.T, flags_t.get(command_queue))
        np.testing.assert_array_equal(ref_dev.astype(np.float32).T, dev_t.get(command_queue))
        np.testing.assert_array_equal(ref_dev.astype(np.float32), fn.buffer("deviations").get(command_queue))
        np.testing.assert_array_equal(ref_noise.astype(np.float32), fn.buffer("noise").get(command_queue))

    def test_amplitude_input_and_params(self, context, command_queue, oracle):
        from katsdpsigproc_amd.rfi import device

        vis = inputs.add_rfi(inputs.generate_data(600, 30, seed=11), seed=12)
        amp = oracle.abs_c64(vis)
        bg = device.BackgroundMedianFilterDeviceTemplate(context, 13, is_amplitude=True)
        ne = device.NoiseEstMADDeviceTemplate(context)
        th = device.ThresholdSumDeviceTemplate(context, n_windows=3, flag_value=5)
        template = device.FlaggerDeviceTemplate(bg, ne, th)  # default: no deviations slot
        out = run_fused(template, command_queue, amp, n_sigma=6.5, threshold_falloff=1.5)
        assert "deviations" not in out
        ref_flags, ref_noise = oracle.flagger_full(
            amp, amplitudes=True, n_sigma=6.5, n_windows=3, threshold_falloff=1.5, flag_value=5
        )
        np.testing.assert_array_equal(ref_flags, out["flags"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])

    def test_degenerate_data(self, context, command_queue, oracle):
        """All-zero baselines (noise NaN, no flags), constant data, quantised data with
        many exact ties in the deviations."""
        rs = np.random.RandomState(4)
        channels, baselines = 512, 16
        vis = inputs.generate_data(channels, baselines, seed=13)
        vis[:, 0] = 0  # all deviations zero -> NaN noise
        vis[:, 1] = 3 + 4j  # constant -> all deviations zero
        vis[:, 2] = (rs.randint(0, 4, channels) + 0j).astype(np.complex64)  # heavy ties
        vis[:, 3] = (rs.randint(0, 50, channels) * 0.25 + 0j).astype(np.complex64)
        vis[100, 3] = 1000.0
        out = run_fused(make_template(context), command_queue, vis, n_sigma=11.0)
        with np.errstate(all="ignore"):
            ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        assert np.isnan(out["noise"][0]) and np.isnan(out["noise"][1])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    def test_full_band_mixed_strips(self, context, command_queue, oracle):
        """4096 channels (the shape that takes the merging median): strips with a NaN
        visibility fall back to the sorted-window path, strips with infinite samples
        stay on the merging path; band edges of the first and last lane in both."""
        channels, baselines = 4096, 16
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=41), seed=42)
        vis[100, 5] = np.nan
        vis[0, 6] = np.nan + 1j
        vis[4095, 7] = np.nan
        vis[200, 9] = np.inf
        vis[3, 10] = np.inf * 1j
        vis[4090, 11] = np.inf
        vis[:, 12] = 0
        out = run_fused(make_template(context), command_queue, vis, n_sigma=11.0)
        with np.errstate(all="ignore"):
            ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("channels, baselines, vis_pad, mode", [
        (4096, 50, 32, "none"), (4096, 33, 512, "full"), (1000, 21, 16, "channel"),
        (8192, 10, 32, "none"), (512, 7, 2048, "none"),
    ])  # fmt: skip
    def test_padded_rows(self, channels, baselines, vis_pad, mode, context, command_queue, oracle):
        """``tuning={"vis_pad": n}``: rows of `vis` padded beyond the baselines (the stride
        the autotuner picks for large blocks) change nothing in the results."""
        from katsdpsigproc_amd.rfi import device

        rs = np.random.RandomState(5)
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=71), seed=72)
        fl = None
        if mode == "channel":
            fl = inputs.channel_mask(channels)
        elif mode == "full":
            fl = (rs.random_sample(vis.shape) < 0.1).astype(np.uint8)
        template = make_template(context, mode.upper(), tuning={"vis_pad": vis_pad})
        fn = template.instantiate(command_queue, channels, baselines, threshold_args={"n_sigma": 9.0})
        assert fn.slots["vis"].dimensions[1].required_padded_size() >= baselines + vis_pad
        out = run_fused(template, command_queue, vis, fl, n_sigma=9.0)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, fl, n_sigma=9.0, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.force_autotune
    def test_autotune_vis_pad(self, context, command_queue, oracle):
        """The search over row paddings runs at the instantiated shape and returns one of
        its candidates; results with the chosen padding match the oracle."""
        from katsdpsigproc_amd.rfi import device

        channels, baselines = 4096, 2048
        template = make_template(context)  # no tuning given: autotuned at instantiation
        fn = template.instantiate(command_queue, channels, baselines, threshold_args={"n_sigma": 11.0})
        assert fn.vis_pad in device.FlaggerDeviceTemplate._VIS_PADS
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=73), seed=74)
        oracle.set_threads(16)
        try:
            ref_flags, ref_noise = oracle.flagger_full(vis, n_sigma=11.0)
        finally:
            oracle.set_threads(1)
        fn.ensure_all_bound()
        fn.buffer("vis").set(command_queue, vis)
        fn()
        np.testing.assert_array_equal(ref_flags, fn.buffer("flags").get(command_queue))
        np.testing.assert_array_equal(ref_noise.astype(np.float32), fn.buffer("noise").get(command_queue))

    @pytest.mark.parametrize("channels", [4096, 8192])
    def test_magnitude_range(self, channels, context, command_queue, oracle):
        """The loaders take a shorter division for batches of ordinary magnitudes
        (2^-63 <= max(|re|, |im|) < 2^65) and the general one otherwise: baselines scaled to
        either side of both limits, and exact zeros, tiny and huge samples sprinkled over
        ordinary data so that batches of either kind alternate inside a strip."""
        rs = np.random.RandomState(61)
        baselines = 24
        vis = inputs.generate_data(channels, baselines, seed=62).astype(np.complex128)
        for bl, scale in enumerate([2.0**-64, 2.0**-62, 2.0**-61, 2.0**62, 2.0**63, 2.0**64,
                                    2.0**66, 2.0**-70, 2.0**-100, 2.0**100]):  # fmt: skip
            vis[:, 4 + bl] *= scale
        vis = vis.astype(np.complex64)
        hit = rs.random_sample(vis.shape) < 0.002
        hit[:, 4:14] = False
        kind = rs.randint(0, 4, vis.shape)
        vis[hit & (kind == 0)] = 0
        vis[hit & (kind == 1)] *= np.float32(2.0**-90)
        vis[hit & (kind == 2)] *= np.float32(2.0**80)
        vis[hit & (kind == 3)] = np.complex64(complex(1.0, 2.0**-80))
        out = run_fused(make_template(context), command_queue, vis, n_sigma=11.0)
        with np.errstate(all="ignore"):
            ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("mode", ["none", "channel"])
    def test_degenerate_full_band(self, mode, context, command_queue, oracle):
        """4096 channels (bit-plane MAD, merging or sorted-window median): all-zero and
        constant baselines, heavily quantised data (hundreds of samples per key bin, ties at
        the median rank, even and odd counts), denormal-sized deviations, one huge outlier."""
        rs = np.random.RandomState(8)
        channels, baselines = 4096, 12
        vis = inputs.generate_data(channels, baselines, seed=51)
        vis[:, 0] = 0
        vis[:, 1] = 3 + 4j
        vis[:, 2] = (rs.randint(0, 4, channels) + 0j).astype(np.complex64)
        vis[:, 3] = (rs.randint(0, 50, channels) * 0.25 + 0j).astype(np.complex64)
        vis[100, 3] = 1000.0
        vis[:, 4] = (rs.randint(0, 3, channels) * 1e-42 + 0j).astype(np.complex64)
        vis[:, 5] = (rs.randint(0, 1000, channels) * 2.0 ** -10 + 0j).astype(np.complex64)
        vis[::2, 6] = vis[1::2, 6]  # pairs of equal samples
        vis[:, 7] = (np.arange(channels) % 7 + 1j * (np.arange(channels) % 5)).astype(np.complex64)
        fl = None
        if mode == "channel":
            fl = (rs.random_sample(channels) < 1 / 16).astype(np.uint8)
        out = run_fused(make_template(context, mode.upper()), command_queue, vis, fl, n_sigma=11.0)
        with np.errstate(all="ignore"):
            ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, fl, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("channels", [4096, 3000, 1024])
    def test_mad_many_float32_ties(self, channels, context, command_queue, oracle):
        """Hundreds of samples whose deviations share ONE float32 value while their
        float64 values differ (constant level L in [1, 2) with isolated dips to tiny,
        distinct amplitudes t: |d| = L - t rounds to L in float32 for t < 2^-25 but not
        in float64). numpy's median is taken on the float64 values, and 1.4826 x median
        then rounds to a different float32 than 1.4826 x L for some baselines -- the
        selection among the ties has to be exact. Even and odd counts, ties above the
        candidate-list capacity (256) and below it."""
        rs = np.random.RandomState(channels)
        baselines = 28
        amp = np.empty((channels, baselines), np.float32)
        for b in range(baselines):
            level = np.float32(1.0 + 0.3 * rs.random_sample())
            amp[:, b] = level
            n_dips = [channels // 3, channels // 3 - 1, 300, 257, 255, 64, 40][b % 7]
            where = 3 * rs.permutation(channels // 3)[:n_dips] + 1
            amp[where, b] = (rs.randint(1, 1 << 16, n_dips) * 2.0 ** -40).astype(np.float32)
            amp[3 * rs.randint(0, channels // 3, 4), b] = 100.0  # something to flag
        vis = amp.astype(np.complex64)
        out = run_fused(make_template(context), command_queue, vis, n_sigma=11.0)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, want_deviations=True)
        # the test has teeth: float32 ties whose float64 median rounds differently
        d32 = np.abs(ref_dev.astype(np.float32))
        naive = np.where(d32 < 50, d32, 0).max(axis=0).astype(np.float64) * 1.4826
        assert np.count_nonzero(naive.astype(np.float32) != ref_noise.astype(np.float32)) >= 3, "no teeth"
        assert ref_flags.sum() > 0
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("kind", ["amplitude", "complex", "long", "short"])
    def test_deviations_of_2_to_minus_150(self, kind, golden, context, command_queue, oracle):
        """Deviations of exactly +-2^-150 (an even-count window over subnormal amplitudes):
        zero as float32, yet counted among the non-zero deviations by rfi/host.py:161. The
        noise estimates are those of the imported reference (golden); they differ from what
        taking such deviations for zeros would give. Amplitude input takes the 4-baseline
        kernel, complex input the ring kernel as well, 8192 channels the long-band kernel,
        256 channels the short-run variant."""
        from katsdpsigproc_amd.rfi import device

        channels = {"long": 8192, "short": 256}.get(kind, 4096)
        amp = inputs.denormal_case(channels)
        is_amp = kind != "complex"
        vis = amp if is_amp else amp.astype(np.complex64)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, amplitudes=is_amp, want_deviations=True)
        assert (np.abs(ref_dev) == 2.0 ** -150).any(axis=0).all()
        if channels == 4096:
            np.testing.assert_array_equal(ref_noise, golden["denormal_noise"])
        naive = np.array([np.median(d[(d > 0) & (d != 2.0 ** -150)]) * 1.4826 for d in np.abs(ref_dev).T])
        assert np.all(naive.astype(np.float32) != ref_noise.astype(np.float32))  # teeth
        if is_amp:
            bg = device.BackgroundMedianFilterDeviceTemplate(context, 13, is_amplitude=True)
            template = device.FlaggerDeviceTemplate(
                bg, device.NoiseEstMADTDeviceTemplate(context, 10240),
                device.ThresholdSumDeviceTemplate(context), keep_deviations=True)
        else:
            template = make_template(context)
        out = run_fused(template, command_queue, vis, n_sigma=11.0)
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @pytest.mark.parametrize("seed", [1, 2, 3])
    def test_full_band_interference_kinds(self, seed, context, command_queue, oracle):
        """4096 channels with narrow-band, broad-band (runs of 2-12 channels: windows 2, 4
        and 8 of SumThreshold fire) and weak interference just around the thresholds."""
        rs = np.random.RandomState(seed)
        channels, baselines = 4096, 20
        vis = inputs.generate_data(channels, baselines, seed=60 + seed)
        for b in range(baselines):
            for _ in range(12):
                c = rs.randint(0, channels - 12)
                w = rs.randint(1, 13)
                vis[c:c + w, b] += rs.uniform(2.0, 9.0) * np.exp(2j * np.pi * rs.rand())
            vis[rs.randint(0, channels, 5), b] *= rs.uniform(20, 60)
        out = run_fused(make_template(context), command_queue, vis, n_sigma=5.0 + seed)
        ref_flags, ref_noise, ref_dev = oracle.flagger_full(vis, n_sigma=5.0 + seed, want_deviations=True)
        np.testing.assert_array_equal(ref_dev.astype(np.float32), out["deviations"])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        assert ref_flags.sum() > 0
        np.testing.assert_array_equal(ref_flags, out["flags"])

    @staticmethod
    def _broad_interference(channels, baselines, seed):
        """Noise with combs -- every 2nd to 4th channel of a stretch of 20 .. 300 raised by a
        factor of 1.5 .. 4, which the median filter leaves standing and only sums over many
        channels find --, a few strong spikes, a stretch longer than the widest window that
        ends up flagged throughout, and a comb up to the band's end."""
        rs = np.random.RandomState(seed)
        vis = inputs.generate_data(channels, baselines, seed=seed + 100)
        for b in range(baselines):
            for _ in range(4):
                length = rs.randint(20, min(300, channels // 2))
                c = rs.randint(0, channels - length)
                vis[c:c + length:rs.randint(2, 5), b] *= rs.uniform(1.5, 4.0)
            vis[rs.randint(0, channels, 3), b] *= rs.uniform(20, 60)
            if channels >= 600 and b % 3 == 0:
                c = rs.randint(0, channels - 400)
                vis[c:c + 300:2, b] *= 40.0  # (across several lanes' runs of 64 channels)
            if channels > 200 and b % 4 == 1:
                vis[channels - 70::3, b] *= 3.0
        return vis

    @pytest.mark.parametrize("n_windows", [5, 6, 7, 8])
    @pytest.mark.parametrize("channels, baselines, mode, width",
                             [(4096, 20, "none", 13), (1000, 13, "full", 13), (273, 9, "channel", 5),
                              (130, 6, "none", 13), (64, 5, "none", 3), (2048, 9, "none", 25)])  # fmt: skip
    def test_wide_windows(self, n_windows, channels, baselines, mode, width, context,
                          command_queue, oracle):  # fmt: skip
        """SumThreshold with 5 .. 8 windows (16 .. 128 channels) in the fused kernel, against
        the oracle (itself pinned for 6 and 8 windows by goldens of the reference)."""
        from katsdpsigproc_amd.rfi import device

        vis = self._broad_interference(channels, baselines, seed=n_windows * 10 + width)
        in_flags = None
        if mode == "channel":
            in_flags = inputs.channel_mask(channels)
        elif mode == "full":
            in_flags = (np.random.RandomState(5).rand(channels, baselines) < 0.05).astype(np.uint8)
        bg = device.BackgroundMedianFilterDeviceTemplate(
            context, width, use_flags=device.BackgroundFlags[mode.upper()])
        th = device.ThresholdSumDeviceTemplate(context, n_windows=n_windows)
        template = device.FlaggerDeviceTemplate(bg, device.NoiseEstMADTDeviceTemplate(context, 10240),
                                                th, fused=True)  # fmt: skip
        for n_sigma, falloff in ((6.0, 1.5), (11.0, 1.5), (6.0, 1.2)):
            out = run_fused(template, command_queue, vis, in_flags, n_sigma=n_sigma,
                            threshold_falloff=falloff)  # fmt: skip
            ref_flags, ref_noise = oracle.flagger_full(
                vis, in_flags, width=width, n_sigma=n_sigma, n_windows=n_windows,
                threshold_falloff=falloff)  # fmt: skip
            np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
            if channels >= 1000 and falloff == 1.5:
                # the widest window has found something the narrower ones do not
                narrow, _ = oracle.flagger_full(vis, in_flags, width=width, n_sigma=n_sigma,
                                                n_windows=n_windows - 1, threshold_falloff=falloff)  # fmt: skip
                assert (ref_flags != narrow).sum() > 0
            np.testing.assert_array_equal(ref_flags, out["flags"])

    def test_wide_windows_degenerate(self, context, command_queue, oracle):
        """8 windows where the quick decisions do not apply: a threshold that is not positive
        (n_sigma 0: every window is summed as the host does it), a huge downward deviation
        (the error bound scales with it) and all-zero baselines."""
        from katsdpsigproc_amd.rfi import device

        vis = self._broad_interference(300, 6, seed=77)
        vis[100, 1] = 0.0
        vis[90:110, 1] *= 1e6
        vis[100, 1] = 0.0  # deviation of about -1e6 noise units
        vis[:, 4] = 0.0
        bg = device.BackgroundMedianFilterDeviceTemplate(context, 13)
        th = device.ThresholdSumDeviceTemplate(context, n_windows=8)
        template = device.FlaggerDeviceTemplate(bg, device.NoiseEstMADTDeviceTemplate(context, 10240),
                                                th, fused=True)  # fmt: skip
        for n_sigma in (0.0, 3.0):
            out = run_fused(template, command_queue, vis, n_sigma=n_sigma, threshold_falloff=1.5)
            ref_flags, ref_noise = oracle.flagger_full(vis, n_sigma=n_sigma, n_windows=8,
                                                       threshold_falloff=1.5)  # fmt: skip
            np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
            np.testing.assert_array_equal(ref_flags, out["flags"])

    def test_ring_kernel_chosen_by_size(self, context, command_queue):
        """Left alone (ksp_flagger_fused_ring_mode 0), a 4096-channel launch without input flags
        takes the ring kernel from 4 strips of 8 baselines per compute unit on, the 4-baseline
        kernel below; the forced modes override that either way."""
        from katsdpsigproc_amd import _lib

        props = context.device
        n_cu = getattr(props, "compute_units", None) or 256
        template = make_template(context, keep_deviations=False)
        rs = np.random.RandomState(3)
        tile = (rs.standard_normal((4096, 64)) + 1j * rs.standard_normal((4096, 64))).astype(np.complex64)
        for baselines, expect_ring in ((32 * n_cu, True), (32 * n_cu - 8, False)):
            fn = template.instantiate(command_queue, 4096, baselines, threshold_args={"n_sigma": 11.0})
            fn.ensure_all_bound()
            fn.buffer("vis").set(command_queue, np.tile(tile, (1, baselines // 64 + 1))[:, :baselines])
            previous = _lib.call("ksp_flagger_fused_ring_mode", 0)
            try:
                fn()
                assert bool(_lib.call("ksp_flagger_fused_last_path") & 4) == expect_ring
                auto = fn.buffer("flags").get(command_queue), fn.buffer("noise").get(command_queue)
                _lib.call("ksp_flagger_fused_ring_mode", -1 if expect_ring else 1)
                fn()
                assert bool(_lib.call("ksp_flagger_fused_last_path") & 4) != expect_ring
                np.testing.assert_array_equal(fn.buffer("flags").get(command_queue), auto[0])
                np.testing.assert_array_equal(fn.buffer("noise").get(command_queue), auto[1])
            finally:
                _lib.call("ksp_flagger_fused_ring_mode", previous)

    def test_unsupported_falls_back_to_sequence(self, context, command_queue):
        from katsdpsigproc_amd.rfi import device

        def six_windows(**kw):
            return device.FlaggerDeviceTemplate(
                device.BackgroundMedianFilterDeviceTemplate(context, 13),
                device.NoiseEstMADTDeviceTemplate(context, 10240),
                device.ThresholdSumDeviceTemplate(context, n_windows=6), **kw)

        fn = six_windows().instantiate(command_queue, 64, 8, threshold_args=dict(n_sigma=11.0))
        assert isinstance(fn, device.FusedFlaggerDevice)  # (round 3: up to 8 windows fuse)
        fn = six_windows().instantiate(command_queue, 8192, 8, threshold_args=dict(n_sigma=11.0))
        assert isinstance(fn, device.FlaggerDevice)  # ... but not on bands above 4096 channels
        fn = make_template(context, width=25).instantiate(
            command_queue, 64, 8, threshold_args=dict(n_sigma=11.0)
        )
        assert isinstance(fn, device.FusedFlaggerDevice)  # (round 3: widths up to 31 fuse)
        fn = make_template(context, noise="MAD").instantiate(
            command_queue, 12289, 8, threshold_args=dict(n_sigma=11.0)
        )
        assert isinstance(fn, device.FlaggerDevice)  # beyond the fused kernels' 12288 channels
        with pytest.raises(ValueError):
            six_windows(fused=True).instantiate(
                command_queue, 8192, 8, threshold_args=dict(n_sigma=11.0)
            )

    def test_config3_size_bit_exact(self, context, command_queue, oracle):
        """4096 channels x 8192 baselines (BASELINE.json config 3 shape) with RFI: every
        flag and every noise value against the oracle."""
        vis = inputs.add_rfi(inputs.generate_data(4096, 8192, seed=21), seed=22)
        out = run_fused(make_template(context, keep_deviations=False), command_queue, vis,
                        n_sigma=11.0)  # fmt: skip
        ref_flags, ref_noise = oracle.flagger_full(vis)
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        assert out["flags"].sum() > 0
        np.testing.assert_array_equal(ref_flags, out["flags"])


@pytest.fixture(scope="module")
def config4_vis():
    """BASELINE.json config 4 (= one GPU's shard of config 5): generate_data(4096, 32768)
    of the reference's script with 1/16 injected interference, 1 GiB of complex64."""
    return inputs.add_rfi_sparse(inputs.generate_data(4096, 32768), seed=3)


def all_cores(oracle):
    try:
        n = len(__import__("os").sched_getaffinity(0))
    except AttributeError:
        n = __import__("os").cpu_count() or 1
    return max(1, min(n, oracle.max_threads(), 64))


class TestFullSize:
    """The benchmarked launch itself (4096 channels x 32768 baselines, 8192 strips through
    the XCD-aware strip order) on RFI-laden data: every flag and every noise value
    against the oracle, deviations on a 4096 x 2048 slab. Mirrors reference
    test/rfi/test_flagger.py:36-132 at BASELINE.json's size."""

    SLAB = slice(30000, 32048)  # baselines whose deviations are compared

    @pytest.mark.parametrize("mode", ["none", "channel"])
    def test_config4_full_size(self, mode, config4_vis, context, command_queue, oracle):
        vis = config4_vis
        fl = inputs.channel_mask(vis.shape[0]) if mode == "channel" else None
        template = make_template(context, mode.upper(), keep_deviations=True)
        out = run_fused(template, command_queue, vis, fl, n_sigma=11.0)
        oracle.set_threads(all_cores(oracle))
        try:
            ref_flags, ref_noise = oracle.flagger_full(vis, fl)
            slab = np.ascontiguousarray(vis[:, self.SLAB])
            _, _, ref_dev = oracle.flagger_full(slab, fl, want_deviations=True)
        finally:
            oracle.set_threads(1)
        np.testing.assert_array_equal(ref_noise.astype(np.float32), out["noise"])
        flagged = int(np.count_nonzero(ref_flags))
        assert flagged > vis.size // 20  # the interference is found, not just nothing
        assert np.array_equal(ref_flags, out["flags"]), (
            f"{int(np.count_nonzero(ref_flags != out['flags']))} flags differ"
        )
        dev = out["deviations"][:, self.SLAB]
        assert np.array_equal(ref_dev.astype(np.float32), dev)
        # the north star's stated tolerance (met with 0): |dev - host| <= 1e-5
        assert np.max(np.abs(dev.astype(np.float64) - ref_dev)) <= 1e-5
        if mode == "channel":
            # masked channels: deviation 0, never flagged (reference rfi/device.py:1073-1076)
            assert not out["flags"][fl != 0].any()
            assert not out["deviations"][fl != 0].any()


class TestStaging:
    """Double-buffered host <-> device staging around the flagger (rfi/staging.py)."""

    @pytest.mark.parametrize("mode, depth", [("none", 2), ("channel", 3), ("none", 1)])
    def test_stream_of_blocks(self, mode, depth, context, oracle):
        from katsdpsigproc_amd.rfi import staging

        channels, baselines, n_blocks = 1024, 96, 7
        template = make_template(context, mode.upper(), keep_deviations=False)
        staged = staging.StagedFlagger(template, context, channels, baselines, depth=depth,
                                       threshold_args=dict(n_sigma=11.0))  # fmt: skip
        mask = inputs.channel_mask(channels) if mode == "channel" else None
        blocks = [inputs.add_rfi(inputs.generate_data(channels, baselines, seed=100 + i), seed=i)
                  for i in range(n_blocks)]  # fmt: skip
        feed = [(b, mask) for b in blocks] if mask is not None else blocks
        # (depth 1 and 2 through copy=True, the others copying the views themselves)
        results = (list(staged.run(feed, copy=True)) if depth <= 2
                   else [(f.copy(), n.copy()) for f, n in staged.run(feed)])
        assert len(results) == n_blocks
        for block, (flags, noise) in zip(blocks, results):
            ref_flags, ref_noise = oracle.flagger_full(block, mask)
            np.testing.assert_array_equal(ref_flags, flags)
            np.testing.assert_array_equal(ref_noise.astype(np.float32), noise)
        staged.finish()

    def test_zero_copy_producer_and_misuse(self, context, oracle):
        from katsdpsigproc_amd.rfi import staging

        template = make_template(context, keep_deviations=False)
        staged = staging.StagedFlagger(template, context, 512, 40, depth=2,
                                       threshold_args=dict(n_sigma=11.0))  # fmt: skip
        with pytest.raises(RuntimeError):
            staged.collect()  # nothing in flight
        vis = inputs.add_rfi(inputs.generate_data(512, 40, seed=5), seed=6)
        host_vis, host_flags = staged.host_buffers()
        assert host_flags is None
        host_vis[...] = vis  # the producer writes pinned memory directly
        assert staged.submit() == 0
        staged.submit(vis * 2)
        with pytest.raises(RuntimeError):
            staged.submit(vis)  # depth blocks already in flight
        with pytest.raises(TypeError):
            staging.StagedFlagger(template, context, 512, 40, threshold_args=dict(n_sigma=11.0)
                                  ).submit(vis, np.zeros(512, np.uint8))  # fmt: skip
        flags, _ = staged.collect()
        np.testing.assert_array_equal(oracle.flagger_full(vis)[0], flags)
        flags, _ = staged.collect()
        np.testing.assert_array_equal(oracle.flagger_full(vis * 2)[0], flags)


class TestCABI:
    """Call the C-ABI directly with ctypes: plain pointers and sizes, no accel layer."""

    def test_fused_flagger_raw(self, oracle):
        from katsdpsigproc_amd import _lib

        lib = _lib.load()
        channels, baselines = 300, 20
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=31), seed=32)
        stride = 24
        host_vis = np.zeros((channels, stride), np.complex64)
        host_vis[:, :baselines] = vis
        d_vis, d_flags, d_noise = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.ksp_malloc(0, host_vis.nbytes, ctypes.byref(d_vis)) == 0
        assert lib.ksp_malloc(0, channels * stride, ctypes.byref(d_flags)) == 0
        assert lib.ksp_malloc(0, baselines * 4, ctypes.byref(d_noise)) == 0
        try:
            assert lib.ksp_memcpy_async(0, d_vis, host_vis.ctypes.data_as(ctypes.c_void_p),
                                        host_vis.nbytes, 0, None) == 0  # fmt: skip
            scales = (ctypes.c_double * 4)(*[pow(1.2, -i) for i in range(4)])
            rc = lib.ksp_flagger_fused(
                0, None, d_vis, None, d_flags, None, d_noise, channels, baselines, stride, 0,
                stride, 0, 13, 0, 0, 1, 11.0, scales, 4, 1, None,
            )  # fmt: skip
            assert rc == 0, _lib.last_error()
            flags = np.empty((channels, stride), np.uint8)
            noise = np.empty(baselines, np.float32)
            assert lib.ksp_memcpy_async(0, flags.ctypes.data_as(ctypes.c_void_p), d_flags,
                                        flags.nbytes, 1, None) == 0  # fmt: skip
            assert lib.ksp_memcpy_async(0, noise.ctypes.data_as(ctypes.c_void_p), d_noise,
                                        noise.nbytes, 1, None) == 0  # fmt: skip
            assert lib.ksp_stream_synchronize(0, None) == 0
        finally:
            for p in (d_vis, d_flags, d_noise):
                lib.ksp_free(0, p)
        ref_flags, ref_noise = oracle.flagger_full(vis)
        np.testing.assert_array_equal(ref_flags, flags[:, :baselines])
        np.testing.assert_array_equal(ref_noise.astype(np.float32), noise)

    def test_error_reporting(self):
        from katsdpsigproc_amd import _lib

        lib = _lib.load()
        rc = lib.ksp_transpose(0, None, None, None, 4, 4, 4, 4, 4)
        assert rc != 0
        assert "NULL" in _lib.last_error()
        with pytest.raises(RuntimeError):
            _lib.call("ksp_background_median_filter", 0, None, ctypes.c_void_p(8),
                      ctypes.c_void_p(8), None, 4, 4, 4, 0, 4, 0, 0, 0)  # fmt: skip
