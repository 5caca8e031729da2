"""Pin the CPU oracle (oracle/rfi_oracle.c) against the reference.

Golden vectors come from the real ``katsdpsigproc.rfi.host`` (tests/golden/make_golden.py);
the known-answer vectors are those of the reference's own tests.
"""

import hashlib

import numpy as np
import pytest

from oracle import rfi_oracle as oracle
from tests import inputs


def digest(a):
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a + 0.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.uint8)


def test_abs_probe(golden):
    """numpy's complex64 abs == mx*sqrt(fma(r,r,1)) ('Amplitude')."""
    out = oracle.abs_c64(inputs.abs_probe())
    np.testing.assert_array_equal(out, golden["abs_probe_out"])


class TestKnownAnswers:
    """Known-answer vectors held by the reference's own tests."""

    vis = np.array([[1.25, 1.5j, 1.0, 2.0, -1.75, 2.0]]).T.astype(np.complex64)
    flags = np.array([0, 0, 1, 0, 0, 4]).T.astype(np.uint8)

    def test_background(self):
        # reference test/rfi/test_background.py:52-55
        out = oracle.BackgroundMedianFilterHost(3)(self.vis)
        ref = np.array([[-0.125, 0.25, -0.5, 0.25, -0.25, 0.125]]).T
        np.testing.assert_array_equal(ref, out)

    def test_background_flags(self):
        # reference test/rfi/test_background.py:57-60
        out = oracle.BackgroundMedianFilterHost(3)(self.vis, self.flags)
        ref = np.array([[-0.125, 0.125, 0.0, 0.125, -0.125, 0.0]]).T
        np.testing.assert_array_equal(ref, out)

    def test_noise(self):
        # reference test/rfi/test_noise_est.py:35-50
        dev = np.array(
            [[0.0, 3.0, 2.4], [1.5, -1.4, 4.6], [0.0, 1.1, 3.3], [5.0, 0.0, -3.1]]
        ).astype(np.float32)
        expected = np.array([3.25, 1.4, 3.2]) * 1.4826
        np.testing.assert_allclose(expected, oracle.NoiseEstMADHost()(dev))

    @pytest.mark.parametrize("cls", ["ThresholdSimpleHost", "ThresholdSumHost"])
    def test_threshold_spikes(self, cls):
        # reference test/rfi/test_threshold.py:44-57
        dev, spikes = inputs.threshold_case()
        noise = np.repeat(10.0, dev.shape[1]).astype(np.float32)
        flags = getattr(oracle, cls)(11.0)(dev, noise)
        np.testing.assert_array_equal(flags.astype(np.bool_), spikes)

    def test_flagger_spikes(self):
        # reference test/rfi/test_flagger.py:55-71
        vis, spikes, input_flags = inputs.flagger_case()
        flagger = oracle.FlaggerHost(
            oracle.BackgroundMedianFilterHost(13),
            oracle.NoiseEstMADHost(),
            oracle.ThresholdSimpleHost(11.0),
        )
        np.testing.assert_array_equal(spikes, flagger(vis))
        flags = flagger(vis, input_flags[:, 0])
        bcast = np.broadcast_to(input_flags[:, 0:1], vis.shape)
        np.testing.assert_array_equal(np.where(bcast, 0, spikes), flags)
        flags = flagger(vis, input_flags)
        np.testing.assert_array_equal(np.where(input_flags, 0, spikes), flags)


@pytest.mark.parametrize("amplitudes", [False, True])
@pytest.mark.parametrize("mode", ["none", "channel", "full"])
def test_background_golden(golden, amplitudes, mode):
    vis_big, flags_big = inputs.background_case()
    vis = oracle.abs_c64(vis_big) if amplitudes else vis_big
    fl = {"none": None, "channel": flags_big[:, 0], "full": flags_big}[mode]
    dev = oracle.BackgroundMedianFilterHost(5, amplitudes)(vis, fl)
    key = f"background_{'amp' if amplitudes else 'cplx'}_{mode}"
    np.testing.assert_array_equal(dev[:, inputs.BACKGROUND_COLS], golden[key + "_cols"])
    assert digest(dev) == str(golden[key + "_sha"])


def test_noise_golden(golden):
    dev32 = inputs.noise_case()
    np.testing.assert_array_equal(oracle.NoiseEstMADHost()(dev32), golden["noise_f32in"])
    dev64 = dev32.astype(np.float64) * 1.000000123
    np.testing.assert_array_equal(oracle.NoiseEstMADHost()(dev64), golden["noise_f64in"])


@pytest.mark.parametrize("name", ["simple", "sum"])
def test_threshold_golden(golden, name):
    dev, _ = inputs.threshold_case()
    cls = {"simple": oracle.ThresholdSimpleHost, "sum": oracle.ThresholdSumHost}[name]
    noise32 = np.linspace(0.0, 50.0, dev.shape[1]).astype(np.float32)
    noise64 = np.linspace(0.0, 50.0, dev.shape[1]) * 1.0000003
    dev64 = dev.astype(np.float64) * 1.0000001
    np.testing.assert_array_equal(
        cls(11.0)(dev, noise32), unpack(golden[f"threshold_{name}_f32"], dev.shape)
    )
    np.testing.assert_array_equal(
        cls(11.0)(dev64, noise64), unpack(golden[f"threshold_{name}_f64"], dev.shape)
    )


def test_threshold_sum_params_golden(golden):
    dev, _ = inputs.threshold_case()
    noise32 = np.linspace(0.0, 50.0, dev.shape[1]).astype(np.float32)
    th = oracle.ThresholdSumHost(7.5, n_windows=5, threshold_falloff=1.35, flag_value=4)
    fl = th(dev, noise32)
    assert set(np.unique(fl)) <= {0, 4}
    np.testing.assert_array_equal(
        (fl != 0).astype(np.uint8), unpack(golden["threshold_sum_f32_params"], dev.shape)
    )


def test_denormal_deviations_golden(golden):
    """Deviations of exactly +-2^-150 (float32 zeros) count as non-zero in the MAD, as in the
    imported reference; the case is built so that this changes the noise estimates."""
    amp = inputs.denormal_case()
    flags, noise, dev = oracle.flagger_full(amp, amplitudes=True, want_deviations=True)
    tiny = np.abs(dev) == 2.0 ** -150
    assert tiny.any(axis=0).all()
    np.testing.assert_array_equal(noise, golden["denormal_noise"])
    np.testing.assert_array_equal(flags, unpack(golden["denormal_flags"], amp.shape))
    # teeth: with those deviations taken for zeros the float32 noise estimates differ
    naive = np.array([np.median(d[(d > 0) & (d != 2.0 ** -150)]) * 1.4826 for d in np.abs(dev).T])
    assert np.all(naive.astype(np.float32) != noise.astype(np.float32))
    flags_c, noise_c = oracle.flagger_full(amp.astype(np.complex64))
    np.testing.assert_array_equal(noise_c, noise)


@pytest.mark.parametrize("n_windows", [6, 8])
@pytest.mark.parametrize("kind", ["f32", "f64"])
def test_threshold_sum_wide_windows_golden(golden, n_windows, kind):
    """6 and 8 windows (window sums of 32 and 128 terms) against the imported reference:
    the oracle adds left to right, numpy.convolve in an order of its own -- on these inputs
    every flag agrees."""
    dev, noise = inputs.threshold_wide_case()
    if kind == "f64":
        dev, noise = dev.astype(np.float64) * 1.0000001, noise.astype(np.float64) * 1.0000003
    fl = oracle.ThresholdSumHost(6.0, n_windows=n_windows)(dev, noise)
    expected = unpack(golden[f"threshold_sum_{kind}_w{n_windows}"], dev.shape)
    assert 0.02 < expected.mean() < 0.6  # interference found, not everything flagged
    np.testing.assert_array_equal(fl, expected)


@pytest.mark.parametrize("name", ["simple", "sum"])
@pytest.mark.parametrize("mode", ["none", "channel", "full"])
def test_flagger_golden(golden, name, mode):
    vis, _, in_flags = inputs.flagger_case()
    fl = {"none": None, "channel": in_flags[:, 0], "full": in_flags}[mode]
    flags, noise = oracle.flagger_full(vis, fl, width=13, threshold=name, n_sigma=11.0)
    np.testing.assert_array_equal(flags, unpack(golden[f"flagger_{name}_{mode}"], vis.shape))
    # the chained classes give the same answer as the one-call C path
    th = {"simple": oracle.ThresholdSimpleHost, "sum": oracle.ThresholdSumHost}[name](11.0)
    chained = oracle.FlaggerHost(
        oracle.BackgroundMedianFilterHost(13), oracle.NoiseEstMADHost(), th
    )
    np.testing.assert_array_equal(flags, chained(vis, fl))


def test_flagger_intermediates_golden(golden):
    vis, _, in_flags = inputs.flagger_case()
    _, noise, dev = oracle.flagger_full(vis, in_flags, width=13, want_deviations=True)
    np.testing.assert_array_equal(dev, golden["flagger_dev_full"])
    np.testing.assert_array_equal(noise, golden["flagger_noise_full"])


@pytest.mark.parametrize("tag", ["cfg1", "cfg1rfi"])
def test_config1_golden(golden, tag):
    """BASELINE.json config 1 (1024 x 2048), plain and RFI-injected."""
    vis = inputs.config1() if tag == "cfg1" else inputs.config1_rfi()
    flags, noise, dev = oracle.flagger_full(vis, width=13, n_sigma=11.0, want_deviations=True)
    np.testing.assert_array_equal(noise, golden[f"{tag}_noise"])
    np.testing.assert_array_equal(dev[:, inputs.CFG1_COLS], golden[f"{tag}_dev_cols"])
    assert digest(dev) == str(golden[f"{tag}_dev_sha"])
    assert int(flags.astype(np.int64).sum()) == int(golden[f"{tag}_flags_count"])
    assert digest(flags) == str(golden[f"{tag}_flags_sha"])


def test_percentile5_matches_numpy():
    """Oracle percentile5 == the expectation the reference's test states with numpy."""
    rs = np.random.RandomState(1)
    ary = np.abs(rs.randn(37, 501)).astype(np.float32)
    for rng in (None, (3, 400), (500, 501)):
        lo, hi = rng if rng else (0, ary.shape[1])
        expected = np.percentile(ary[:, lo:hi], [0, 100, 25, 75, 50], axis=1, method="lower")
        np.testing.assert_array_equal(expected.astype(np.float32), oracle.percentile5(ary, rng))
