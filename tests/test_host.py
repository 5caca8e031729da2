"""The package's NumPy host classes (katsdpsigproc_amd.rfi.host) against golden vectors
from the reference's host classes and against the reference's own known answers."""

import numpy as np
import pytest

from katsdpsigproc_amd.rfi import host
from tests import inputs


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.uint8)


def test_known_answers():
    vis = np.array([[1.25, 1.5j, 1.0, 2.0, -1.75, 2.0]]).T.astype(np.complex64)
    flags = np.array([0, 0, 1, 0, 0, 4]).astype(np.uint8)
    bg = host.BackgroundMedianFilterHost(3)
    np.testing.assert_array_equal(bg(vis), np.array([[-0.125, 0.25, -0.5, 0.25, -0.25, 0.125]]).T)
    np.testing.assert_array_equal(
        bg(vis, flags), np.array([[-0.125, 0.125, 0.0, 0.125, -0.125, 0.0]]).T
    )
    dev = np.array([[0.0, 3.0, 2.4], [1.5, -1.4, 4.6], [0.0, 1.1, 3.3], [5.0, 0.0, -3.1]],
                   np.float32)  # fmt: skip
    np.testing.assert_allclose(host.NoiseEstMADHost()(dev), np.array([3.25, 1.4, 3.2]) * 1.4826)


@pytest.mark.parametrize("amplitudes", [False, True])
@pytest.mark.parametrize("mode", ["none", "channel", "full"])
def test_background_golden(golden, amplitudes, mode):
    vis_big, flags_big = inputs.background_case()
    vis = np.abs(vis_big) if amplitudes else vis_big
    fl = {"none": None, "channel": flags_big[:, 0], "full": flags_big}[mode]
    dev = host.BackgroundMedianFilterHost(5, amplitudes)(vis, fl)
    assert dev.dtype == np.float64
    key = f"background_{'amp' if amplitudes else 'cplx'}_{mode}_cols"
    np.testing.assert_array_equal(dev[:, inputs.BACKGROUND_COLS], golden[key])


def test_noise_golden(golden):
    dev32 = inputs.noise_case()
    np.testing.assert_array_equal(host.NoiseEstMADHost()(dev32), golden["noise_f32in"])
    np.testing.assert_array_equal(
        host.NoiseEstMADHost()(dev32.astype(np.float64) * 1.000000123), golden["noise_f64in"]
    )
    with pytest.warns(RuntimeWarning):
        out = host.NoiseEstMADHost()(np.zeros((4, 2), np.float32))
    assert np.all(np.isnan(out))


@pytest.mark.parametrize("name", ["simple", "sum"])
def test_threshold_golden(golden, name):
    dev, spikes = inputs.threshold_case()
    cls = {"simple": host.ThresholdSimpleHost, "sum": host.ThresholdSumHost}[name]
    noise32 = np.linspace(0.0, 50.0, dev.shape[1]).astype(np.float32)
    noise64 = np.linspace(0.0, 50.0, dev.shape[1]) * 1.0000003
    np.testing.assert_array_equal(
        cls(11.0)(dev, noise32), unpack(golden[f"threshold_{name}_f32"], dev.shape)
    )
    np.testing.assert_array_equal(
        cls(11.0)(dev.astype(np.float64) * 1.0000001, noise64),
        unpack(golden[f"threshold_{name}_f64"], dev.shape),
    )
    noise = np.repeat(10.0, dev.shape[1]).astype(np.float32)
    np.testing.assert_array_equal(cls(11.0)(dev, noise).astype(np.bool_), spikes)


def test_threshold_sum_params_golden(golden):
    dev, _ = inputs.threshold_case()
    noise32 = np.linspace(0.0, 50.0, dev.shape[1]).astype(np.float32)
    fl = host.ThresholdSumHost(7.5, n_windows=5, threshold_falloff=1.35, flag_value=4)(dev, noise32)
    assert set(np.unique(fl)) <= {0, 4}
    np.testing.assert_array_equal(
        (fl != 0).astype(np.uint8), unpack(golden["threshold_sum_f32_params"], dev.shape)
    )


@pytest.mark.parametrize("name", ["simple", "sum"])
@pytest.mark.parametrize("mode", ["none", "channel", "full"])
def test_flagger_golden(golden, name, mode):
    vis, _, in_flags = inputs.flagger_case()
    fl = {"none": None, "channel": in_flags[:, 0], "full": in_flags}[mode]
    th = {"simple": host.ThresholdSimpleHost, "sum": host.ThresholdSumHost}[name](11.0)
    flagger = host.FlaggerHost(host.BackgroundMedianFilterHost(13), host.NoiseEstMADHost(), th)
    np.testing.assert_array_equal(
        flagger(vis, fl), unpack(golden[f"flagger_{name}_{mode}"], vis.shape)
    )
