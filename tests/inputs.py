"""Seeded input generators shared by the golden-vector script and the tests.

Each generator reproduces the set-up of one of the reference's own tests (same seed,
same order of RandomState calls), so that the inputs need not be stored:
``np.random.RandomState`` legacy streams are frozen by NumPy's compatibility policy.
"""

import numpy as np

BACKGROUND_COLS = [0, 1, 17, 99, 100, 312]
CFG1_COLS = [0, 1, 2, 3, 511, 1024, 2047]


def complex_normal(rs, size):
    """Circularly symmetric Gaussian, as reference test/__init__.py:31-42."""
    return rs.normal(0.0, 1.0, size) + 1j * rs.normal(0.0, 1.0, size)


def abs_probe():
    """complex64 probe values for numpy's abs: wide dynamic range plus specials."""
    rs = np.random.RandomState(5)
    n = 4096
    re = rs.standard_normal(n) * np.exp(rs.uniform(-20, 20, n))
    im = rs.standard_normal(n) * np.exp(rs.uniform(-20, 20, n))
    z = (re + 1j * im).astype(np.complex64)
    specials = np.array(
        [
            0, 1, 1j, -1, -1j, complex(-0.0, -0.0), complex(3, 4), complex(1e-45, 1e-45),
            complex(1e-45, 0), complex(3e38, 3e38), complex(3e38, 1e38), complex(1e-30, 1e-38),
            complex(np.inf, 1), complex(1, -np.inf), complex(np.inf, np.inf),
            complex(1e-20, 1e20), complex(1.1754944e-38, 1.1754944e-38),
        ],
        dtype=np.complex64,
    )  # fmt: skip
    return np.concatenate([z, specials])


def background_case():
    """reference test/rfi/test_background.py:33-45 (417x313, 10 % flags, block of 4s)."""
    shape = (417, 313)
    rs = np.random.RandomState(seed=1)
    vis_big = complex_normal(rs, size=shape).astype(np.complex64)
    flags_big = (rs.random_sample(shape) < 0.1).astype(np.uint8)
    flags_big[100:110, 0:100] = 4
    return vis_big, flags_big


def noise_case():
    """reference test/rfi/test_noise_est.py:39-43 (117x273 standard normal float32)."""
    rs = np.random.RandomState(seed=1)
    return rs.standard_normal((117, 273)).astype(np.float32)


def threshold_case():
    """reference test/rfi/test_threshold.py:32-41 (117x273, 25 % spikes of +200)."""
    shape = (117, 273)
    rs = np.random.RandomState(seed=1)
    spikes = rs.random_sample(shape) < 0.25
    deviations = rs.standard_normal(shape).astype(np.float32) * 10.0
    deviations[spikes] += 200.0
    return deviations, spikes


def flagger_case():
    """reference test/rfi/test_flagger.py:36-52 (117x131, 1/16 RFI, 1/16 input flags = 2)."""
    shape = (117, 131)
    rs = np.random.RandomState(seed=1)
    vis = complex_normal(rs, size=shape)
    spikes = rs.random_sample(shape) < 1.0 / 16.0
    spikes = spikes.astype(np.uint8)
    rfi_amp = rs.random_sample(shape) * 20.0 + 50.0
    rfi_phase = rs.random_sample(shape) * (2j * np.pi)
    rfi = rfi_amp * np.exp(rfi_phase)
    vis += spikes * rfi
    vis = vis.astype(np.complex64)
    input_flags = (rs.random_sample(shape) < 1.0 / 16.0).astype(np.uint8) * 2
    return vis, spikes, input_flags


def generate_data(channels, baselines, seed=1):
    """reference scripts/rfiflagtest.py:35-44."""
    rs = np.random.RandomState(seed=seed)
    out = np.empty((channels, baselines), np.complex64)
    for i in range(channels):
        real = rs.standard_normal(size=baselines).astype(np.float32)
        imag = rs.standard_normal(size=baselines).astype(np.float32)
        out[i] = real + 1j * imag
    return out


def add_rfi(vis, seed=3, fraction=1.0 / 16.0):
    """Spikes as in test/rfi/test_flagger.py:42-50, row by row to bound memory."""
    rs = np.random.RandomState(seed=seed)
    out = np.array(vis, dtype=np.complex64, copy=True)
    for i in range(vis.shape[0]):
        s = rs.random_sample(vis.shape[1]) < fraction
        amp = rs.random_sample(vis.shape[1]) * 20.0 + 50.0
        phase = rs.random_sample(vis.shape[1]) * (2j * np.pi)
        out[i] = (out[i].astype(np.complex128) + s * (amp * np.exp(phase))).astype(np.complex64)
    return out


def config1():
    """BASELINE.json config 1: generate_data(1024, 2048)."""
    return generate_data(1024, 2048)


def config1_rfi():
    """Config 1 with injected RFI so that 'flags bit-identical' is not vacuous."""
    return add_rfi(config1())


def channel_mask(channels, seed=2, fraction=1.0 / 16.0):
    """Per-channel input-flag mask of SURVEY 8(d) / config 5."""
    return (np.random.RandomState(seed).random_sample(channels) < fraction).astype(np.uint8)


def add_rfi_sparse(vis, seed=3, fraction=1.0 / 16.0, block=256):
    """Same kind of interference as :func:`add_rfi` (amplitude U(50, 70), random phase on
    a random `fraction` of the samples) for arrays of 10^8 samples: amplitudes and
    phases are drawn for the hit samples only, in blocks of rows, in place of three
    full-size float64 draws. Modifies and returns `vis` (complex64)."""
    rs = np.random.RandomState(seed=seed)
    for r0 in range(0, vis.shape[0], block):
        part = vis[r0 : r0 + block]
        hit = rs.random_sample(part.shape) < fraction
        n = int(np.count_nonzero(hit))
        amp = rs.random_sample(n) * 20.0 + 50.0
        phase = rs.random_sample(n) * (2.0 * np.pi)
        part[hit] += (amp * np.exp(1j * phase)).astype(np.complex64)
    return vis


def threshold_wide_case(seed=11):
    """Deviations for SumThreshold with 6 and 8 windows: 273 channels x 117 baselines of
    unit noise (float32) with runs of 3..40 channels raised by 2.5..6 (broad, weak
    interference that only the wide windows can find) and a few strong spikes; noise
    estimates near 1. Returns (deviations [C][B] float32, noise [B] float32)."""
    rs = np.random.RandomState(seed)
    channels, baselines = 273, 117
    dev = rs.standard_normal((channels, baselines)).astype(np.float32)
    for b in range(baselines):
        for _ in range(3):
            start = rs.randint(0, channels - 40)
            length = rs.randint(3, 41)
            dev[start : start + length, b] += np.float32(2.5 + 3.5 * rs.random_sample())
        dev[rs.randint(0, channels, 2), b] += 40.0
    noise = (0.9 + 0.2 * rs.random_sample(baselines)).astype(np.float32)
    return dev, noise


def denormal_case(channels=4096):
    """Amplitudes that are small multiples of 2^-149 (float32 subnormals), 8 baselines: a
    constant level of 20 units with two isolated spikes, and at one band edge a pattern under
    which an even-count window yields a deviation of exactly +-2^-150 -- not zero, but zero
    once rounded to float32 -- whose being counted among the non-zero deviations changes the
    MAD (1 or 2 units of noise with it, 2 or 3 without: patterns found by search). Baselines
    0-3 carry the pattern at the lower edge, 4-7 mirrored at the upper one.
    Returns float32 [channels][8]."""
    patterns = [[21, 18, 19, 19, 19, 18, 20, 22, 21, 21], [22, 22, 21, 22, 22, 22, 18, 22, 21, 20],
                [20, 20, 21, 20, 21, 22, 22, 20, 19, 20], [19, 19, 18, 18, 20, 22, 21, 21, 20, 19]]
    units = np.full((channels, 8), 20.0)
    for b in range(8):
        edge = np.array(patterns[b % 4], dtype=np.float64)
        if b < 4:
            units[:10, b] = edge
            units[100, b], units[200, b] = 29, 23
        else:
            units[-10:, b] = edge[::-1]
            units[channels - 101, b], units[channels - 201, b] = 29, 23
    return (units * 2.0 ** -149).astype(np.float32)
