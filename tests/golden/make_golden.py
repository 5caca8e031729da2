#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the REAL reference.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference/src python3 tests/golden/make_golden.py

It imports ``katsdpsigproc.rfi.host`` (reference: src/katsdpsigproc/rfi/host.py) and
runs it on seeded inputs that reproduce the reference's own test set-ups
(test/rfi/test_background.py:33-45, test/rfi/test_noise_est.py:35-43,
test/rfi/test_threshold.py:32-41,61-70, test/rfi/test_flagger.py:36-52) and
BASELINE.json config 1 (scripts/rfiflagtest.py:35-44). Only inputs that cannot be
regenerated from a seed, and outputs, are stored. Large outputs are stored as a
sha256 of their bytes plus a small slice (the oracle must match bit for bit, so a
digest is a complete check; the slice is there to make failures debuggable).

Versions used for the committed fixtures: numpy 2.2.6, pandas 2.3.3, Python 3.10.12.
"""

import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from katsdpsigproc.rfi import host  # noqa: E402  (the reference)

from tests import inputs  # noqa: E402  (seeded input generators shared with the tests)


def digest(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a + 0.0  # normalise -0.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def main() -> None:
    out = {}

    # (iv) numpy complex64 abs probe: wide dynamic range + special values
    probe = inputs.abs_probe()
    with np.errstate(all="ignore"):
        out["abs_probe_out"] = np.abs(probe)

    # (ii-a) background 417x313, width 5, {complex, amplitude} x {none, channel, full}
    vis_big, flags_big = inputs.background_case()
    cols = inputs.BACKGROUND_COLS
    for amplitudes in (False, True):
        vis = np.abs(vis_big) if amplitudes else vis_big
        bg = host.BackgroundMedianFilterHost(5, amplitudes)
        for mode in ("none", "channel", "full"):
            if mode == "none":
                dev = bg(vis)
            elif mode == "channel":
                dev = bg(vis, flags_big[:, 0])
            else:
                dev = bg(vis, flags_big)
            key = f"background_{'amp' if amplitudes else 'cplx'}_{mode}"
            assert dev.dtype == np.float64
            out[key + "_sha"] = np.array(digest(dev))
            out[key + "_cols"] = dev[:, cols]

    # (ii-b) noise 117x273, float32 input (as the reference test) and float64 input
    dev32 = inputs.noise_case()
    out["noise_f32in"] = host.NoiseEstMADHost()(dev32)
    out["noise_f64in"] = host.NoiseEstMADHost()(dev32.astype(np.float64) * 1.000000123)

    # (ii-c) thresholds 117x273: float32 deviations with float32 noise (reference test)
    # and float64 deviations with float64 noise (what FlaggerHost feeds them)
    dev_t, _spikes = inputs.threshold_case()
    noise32 = np.linspace(0.0, 50.0, dev_t.shape[1]).astype(np.float32)
    noise64 = np.linspace(0.0, 50.0, dev_t.shape[1]) * 1.0000003
    dev64 = dev_t.astype(np.float64) * 1.0000001
    for name, cls in (("simple", host.ThresholdSimpleHost), ("sum", host.ThresholdSumHost)):
        out[f"threshold_{name}_f32"] = np.packbits(cls(11.0)(dev_t, noise32).astype(np.bool_))
        out[f"threshold_{name}_f64"] = np.packbits(cls(11.0)(dev64, noise64).astype(np.bool_))
    # non-default parameters for the sum threshold
    th = host.ThresholdSumHost(7.5, n_windows=5, threshold_falloff=1.35, flag_value=4)
    fl = th(dev_t, noise32)
    assert set(np.unique(fl)) <= {0, 4}
    out["threshold_sum_f32_params"] = np.packbits(fl.astype(np.bool_))
    # 6 and 8 windows (sums of 32 and 128 terms, which numpy.convolve does not promise to
    # add left to right): float32 and float64 deviations with broad interference added, so
    # that the wide windows decide flags the narrow ones do not
    wide32, wide_noise32 = inputs.threshold_wide_case()
    wide64 = wide32.astype(np.float64) * 1.0000001
    wide_noise64 = wide_noise32.astype(np.float64) * 1.0000003
    for n_windows in (6, 8):
        th = host.ThresholdSumHost(6.0, n_windows=n_windows)
        out[f"threshold_sum_f32_w{n_windows}"] = np.packbits(th(wide32, wide_noise32).astype(np.bool_))
        out[f"threshold_sum_f64_w{n_windows}"] = np.packbits(th(wide64, wide_noise64).astype(np.bool_))

    # subnormal amplitudes with deviations of exactly +-2^-150 (zero as float32, counted by
    # host.py:161): noise estimates, as amplitudes and as complex visibilities
    den = inputs.denormal_case()
    bg_den = host.BackgroundMedianFilterHost(13, True)
    dev_den = bg_den(den)
    assert np.count_nonzero(np.abs(dev_den) == 2.0 ** -150) >= 8
    out["denormal_noise"] = host.NoiseEstMADHost()(dev_den)
    dev_cplx = host.BackgroundMedianFilterHost(13)(den.astype(np.complex64))
    assert np.array_equal(dev_cplx, dev_den)
    out["denormal_flags"] = np.packbits(
        host.ThresholdSumHost(11.0)(dev_den, out["denormal_noise"]).astype(np.bool_))

    # (ii-d) flagger 117x131 with injected RFI, three flag modes, Simple and Sum
    vis_f, _sp, in_flags = inputs.flagger_case()
    for name, th in (
        ("simple", host.ThresholdSimpleHost(11.0)),
        ("sum", host.ThresholdSumHost(11.0)),
    ):
        flagger = host.FlaggerHost(
            host.BackgroundMedianFilterHost(13), host.NoiseEstMADHost(), th
        )
        out[f"flagger_{name}_none"] = np.packbits(flagger(vis_f).astype(np.bool_))
        out[f"flagger_{name}_channel"] = np.packbits(
            flagger(vis_f, in_flags[:, 0]).astype(np.bool_)
        )
        out[f"flagger_{name}_full"] = np.packbits(flagger(vis_f, in_flags).astype(np.bool_))
    bg13 = host.BackgroundMedianFilterHost(13)
    dev_f = bg13(vis_f, in_flags)
    out["flagger_dev_full"] = dev_f
    out["flagger_noise_full"] = host.NoiseEstMADHost()(dev_f)

    # (iii) BASELINE config 1: 1024 x 2048, width 13, 11 sigma, 4 windows; plain and with RFI
    for tag, vis in (("cfg1", inputs.config1()), ("cfg1rfi", inputs.config1_rfi())):
        bg = host.BackgroundMedianFilterHost(13)
        dev = bg(vis)
        noise = host.NoiseEstMADHost()(dev)
        flags = host.ThresholdSumHost(11.0)(dev, noise)
        out[f"{tag}_noise"] = noise
        out[f"{tag}_dev_sha"] = np.array(digest(dev))
        out[f"{tag}_dev_cols"] = dev[:, inputs.CFG1_COLS]
        out[f"{tag}_flags_sha"] = np.array(digest(flags))
        out[f"{tag}_flags_count"] = np.array(int(flags.astype(np.int64).sum()))
        out[f"{tag}_flags_cols"] = np.packbits(flags[:, inputs.CFG1_COLS].astype(np.bool_))
        print(tag, "flagged", int(flags.sum()), "of", flags.size)

    path = os.path.join(HERE, "rfi_host_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
