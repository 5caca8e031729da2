import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katsdpsigproc_amd import accel
from katsdpsigproc_amd.rfi import device
from oracle import rfi_oracle as oracle
from tests import inputs
ctx = accel.create_some_context(False); q = ctx.create_command_queue()
vis, spikes, in_flags = inputs.flagger_case()
for th in ("simple", "sum"):
    t = device.FlaggerDeviceTemplate(device.BackgroundMedianFilterDeviceTemplate(ctx, 13), device.NoiseEstMADTDeviceTemplate(ctx, 10240),
        device.ThresholdSimpleDeviceTemplate(ctx, False) if th == "simple" else device.ThresholdSumDeviceTemplate(ctx),
        keep_deviations=True)
    fn = t.instantiate(q, *vis.shape, threshold_args={"n_sigma": 11.0}); fn.ensure_all_bound()
    fn.buffer("vis").set(q, vis); fn()
    flags = fn.buffer("flags").get(q); noise = fn.buffer("noise").get(q); dev = fn.buffer("deviations").get(q)
    rf, rn, rd = oracle.flagger_full(vis, threshold=th, want_deviations=True)
    print(th, "dev mismatch", np.sum(dev != rd.astype(np.float32)), "noise mismatch", np.sum(noise != rn.astype(np.float32)))
    bad = np.argwhere(flags != rf)
    print(" flag mismatches", len(bad), "cols", np.unique(bad[:, 1])[:40], "rows", np.unique(bad[:,0])[:20])
    print(" device-only", int(np.sum((flags != 0) & (rf == 0))), "oracle-only", int(np.sum((flags == 0) & (rf != 0))))
    bn = np.flatnonzero(noise != rn.astype(np.float32))[:10]
    print(" noise idx", bn, noise[bn], rn[bn])
