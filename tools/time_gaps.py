import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from katsdpsigproc_amd import accel
from katsdpsigproc_amd.rfi import device
ctx = accel.create_some_context(False); q = ctx.create_command_queue()
t = device.FlaggerDeviceTemplate(device.BackgroundMedianFilterDeviceTemplate(ctx, 13),
    device.NoiseEstMADTDeviceTemplate(ctx, 16384), device.ThresholdSumDeviceTemplate(ctx), fused=True)
fn = t.instantiate(q, 4096, 32768, threshold_args={"n_sigma": 11.0}); fn.ensure_all_bound()
rs = np.random.RandomState(1)
block = (rs.standard_normal((4096, 4096)).astype(np.float32) + 1j * rs.standard_normal((4096, 4096)).astype(np.float32)).astype(np.complex64)
fn.buffer("vis").set(q, np.tile(block, (1, 8)))
def run(n, markers, events):
    for _ in range(100): fn()
    q.finish()
    a = q.enqueue_marker(); t0 = time.perf_counter()
    for _ in range(n):
        if events: fn.profile_next_run()
        fn()
        if markers: q.enqueue_marker()
    b = q.enqueue_marker(); q.finish()
    return 1e3 * b.time_since(a) / n, 1e3 * (time.perf_counter() - t0) / n
for rep in range(2):
    for markers, events in ((False, False), (True, False), (False, True), (True, True)):
        d, w = run(300, markers, events)
        print(f"markers={markers!s:5} kernel-events={events!s:5}: device {d:.4f} ms/step, wall {w:.4f} ms/step", flush=True)
