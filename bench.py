#!/usr/bin/env python3
"""Benchmark of the RFI-flagging hot path on MI355X.

One "step" = one pass of the full flagger (median-filter background, MAD noise
estimate, SumThreshold) over one block of synthetic visibilities that is already
resident in HBM: 4096 channels x 32768 baselines of complex64 per GPU
(BASELINE.json config 4; with N GPUs the baselines are sharded, N x 32768 in total,
config 5, and every step broadcasts the per-channel flag mask over RCCL). Prints ONE
JSON line (see the task contract): whole-job samples/s, the HBM roofline of the step
(zero-fill of the flags + the fused kernel) from HIP-event timing, the same workload
with injected interference (``rfi_variant``), a check of the timed launch's output
against the CPU oracle, the bring-up shapes of BASELINE.json configs 2 and 3, and a CPU
baseline (the oracle's C restatement of rfi.host, timed on this box's cores).

    python bench.py                      # 1 GPU
    python bench.py --gpus N             # starts N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   # the same

Parameters follow scripts/rfiflagtest.py of the reference: width 13, 11 sigma,
4 windows, falloff 1.2, RandomState(seed=1) standard-normal real/imag.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHANNELS = 4096
BASELINES_PER_GPU = 32768
WIDTH = 13
N_SIGMA = 11.0
PREHEAT_STEPS = 100  # x 0.45 ms: the clock ramp takes about 40 launches
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGORITHMIC_BYTES_PER_SAMPLE = 9  # 8 B complex64 read + 1 B flag written
CHECK_BASELINES = 512  # slice of the timed launch's output compared with the oracle


def synth_block(channels: int, baselines: int, seed: int) -> np.ndarray:
    """generate_data of the reference (scripts/rfiflagtest.py:35-44), any seed."""
    rs = np.random.RandomState(seed=seed)
    out = np.empty((channels, baselines), np.complex64)
    for i in range(channels):
        real = rs.standard_normal(size=baselines).astype(np.float32)
        imag = rs.standard_normal(size=baselines).astype(np.float32)
        out[i].real = real
        out[i].imag = imag
    return out


def inject_rfi(vis: np.ndarray, seed: int = 3, fraction: float = 1.0 / 16.0, block: int = 256):
    """Interference as reference test/rfi/test_flagger.py:42-50 adds it (a random
    `fraction` of the samples, amplitude U(50, 70), random phase), drawn for the hit
    samples only and in blocks of rows so that 10^8 samples take seconds. In place."""
    rs = np.random.RandomState(seed=seed)
    for r0 in range(0, vis.shape[0], block):
        part = vis[r0 : r0 + block]
        hit = rs.random_sample(part.shape) < fraction
        n = int(np.count_nonzero(hit))
        amp = rs.random_sample(n) * 20.0 + 50.0
        phase = rs.random_sample(n) * (2.0 * np.pi)
        part[hit] += (amp * np.exp(1j * phase)).astype(np.complex64)
    return vis


def host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def numpy_baseline():
    """The NumPy restatement of the reference's host path (katsdpsigproc_amd.rfi.host, checked
    against the golden vectors of the imported reference) on BASELINE.json config 1, the shape
    scripts/rfiflagtest.py times on the host: 1024 channels x 2048 baselines, one warm-up call,
    one timed call, one core."""
    from katsdpsigproc_amd.rfi import host

    channels, baselines = 1024, 2048
    vis = synth_block(channels, baselines, 1)
    flagger = host.FlaggerHost(host.BackgroundMedianFilterHost(WIDTH), host.NoiseEstMADHost(),
                               host.ThresholdSumHost(N_SIGMA))  # fmt: skip
    flagger(vis[:, :64])  # warm-up (imports, allocator)
    t0 = time.perf_counter()
    flagger(vis)
    dt = time.perf_counter() - t0
    return {
        "value": channels * baselines / dt,
        "unit": "samples/s",
        "cores": 1,
        "kind": "numpy",
        "sample": f"{channels} ch x {baselines} bl complex64 (BASELINE config 1), one call, {dt:.2f} s",
    }


def cpu_baseline(budget_s: float = 20.0):
    """Time the oracle (C restatement of rfi.host.FlaggerHost) on one core.

    The sample is the same workload cut down in baselines: 4096 channels x as many
    baselines as fit the time budget (start with 512, grow to at most 8192).
    """
    from oracle import rfi_oracle as oracle

    threads = max(1, min(host_cores(), oracle.max_threads(), 64))
    oracle.set_threads(1)
    baselines = 512
    vis = synth_block(CHANNELS, baselines, 1)
    t0 = time.perf_counter()
    oracle.flagger_full(vis, width=WIDTH, n_sigma=N_SIGMA)
    dt = time.perf_counter() - t0
    # one bigger, timed run sized for ~budget/2 seconds
    scale = max(1, min(16, int(0.5 * budget_s / max(dt, 1e-3))))
    if scale > 1:
        baselines *= scale
        vis = synth_block(CHANNELS, baselines, 1)
        t0 = time.perf_counter()
        oracle.flagger_full(vis, width=WIDTH, n_sigma=N_SIGMA)
        dt = time.perf_counter() - t0
    samples = CHANNELS * baselines
    single = samples / dt
    dt_all = dt
    if threads > 1:
        # all cores this process may use, for context (OpenMP over baselines)
        oracle.set_threads(threads)
        t0 = time.perf_counter()
        oracle.flagger_full(vis, width=WIDTH, n_sigma=N_SIGMA)
        dt_all = time.perf_counter() - t0
        oracle.set_threads(1)
    out = {
        "value": single,
        "unit": "samples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{CHANNELS} ch x {baselines} bl complex64, one call, {dt:.2f} s",
        "host_cpus": os.cpu_count(),
        "affinity_cores": host_cores(),  # what this process may run on (sched_getaffinity)
    }
    if threads > 1:  # (a process confined to one core has no all-cores figure)
        out["all_cores_value"] = samples / dt_all
        out["all_cores"] = threads
    try:
        out["numpy_path"] = numpy_baseline()
    except Exception as exc:  # the headline must not depend on it
        out["numpy_path"] = {"error": repr(exc)}
    return out


def check_against_oracle(vis_slice, mask, flags_slice, noise_slice) -> dict:
    """Compare a slice of a timed launch's output with the CPU oracle (outside the timed
    region; the oracle is the checker, never the thing measured). Exits loudly on a
    mismatch: a fast kernel with different results is not a result."""
    from oracle import rfi_oracle as oracle

    oracle.set_threads(max(1, min(host_cores(), oracle.max_threads(), 64)))
    try:
        ref_flags, ref_noise = oracle.flagger_full(vis_slice, mask, width=WIDTH, n_sigma=N_SIGMA)
    finally:
        oracle.set_threads(1)
    flags_equal = bool(np.array_equal(ref_flags, flags_slice))
    noise_equal = bool(np.array_equal(ref_noise.astype(np.float32), noise_slice, equal_nan=True))
    if not (flags_equal and noise_equal):
        raise SystemExit(
            f"bench.py: GPU output differs from the oracle (flags_equal={flags_equal}, "
            f"noise_equal={noise_equal}) on the first {vis_slice.shape[1]} baselines"
        )
    return {
        "against": "oracle (C restatement of rfi.host.FlaggerHost)",
        "slice": f"{vis_slice.shape[0]} ch x {vis_slice.shape[1]} bl of the timed block",
        "flags_equal": flags_equal,
        "noise_equal": noise_equal,
        "flags_set_in_slice": int(np.count_nonzero(ref_flags)),
    }


def measured_traffic(channels, baselines, use_flags, args):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json, collected by tools/pmc_mem.sh as the micro-architecture
    guide prescribes); None when no pass exists for this exact workload."""
    if args.sequence or args.keep_deviations:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            for entry in json.load(f):
                if (entry["channels"], entry["baselines"], entry["use_flags"]) == \
                        (channels, baselines, use_flags):
                    return entry["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


class _StdoutToStderr:
    """Route file descriptor 1 to stderr for a while: RCCL prints a version banner on
    stdout when the first communicator is created, and stdout must carry exactly one
    JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child
    `torch.distributed.run` and relay rank 0's JSON line. This process never touches
    the GPU (nothing here imports torch or the HIP library)."""
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
        f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
        "--master-port", str(free_port()), os.path.abspath(__file__),
    ] + argv  # fmt: skip
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for text in proc.stdout.splitlines():
        if text.startswith("{") and '"metric"' in text:
            line = text
        elif text.strip():
            print(text, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def time_op(queue, fn, reps: int = 100, preheat_s: float = 0.04) -> float:
    """Average device seconds per call of `fn` (HIP events on the queue's stream), after
    enough untimed calls to have the device at its sustained clocks."""
    fn()
    queue.finish()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preheat_s:
        for _ in range(10):
            fn()
        queue.finish()
    a = queue.enqueue_marker()
    for _ in range(reps):
        fn()
    b = queue.enqueue_marker()
    queue.finish()
    return b.time_since(a) / reps


def bringup_configs(context, queue, vis_host) -> dict:
    """BASELINE.json configs 2 and 3 (the bring-up shapes), one GPU, HIP-event timing,
    algorithmic bytes of SURVEY.md 8(d). The 64-512 MiB arrays partly live in the
    256 MiB Infinity Cache between repeats, so these are upper bounds for larger inputs."""
    from katsdpsigproc_amd import percentile, transpose
    from katsdpsigproc_amd.rfi import device

    from oracle import rfi_oracle as oracle  # the checker of each leg's output, never timed

    def entry(seconds, nbytes, note, checked):
        gbs = nbytes / seconds / 1e9
        return {"ms": 1e3 * seconds, "algorithmic_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                "algorithmic_bytes": note, "output_checked": checked}  # fmt: skip

    def must(ok: bool, what: str) -> str:
        if not ok:
            raise SystemExit(f"bench.py: bring-up leg differs from its reference: {what}")
        return what

    SL = 64  # baselines (or rows) of each leg's output compared with the oracle / NumPy

    out = {}
    n = 4096
    rs = np.random.RandomState(1)
    src = np.abs(rs.standard_normal((n, n))).astype(np.float32)
    op = transpose.TransposeTemplate(context, np.float32, "float").instantiate(queue, (n, n))
    op.ensure_all_bound()
    op.buffer("src").set(queue, src)
    seconds = time_op(queue, op)
    out["config2_transpose_4096x4096_f32"] = entry(
        seconds, 8 * n * n, "8 B/element",
        must(np.array_equal(op.buffer("dest").get(queue), src.T), "whole output == src.T"))
    op = percentile.Percentile5Template(context, n, is_amplitude=True).instantiate(queue, (n, n))
    op.ensure_all_bound()
    op.buffer("src").set(queue, src)
    seconds = time_op(queue, op)
    out["config2_percentile5_4096x4096_f32"] = entry(
        seconds, 4 * n * n, "4 B/element",
        must(np.array_equal(op.buffer("dest").get(queue), oracle.percentile5(src)),
             "whole output == oracle.percentile5 (NumPy 'lower' percentiles)"))
    del op
    C, B = 4096, 8192
    block = np.ascontiguousarray(vis_host[:, :B])
    bg = device.BackgroundMedianFilterDeviceTemplate(context, WIDTH).instantiate(queue, C, B)
    bg.ensure_all_bound()
    bg.buffer("vis").set(queue, block)
    seconds = time_op(queue, bg)
    dev = bg.buffer("deviations").get(queue)
    ref_dev = oracle.BackgroundMedianFilterHost(WIDTH)(block[:, :SL])
    out["config3_background_4096x8192_c64"] = entry(
        seconds, 12 * C * B, "12 B/sample",
        must(np.array_equal(dev[:, :SL], ref_dev.astype(np.float32)),
             f"first {SL} baselines == float32(oracle deviations)"))
    for name, tmpl in (
        ("config3_noise_mad_4096x8192", device.NoiseEstMADDeviceTemplate(context)),
        ("config3_noise_mad_t_4096x8192", device.NoiseEstMADTDeviceTemplate(context, 10240)),
    ):
        ne = tmpl.instantiate(queue, C, B)
        ne.ensure_all_bound()
        ne.buffer("deviations").set(queue, np.ascontiguousarray(dev.T) if tmpl.transposed else dev)
        seconds = time_op(queue, ne)
        ref_noise = oracle.NoiseEstMADHost()(np.ascontiguousarray(dev[:, :SL]))
        out[name] = entry(
            seconds, 4 * C * B, "4 B/sample",
            must(np.array_equal(ne.buffer("noise").get(queue)[:SL], ref_noise.astype(np.float32)),
                 f"first {SL} baselines == float32(oracle MAD of the float32 deviations)"))
    del dev, bg, ne
    # the reference-shaped five-kernel sequence and the fused kernel on config 3's block
    for fused in (False, True):
        template = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(context, WIDTH),
            device.NoiseEstMADTDeviceTemplate(context, 10240),
            device.ThresholdSumDeviceTemplate(context),
            fused=fused,
        )
        fn = template.instantiate(queue, C, B, threshold_args={"n_sigma": N_SIGMA})
        fn.ensure_all_bound()
        fn.buffer("vis").set(queue, block)
        key = "flagger_fused_4096x8192" if fused else "flagger_sequence_5_kernels_4096x8192"
        seconds = time_op(queue, fn)
        ref_flags, _ = oracle.flagger_full(block[:, :SL], width=WIDTH, n_sigma=N_SIGMA)
        out[key] = entry(
            seconds, 9 * C * B, "9 B/sample",
            must(np.array_equal(fn.buffer("flags").get(queue)[:, :SL], ref_flags),
                 f"flags of the first {SL} baselines == oracle"))
    del fn
    # configurations the fused kernels refused before round 3 (they took the five-kernel
    # sequence): a median window of 25 channels; SumThreshold with 8 windows (up to 128 channels)
    for key, width, n_windows in (("flagger_fused_width25_4096x8192", 25, 4),
                                  ("flagger_fused_8_windows_4096x8192", WIDTH, 8)):  # fmt: skip
        template = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(context, width),
            device.NoiseEstMADTDeviceTemplate(context, 10240),
            device.ThresholdSumDeviceTemplate(context, n_windows=n_windows),
            fused=True,
        )
        fn = template.instantiate(queue, C, B, threshold_args={"n_sigma": N_SIGMA})
        fn.ensure_all_bound()
        fn.buffer("vis").set(queue, block)
        seconds = time_op(queue, fn)
        ref_flags, _ = oracle.flagger_full(block[:, :SL], width=width, n_sigma=N_SIGMA,
                                           n_windows=n_windows)  # fmt: skip
        out[key] = entry(
            seconds, 9 * C * B, "9 B/sample",
            must(np.array_equal(fn.buffer("flags").get(queue)[:, :SL], ref_flags),
                 f"flags of the first {SL} baselines == oracle"))
    del fn
    # the reference script's wider presets (scripts/rfiflagtest.py:190-195): the same
    # number of samples laid out as 8192 and 10240 channels (fused long-band kernels)
    for channels in (8192, 10240):
        baselines = (C * B) // channels
        long_block = np.ascontiguousarray(block.reshape(-1)[: channels * baselines]).reshape(
            channels, baselines)
        template = device.FlaggerDeviceTemplate(
            device.BackgroundMedianFilterDeviceTemplate(context, WIDTH),
            device.NoiseEstMADTDeviceTemplate(context, 10240),
            device.ThresholdSumDeviceTemplate(context),
            fused=True,
        )
        fn = template.instantiate(queue, channels, baselines, threshold_args={"n_sigma": N_SIGMA})
        fn.ensure_all_bound()
        fn.buffer("vis").set(queue, long_block)
        seconds = time_op(queue, fn)
        ref_flags, _ = oracle.flagger_full(long_block[:, :SL], width=WIDTH, n_sigma=N_SIGMA)
        out[f"flagger_fused_{channels}x{baselines}"] = entry(
            seconds, 9 * channels * baselines, "9 B/sample",
            must(np.array_equal(fn.buffer("flags").get(queue)[:, :SL], ref_flags),
                 f"flags of the first {SL} baselines == oracle"))
    return out


def main() -> int:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=20)
    parser.add_argument("--preheat", type=int, default=PREHEAT_STEPS,
                        help="untimed steps before the warm-up that bring the device from idle to"
                             " its sustained clocks (the first ones are timed as `cold_start`)")  # fmt: skip
    parser.add_argument("--baselines", type=int, default=BASELINES_PER_GPU,
                        help="baselines per GPU (default: the benchmark configuration)")  # fmt: skip
    parser.add_argument("--channels", type=int, default=CHANNELS)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-extras", action="store_true",
                        help="skip the RFI-laden variant and the config 2/3 legs (N = 1 only)")  # fmt: skip
    parser.add_argument("--keep-deviations", action="store_true",
                        help="also write the deviations slot (13 B/sample variant)")  # fmt: skip
    parser.add_argument("--sequence", action="store_true",
                        help="time the reference-shaped 5-kernel sequence instead of the fused kernel")  # fmt: skip
    args = parser.parse_args()

    # KSP_BENCH_FORCE_LAUNCH=1 takes the launcher route for N = 1 too (rehearsal of the
    # self-launch on a one-GPU box, together with KSP_BENCH_FORCE_DIST=1)
    if "WORLD_SIZE" not in os.environ and (
        args.gpus > 1 or os.environ.get("KSP_BENCH_FORCE_LAUNCH") == "1"
    ):
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    torch = None
    # KSP_BENCH_FORCE_DIST=1 exercises the torch.distributed code path with a single
    # rank too (used to rehearse the multi-GPU path on a one-GPU box)
    use_dist = world > 1 or os.environ.get("KSP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))  # fmt: skip
            # create the communicator now (its banner goes to stderr with the redirect)
            dist.barrier()

    from katsdpsigproc_amd import accel, hip
    from katsdpsigproc_amd.rfi import device

    devices = hip.Device.get_devices()
    if not devices:
        raise SystemExit("no HIP device: this benchmark needs an MI355X (no CPU fallback)")
    context = devices[local_rank if world > 1 else 0].make_context()
    if use_dist:
        # run on torch's current stream so the RCCL broadcast orders with the kernel
        queue = hip.CommandQueue(context, stream=torch.cuda.current_stream().cuda_stream)
    else:
        queue = context.create_command_queue()

    channels, baselines = args.channels, args.baselines
    use_flags = device.BackgroundFlags.CHANNEL if use_dist else device.BackgroundFlags.NONE
    template = device.FlaggerDeviceTemplate(
        device.BackgroundMedianFilterDeviceTemplate(context, WIDTH, use_flags=use_flags),
        device.NoiseEstMADTDeviceTemplate(context, 10240),
        device.ThresholdSumDeviceTemplate(context),
        fused=not args.sequence,
        keep_deviations=args.keep_deviations,
    )
    fn = template.instantiate(queue, channels, baselines, threshold_args={"n_sigma": N_SIGMA})
    fn.ensure_all_bound()

    # synthetic input, resident in HBM before the timed region
    vis = synth_block(channels, baselines, seed=1 + rank)
    fn.buffer("vis").set(queue, vis)
    n_check = min(CHECK_BASELINES, baselines)
    mask = None
    pipe = None
    if use_dist:
        # The channel mask is rank 0's to decide and changes from block to block in a
        # live system, so every step broadcasts it (4 KiB, RCCL over xGMI). The broadcast
        # for step k + 1 runs on its own stream while step k's kernel is busy: two mask
        # buffers alternate under the flagger's input_flags slot, events order
        # "broadcast into buffer i" before "kernel reads buffer i" before the next
        # broadcast into it.
        mask = (np.random.RandomState(2).random_sample(channels) < 1.0 / 16.0).astype(np.uint8)
        first = fn.buffer("input_flags")
        second = accel.DeviceArray(context, first.shape, first.dtype, first.padded_shape)
        bufs = [first, second]
        for buf in bufs:
            if rank == 0:
                buf.set(queue, mask)
            else:
                buf.zero(queue)
        queue.finish()
        dev_t = torch.device("cuda", local_rank)
        comm = torch.cuda.Stream(device=dev_t)
        pipe = {
            "bufs": bufs,
            "tensors": [torch.as_tensor(b.buffer, device=dev_t) for b in bufs],
            "comm": comm,
            # the same stream as a command queue: events that only order the two streams
            # (no time stamp, no cache write-back when recorded) go through it
            "comm_queue": hip.CommandQueue(context, stream=comm.cuda_stream),
            "ready": [None, None],  # broadcast into buffer i done
            "free": [None, None],   # kernel reading buffer i done
            "k": 0,
        }
        with torch.cuda.stream(comm):
            dist.broadcast(pipe["tensors"][0], src=0)
        # (the "ready" edge follows a broadcast, i.e. possibly a peer's writes over xGMI: a
        # normal event, whose record flushes; the "free" edge orders a local kernel before
        # a later broadcast and needs no more than ordering)
        pipe["ready"][0] = pipe["comm_queue"].enqueue_marker()
        pipe["free"][1] = queue.enqueue_marker(ordering_only=True)

    def step() -> None:
        if pipe is None:
            fn()
            return
        i = pipe["k"] % 2
        j = 1 - i
        pipe["k"] += 1
        queue.enqueue_wait_for_events([pipe["ready"][i]])
        fn.bind(input_flags=pipe["bufs"][i])
        fn()
        pipe["free"][i] = queue.enqueue_marker(ordering_only=True)
        pipe["comm_queue"].enqueue_wait_for_events([pipe["free"][j]])  # the kernel that last read buffer j
        with torch.cuda.stream(pipe["comm"]):
            dist.broadcast(pipe["tensors"][j], src=0)
        pipe["ready"][j] = pipe["comm_queue"].enqueue_marker()

    def sync() -> None:
        queue.finish()
        if torch is not None:
            torch.cuda.synchronize()

    def timed(steps: int, warmup: int, barrier: bool):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronise.
        Returns (wall seconds, device seconds): the latter between two HIP events on the
        flagger's stream, recorded before the first and after the last timed step -- none
        in between: an event per step (or per kernel) costs the stream 3-10 us of its own."""
        for _ in range(warmup):
            step()
        sync()
        if barrier and dist is not None:
            dist.barrier()
        sync()
        first = queue.enqueue_marker()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        last = queue.enqueue_marker()
        sync()
        if barrier and dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
        return wall, last.time_since(first)

    def sampled(steps: int):
        """Per-launch figures of `steps` further (untimed) steps: device seconds of each
        step (zero-fill + kernel, events between the steps) and, for the fused path, of
        each kernel by itself (events around the kernel)."""
        marks = [queue.enqueue_marker()]
        kernel_events = []
        for _ in range(steps):
            if not args.sequence:
                kernel_events.append(fn.profile_next_run())
            step()
            marks.append(queue.enqueue_marker())
        sync()
        step_s = [b.time_since(a) for a, b in zip(marks[:-1], marks[1:])]
        kernel_s = [stop.time_since(start) for start, stop in kernel_events]
        return step_s, kernel_s or step_s

    # From idle the device takes some tens of milliseconds of load to reach its sustained
    # clocks (launches right after set-up run up to 15 % longer). A continuously fed
    # flagger lives in the sustained state, so that is what the timed region measures;
    # the first launches are reported beside it as `cold_start`.
    # the reference script's own method (scripts/rfiflagtest.py:88-107): one warm-up call,
    # then repeats timed one by one, from an idle device -- median of 10
    single_shot = None
    if args.preheat > 0 and pipe is None:
        step()
        sync()
        s_step, _ = sampled(10)
        single_shot = {"method": "one warm-up call, then 10 calls timed one by one from idle (median)",
                       "step_device_ms": 1e3 * float(np.median(s_step)),
                       "frac": channels * baselines * (ALGORITHMIC_BYTES_PER_SAMPLE + (4 if args.keep_deviations else 0))
                               / float(np.median(s_step)) / 1e9 / HBM_PEAK_GBS}  # fmt: skip
    cold = None
    if args.preheat > 0:
        n_cold = min(20, args.preheat)
        c_step, c_kernel = sampled(n_cold)
        cold = {"launches": n_cold, "step_device_ms": 1e3 * float(np.mean(c_step)),
                "kernel_ms": 1e3 * float(np.mean(c_kernel))}  # fmt: skip
        for _ in range(args.preheat - n_cold):
            step()
    elapsed, device_total = timed(args.steps, args.warmup, True)
    step_s, kernel_s = sampled(min(args.steps, 50))
    stats = [elapsed, device_total / args.steps, float(np.mean(kernel_s)),
             float(np.max(step_s)), float(np.max(kernel_s)),
             -float(np.min(step_s)), -float(np.min(kernel_s)), float(np.mean(step_s))]  # fmt: skip
    if dist is not None:
        t = torch.tensor(stats, dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        stats = [float(x) for x in t]
    elapsed, step_mean, kernel_mean, step_max, kernel_max, step_min, kernel_min, step_sampled = stats
    step_min, kernel_min = -step_min, -kernel_min

    # the timed launch's own output against the oracle (every rank checks its block)
    verified = None
    if not args.sequence:
        flags_out = fn.buffer("flags").get(queue)
        noise_out = fn.buffer("noise").get(queue)
        verified = check_against_oracle(
            np.ascontiguousarray(vis[:, :n_check]), mask,
            flags_out[:, :n_check], noise_out[:n_check],
        )  # fmt: skip
        verified["flagged_fraction_of_block"] = float(np.count_nonzero(flags_out)) / flags_out.size
        del flags_out
        if pipe is not None:
            # every rank: both mask buffers hold what rank 0 broadcast
            got = [b.get(queue) for b in pipe["bufs"]]
            ok = all(np.array_equal(g, mask) for g in got)
            if not ok:
                raise SystemExit(f"bench.py: rank {rank}: a broadcast channel mask differs from rank 0's")
            verified["mask_buffers_equal_on_rank"] = rank

    samples_per_gpu = channels * baselines
    n_bytes = ALGORITHMIC_BYTES_PER_SAMPLE + (4 if args.keep_deviations else 0)
    value = samples_per_gpu * world * args.steps / elapsed

    result = None
    if rank == 0:
        achieved = samples_per_gpu * world * n_bytes / step_mean / 1e9
        peak = HBM_PEAK_GBS * world
        result = {
            "metric": "visibility samples/s (baselines x channels) through full RFI flagger",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat_steps": args.preheat,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 amplitude / f64 deviations",
            "data": "synthetic",
            "config": {
                "workload": f"full SumThreshold flagger, {channels} ch x {baselines} bl per GPU"
                            f" ({channels} x {baselines * world} total), complex64,"
                            f" width {WIDTH}, {N_SIGMA} sigma, 4 windows",
                "path": "sequence (5 kernels)" if args.sequence else "fused single-pass kernel",
                "use_flags": use_flags.name,
                "keep_deviations": bool(args.keep_deviations),
                "sharding": f"baselines over {world} GPU(s)"
                            + (", RCCL broadcast of the channel mask per step; N > 1 runs"
                               " BackgroundFlags.CHANNEL (1/16 of the channels masked), N = 1"
                               " runs without input flags" if use_dist else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "sequence" if args.sequence
                          else ("flagger_fused_kernel (4-baseline strips: a channel mask, or the"
                                " deviations output, rules the ring kernel out) + zero-fill of flags"
                                " (one step)" if use_dist or args.keep_deviations
                                else "flagger_ring_kernel (persistent, 8-baseline strips) + zero-fill"
                                     " of flags (one step)"),
                "achieved": achieved,
                "peak": peak,
                "unit": "GB/s",
                "frac": achieved / peak,
                "algorithmic_bytes_per_sample": n_bytes,
                "step_device_ms": 1e3 * step_mean,
                "how": "achieved = algorithmic bytes of one step / step_device_ms, the device time"
                       " between two HIP events around the whole timed region / steps; `sampled`"
                       " = 50 further launches with events per step and around each kernel",
                "sampled": {
                    "step_device_ms": {"mean": 1e3 * step_sampled, "min": 1e3 * step_min,
                                       "max": 1e3 * step_max},
                    "kernel_ms": {"mean": 1e3 * kernel_mean, "min": 1e3 * kernel_min,
                                  "max": 1e3 * kernel_max},
                },
                "frac_kernel_only": samples_per_gpu * n_bytes / kernel_mean / 1e9 / HBM_PEAK_GBS,
                "traffic": measured_traffic(channels, baselines, use_flags.name, args),
                "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC passes of the same launch,"
                                  " tools/pmc_mem.sh; not read in this run)",
                "cold_start": cold,
                "single_shot": single_shot,
            },
            "verified": verified,
        }  # fmt: skip

    extras = world == 1 and not use_dist and not args.no_extras and not args.sequence
    if extras:
        # the same launch on RFI-laden input: SumThreshold's windows are summed and the
        # flagged bytes written (the plain input of the reference's script gives 0 flags)
        inject_rfi(vis, seed=3)
        fn.buffer("vis").set(queue, vis)
        r_elapsed, r_device = timed(args.steps, args.preheat + args.warmup, False)
        r_step, r_kernel = sampled(min(args.steps, 50))
        flags_out = fn.buffer("flags").get(queue)
        noise_out = fn.buffer("noise").get(queue)
        r_verified = check_against_oracle(
            np.ascontiguousarray(vis[:, :n_check]), None,
            flags_out[:, :n_check], noise_out[:n_check],
        )  # fmt: skip
        result["rfi_variant"] = {
            "input": "the same block + interference on 1/16 of the samples (amplitude U(50,70))",
            "ms_per_step": 1e3 * r_elapsed / args.steps,
            "step_device_ms": 1e3 * r_device / args.steps,
            "kernel_ms": 1e3 * float(np.mean(r_kernel)),
            "frac": samples_per_gpu * n_bytes / (r_device / args.steps) / 1e9 / HBM_PEAK_GBS,
            "flagged_fraction": float(np.count_nonzero(flags_out)) / flags_out.size,
            "verified": r_verified,
        }
        del flags_out
        result["bringup"] = bringup_configs(context, queue, vis)
    del vis
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
