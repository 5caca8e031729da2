#!/usr/bin/env python3
"""Benchmark of the RFI-flagging hot path on MI355X.

One "step" = one pass of the full flagger (median-filter background, MAD noise
estimate, SumThreshold) over one block of synthetic visibilities that is already
resident in HBM: 4096 channels x 32768 baselines of complex64 per GPU
(BASELINE.json config 4; with N GPUs the baselines are sharded, N x 32768 in total,
config 5). Prints ONE JSON line (see the task contract): whole-job samples/s, the
HBM roofline of the dominant kernel from HIP-event timing, and a CPU baseline (the
oracle's C restatement of rfi.host, timed on this box's cores).

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N

Parameters follow scripts/rfiflagtest.py of the reference: width 13, 11 sigma,
4 windows, falloff 1.2, RandomState(seed=1) standard-normal real/imag.
"""

import argparse
import json
import os
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
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGORITHMIC_BYTES_PER_SAMPLE = 9  # 8 B complex64 read + 1 B flag written


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


def cpu_baseline(budget_s: float = 20.0):
    """Time the oracle (C restatement of rfi.host.FlaggerHost) on one core.

    The sample is the same workload cut down in baselines: 4096 channels x as many
    baselines as fit the time budget (start with 512, grow to at most 8192).
    """
    from oracle import rfi_oracle as oracle

    try:
        cores_available = len(os.sched_getaffinity(0))
    except AttributeError:
        cores_available = os.cpu_count() or 1
    threads = max(1, min(cores_available, oracle.max_threads(), 64))
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
    # all cores, for context (OpenMP over baselines)
    oracle.set_threads(threads)
    t0 = time.perf_counter()
    oracle.flagger_full(vis, width=WIDTH, n_sigma=N_SIGMA)
    dt_all = time.perf_counter() - t0
    oracle.set_threads(1)
    return {
        "value": single,
        "unit": "samples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{CHANNELS} ch x {baselines} bl complex64, one call, {dt:.2f} s",
        "all_cores_value": samples / dt_all,
        "all_cores": threads,
        "host_cpus": os.cpu_count(),
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


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=20)
    parser.add_argument("--warmup", type=int, default=3)
    parser.add_argument("--baselines", type=int, default=BASELINES_PER_GPU,
                        help="baselines per GPU (default: the benchmark configuration)")  # fmt: skip
    parser.add_argument("--channels", type=int, default=CHANNELS)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--keep-deviations", action="store_true",
                        help="also write the deviations slot (13 B/sample variant)")  # fmt: skip
    parser.add_argument("--sequence", action="store_true",
                        help="time the reference-shaped 5-kernel sequence instead of the fused kernel")  # fmt: skip
    args = parser.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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
    del vis
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
        pipe = {
            "bufs": bufs,
            "tensors": [torch.as_tensor(b.buffer, device=dev_t) for b in bufs],
            "ready": [torch.cuda.Event(), torch.cuda.Event()],  # broadcast into buffer i done
            "free": [torch.cuda.Event(), torch.cuda.Event()],   # kernel reading buffer i done
            "comm": torch.cuda.Stream(device=dev_t),
            "compute": torch.cuda.current_stream(),
            "k": 0,
        }
        with torch.cuda.stream(pipe["comm"]):
            dist.broadcast(pipe["tensors"][0], src=0)
            pipe["ready"][0].record(pipe["comm"])
        pipe["free"][1].record(pipe["compute"])

    def step() -> None:
        if pipe is None:
            fn()
            return
        i = pipe["k"] % 2
        j = 1 - i
        pipe["k"] += 1
        pipe["compute"].wait_event(pipe["ready"][i])
        fn.bind(input_flags=pipe["bufs"][i])
        fn()
        pipe["free"][i].record(pipe["compute"])
        with torch.cuda.stream(pipe["comm"]):
            pipe["comm"].wait_event(pipe["free"][j])  # the kernel that last read buffer j
            dist.broadcast(pipe["tensors"][j], src=0)
            pipe["ready"][j].record(pipe["comm"])

    def sync() -> None:
        queue.finish()
        if torch is not None:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    start_evt = queue.enqueue_marker()
    kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if not args.sequence:
            # HIP events recorded around the flagger kernel itself, on its stream
            kernel_events.append(fn.profile_next_run())
        step()
    end_evt = queue.enqueue_marker()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    device_s = end_evt.time_since(start_evt)
    if kernel_events:
        device_s = sum(stop.time_since(start) for start, stop in kernel_events)
    if dist is not None:
        t = torch.tensor([elapsed, device_s], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, device_s = float(t[0]), float(t[1])

    samples_per_step = channels * baselines * world
    value = samples_per_step * args.steps / elapsed
    # dominant kernel: flagger_fused_kernel; its average launch duration comes from the
    # HIP events armed around every launch (the zero-fill of the flags, a separate
    # memset kernel of ~19 us, is part of the step but not of this kernel)
    kernel_s = device_s / args.steps
    n_bytes = ALGORITHMIC_BYTES_PER_SAMPLE + (4 if args.keep_deviations else 0)
    achieved = channels * baselines * n_bytes / kernel_s / 1e9

    if rank == 0:
        result = {
            "metric": "visibility samples/s (baselines x channels) through full RFI flagger",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
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
                            + (", RCCL broadcast of the channel mask per step" if use_dist else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "sequence" if args.sequence else "flagger_fused_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_sample": n_bytes,
                "kernel_ms": 1e3 * kernel_s,
                "traffic": measured_traffic(channels, baselines, use_flags.name, args),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
