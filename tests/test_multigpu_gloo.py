"""world_size-2 test of the multi-GPU path on CPU (gloo): baseline sharding, the
channel-mask broadcast and the optional noise all-gather.

The per-rank compute is stood in for by the CPU oracle (allowed in tests); what is
under test is the partitioning and the collectives of katsdpsigproc_amd.multigpu, i.e.
that flagging each shard with the broadcast mask reproduces the unsharded result.
"""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharding_ranges():
    from katsdpsigproc_amd.multigpu import BaselineSharding

    for baselines, world in [(262144, 8), (32768, 1), (131, 2), (8, 3), (0, 2), (1000, 7)]:
        ranges = BaselineSharding(baselines, world, 0).all_ranges()
        assert ranges[0][0] == 0 and ranges[-1][1] == baselines
        for (a0, b0), (a1, b1) in zip(ranges, ranges[1:]):
            assert b0 == a1 and a0 <= b0
        for a, b in ranges[:-1]:
            assert a % 8 == 0 and (b % 8 == 0 or b == baselines)
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) < 16 or baselines < 8 * world
    s = BaselineSharding(262144, 8, 3)
    assert (s.start, s.stop, s.count) == (3 * 32768, 4 * 32768, 32768)
    with pytest.raises(ValueError):
        BaselineSharding(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from katsdpsigproc_amd.multigpu import (
        BaselineSharding, all_gather_noise, broadcast_channel_mask,
    )  # fmt: skip
    from oracle import rfi_oracle as oracle
    from tests import inputs

    dist.init_process_group(
        "gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world
    )
    try:
        channels, baselines = 256, 42  # 42 baselines -> strips of 8: shards 24 + 18
        vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=3), seed=4)
        sharding = BaselineSharding(baselines, world, rank)
        # only rank 0 knows the mask; the others start with garbage
        mask = inputs.channel_mask(channels) if rank == 0 else np.full(channels, 255, np.uint8)
        broadcast_channel_mask(mask, src=0)
        np.testing.assert_array_equal(mask, inputs.channel_mask(channels))
        shard = np.ascontiguousarray(vis[:, sharding.start : sharding.stop])
        flags, noise = oracle.flagger_full(shard, mask)
        full_noise = all_gather_noise(noise.astype(np.float32), sharding)
        np.save(os.path.join(tmpdir, f"flags_{rank}.npy"), flags)
        if rank == 0:
            np.save(os.path.join(tmpdir, "noise.npy"), full_noise)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_flagger_gloo(tmp_path):
    import torch.multiprocessing as mp

    from oracle import rfi_oracle as oracle
    from tests import inputs

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    channels, baselines = 256, 42
    vis = inputs.add_rfi(inputs.generate_data(channels, baselines, seed=3), seed=4)
    ref_flags, ref_noise = oracle.flagger_full(vis, inputs.channel_mask(channels))
    flags = np.concatenate(
        [np.load(os.path.join(tmp_path, f"flags_{r}.npy")) for r in range(world)], axis=1
    )
    np.testing.assert_array_equal(ref_flags, flags)
    assert flags.sum() > 0
    np.testing.assert_array_equal(
        ref_noise.astype(np.float32), np.load(os.path.join(tmp_path, "noise.npy"))
    )
