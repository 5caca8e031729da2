"""Sharding the flagger over the GPUs of one node.

Every stage of the flagger is independent per baseline, so N GPUs each take a
contiguous range of baselines (a dense ``[channels][baselines / N]`` block of their own)
and run the single-GPU flagger on it; outputs stay sharded. The only shared input is
the per-channel flag mask (``BackgroundFlags.CHANNEL``): rank 0 owns it and it is
broadcast -- 1 byte per channel -- with ``torch.distributed`` (backend ``"nccl"`` is
RCCL over xGMI on ROCm; ``"gloo"`` on CPU for tests). Optionally the per-baseline
noise estimates can be all-gathered. There is no other exchange.

One process per GPU; the reference has no multi-device support at all
(doc/user/init.rst:4-6), so this module has no counterpart there.
"""

from typing import List, Optional, Tuple

import numpy as np

#: shard boundary in baselines: 64-byte row segments, and a multiple of the fused
#: kernel's 4-baseline strip so that no strip straddles two GPUs
STRIP = 8


class BaselineSharding:
    """Contiguous, strip-aligned partition of `baselines` over `world_size` ranks."""

    def __init__(self, baselines: int, world_size: int, rank: int, align: int = STRIP) -> None:
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        if baselines < 0 or align < 1:
            raise ValueError("bad baselines/align")
        self.baselines = baselines
        self.world_size = world_size
        self.rank = rank
        self.align = align
        units = -(-baselines // align)  # strips, the last one possibly partial
        base, extra = divmod(units, world_size)
        bounds = [0]
        for r in range(world_size):
            bounds.append(bounds[-1] + (base + (1 if r < extra else 0)) * align)
        self._bounds = [min(b, baselines) for b in bounds]

    def range_of(self, rank: int) -> Tuple[int, int]:
        return self._bounds[rank], self._bounds[rank + 1]

    @property
    def start(self) -> int:
        return self._bounds[self.rank]

    @property
    def stop(self) -> int:
        return self._bounds[self.rank + 1]

    @property
    def count(self) -> int:
        return self.stop - self.start

    def all_ranges(self) -> List[Tuple[int, int]]:
        return [self.range_of(r) for r in range(self.world_size)]


def _dist():
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    return dist


def broadcast_channel_mask(mask, src: int = 0, group=None):
    """Broadcast the per-channel flag mask from rank `src` to every rank, in place.

    `mask` is either a numpy uint8 array (host path: works with any backend, used by the
    gloo tests) or a :class:`~katsdpsigproc_amd.accel.DeviceArray` bound to the
    flagger's ``input_flags`` slot (device path: RCCL writes straight into the buffer
    the kernel reads; run the flagger on torch's current stream so the two order).
    """
    import torch

    dist = _dist()
    if isinstance(mask, np.ndarray):
        if mask.dtype != np.uint8:
            raise TypeError("channel mask must be uint8")
        tensor = torch.from_numpy(mask)
        if dist.get_backend(group) == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            staged = tensor.to(dev)
            dist.broadcast(staged, src=src, group=group)
            mask[...] = staged.cpu().numpy()
        else:
            dist.broadcast(tensor, src=src, group=group)
        return mask
    # DeviceArray: zero-copy view through __cuda_array_interface__
    buffer = mask.buffer
    tensor = torch.as_tensor(buffer, device=torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(tensor, src=src, group=group)
    return mask


def all_gather_noise(noise_local: np.ndarray, sharding: BaselineSharding, group=None) -> np.ndarray:
    """Collect every rank's per-baseline noise into one array (optional; host arrays)."""
    import torch

    dist = _dist()
    longest = max(b - a for a, b in sharding.all_ranges())
    padded = np.zeros(longest, np.float32)
    padded[: sharding.count] = noise_local
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = torch.from_numpy(padded).to(dev)
        parts = [torch.empty_like(mine) for _ in range(sharding.world_size)]
        dist.all_gather(parts, mine, group=group)
        parts = [p.cpu() for p in parts]
    else:
        mine = torch.from_numpy(padded)
        parts = [torch.empty_like(mine) for _ in range(sharding.world_size)]
        dist.all_gather(parts, mine, group=group)
    out = np.empty(sharding.baselines, np.float32)
    for (a, b), part in zip(sharding.all_ranges(), parts):
        out[a:b] = part.numpy()[: b - a]
    return out
