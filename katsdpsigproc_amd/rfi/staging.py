"""Blocks of visibilities streamed through the flagger with the PCIe transfers overlapped.

The piece of an ingest pipeline that sits either side of the flagger: the reference gives
callers ``DeviceArray.set_async`` / ``get_async`` plus events (reference accel.py:573-586,
doc/user/sync.rst:24-42) and leaves the arrangement to them; ``FlaggerHostFromDevice``
(reference rfi/device.py:1169-1222) allocates, copies, runs and copies back serially on
every call. :class:`StagedFlagger` keeps `depth` sets of device and pinned host buffers
and three in-order queues -- upload, compute, download -- ordered by events only:

    upload[k]:   wait(compute done with set k's previous block)   H2D vis
    compute[k]:  wait(upload[k])                                   flagger
    download[k]: wait(compute[k])                                  D2H flags (+ noise)

so block n + 1 uploads while block n is flagged and block n - 1 downloads. The host
never blocks except to hand back finished flags (or when it runs `depth` blocks ahead).
"""

from typing import Any, Iterable, Iterator, List, Mapping, Optional, Tuple

import numpy as np

from . import device


class StagedFlagger:
    """
    Parameters
    ----------
    template
        A :class:`~katsdpsigproc_amd.rfi.device.FlaggerDeviceTemplate`
    context
        Context to allocate in; three command queues are created on it
    channels, baselines
        Shape of every block
    depth
        Number of blocks that can be in flight (2 = double buffering)
    threshold_args, background_args, noise_est_args
        As for ``FlaggerDeviceTemplate.instantiate``
    """

    def __init__(self, template: "device.FlaggerDeviceTemplate", context, channels: int,
                 baselines: int, depth: int = 2, threshold_args: Mapping[str, Any] = {},
                 background_args: Mapping[str, Any] = {},
                 noise_est_args: Mapping[str, Any] = {}) -> None:  # fmt: skip
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.template = template
        self.context = context
        self.shape = (channels, baselines)
        self.depth = depth
        self.upload = context.create_command_queue()
        self.compute = context.create_command_queue()
        self.download = context.create_command_queue()
        self.uses_input_flags = bool(template.background.use_flags)
        self._sets: List[dict] = []
        for _ in range(depth):
            fn = template.instantiate(self.compute, channels, baselines, background_args,
                                      noise_est_args, threshold_args)  # fmt: skip
            fn.ensure_all_bound()
            entry = {
                "fn": fn,
                "vis": fn.slots["vis"].allocate_host(context),
                "flags": fn.slots["flags"].allocate_host(context),
                "noise": fn.slots["noise"].allocate_host(context),
                "computed": None,    # event: the flagger has finished with this set's vis
                "downloaded": None,  # event: this set's results are in host memory
            }
            if self.uses_input_flags:
                entry["input_flags"] = fn.slots["input_flags"].allocate_host(context)
            self._sets.append(entry)
        self._submitted = 0
        self._collected = 0

    # ------------------------------------------------------------------ one block
    def host_buffers(self) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Pinned host arrays (vis, input_flags or None) of the set the NEXT
        :meth:`submit` will use: a producer can write a block straight into them (and
        then call ``submit()`` with no arguments) instead of having it copied."""
        entry = self._sets[self._submitted % self.depth]
        self._wait_host(entry)
        return entry["vis"], entry.get("input_flags")

    def _wait_host(self, entry: dict) -> None:
        """The set's host buffers may be rewritten once its upload has been consumed by
        the flagger and its results have been collected."""
        if self._submitted - self._collected >= self.depth:
            raise RuntimeError(f"{self.depth} blocks are in flight: collect() one first")
        if entry["computed"] is not None:
            entry["computed"].wait()  # its previous upload is long done by then

    def submit(self, vis: Optional[np.ndarray] = None,
               input_flags: Optional[np.ndarray] = None) -> int:  # fmt: skip
        """Queue one block (copied into the pinned buffers unless already written there
        through :meth:`host_buffers`); returns its sequence number. Never waits for the
        device unless `depth` blocks are already in flight."""
        entry = self._sets[self._submitted % self.depth]
        self._wait_host(entry)
        if vis is not None:
            if vis.shape != self.shape:
                raise ValueError(f"block has shape {vis.shape}, expected {self.shape}")
            entry["vis"][...] = vis
        if self.uses_input_flags:
            if input_flags is not None:
                entry["input_flags"][...] = input_flags
        elif input_flags is not None:
            raise TypeError("channel flags were provided but not included in the template")
        fn = entry["fn"]
        # the flagger's previous read of this set's vis is ordered before the overwrite
        if entry["computed"] is not None:
            self.upload.enqueue_wait_for_events([entry["computed"]])
        fn.buffer("vis").set_async(self.upload, entry["vis"])
        if self.uses_input_flags:
            fn.buffer("input_flags").set_async(self.upload, entry["input_flags"])
        uploaded = self.upload.enqueue_marker()
        waits = [uploaded]
        if entry["downloaded"] is not None:
            waits.append(entry["downloaded"])  # flags/noise buffers are being read out
        self.compute.enqueue_wait_for_events(waits)
        fn()
        entry["computed"] = self.compute.enqueue_marker()
        self.download.enqueue_wait_for_events([entry["computed"]])
        fn.buffer("flags").get_async(self.download, entry["flags"])
        fn.buffer("noise").get_async(self.download, entry["noise"])
        entry["downloaded"] = self.download.enqueue_marker()
        self._submitted += 1
        return self._submitted - 1

    def collect(self, copy: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """(flags, noise) of the oldest block in flight, waiting for it if necessary.

        Without `copy` the arrays are VIEWS of this block's pinned buffers, which the download
        of the block submitted `depth` submissions later overwrites: they are valid until
        that :meth:`submit` (in :meth:`run`: until the generator is advanced again). Pass
        ``copy=True``, or copy what has to live longer."""
        if self._collected >= self._submitted:
            raise RuntimeError("no block is in flight")
        entry = self._sets[self._collected % self.depth]
        entry["downloaded"].wait()
        for queue in (self.upload, self.download):
            queue.release_host_references()
        self._collected += 1
        if copy:
            return entry["flags"].copy(), entry["noise"].copy()
        return entry["flags"], entry["noise"]

    # ------------------------------------------------------------------ a stream
    def run(self, blocks: Iterable, copy: bool = False) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        """Flag every block of `blocks` (arrays, or ``(vis, input_flags)`` pairs), yielding
        (flags, noise) in order while keeping up to `depth` blocks in flight. What is yielded
        is only valid until the generator is advanced (see :meth:`collect`) unless `copy` is
        set -- ``list(staged.run(blocks))`` needs ``copy=True``."""
        for block in blocks:
            if self._submitted - self._collected >= self.depth:
                yield self.collect(copy)
            if isinstance(block, tuple):
                self.submit(*block)
            else:
                self.submit(block)
        while self._collected < self._submitted:
            yield self.collect(copy)

    def finish(self) -> None:
        for queue in (self.upload, self.compute, self.download):
            queue.finish()
