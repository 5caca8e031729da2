"""Hand-off of contended resources (device buffers, queues) between asyncio tasks.

Counterpart of the reference's ``resource`` module (reference:
src/katsdpsigproc/resource.py:31-253). A :class:`Resource` is passed from one user to
the next through futures whose values are lists of *device events*: a task that wants
the resource calls :meth:`Resource.acquire` (which only takes a place in the queue),
awaits :meth:`ResourceAllocation.wait` for the events of the previous user, makes its
own device work wait for them (``command_queue.enqueue_wait_for_events``) or waits on
the host (:meth:`ResourceAllocation.wait_events`), and finally calls
:meth:`ResourceAllocation.ready` with the events that mark its own work. Nothing here
touches a device: events only need a blocking ``wait()``.
"""

import asyncio
import collections
import logging
from typing import Awaitable, Deque, Generic, Iterable, List, Optional, TypeVar

from .abc import AbstractEvent

_logger = logging.getLogger(__name__)
_T = TypeVar("_T")


async def wait_until(future: Awaitable[_T], when: float,
                     loop: Optional[asyncio.AbstractEventLoop] = None) -> _T:  # fmt: skip
    """Await `future`, giving up (``asyncio.TimeoutError``, future cancelled) once the
    loop's clock reaches the absolute time `when`."""
    if loop is None:
        loop = asyncio.get_event_loop()
    task = asyncio.ensure_future(future, loop=loop)
    woken: "asyncio.Future[None]" = loop.create_future()

    def wake(*_args) -> None:
        if not woken.done():
            woken.set_result(None)

    timer = loop.call_at(when, wake)
    task.add_done_callback(wake)
    try:
        await woken
        if task.done():
            return task.result()
        task.remove_done_callback(wake)
        task.cancel()
        raise asyncio.TimeoutError()
    finally:
        timer.cancel()


async def async_wait_for_events(events: Iterable[AbstractEvent],
                                loop: Optional[asyncio.AbstractEventLoop] = None) -> None:  # fmt: skip
    """Wait for device events without blocking the event loop (the blocking waits run in
    the default executor)."""

    def block(pending: List[AbstractEvent]) -> None:
        for event in pending:
            event.wait()
        # drop the references in the worker thread BEFORE the awaiting task can drop
        # its own, so that an event is never destroyed by a thread that is unaware of it
        pending.clear()

    if loop is None:
        loop = asyncio.get_event_loop()
    pending = list(events)
    if pending:
        await loop.run_in_executor(None, block, pending)


class ResourceAllocation(Generic[_T]):
    """A place in a resource's queue (made by :meth:`Resource.acquire`, never directly).

    As a context manager it yields the resource's value and, should the block leave
    without :meth:`ready` having been called, passes the exception on to the next user
    (or releases the resource with a warning if there was none).
    """

    def __init__(self, start: "asyncio.Future[List[AbstractEvent]]",
                 end: "asyncio.Future[List[AbstractEvent]]", value: _T,
                 loop: asyncio.AbstractEventLoop) -> None:  # fmt: skip
        self._start = start
        self._end = end
        self._loop = loop
        self.value = value

    def wait(self) -> "asyncio.Future[List[AbstractEvent]]":
        """Future for the device events that must complete before the resource is used."""
        return self._start

    async def wait_events(self) -> None:
        """Wait, on the host, until the previous user's device work has finished."""
        await async_wait_for_events(await self._start, loop=self._loop)

    def ready(self, events: Optional[List[AbstractEvent]] = None) -> None:
        """Release the resource to the next user, who must honour `events` first. Call
        it only once the resource has actually been obtained (after :meth:`wait`)."""
        self._end.set_result(list(events) if events is not None else [])

    def __enter__(self) -> _T:
        return self.value

    def __exit__(self, exc_type, exc_value, exc_tb) -> None:
        if self._end.done():
            return
        if exc_value is not None:
            self._end.set_exception(exc_value)
            self._end.exception()  # mark as retrieved: it propagates from the block too
        else:
            _logger.warning("Resource allocation was not explicitly made ready")
            self.ready()


class Resource(Generic[_T]):
    """A value that one task at a time may use, handed on in acquisition order."""

    def __init__(self, value: _T, loop: Optional[asyncio.AbstractEventLoop] = None) -> None:
        if loop is None:
            loop = asyncio.get_event_loop()
        self._loop = loop
        self._tail: "asyncio.Future[List[AbstractEvent]]" = loop.create_future()
        self._tail.set_result([])
        self.value = value

    def acquire(self) -> ResourceAllocation[_T]:
        """Queue up for the resource (does not wait): see :class:`ResourceAllocation`."""
        previous, self._tail = self._tail, self._loop.create_future()
        return ResourceAllocation(previous, self._tail, self.value, self._loop)


class JobQueue:
    """In-flight asynchronous jobs, oldest first."""

    def __init__(self) -> None:
        self._jobs: Deque["asyncio.Future"] = collections.deque()

    def add(self, job: Awaitable) -> None:
        """Append a job; a coroutine is wrapped in a task."""
        self._jobs.append(asyncio.ensure_future(job))

    def clean(self) -> None:
        """Drop finished jobs from the front, re-raising what they raised."""
        while self._jobs and self._jobs[0].done():
            self._jobs.popleft().result()

    async def finish(self, max_remaining: int = 0) -> None:
        """Await jobs from the front until at most `max_remaining` are left."""
        while len(self._jobs) > max_remaining:
            await self._jobs.popleft()

    def __len__(self) -> int:
        return len(self._jobs)

    def __bool__(self) -> bool:
        return bool(self._jobs)

    def __contains__(self, item) -> bool:
        return item in self._jobs


__all__ = ["wait_until", "async_wait_for_events", "Resource", "ResourceAllocation", "JobQueue"]
