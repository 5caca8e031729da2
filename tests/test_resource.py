"""resource.py: ordering of acquisitions, event hand-off, error propagation, JobQueue,
wait_until (behaviours of reference test/test_resource.py:73-209; no device needed)."""

import asyncio
import logging

import pytest

from katsdpsigproc_amd import resource


class DummyEvent:
    """A device event that is complete once `complete()` has been called."""

    def __init__(self):
        self._done = asyncio.get_event_loop().create_future()
        self.waits = 0

    def complete(self):
        self._done.get_loop().call_soon_threadsafe(self._done.set_result, None)

    def wait(self):  # called from an executor thread
        self.waits += 1
        fut = asyncio.run_coroutine_threadsafe(self._wait(), self._done.get_loop())
        fut.result()

    async def _wait(self):
        await self._done


def run(coro):
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    try:
        return loop.run_until_complete(coro)
    finally:
        loop.close()
        asyncio.set_event_loop(None)


def test_acquisitions_are_served_in_order_with_events():
    async def scenario():
        res = resource.Resource("buffer")
        log = []
        e0 = DummyEvent()

        async def user(name, alloc, release_events):
            with alloc as value:
                events = await alloc.wait()
                log.append((name, value, list(events)))
                alloc.ready(release_events)

        a0, a1, a2 = res.acquire(), res.acquire(), res.acquire()
        # started out of order on purpose: the order of acquire() calls decides
        t2 = asyncio.ensure_future(user("third", a2, []))
        t1 = asyncio.ensure_future(user("second", a1, None))
        await asyncio.sleep(0)
        assert log == []  # nobody may run before the first user is done
        await user("first", a0, [e0])
        await asyncio.gather(t1, t2)
        assert [x[0] for x in log] == ["first", "second", "third"]
        assert log[0][2] == [] and log[1][2] == [e0] and log[2][2] == []
        assert all(x[1] == "buffer" for x in log)

    run(scenario())


def test_wait_events_blocks_on_the_previous_users_device_work():
    async def scenario():
        res = resource.Resource(None)
        first, second = res.acquire(), res.acquire()
        event = DummyEvent()
        await first.wait()
        first.ready([event])
        waiter = asyncio.ensure_future(second.wait_events())
        await asyncio.sleep(0.05)
        assert not waiter.done()
        event.complete()
        await asyncio.wait_for(waiter, 5)
        assert event.waits == 1
        await resource.async_wait_for_events([])  # nothing to wait for: returns at once

    run(scenario())


def test_exception_in_a_user_reaches_the_next_one(caplog):
    async def scenario():
        res = resource.Resource(7)
        first, second, third = res.acquire(), res.acquire(), res.acquire()
        with pytest.raises(ValueError):
            with first:
                await first.wait()
                raise ValueError("boom")
        with pytest.raises(ValueError, match="boom"):
            await second.wait()
        # leaving the block without ready(): released with a warning
        with caplog.at_level(logging.WARNING, logger="katsdpsigproc_amd.resource"):
            with second:
                pass
        assert "not explicitly made ready" in caplog.text
        assert await third.wait() == []

    run(scenario())


def test_job_queue():
    async def scenario():
        queue = resource.JobQueue()
        gates = [asyncio.get_event_loop().create_future() for _ in range(3)]

        async def job(i):
            await gates[i]
            if i == 1:
                raise RuntimeError("job 1 failed")
            return i

        for i in range(3):
            queue.add(job(i))
        assert len(queue) == 3 and queue
        queue.clean()
        assert len(queue) == 3  # nothing finished yet
        gates[0].set_result(None)
        await asyncio.sleep(0)
        await asyncio.sleep(0)
        queue.clean()
        assert len(queue) == 2
        gates[1].set_result(None)
        gates[2].set_result(None)
        with pytest.raises(RuntimeError, match="job 1 failed"):
            await queue.finish(max_remaining=1)
        assert len(queue) == 1
        await queue.finish()
        assert not queue and len(queue) == 0

    run(scenario())


def test_wait_until():
    async def scenario():
        loop = asyncio.get_event_loop()
        fast = loop.create_future()
        loop.call_later(0.01, fast.set_result, 42)
        assert await resource.wait_until(fast, loop.time() + 5) == 42
        slow = loop.create_future()
        with pytest.raises(asyncio.TimeoutError):
            await resource.wait_until(slow, loop.time() + 0.02)
        assert slow.cancelled()

        async def fails():
            raise KeyError("x")

        with pytest.raises(KeyError):
            await resource.wait_until(fails(), loop.time() + 5)

    run(scenario())
