"""The autotuner: the search (:func:`tune.autotune`) and the sqlite cache behind
``@tune.autotuner`` -- behaviours of reference test/test_tune.py:30-214, no GPU."""

import sqlite3
import threading
import types

import pytest

from katsdpsigproc_amd import tune

pytestmark = pytest.mark.real_autotuner


# ------------------------------------------------------------------------ the search
def test_search_tries_every_combination_and_picks_the_lowest_score():
    seen, lock = [], threading.Lock()

    def generate(a, b):
        with lock:
            seen.append((a, b))
        return lambda iters: a * b

    best = tune.autotune(generate, time_limit=0.001, a=[1, 2], b=[7, 3])
    assert sorted(seen) == [(1, 3), (1, 7), (2, 3), (2, 7)]
    assert best == {"a": 1, "b": 3}


def test_search_passes_iteration_counts():
    calls = []
    tune.autotune(lambda x: lambda iters: calls.append(iters) or 1.0, time_limit=0.001, x=[5])
    assert calls[0] == 1 and calls[1] == 1 and calls[2] >= 3  # warm-up, sizing, scored run


def test_search_with_nothing_to_try():
    with pytest.raises(ValueError):
        tune.autotune(lambda x, y: lambda iters: 0, x=[1, 2], y=[])


class Broken(RuntimeError):
    pass


def test_search_skips_candidates_that_fail_or_decline():
    def generate(x):
        if x == 1:
            raise Broken("x = 1")
        if x == 4:
            return None  # known to be unsuitable

        def measure(iters):
            if x == 3:
                raise Broken("x = 3")
            return -x

        return measure

    assert tune.autotune(generate, x=[0, 1, 2, 3, 4]) == {"x": 2}


def always_fails(x):
    raise Broken(f"x = {x}")


def test_search_reraises_the_last_failure_when_nothing_runs():
    with pytest.raises(Broken, match=r"^x = 3$") as info:
        tune.autotune(always_fails, x=[1, 2, 3])
    assert info.traceback[-1].name == "always_fails"  # the original site, not a re-raise


def test_make_measure_averages_over_the_tuning_queue():
    class Queue:
        def __init__(self):
            self.calls = 0

        def start_tuning(self):
            self.calls = 0

        def stop_tuning(self):
            return 0.5 * self.calls

    queue = Queue()

    def enqueue():
        queue.calls += 1

    assert tune.make_measure(queue, enqueue)(8) == pytest.approx(0.5)


# ------------------------------------------------------------------------- the cache
def make_context(name="mock device", platform="mock platform", version="mock version"):
    device = types.SimpleNamespace(name=name, platform_name=platform, driver_version=version)
    return types.SimpleNamespace(device=device)


class Tuned:
    """A class with autotune methods whose "tuning" is scripted by the test."""

    autotune_version = 3
    calls = []
    answer = {}
    fail = False

    @classmethod
    @tune.autotuner(test={"a": 3, "b": -1})
    def autotune(cls, context, param, flag=True):
        if cls.fail:
            raise RuntimeError("tuning was not expected to run")
        cls.calls.append((context.device.name, param, flag))
        return dict(cls.answer)

    @classmethod
    @tune.autotuner(test={"a": 3, "b": -1})
    def autotune_no_args(cls, context):
        cls.calls.append((context.device.name,))
        return dict(cls.answer)


@pytest.fixture
def database(monkeypatch):
    """One in-memory database that survives the close after every lookup."""
    conn = sqlite3.connect(":memory:")
    monkeypatch.setattr(tune, "_open_db", lambda: conn)
    monkeypatch.setattr(tune, "_close_db", lambda c: None)
    monkeypatch.delenv("KATSDPSIGPROC_TUNE_MATCH", raising=False)
    Tuned.calls, Tuned.answer, Tuned.fail = [], {}, False
    yield conn
    conn.close()


def test_result_is_cached_per_key_and_device(database):
    first = {"a": 1, "b": 2}
    Tuned.answer = first
    ctx = make_context()
    assert Tuned.autotune(ctx, "xyz") == first
    assert Tuned.autotune(ctx, "xyz") == first
    assert Tuned.calls == [("mock device", "xyz", True)]  # tuned once
    # another argument value, another default, another device: each tunes again
    Tuned.answer = {"a": 5, "b": 6}
    assert Tuned.autotune(ctx, "zzz") == {"a": 5, "b": 6}
    assert Tuned.autotune(ctx, "xyz", flag=False) == {"a": 5, "b": 6}
    other = make_context("another device", "another platform", "another version")
    Tuned.answer = {"a": 3, "b": 4}
    assert Tuned.autotune(other, "xyz") == {"a": 3, "b": 4}
    assert len(Tuned.calls) == 4
    assert Tuned.autotune(ctx, "xyz") == first  # and the first entry is still there
    # the table name carries the class, the method and autotune_version
    tables = [r[0] for r in database.execute("SELECT name FROM sqlite_master WHERE type='table'")]
    assert tables == ["tests_test_tune_Tuned_autotune__3"]


def test_nearest_match_relaxes_driver_then_platform_then_device(database, monkeypatch):
    Tuned.answer = one = {"a": 1, "b": 2}
    Tuned.autotune(make_context(), "xyz")
    Tuned.answer = two = {"a": 3, "b": 4}
    Tuned.autotune(make_context("another device", "another platform", "another version"), "xyz")
    Tuned.fail = True
    with pytest.raises(RuntimeError):  # exact matching: a new driver version tunes again
        Tuned.autotune(make_context(version="abc"), "xyz")
    monkeypatch.setenv("KATSDPSIGPROC_TUNE_MATCH", "nearest")
    assert Tuned.autotune(make_context(version="abc"), "xyz") == one
    assert Tuned.autotune(make_context("another device", "abc", "abc"), "xyz") == two
    assert Tuned.autotune(make_context(), "xyz") == one
    assert Tuned.autotune(make_context("x", "y", "z"), "xyz") in (one, two)  # any will do
    with pytest.raises(RuntimeError):  # but never one recorded for other arguments
        Tuned.autotune(make_context("x", "y", "z"), "other")
    monkeypatch.setenv("KATSDPSIGPROC_TUNE_MATCH", "bogus")  # falls back to exact
    with pytest.raises(RuntimeError):
        Tuned.autotune(make_context(version="abc"), "xyz")


def test_method_without_key_arguments(database):
    Tuned.answer = {"a": 1, "b": 2}
    ctx = make_context()
    assert Tuned.autotune_no_args(ctx) == {"a": 1, "b": 2}
    assert Tuned.autotune_no_args(ctx) == {"a": 1, "b": 2}
    assert Tuned.calls == [("mock device",)]


def test_database_file_from_environment(tmp_path, monkeypatch):
    path = tmp_path / "sub" / "tuning.db"
    path.parent.mkdir()
    monkeypatch.setenv("KATSDPSIGPROC_TUNE_DB", str(path))
    Tuned.calls, Tuned.answer, Tuned.fail = [], {"a": 9, "b": 8}, False
    assert Tuned.autotune(make_context(), "p") == {"a": 9, "b": 8}
    assert path.exists()
    Tuned.fail = True  # a second process would find it on disk
    assert Tuned.autotune(make_context(), "p") == {"a": 9, "b": 8}


def test_unusable_cache_does_not_stop_tuning(tmp_path, monkeypatch, caplog):
    Tuned.calls, Tuned.answer, Tuned.fail = [], {"a": 2, "b": 3}, False
    monkeypatch.setenv("KATSDPSIGPROC_TUNE_DB", str(tmp_path / "no" / "such" / "dir" / "t.db"))
    assert Tuned.autotune(make_context(), "p") == {"a": 2, "b": 3}  # tuned, not recorded
    assert Tuned.autotune(make_context(), "p") == {"a": 2, "b": 3}
    assert len(Tuned.calls) == 2
    # a database that can be read but not written
    path = tmp_path / "ro.db"
    monkeypatch.setenv("KATSDPSIGPROC_TUNE_DB", str(path))
    sqlite3.connect(str(path)).close()
    path.chmod(0o444)
    import os
    if os.geteuid() != 0:  # (root writes through the mode bits)
        assert Tuned.autotune(make_context(), "q") == {"a": 2, "b": 3}


def test_key_values_are_adapted_for_sqlite():
    import enum

    import numpy as np

    class Colour(enum.Enum):
        RED = 1

    assert tune.adapt_value(Colour.RED) == "RED"
    assert tune.adapt_value(np.dtype(np.float32)) == repr(np.dtype(np.float32))
    assert tune.adapt_value(np.float32) == repr(np.float32)
    assert tune.adapt_value(True) == 1 and tune.adapt_value(7) == 7


def test_stub_and_force_replacements():
    Tuned.calls, Tuned.answer, Tuned.fail = [], {"a": 1, "b": 1}, False
    raw = Tuned.autotune.__wrapped__
    ctx = make_context()
    assert tune.stub_autotuner({"a": 3, "b": -1}, raw, Tuned, ctx, "p") == {"a": 3, "b": -1}
    assert Tuned.calls == []
    assert tune.force_autotuner({"a": 3, "b": -1}, raw, Tuned, ctx, "p") == {"a": 1, "b": 1}
    assert Tuned.autotune.test == {"a": 3, "b": -1}
