"""Autotuning of launch geometry with an on-disk cache.

Counterpart of the reference's ``tune.py`` (reference: src/katsdpsigproc/tune.py:75-448)
with the same surface, so templates and tests read the same way:

* ``@autotuner(test={...})`` wraps a ``classmethod autotune(cls, context, *key_args)``.
  The call is answered from a sqlite database keyed on the class, the arguments and the
  device (name, platform, driver version); on a miss the wrapped function runs -- it
  normally calls :func:`autotune` -- and its result is stored.
* :func:`autotune` times every combination of the candidate parameters it is given and
  returns the fastest; :func:`make_measure` turns "enqueue this operation" into the
  scoring function it wants, timed by a tuning command queue (HIP events).
* ``KATSDPSIGPROC_TUNE_DB`` names the database file (default: the user cache directory),
  ``KATSDPSIGPROC_TUNE_MATCH=nearest`` accepts an entry of another driver version /
  platform / device when there is no exact one.
* :func:`stub_autotuner` / :func:`force_autotuner` are drop-in replacements of
  :func:`autotuner_impl` for tests (return the ``test=`` value / tune without the cache);
  the wrapper looks ``autotuner_impl`` up at call time so that patching it works.

What there is to tune on MI355X: the kernels fix their wavefront-level geometry at
compile time, the launchers expose the remaining choices (how a baseline's channels are
cut into segments for the median filter, values per thread for SumThreshold) as
arguments, and the templates' ``autotune`` methods search those.
"""

import concurrent.futures
import enum
import functools
import inspect
import itertools
import logging
import os
import sqlite3
import time
from typing import Any, Callable, Dict, Mapping, Optional, Sequence

import numpy as np

_logger = logging.getLogger(__name__)


def _match_mode() -> str:
    mode = os.environ.get("KATSDPSIGPROC_TUNE_MATCH", "exact")
    if mode not in ("exact", "nearest"):
        _logger.debug("KATSDPSIGPROC_TUNE_MATCH=%r is neither 'exact' nor 'nearest'", mode)
        mode = "exact"
    return mode


def adapt_value(value: Any) -> Any:
    """A lookup-key value in a form sqlite can store (types, dtypes and enums by name)."""
    if isinstance(value, (type, np.dtype)):
        return repr(value)
    if isinstance(value, enum.Enum):
        return value.name
    if isinstance(value, (bool, np.bool_)):
        return int(value)
    return value


# ------------------------------------------------------------------------ the cache
_DEVICE_KEYS = ("device_version", "device_platform", "device_name")  # dropped in this order


def _key_columns(fn: Callable, args: Sequence, kwargs: Mapping) -> Dict[str, Any]:
    """Database key of one ``autotune(cls, context, ...)`` call: every argument after the
    context by name, plus the identity of the context's device."""
    bound = inspect.signature(fn).bind(*args, **kwargs)
    bound.apply_defaults()
    names = list(bound.arguments)
    columns = {"arg_" + name: adapt_value(bound.arguments[name]) for name in names[2:]}
    device = bound.arguments[names[1]].device
    columns["device_name"] = device.name
    columns["device_platform"] = device.platform_name
    columns["device_version"] = device.driver_version
    return columns


def _select(conn: sqlite3.Connection, table: str, columns: Mapping[str, Any]):
    where = " AND ".join(f"{name}=?" for name in columns)
    sql = f"SELECT * FROM {table}" + (f" WHERE {where}" if where else "")
    return conn.execute(sql, list(columns.values())).fetchone()


def _fetch(conn: sqlite3.Connection, table: str, columns: Mapping[str, Any]):
    """The stored result for `columns` (``value_*`` columns, prefix removed) or None."""
    attempts = [dict(columns)]
    if _match_mode() == "nearest":
        relaxed = dict(columns)
        for name in _DEVICE_KEYS:
            relaxed.pop(name, None)
            attempts.append(dict(relaxed))
    for attempt in attempts:
        try:
            row = _select(conn, table, attempt)
        except sqlite3.Error:  # e.g. the table does not exist yet
            _logger.debug("tuning query failed", exc_info=True)
            return None
        if row is not None:
            return {k[len("value_"):]: row[k] for k in row.keys() if k.startswith("value_")}
    return None


def _save(conn: sqlite3.Connection, table: str, columns: Mapping[str, Any],
          values: Mapping[str, Any]) -> None:  # fmt: skip
    names = list(columns) + list(values)
    conn.execute(
        f"CREATE TABLE IF NOT EXISTS {table} ("
        + ", ".join(f"{n} NOT NULL" for n in names)
        + f", PRIMARY KEY ({', '.join(columns)}) ON CONFLICT REPLACE)"
    )
    with conn:
        conn.execute(
            f"INSERT OR REPLACE INTO {table}({', '.join(names)}) "
            f"VALUES ({', '.join('?' for _ in names)})",
            list(columns.values()) + list(values.values()),
        )


def _open_db() -> sqlite3.Connection:
    path = os.environ.get("KATSDPSIGPROC_TUNE_DB")
    if path is None:
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        directory = os.path.join(base, "katsdpsigproc_amd")
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, "tuning.db")
    return sqlite3.connect(path)


def _close_db(conn: sqlite3.Connection) -> None:
    conn.close()  # a function of its own so that tests can patch it


def autotuner_impl(test: Mapping[str, Any], fn: Callable, *args: Any, **kwargs: Any):
    """Answer an autotune call from the cache, running and recording it on a miss."""
    cls = args[0]
    name = f"{cls.__module__}.{cls.__name__}.{fn.__name__}"
    table = name.replace(".", "_") + "__" + str(getattr(cls, "autotune_version", 0))
    columns = _key_columns(fn, args, kwargs)
    try:
        conn = _open_db()
        conn.row_factory = sqlite3.Row
    except (OSError, sqlite3.Error) as exc:
        # an unwritable cache directory must not stop the computation: tune, do not record
        _logger.warning("tuning cache unavailable (%s): tuning %s without it", exc, name)
        return fn(*args, **kwargs)
    try:
        result = _fetch(conn, table, columns)
        if result is None:
            _logger.info("autotuning %s for %s", name, columns)
            result = fn(*args, **kwargs)
            try:
                _save(conn, table, columns, {"value_" + k: v for k, v in result.items()})
            except sqlite3.Error as exc:  # read-only or locked database: keep the result
                _logger.warning("could not record the tuning of %s: %s", name, exc)
        else:
            _logger.debug("tuning cache hit for %s %s", name, columns)
    finally:
        _close_db(conn)
    return result


def force_autotuner(test: Mapping[str, Any], fn: Callable, *args: Any, **kwargs: Any):
    """Stand-in for :func:`autotuner_impl`: always tune, never touch the cache."""
    return fn(*args, **kwargs)


def stub_autotuner(test: Mapping[str, Any], fn: Callable, *args: Any, **kwargs: Any):
    """Stand-in for :func:`autotuner_impl`: return the ``test=`` value, tune nothing."""
    return test


def autotuner(test: Mapping[str, Any]) -> Callable:
    """Make ``autotune(cls, context, *named key arguments)`` a cached tuning function.

    `test` is what :func:`stub_autotuner` answers instead (a valid, not necessarily fast,
    configuration).
    """

    def decorate(fn: Callable) -> Callable:
        @functools.wraps(fn)
        def wrapper(*args: Any, **kwargs: Any):
            return autotuner_impl(dict(test), fn, *args, **kwargs)  # looked up at call time

        wrapper.test = dict(test)
        return wrapper

    return decorate


# ---------------------------------------------------------------------- the search
def make_measure(queue, function: Callable[[], None]) -> Callable[[int], float]:
    """Scoring function for :func:`autotune`: seconds per call of `function`, measured by
    the tuning command queue `queue` over the requested number of calls."""

    def measure(iters: int) -> float:
        queue.start_tuning()
        for _ in range(iters):
            function()
        return queue.stop_tuning() / iters

    return measure


def autotune(generate: Callable[..., Optional[Callable[[int], float]]], time_limit: float = 0.1,
             threads: Optional[int] = None, **kwargs: Any) -> Mapping[str, Any]:  # fmt: skip
    """Try every combination of the iterables in `kwargs` and return the best one.

    ``generate(**combination)`` builds the candidate and returns its scoring function
    (``score = f(iterations)``, lower is better) or None to skip it; candidates are built
    `threads` at a time (building may compile), scored one after another: a warm-up
    call, one timed call to size the run, then as many iterations as fit `time_limit`.
    A candidate that raises is skipped; if none survives the last exception is raised
    (``ValueError`` if there was nothing to try).
    """
    names = list(kwargs)
    combos = [dict(zip(names, values)) for values in itertools.product(*kwargs.values())]
    if threads is None:
        threads = os.cpu_count() or 1
    best: Optional[Mapping[str, Any]] = None
    best_score = float("inf")
    failure: Optional[BaseException] = None
    with concurrent.futures.ThreadPoolExecutor(max(1, threads)) as pool:
        for first in range(0, len(combos), max(1, threads)):
            batch = combos[first : first + max(1, threads)]
            built = [pool.submit(generate, **combo) for combo in batch]
            for combo, future in zip(batch, built):
                try:
                    measure = future.result()
                    if measure is None:
                        continue
                    measure(1)  # warm-up
                    t0 = time.perf_counter()
                    measure(1)
                    once = max(time.perf_counter() - t0, 1e-4)
                    iters = max(3, int(time_limit / once))
                    score = measure(iters)
                    _logger.debug("%s: %g s per call over %d calls", combo, score, iters)
                    if score < best_score:
                        best, best_score = combo, score
                except Exception as exc:  # a candidate that cannot run is not a candidate
                    failure = exc
                    _logger.debug("candidate %s failed", combo, exc_info=True)
    if best is None:
        if failure is not None:
            raise failure
        raise ValueError("No options to test")
    return best


def fixed_geometry(what: str, tuning, reference_keys):
    """`tuning=` of an operation whose gfx950 kernel has ONE launch geometry.

    The reference tunes workgroup shapes of kernels it generates from source (`reference_keys`
    names its parameters); the hand-written kernel behind `what` has nothing to choose. A
    caller's values for the reference's keys are therefore accepted -- call sites keep working --
    and have no effect; any other key is a mistake and raises ``ValueError``. Returns the
    mapping to store as ``template.tuning``: what the caller passed, or ``{}`` (nothing was
    tuned, nothing is cached)."""
    if tuning is None:
        return {}
    unknown = sorted(set(tuning) - set(reference_keys))
    if unknown:
        raise ValueError(
            f"{what}: unknown tuning parameter(s) {unknown}. Its kernel has a fixed geometry; only "
            f"the reference's {sorted(reference_keys)} are accepted, without effect")
    return dict(tuning)
