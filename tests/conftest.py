import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line(
        "markers", "force_autotune: run the real autotuning search (no cache) in this test"
    )
    config.addinivalue_line(
        "markers", "real_autotuner: leave tune.autotuner_impl (the sqlite cache) in place"
    )


@pytest.fixture(autouse=True)
def patch_autotune(request, monkeypatch):
    """Templates answer ``autotune`` with their ``test=`` configuration under pytest, or
    run the real search without the on-disk cache under ``@pytest.mark.force_autotune``
    (the arrangement of the reference's pytest plugin, pytest_plugin.py:30-35)."""
    from katsdpsigproc_amd import tune

    if request.node.get_closest_marker("force_autotune"):
        monkeypatch.setattr(tune, "autotuner_impl", tune.force_autotuner)
    elif not request.node.get_closest_marker("real_autotuner"):
        monkeypatch.setattr(tune, "autotuner_impl", tune.stub_autotuner)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "rfi_host_golden.npz")
    return np.load(path, allow_pickle=False)
