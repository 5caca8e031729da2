"""bench.py must be runnable as `python bench.py --gpus N`: without a launcher's
environment it starts the N ranks itself and relays rank 0's JSON line (no GPU needed
to check that: the child process is faked)."""

import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KSP_BENCH_FORCE_LAUNCH"):
        monkeypatch.delenv(var, raising=False)
    import bench as module

    return module


def test_self_launch_command_and_relay(bench, monkeypatch, capsys):
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"] = cmd
        seen["env"] = env
        line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 4})
        return types.SimpleNamespace(returncode=0, stdout="NCCL banner\n" + line + "\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    assert bench.main() == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1 :] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    lines = [x for x in out.out.splitlines() if x.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 4  # exactly one JSON line
    assert "NCCL banner" in out.err


def test_self_launch_propagates_failure(bench, monkeypatch, capsys):
    monkeypatch.setattr(
        subprocess, "run",
        lambda cmd, env=None, stdout=None, text=None: types.SimpleNamespace(returncode=3, stdout=""),
    )
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    assert bench.main() == 3
    assert capsys.readouterr().out == ""


def test_rank_mismatch_is_an_error(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit):
        bench.main()


def test_inject_rfi_is_deterministic_and_sparse(bench):
    import numpy as np

    a = bench.inject_rfi(bench.synth_block(64, 512, 1), seed=3)
    b = bench.inject_rfi(bench.synth_block(64, 512, 1), seed=3)
    assert np.array_equal(a, b)
    strong = np.abs(a) > 40
    assert 0.04 < strong.mean() < 0.09
