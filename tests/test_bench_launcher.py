"""bench.py's own rank launcher (`python bench.py --gpus N` outside torch.distributed.run): host logic only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


def test_more_gpus_than_visible_fails_loudly():
    """Fewer visible GPUs than --gpus: a non-zero exit and a message, never a silent n_gpus = 1 line."""
    import torch
    n = torch.cuda.device_count() + 1
    if n == 1:
        n = 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSW_BENCH_SAME_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_gpus_flag_must_match_world_size():
    """Under torch.distributed.run the two must agree (the driver passes both)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_launcher_starts_one_child_per_rank(tmp_path, monkeypatch):
    """launch_ranks: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's stdout relayed,
    a failing rank turns into a non-zero exit."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    class FakeProc:
        def __init__(self, argv, env=None, stdout=None):
            self.rank = int(env["RANK"])
            calls.append((argv, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
            self.returncode = 0 if self.rank != fail_rank[0] else 3

        def communicate(self):
            return (b'{"n_gpus": 3}\n', None)

        def wait(self, timeout=None):
            return self.returncode

        def kill(self):
            pass

    fail_rank = [-1]
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setenv("HSW_BENCH_SAME_DEVICE", "1")          # skip the visible-GPU check (none here)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)

    class A:
        gpus = 3
    assert bench.launch_ranks(A, ["--gpus", "3", "--steps", "2"]) == 0
    assert [c[1]["RANK"] for c in calls] == ["0", "1", "2"]
    assert all(c[1]["WORLD_SIZE"] == "3" and c[1]["MASTER_ADDR"] == "127.0.0.1" for c in calls)
    assert len({c[1]["MASTER_PORT"] for c in calls}) == 1
    assert all(c[0][-4:] == ["--gpus", "3", "--steps", "2"] for c in calls)
    calls.clear()
    fail_rank[0] = 1
    assert bench.launch_ranks(A, ["--gpus", "3"]) != 0


def test_gpus_8_with_one_visible_gpu_exits_2_with_the_message(monkeypatch, capsys):
    """The driver's 8-GPU run on a box that shows one GPU must not degrade to a 1-GPU line: exit code 2 and a
    message, no rank started."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    started = []
    monkeypatch.setattr(bench.subprocess, "Popen", lambda *a, **k: started.append(a))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("HSW_BENCH_SAME_DEVICE", raising=False)

    class A:
        gpus = 8
    assert bench.launch_ranks(A, ["--gpus", "8"]) == 2
    assert started == []
    assert "--gpus 8 asked for, but only 1 GPU(s) visible" in capsys.readouterr().err
