"""bench.py started the way the driver starts it for N > 1 -- `python bench.py --gpus N`, no launcher -- has to
start its N ranks itself, as fresh child processes, before anything touches HIP (CPU test of the launcher logic;
the ranks themselves need GPUs), and the C ABI's query slicing has to be the partitioner's."""
import os
import subprocess
import sys

import pytest

import sdf_amd as S
from sdf_amd import partition as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("bench")


def test_gpus_n_without_a_launcher_starts_the_ranks(monkeypatch):
    bench = _load_bench(monkeypatch)
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ei:
        bench._self_launch_before_hip()
    assert ei.value.code == 7                       # the children's status is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]   # the ranks see the same arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus=2"])
    with pytest.raises(SystemExit):
        bench._self_launch_before_hip()
    assert "--nproc-per-node=2" in seen["cmd"]


def test_no_self_launch_under_a_launcher_or_on_one_gpu(monkeypatch):
    bench = _load_bench(monkeypatch)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("must not start child processes"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    bench._self_launch_before_hip()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench._self_launch_before_hip()
    monkeypatch.setenv("WORLD_SIZE", "8")            # torchrun's environment: this process IS a rank
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    bench._self_launch_before_hip()


def test_the_launcher_runs_before_torch_is_imported():
    """Textual guard: nothing that can initialise HIP is imported above the self-launch call."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("_self_launch_before_hip()\n", src.index("def _self_launch_before_hip"))]
    for forbidden in ("import torch", "import sdf_amd", "import numpy"):
        assert forbidden not in head.replace("torch.distributed.run", "")


def test_query_slice_of_the_c_abi_is_the_partitioners():
    for n in (0, 1, 7, 64, 1000, 10_000_019):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                lo, cnt = S.query_slice(n, r, world)
                assert (lo, lo + cnt) == P.query_slice(r, world, n)
                assert lo == covered
                covered += cnt
            assert covered == n
    assert S.query_slice(10, 5, 4) == (0, 0)        # a rank outside the world owns nothing
