"""`python bench.py --gpus N` without a launcher spawns its own ranks (round-4 verdict: the plain command died on an
assertion at N > 1).  No GPU needed: the child process is mocked, only the command line and the control flow are checked."""
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_command_is_the_drivers_multi_gpu_form():
    args = bench.parse(["--gpus", "8", "--steps", "7", "--warmup", "2"])
    cmd = bench.self_launch_command(args, ["--gpus", "8", "--steps", "7", "--warmup", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]


def test_plain_command_spawns_ranks_when_no_launcher_set_world_size(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "3"])
    assert e.value.code == 0
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_under_a_launcher_nothing_is_spawned(monkeypatch):
    """WORLD_SIZE set = a launcher owns the ranks: main() must go on to the GPU check (which fails here), not re-exec."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("spawned under a launcher"))
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only control-flow test")
    with pytest.raises(AssertionError, match="needs MI355X"):
        bench.main(["--gpus", "2"])


def test_config_and_strong_scaling_flags():
    """`--config` picks a BASELINE configuration, `--global-batch` its strong-scaling form (round-5 verdict: configs 4 / 5 are
    DEFINED as a global batch of 16 / 8 over 8 GPUs); the default stays config 2, weak, 32 clouds per GPU."""
    from deltaconv_amd.configs import CONFIGS
    a = bench.parse([])
    assert (a.config, a.points, a.k, a.strong, a.sync_bn) == ("C2", 1024, 20, False, False)
    assert [bench.per_rank_clouds(a, w) for w in (1, 2, 4, 8)] == [32, 32, 32, 32]
    a = bench.parse(["--config", "C4", "--global-batch", "16"])
    assert (a.points, a.k, a.strong, a.sync_bn) == (2048, 20, True, True)
    assert [bench.per_rank_clouds(a, w) for w in (1, 2, 4, 8)] == [16, 8, 4, 2]
    a = bench.parse(["--config", "C5", "--global-batch", "8", "--no-sync-bn"])
    assert (a.points, a.k, a.sync_bn) == (4096, 30, False) and bench.per_rank_clouds(a, 8) == 1
    with pytest.raises(SystemExit):
        bench.per_rank_clouds(bench.parse(["--global-batch", "12"]), 8)
    assert bench.per_rank_clouds(bench.parse(["--config", "C3", "--batch", "4"]), 2) == 4
    for name, cfg in CONFIGS.items():        # the table is BASELINE.json's: sizes of SURVEY.md section 8
        assert cfg["B"] * cfg["N"] in (32768, 65536) and cfg["k"] in (20, 30)
