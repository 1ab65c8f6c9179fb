"""bench.py through the driver's launcher (`python -m torch.distributed.run ... bench.py --gpus N`) on the
one GPU of the test box, with --force-dist: RCCL process group, barrier, flat gradient all-reduce and the
JSON contract are all exercised (with a single rank)."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def test_bench_under_torchrun_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "2", "--no-cpu-baseline", "--force-dist"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1


def test_graph_replay_with_gradient_reducer_and_sync_bn():
    """tests/dist/gpu_worker.py under the launcher: (1) HIP-graph step + reduce_gradients() + optimizer for 3 steps is
    bit-identical to eager; (2) the synchronised-BatchNorm path (split statistics kernels, all-reduced fp64 sums)
    agrees with per-rank statistics on one rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist", "gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["graph_dp_mismatches"] == []
    assert d["two_graph_dp_mismatches"] == []
    assert d["sync_graph_mismatches"] == []          # sync_bn step with its collectives captured in one graph == eager
    # (gradients: the synchronised path runs the composed blocks, the per-rank path the fused layer nodes; with
    # max-aggregation a near-tie can pick another neighbour under different fp32 rounding -- measured 9e-3)
    assert d["sync_logits_err"] < 1e-4 and d["sync_grad_err"] < 3e-2 and d["sync_running_err"] < 1e-5, d


def test_sync_bn_world2_on_one_gpu_equals_single_process():
    """Two ranks (gloo, both on the one GPU of the test box), each with half of the batch and sync_bn=True: the HIP
    synchronised-BatchNorm path (split statistics kernels, all-reduced fp64 sums, gradients averaged by the flat
    all-reduce) reproduces ONE process on the full batch -- /root/reference/deltaconv/nn/nonlin.py:24-35 semantics of the
    global batch: logits 2e-4, gradients (L1 per parameter) 5e-3, running statistics 1e-5; the replicas stay bit-equal."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tests", "dist", "gpu_worker2.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["logits_err"] < 2e-4 and d["grad_l1_err"] < 5e-3 and d["running_err"] < 1e-5, d
    assert d["replicas_equal"] and d["params_after_step_err"] < 2e-2, d     # (one SGD step from zero-initialised biases: = the gradient error)


def test_plain_command_self_launches_its_ranks():
    """`python bench.py --gpus N` without a launcher spawns its own ranks (round-4 verdict: it died on an assertion).  One GPU
    here, so the same path is taken with `--gpus 1 --spawn`: the child runs under torch.distributed.run with WORLD_SIZE = 1 and
    rank 0 prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-exact-chain", "--force-dist"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"


@pytest.mark.parametrize("config,gb", [("C4", 2), ("C5", 2)])
def test_strong_scaling_form_under_the_launcher(config, gb):
    """`bench.py --config C4|C5 --global-batch B` (round 6: BASELINE configs 4 / 5 are a global batch over 8 GPUs) through the
    launcher with one rank and --force-dist: global-batch / world clouds per rank, "scaling": "strong", BatchNorm statistics
    over the global batch (the step is ONE captured graph with its collectives), the segmentation nets, SGD / Adam."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29561" if config == "C4" else "29562", os.path.join(ROOT, "bench.py"), "--gpus", "1",
           "--config", config, "--global-batch", str(gb), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-exact-chain",
           "--force-dist"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["config"]["name"] == config and d["config"]["global_batch"] == gb and d["value"] > 0
    assert "BatchNorm statistics over the global batch" in d["config"]["workload"]
    assert "one HIP graph" in d["config"]["workload"], d["config"]["workload"]
