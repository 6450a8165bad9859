"""GPU tests (-m gpu), round 6: data parallelism of the ResNet-50 backbone swap (BASELINE.json
configs[4] names 8 GPUs; src/pascal/pascal_train_resnet.py:37-50, slim clones src/slim_dir/deployment/model_deploy.py:
222-225,436-446).  Two rank processes share cuda:0 (gloo moves the device tensors; RCCL refuses two ranks on one device):
tests/dp_resnet_worker.py states what is checked.  Then `bench.py --model resnet50 --gpus 2` as the driver would start it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "Y2_FORCE_DIST"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("dtype,strategy", [("f16", "allreduce"), ("f16", "rs_ag"), ("f32", "allreduce")])
def test_resnet_swap_two_ranks_on_one_gpu(dtype, strategy):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(_env(), Y2_TEST_DTYPE=dtype, Y2_DP_STRATEGY=strategy, Y2_NO_WGRAD_OVERLAP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_resnet_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "resnet dp2 ok " + dtype in r.stdout, r.stdout[-2000:]
    if dtype == "f16":
        assert "fused_fc1=True" in r.stdout         # the operand all-gather path is what ran


def test_bench_resnet50_two_ranks_child_tree_on_one_gpu():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--model", "resnet50", "--gpus", "2", "--all-ranks-on-gpu0",
           "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "8"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert "all-gather" in d["config"]["grad_exchange"]["yolo_fc1"]
    assert d["steps_applied"] == 2 and d["steps_skipped_by_overflow_guard"] == 0
    assert np.isfinite(d["value"]) and abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
