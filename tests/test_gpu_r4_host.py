"""GPU tests (-m gpu) of round 4's host-side items (VERDICT r3 "next" 5-6)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_backend_runs_the_sliced_allreduce_path():
    """torch.distributed backend "nccl" (= RCCL) at world size 1 in a fresh child process, Y2_FORCE_DIST=1: the
    multi-GPU call sequence (backward marks, communication stream, both collective strategies, optimizer with
    grad_mult = 1 / world) gives the same bits as the single-process fused train_op (tests/rccl_worker.py).
    The world-size-2 semantics run under gloo (tests/test_dp_gloo.py, test_grad_reducer_two_ranks_on_one_gpu)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("Y2_FORCE_DIST", "Y2_DP_STRATEGY", "Y2_DP_CUTS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "rccl world-1 ok" in r.stdout
