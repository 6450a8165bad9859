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


def test_forward_graph_replays_the_detection_forward():
    """engine.ForwardGraph: pascal_detect_darknet.py's forward (core with moving statistics, head with batch statistics,
    one image) captured into ONE HIP graph -- replays on changing inputs give the bits of the eager launches, float32
    and uint8 inputs, and leave the moving statistics alone."""
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E, synthetic
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 4)]
    head = [(3, core[-1][2], 256, 0), (1, 256, 30, 0)]
    size = 224
    for dtype in ("f16", "f32"):
        net = E.Network(core + head, 1, size, size, dtype=dtype, core_layers=len(core), training=False)
        net.init_params(3)
        state0 = net.state.clone()
        g = net.forward_graph(False, True)
        gu = net.forward_graph(False, True, uint8=True)
        rng = np.random.default_rng(0)
        for i in range(3):
            x = torch.as_tensor(synthetic.images(1, size, 50 + i)).cuda()
            want = net.forward(x, False, True).clone()
            got = g(x)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (dtype, i)
            u = torch.as_tensor(rng.integers(0, 256, (1, size, size, 3), dtype=np.uint8)).cuda()
            want_u = net.forward(u, False, True).clone()
            got_u = gu(u)
            torch.cuda.synchronize()
            assert torch.equal(got_u, want_u), (dtype, i, "uint8")
        assert torch.equal(net.state, state0)


def test_forward_graph_follows_a_parameter_reload():
    """a ForwardGraph replays the filter packs of its capture; load_params / init_params re-pack only at the
    next eager forward.  The replay now refreshes the packs first: a graph captured on one set of parameters gives the
    eager bits of ANOTHER set loaded afterwards, with no eager forward in between."""
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E, synthetic
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 4)]
    head = [(3, core[-1][2], 256, 0), (1, 256, 30, 0)]
    size = 224
    net = E.Network(core + head, 1, size, size, dtype="f16", core_layers=len(core), training=False)
    net.init_params(3)
    g = net.forward_graph(False, True)
    x = torch.as_tensor(synthetic.images(1, size, 60)).cuda()
    first = g(x).clone()
    torch.cuda.synchronize()
    other = E.Network(core + head, 1, size, size, dtype="f16", core_layers=len(core), training=False)
    other.init_params(4)
    net.load_params(other.export_params())
    got = g(x).clone()                       # no eager forward since the reload
    torch.cuda.synchronize()
    want = other.forward(x, False, True)
    torch.cuda.synchronize()
    assert not torch.equal(first, want)
    assert torch.equal(got, want)
    got2 = g(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(got2, want)
