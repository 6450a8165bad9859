"""world_size-2 data-parallel semantics on CPU (gloo): the flat gradient buffer is SUM
all-reduced in layer slices and scaled by 1/world in the optimizer -- slim's clone
semantics (src/slim_dir/deployment/model_deploy.py:222-225,436-446)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nn_ref as R, optim_ref as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SPEC = [(3, 32, 32, False), (1, 32, 64, True), (3, 64, 32, False)]


def _flat(grads):
    return np.concatenate([g[k].ravel() for g in grads for k in ("W", "b", "gamma", "beta")]).astype(np.float32)


def _rank_grads(rank):
    rng = np.random.default_rng(100 + rank)           # per-rank data shard
    params = R.init_params(SPEC, seed=0)              # identical replicas
    x = rng.uniform(-1, 1, (2, 6, 6, 32)).astype(np.float32)
    out, caches, _ = R.run_stack(x, params, SPEC, True, np.float64)
    dout = out / out.size                             # d(mean(out^2)/2)/d out
    _, grads = R.run_stack_backward(params, caches, dout, np.float64)
    return params, _flat(grads)


def _worker(rank, world, port, q, strategy="allreduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensorflow_yolo2_amd import trainer
    params, g = _rank_grads(rank)
    flat = torch.tensor(g)
    # offsets of each layer's filter inside the flat buffer
    offs, o = [], 0
    for (k, ci, co, _p) in SPEC:
        offs.append(o)
        o += k * k * ci * co + 3 * co
    ranges = [trainer.slice_range(offs, flat.numel(), len(SPEC), lo, hi) for (lo, hi) in [(2, 3), (0, 2)]]
    trainer.reduce_flat(flat, ranges, dist, strategy)
    p0 = _flat([{k: p[k] for k in ("W", "b", "gamma", "beta")} for p in params])
    var, m, v = O.adam_step(p0, np.zeros_like(p0), np.zeros_like(p0), flat.numpy() * (1.0 / world), 1)
    q.put((rank, flat.numpy().copy(), var))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("strategy", ["allreduce", "rs_ag"])
def test_sliced_allreduce_equals_mean_of_replica_gradients(strategy):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, strategy)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r, flat, var = q.get(timeout=120)
        res[r] = (flat, var)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, g1 = _rank_grads(0)[1], _rank_grads(1)[1]
    np.testing.assert_allclose(res[0][0], g0 + g1, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(res[0][0], res[1][0])          # replicas stay identical
    np.testing.assert_array_equal(res[0][1], res[1][1])
    p0 = _flat([{k: p[k] for k in ("W", "b", "gamma", "beta")} for p in _rank_grads(0)[0]])
    exp, _, _ = O.adam_step(p0, np.zeros_like(p0), np.zeros_like(p0), (g0 + g1) / 2, 1)
    np.testing.assert_allclose(res[0][1], exp, rtol=1e-6, atol=1e-7)
