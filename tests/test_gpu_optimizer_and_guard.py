"""GPU tests (-m gpu): the optimizers behind tf.train.*Optimizer().minimize (src/pascal/pascal_train_darknet.py:49-51,
src/imagenet/imagenet_train_darknet.py:58) -- overflow-guarded Adam / Momentum of the half-precision modes, the loss scaler,
the fused train op (backward + guarded update + filter re-pack in one call), bit-reproducible backward passes, and partial
backward passes that leave no stale gradients behind."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import nn_ref as R, loss_ref as L, optim_ref as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def l2err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


# ---------------------------------------------------------------- overflow-guarded optimizers
def test_guarded_adam_matches_oracle_and_skips_on_overflow():
    from tensorflow_yolo2_amd import engine as E
    spec = [(3, 32, 32, 0), (1, 32, 30, 0)]
    net = E.Network(spec, 1, 8, 8, dtype="f16", training=True)
    net.init_params(1)
    opt = E.AdamOptimizer(net)                                  # f16 => guarded by default
    assert opt.guard and opt.scaler.scale == 1024.0
    rng = np.random.default_rng(0)
    p = net.params.cpu().numpy().copy()
    m = np.zeros_like(p); v = np.zeros_like(p)
    for t in range(1, 4):
        g = rng.standard_normal(p.shape).astype(np.float32) * 1e-2
        net.grads.copy_(torch.as_tensor(g))
        opt.step(full_check=True)
        p, m, v = O.adam_step(p, m, v, g, t)
        assert np.abs(net.params.cpu().numpy() - p).max() < 1e-6
    assert opt.scaler.state() == (0, 3, 0)
    # an inf anywhere in the buffer: the step is skipped as a whole, nothing is poisoned
    before = (net.params.clone(), opt.m.clone(), opt.v.clone())
    g = rng.standard_normal(p.shape).astype(np.float32)
    g[g.size // 2] = np.inf                                     # an arbitrary element: the full scan
    net.grads.copy_(torch.as_tensor(g))
    opt.step(full_check=True)
    assert torch.equal(net.params, before[0]) and torch.equal(opt.m, before[1]) and torch.equal(opt.v, before[2])
    assert opt.scaler.state() == (1, 3, 1)
    g[g.size // 2] = np.nan
    net.grads.copy_(torch.as_tensor(g))
    opt.step(full_check=True)                                                  # the host saw the first overflow: scale halved
    assert opt.scaler.scale == 512.0 and net.grad_scale == 512.0
    assert torch.isfinite(net.params).all() and opt.scaler.state() == (1, 3, 2)
    # a clean step resumes at t = 4 with the bias correction of t = 4
    g = rng.standard_normal(p.shape).astype(np.float32) * 1e-2
    net.grads.copy_(torch.as_tensor(g))
    opt.step()                                                  # sentinel scan (the production default)
    p, m, v = O.adam_step(p, m, v, g, 4)
    assert np.abs(net.params.cpu().numpy() - p).max() < 1e-6
    assert opt.scaler.state()[:2] == (0, 4)
    # momentum form
    mo = E.MomentumOptimizer(net, 1e-3, 0.9)
    acc = np.zeros_like(p)
    pm = net.params.cpu().numpy().copy()
    net.grads.copy_(torch.as_tensor(g))
    mo.step()
    pm, acc = O.momentum_step(pm, acc, g)
    assert np.abs(net.params.cpu().numpy() - pm).max() < 1e-6
    g[0] = -np.inf                                              # the first filter is a sentinel range
    net.grads.copy_(torch.as_tensor(g))
    keep = net.params.clone()
    mo.step()
    assert torch.equal(net.params, keep)


def test_f16_training_survives_a_forced_overflow():
    """a loss scale far too large overflows fp16 dY: the guarded step is skipped, the scale backs off and
    training continues with finite parameters (a fixed 1024 had no such safety net)"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    n, size, S = 4, 64, 2
    tr = DetectorTrainer(n, size, dtype="f16", core_spec=core, head_spec=head, grad_scale=2.0 ** 30)
    x = dev(synthetic.images(n, size, 1))
    lab = dev(synthetic.det_labels(n, size, S, 2))
    for _ in range(40):
        tr.step(x, lab)
    torch.cuda.synchronize()
    found, steps, skipped = tr.opt.scaler.state()
    assert skipped >= 1 and steps >= 1, (found, steps, skipped)
    assert tr.opt.scaler.scale < 2.0 ** 30
    assert torch.isfinite(tr.net.params).all() and torch.isfinite(tr.opt.m).all() and torch.isfinite(tr.opt.v).all()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_fused_train_op_equals_backward_then_step(dtype):
    """y2_backward_adam / _momentum (train_op as one call, the upper layers' update overlapped with the first
    layer's gradient kernel) leave params, slots and guard words bit-identical to y2_backward + y2_grad_check +
    y2_*_step_packed, clean steps and an overflowing one alike (pascal_train_darknet.py:49-51)."""
    from tensorflow_yolo2_amd import engine as E, synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    n, size, S = 4, 128, 4
    a = DetectorTrainer(n, size, dtype=dtype, core_spec=core, head_spec=head, seed=5)
    b = DetectorTrainer(n, size, dtype=dtype, core_spec=core, head_spec=head, seed=5)
    assert torch.equal(a.net.params, b.net.params)
    lab = dev(synthetic.det_labels(n, size, S, 2))

    # the second trainer takes the first one's gradient buffer and only runs the separate check + step on it: the
    # comparison is then about the optimizer forms alone (written when split-K float atomics still made two backward
    # passes differ in the last bits; they have been order-fixed slab sums since round 2)
    for it in range(4):
        x = dev(synthetic.images(n, size, 10 + it))
        if it == 2 and dtype == "f16":       # one overflowing step: both forms must skip it
            for tr in (a, b):
                tr.opt.scaler._apply(2.0 ** 40)
        a.step(x, lab)
        b.net.grads.copy_(a.net.grads)
        b.opt.step()
        torch.cuda.synchronize()
        assert torch.equal(a.net.params, b.net.params), it
        assert torch.equal(a.opt.m, b.opt.m) and torch.equal(a.opt.v, b.opt.v), it
        if dtype == "f16":
            assert a.opt.scaler.state() == b.opt.scaler.state(), it
            if it == 2:
                assert a.opt.scaler.state()[2] == 1 and not torch.isfinite(a.net.grads).all()
                for tr in (a, b):
                    tr.opt.scaler._apply(1024.0)
    assert a.opt.scaler is None or a.opt.scaler.state()[1] == 3
    # the packed filter copies followed the fused update: a forward agrees with a context that re-packs from scratch
    x = dev(synthetic.images(n, size, 3))
    ya = a.net.forward(x, True, True).clone()
    b.net.params_changed()
    yb = b.net.forward(x, True, True)
    assert torch.equal(ya, yb)
    # momentum form
    ma, mb = E.MomentumOptimizer(a.net), E.MomentumOptimizer(b.net)
    _, (_, _, _, dnet) = a.forward_loss(x, lab, True, True)
    ma.backward_step(dnet)
    b.net.grads.copy_(a.net.grads)
    mb.step()
    assert torch.equal(a.net.params, b.net.params) and torch.equal(ma.accum, mb.accum)


# ---------------------------------------------------------------- fused optimizer + filter re-pack
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("kind", ["adam", "momentum"])
def test_fused_optimizer_repack_equals_separate_passes(dtype, kind):
    """y2_*_step_packed: same parameters and slots as the flat step bit for bit, and the packed filter copies it
    leaves behind give the same forward as a context that re-packs from the updated parameters."""
    from tensorflow_yolo2_amd import engine as E, synthetic
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 4)]
    head = [(3, core[-1][2], 256, 0), (1, 256, 30, 0)]
    spec = core + head
    n, size = 2, 64
    x = dev(synthetic.images(n, size, 3))
    nets, opts = [], []
    for fused in (True, False):
        net = E.Network(spec, n, size, size, dtype=dtype, core_layers=len(core), training=True)
        net.init_params(5)
        cls = E.AdamOptimizer if kind == "adam" else E.MomentumOptimizer
        opts.append(cls(net, guard=False, fused_pack=fused))
        nets.append(net)
    rng = np.random.default_rng(0)
    for step in range(3):
        g = torch.as_tensor(rng.standard_normal(nets[0].n_params).astype(np.float32) * 1e-2).cuda()
        outs = []
        for net, opt in zip(nets, opts):
            net.grads.copy_(g)
            opt.step()
            outs.append(net.forward(x, True, True).clone())
        assert torch.equal(nets[0].params, nets[1].params), step
        assert torch.equal(opts[0].m, opts[1].m) if kind == "adam" else torch.equal(opts[0].accum, opts[1].accum)
        assert torch.equal(outs[0], outs[1]), step
    # guarded form: a flagged step leaves parameters AND packed copies alone
    net = nets[0]
    opt = (E.AdamOptimizer if kind == "adam" else E.MomentumOptimizer)(net, guard=True, fused_pack=True)
    before = net.forward(x, True, True).clone()
    g = torch.zeros_like(net.grads); g[0] = float("inf")
    net.grads.copy_(g)
    keep = net.params.clone()
    opt.step(full_check=True)
    assert torch.equal(net.params, keep) and torch.equal(net.forward(x, True, True), before)


def test_backward_is_bit_reproducible():
    """split-K partial tiles of the weight gradients go through a slab and a fixed-order sum (no float atomics on the
    C4 path), the first layer's partials are added in a fixed order: two backward passes over the same forward state
    give bit-identical gradient buffers"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    n, size = 16, 416       # large enough for every register-filter convolution form to be selected
    tr = DetectorTrainer(n, size, dtype="f16", seed=2)
    x = dev(synthetic.images(n, size, 5))
    lab = dev(synthetic.det_labels(n, size, size // 32, 6))
    _, (loss, ious, mask, dnet) = tr.forward_loss(x, lab, True, True)
    tr.net.backward(dnet)
    g0 = tr.net.grads.clone()
    tr.net.grads.fill_(float("nan"))            # nothing may rely on a pre-zeroed buffer either
    tr.net.backward(dnet)
    torch.cuda.synchronize()
    assert torch.isfinite(tr.net.grads).all()
    assert torch.equal(tr.net.grads, g0)


def test_partial_backward_zeroes_the_gradients_below_its_range():
    """y2_backward(layer_lo > 0) from the top: the gradients of the layers below layer_lo are ZERO afterwards, not the
    previous step's (a full-buffer optimizer step must not re-apply stale values); a second call that continues
    downwards completes the buffer to what one full pass writes."""
    from tensorflow_yolo2_amd import engine as E
    spec = [(3, 3, 32, 1), (3, 32, 64, 1), (1, 64, 32, 0), (3, 32, 64, 0), (3, 64, 30, 0)]
    rng = np.random.default_rng(0)
    net = E.Network(spec, 4, 32, 32, dtype="f32", training=True)
    net.init_params(1)
    x = dev(rng.uniform(-1, 1, (4, 32, 32, 3)))
    out = net.forward(x, True, True)
    dout = dev(rng.standard_normal(tuple(out.shape)))
    net.backward(dout)
    full = net.grads.clone()
    assert float(full.abs().max()) > 0
    lo = net._offsets[3][0]
    net.forward(x, True, True)
    net.backward(dout, 3, len(spec))
    g = net.grads.clone()
    assert float(g[:lo].abs().max()) == 0.0                      # stale values of layers 0..2 are gone
    # (a slice boundary runs the standalone BN-backward reduce where the full pass rides in the dgrad epilogue:
    #  other partial sums of the same reduction, fp32 round-off apart)
    atol = 1e-5 * float(full.abs().max())
    np.testing.assert_allclose(g[lo:].cpu().numpy(), full[lo:].cpu().numpy(), rtol=1e-4, atol=atol)
    net.backward(None, 0, 3)                                     # continue downwards
    np.testing.assert_allclose(net.grads.cpu().numpy(), full.cpu().numpy(), rtol=1e-4, atol=atol)
