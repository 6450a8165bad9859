"""GPU tests (-m gpu) added in round 2 for the host side of the boundary: variable sharing by scope
(darknet.py:144,187 reuse=True), UPDATE_OPS semantics of the BN moving statistics, initial values
(darknet.py:10-17), the classifier at its real 224x224 / 7x7 average-pool geometry, accuracy
(imagenet_train_darknet.py:60-61), overflow-guarded optimizers, snapshots with optimizer slots and the
two caller scripts run through their main()."""
import os

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


# ---------------------------------------------------------------- a-1: initial values
def test_init_params_distribution_and_seeding():
    """weight_variable: truncated normal(0, 0.1) re-drawn beyond 2 sigma => |w| <= 0.2, std = 0.1 * 0.8796;
    bias_variable: 0.1; BN gamma/beta/moving_mean/moving_var = 1/0/0/1 (darknet.py:10-17, tf.layers defaults).
    Same seed => identical values in another context; another seed => different values."""
    from tensorflow_yolo2_amd import engine as E
    spec = E.CORE_SPEC + E.det_head_spec(30)
    a = E.Network(spec, 1, 64, 64, dtype="f32", core_layers=18, training=False)
    b = E.Network(spec, 2, 96, 96, dtype="f16", core_layers=18, training=False)
    c = E.Network(spec, 1, 64, 64, dtype="f32", core_layers=18, training=False)
    a.init_params(7); b.init_params(7); c.init_params(8)
    pa, pb, pc = a.export_params(), b.export_params(), c.export_params()
    trunc_std = 0.1 * 0.87962566           # std of N(0,1) truncated at +-2
    for l, (k, ci, co, _p) in enumerate(spec):
        w = pa[l]["W"].astype(np.float64).ravel()
        n = w.size
        assert np.abs(w).max() <= 0.2 + 1e-7, l
        assert abs(w.mean()) < 5 * 0.1 / np.sqrt(n) + 1e-4, (l, w.mean())
        assert abs(w.std() - trunc_std) < 6 * trunc_std / np.sqrt(2 * n) + 2e-4, (l, w.std())
        if n > 10000:                       # two-sided tail mass beyond 1 sigma of the parent normal: 0.2846/0.9545*... -> 0.3025
            frac = (np.abs(w) > 0.1).mean()
            assert abs(frac - (1 - 0.682689 / 0.954500)) < 0.01, (l, frac)
        assert (pa[l]["b"] == np.float32(0.1)).all()
        assert (pa[l]["gamma"] == 1).all() and (pa[l]["beta"] == 0).all()
        assert (pa[l]["moving_mean"] == 0).all() and (pa[l]["moving_var"] == 1).all()
        np.testing.assert_array_equal(pa[l]["W"], pb[l]["W"])          # seed, not shape or dtype, decides
        assert not np.array_equal(pa[l]["W"], pc[l]["W"])
    # layers draw from different streams
    assert not np.array_equal(pa[18]["W"], pa[19]["W"])


# ---------------------------------------------------------------- UPDATE_OPS semantics
def test_moving_statistics_move_only_with_the_train_op():
    """reference: UPDATE_OPS hang off train_op (pascal_train_darknet.py:49-51).  A forward that only
    evaluates the output / loss leaves the moving statistics alone -- including the detect script's head,
    which normalises with batch statistics (is_training default) but never updates."""
    from tensorflow_yolo2_amd import engine as E, synthetic
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    spec = core + head
    n, size = 2, 64
    params = R.init_params(spec, seed=3)
    x = synthetic.images(n, size, 5)
    net = E.Network(spec, n, size, size, dtype="f32", core_layers=len(core), training=True)
    net.load_params(params)
    net.forward(dev(x), True, True)                               # update_moving defaults to False
    st = net.export_params()
    for l in range(len(spec)):
        assert (st[l]["moving_mean"] == params[l]["moving_mean"]).all()
        assert (st[l]["moving_var"] == params[l]["moving_var"]).all()
    net.update_moving_stats()                                      # the deferred UPDATE_OPS
    _, _, movings = R.run_stack(x, params, spec, True, np.float64)
    st = net.export_params()
    for l, mv in enumerate(movings):
        assert relerr(st[l]["moving_mean"], mv[0]) < 1e-4 and relerr(st[l]["moving_var"], mv[1]) < 1e-4
    net.update_moving_stats()                                      # applies once per forward
    st2 = net.export_params()
    for l in range(len(spec)):
        np.testing.assert_array_equal(st2[l]["moving_var"], st[l]["moving_var"])
    # fused form == deferred form
    net2 = E.Network(spec, n, size, size, dtype="f32", core_layers=len(core), training=True)
    net2.load_params(params)
    net2.forward(dev(x), True, True, update_moving=True)
    st3 = net2.export_params()
    for l in range(len(spec)):
        np.testing.assert_array_equal(st3[l]["moving_mean"], st[l]["moving_mean"])
        np.testing.assert_array_equal(st3[l]["moving_var"], st[l]["moving_var"])
    # detect-time flags (core infer, head batch statistics): nothing moves
    net2.forward(dev(x), False, True)
    st4 = net2.export_params()
    for l in range(len(spec)):
        np.testing.assert_array_equal(st4[l]["moving_var"], st3[l]["moving_var"])


# ---------------------------------------------------------------- reuse=True shares LIVE variables
def test_reuse_true_graphs_see_each_train_step():
    """VERDICT r1 weak #6 / ADVICE: train step -> eval -> train step -> eval; every eval of the reuse=True
    detect graph must equal a fresh forward of the trainer's CURRENT variables (the reference's
    validate-every-25-iterations pattern, imagenet_train_darknet.py:117-120)."""
    from tensorflow_yolo2_amd import config as cfg, engine as E, synthetic
    from tensorflow_yolo2_amd.yolo2_nets import darknet, net_utils
    darknet.reset_default_graph()
    darknet.set_default_dtype("f32")
    try:
        n, size, S, B = 2, 64, 2, 2
        x = dev(synthetic.images(n, size, 1))
        xd = dev(synthetic.images(1, size, 9))                 # the detect graph runs another batch size
        labels = synthetic.det_labels(n, size, S, 2)
        # detect graph FIRST (the order that went stale in round 1), train graph on the same scopes after
        core_d = darknet.darknet19_core(xd, is_training=False)
        det = darknet.darknet19_detection(core_d, 30).reshape([-1, S, S, 30])
        p0 = det.eval().clone()
        core_t = darknet.darknet19_core(x, is_training=True, reuse=True)
        grid = darknet.darknet19_detection(core_t, 30, reuse=True).reshape([-1, S, S, 30])
        opt = net_utils.AdamOptimizer()
        seen = [p0]
        for it in range(2):
            loss, _, _ = net_utils.get_loss(grid, labels, 20, n, size, S, B, cfg.yolo_grid_offset(S, B))
            opt.minimize(loss)()
            pred = det.eval().clone()
            # a fresh context loaded with the training graph's current variables
            tnet = grid.network
            fresh = E.Network(tnet.spec, 1, size, size, dtype="f32", core_layers=18, training=False)
            fresh.load_params(tnet.export_params())
            want = fresh.forward(xd, False, True)
            np.testing.assert_array_equal(pred.cpu().numpy().reshape(-1), want.cpu().numpy().reshape(-1))
            assert not torch.equal(pred, seen[-1])             # and it did move
            seen.append(pred)
        # one parameter buffer behind both graphs, one Adam state
        assert det.network.params.data_ptr() == grid.network.params.data_ptr()
        assert det.network.state.data_ptr() == grid.network.state.data_ptr()
        assert len(opt._opt) == 1
        # the backbone alone (a leading part of the chain) is a view of the same buffer
        core_only = darknet.darknet19_core(xd, is_training=False, reuse=True)
        feat = core_only.eval()
        assert core_only.network.params.data_ptr() == grid.network.params.data_ptr()
        assert core_only.network.n_params < grid.network.n_params and feat.shape == (1, 2, 2, 1024)
    finally:
        darknet.reset_default_graph()
        darknet.set_default_dtype("f16")


def test_scope_store_extends_when_the_head_joins_later():
    """backbone graph built and run first, detection head declared afterwards: the backbone's variables
    (already evaluated) move into the longer flat buffer and stay shared"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.yolo2_nets import darknet
    darknet.reset_default_graph()
    darknet.set_default_dtype("f32")
    try:
        x = dev(synthetic.images(1, 64, 3))
        core = darknet.darknet19_core(x, is_training=False)
        f0 = core.eval().clone()
        w0 = core.network.layer_views(0)["W"].clone()
        det = darknet.darknet19_detection(core, 30)
        out = det.eval()
        assert out.shape == (1, 2, 2, 30)
        assert core.network.params.data_ptr() == det.network.params.data_ptr()
        assert torch.equal(core.network.layer_views(0)["W"], w0)
        assert torch.equal(core.eval(), f0)
    finally:
        darknet.reset_default_graph()
        darknet.set_default_dtype("f16")


# ---------------------------------------------------------------- a-8 at the real geometry + accuracy
def test_classifier_224_avgpool7_vs_oracle():
    """darknet19() at configs[2]'s geometry (224x224 -> 7x7x1000 -> average_pooling2d(7,7) -> [N,1000]),
    batch 4, f32 mode: logits, softmax-CE loss, accuracy and gradients against the oracle."""
    from tensorflow_yolo2_amd import engine as E, synthetic, _lib
    n, size = 4, 224
    spec = E.CORE_SPEC + E.CLS_HEAD_SPEC
    params = R.init_params(spec, seed=2)
    x = synthetic.images(n, size, 11)
    labels = synthetic.cls_labels(n, 12)
    logits_ref, ctx, _ = R.darknet19(x, params, True, np.float64, spec=R.CORE_SPEC + R.CLS_HEAD_SPEC, pool_k=7)
    loss_ref, dl = R.sparse_softmax_cross_entropy_mean(logits_ref, labels)
    _, rg = R.darknet19_backward(params, ctx, dl, np.float64)
    net = E.Network(spec, n, size, size, dtype="f32", tail=_lib.Y2_TAIL_AVGPOOL, tail_k=7, training=True)
    net.load_params(params)
    logits = net.forward(dev(x), True, True)
    assert logits.shape == (n, 1000)
    e = relerr(logits.cpu().numpy(), logits_ref)
    assert e < 1e-3, e
    loss, dlog = E.softmax_cross_entropy(logits, torch.as_tensor(labels).cuda())
    assert abs(loss.item() - loss_ref) < 1e-3 * loss_ref
    net.backward(dlog)
    g = net.export_grads()
    errs = {l: l2err(g[l]["W"], rg[l]["W"]) for l in (0, 9, 18)}
    print("classifier 224 f32 vs float64 oracle: logits %.2e, dW l2 errors %s" % (e, errs))
    # the last layer sees only dlogits and its own input: tight.  Below it ONE leaky-slope / arg-max decision
    # of a near-tie element that falls the other way than in float64 moves every upstream gradient by ~1 %
    # through the batch-norms over 196-pixel batches (same finding as test_full_detector_f32_vs_oracle_224;
    # observed here: 2.5e-4 at layer 18, 1.0e-2 at layers 9 and 0)
    assert errs[18] < 2e-3 and errs[9] < 3e-2 and errs[0] < 3e-2, errs
    for l in (0, 9):
        a_, b_ = g[l]["W"].ravel().astype(np.float64), rg[l]["W"].ravel().astype(np.float64)
        assert float(a_ @ b_ / (np.linalg.norm(a_) * np.linalg.norm(b_))) > 0.9995, l
    # accuracy: argmax with lowest-index ties
    lab = torch.as_tensor(labels).cuda()
    want = float((logits.argmax(1).cpu().numpy() == labels).mean())
    assert float(E.accuracy(logits, lab)) == want
    forced = logits.clone()
    forced[torch.arange(n), lab.long()] = 1e9                   # every row right
    assert float(E.accuracy(forced, lab)) == 1.0
    ties = torch.zeros((3, 1000), device="cuda")
    assert float(E.accuracy(ties, torch.tensor([0, 0, 5], dtype=torch.int32).cuda())) == pytest.approx(2.0 / 3.0)


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


def test_ab_switches_select_equivalent_paths(tmp_path):
    """Every Y2_* A/B switch of DESIGN section 5 selects between two implementations of the SAME arithmetic: one
    detector train step (batch 16, f16) under each switch gives the loss of the default path to 1e-3 (f16 outputs of differently ordered sums) and its gradient
    buffer / updated parameters to f16 round-off; the pure scheduling switches give the same bits (different tilings and summation orders move near-tie decisions)."""
    import subprocess, sys
    worker = os.path.join(ROOT, "tests", "switch_worker.py")

    def run(env_extra, tag, labels=None):
        env = dict(os.environ)
        env.update(env_extra)
        out = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, worker, out] + ([labels] if labels else []), check=True, env=env, timeout=600)
        return np.load(out)

    # The loss jumps by O(1 / batch) where a perturbation of the output moves a cell's responsible-box choice (the
    # arg-max of two IoUs, net_utils.py:300-305).  Round 4 widened the loss bound to 3e-2 to live with that; instead the
    # object cells whose two IoUs lie within 20 % of each other in the default run are now DROPPED from the labels (a
    # cell's IoUs depend on its own label only, and the forward pass on none): a switch then cannot flip a responsible box
    # (the 1e-2 output perturbation moves an IoU by about as much), object_mask must come out IDENTICAL and the loss is
    # held to 1e-3 again.
    from tensorflow_yolo2_amd import synthetic
    probe = run({}, "probe")
    labels = synthetic.det_labels(16, 416, 13, 8)
    obj = probe["response"] > 0
    margin = np.abs(probe["ious"][..., 0] - probe["ious"][..., 1])
    big = np.maximum(probe["ious"][..., 0], probe["ious"][..., 1])
    # (a randomly initialised head predicts boxes that barely overlap their ground truth: the IoUs are a few percent, so the
    #  margin is taken relative to the larger one -- the perturbation is relative too)
    # ... and absolute as well: a box that barely touches its ground truth (IoU 0.005) loses the overlap altogether
    tied = obj & ((margin < 0.2 * big) | (margin < 0.02))
    labels[tied] = 0.0
    print("object cells %d, dropped as near-tied %d" % (int(obj.sum()), int(tied.sum())))
    assert int((labels[..., 0] > 0).sum()) >= 4
    run_seed = str(tmp_path / "labels.npy")
    np.save(run_seed, labels)
    base = run({}, "base", run_seed)
    assert tuple(base["ctrl"]) == (0, 1, 0)
    switches = [{"Y2_NO_CONV_RF": "1"}, {"Y2_NO_WGRAD_SLAB": "1"}, {"Y2_XCD_CONV": "0", "Y2_XCD_WGRAD": "0"},
                {"Y2_NO_BN_FIN_FUSE": "1"}, {"Y2_NO_FUSED_TRAIN_OP": "1"}, {"Y2_NO_BNBWD_FUSE": "1"},
                {"Y2_NO_WGRAD_OVERLAP": "1"}, {"Y2_HALO_COMPACT": "1"}, {"Y2_HALOQ_1X1": "1"}, {"Y2_NO_HALOQ_52": "1"},
                {"Y2_NO_CONV1_GRAM": "1"}, {"Y2_LEGACY_TILES": "1"}, {"Y2_NO_KSPLIT": "1"},   # round 4: Gram-matrix statistics, tile cost model, K split of small launches
                {"Y2_CONV1_YSEL": "1"}]     # first layer: arg-max conv outputs kept (ysel) instead of 3 index bits + the linear S2
    for sw in switches:
        r = run(sw, "_".join(sw), run_seed)
        assert tuple(r["ctrl"]) == (0, 1, 0), sw
        if not np.array_equal(r["mask"], base["mask"]):
            for c in np.argwhere((r["mask"] != base["mask"]).any(-1)):
                print("mask differs at", tuple(c), "ious base", base["ious"][tuple(c)], "switch", r["ious"][tuple(c)],
                      "response", base["response"][tuple(c)])
        assert np.array_equal(r["mask"], base["mask"]), sw        # index work: the same responsible boxes
        el = abs(float(r["loss"][4]) - float(base["loss"][4])) / abs(float(base["loss"][4]))
        eg, ep = l2err(r["grads"], base["grads"]), l2err(r["params"], base["params"])
        print("switch", sw, "loss %.2e grads %.2e params %.2e" % (el, eg, ep))
        # a different summation order moves f16 outputs by one ulp at layer 2; the randomly initialised 22-layer
        # network amplifies that ~1.4x per layer (scripts/diag_switch_forward.py: 5e-6 -> 1.3e-2 at the output), so
        # the implementation switches are held to the loss (1e-3: the responsible boxes are fixed, above) and to a loose
        # gradient bound here -- their kernels are checked against the oracle one by one elsewhere; the scheduling
        # switches below must give the same bits
        assert el < 1e-3 and eg < 0.5 and ep < 2e-2, (sw, el, eg, ep)
        # (Y2_HALO_COMPACT: the conflict-free LDS image of conv_haloq -- other addresses, the same products in the
        #  same order)
        if any(k in sw for k in ("Y2_NO_FUSED_TRAIN_OP", "Y2_XCD_CONV", "Y2_NO_WGRAD_OVERLAP", "Y2_NO_BN_FIN_FUSE")):
            assert el == 0.0 and eg == 0.0 and ep == 0.0, sw      # scheduling / same-order switches: the same bits
        if "Y2_HALO_COMPACT" in sw:
            # the same products in the same order in every launch it touches -- but the compact image has no K-split form
            # (round 4: the 13x13 dgrads of this batch-16 step split their K range and leave the BN-backward reduce to the
            # standalone kernel): other partial sums in the backward pass, the forward pass bit-identical
            assert el == 0.0 and eg < 1e-2, sw
        if "Y2_CONV1_YSEL" in sw:
            # the forward pass is the same arithmetic (the same window maximum); the first layer's sum of g * y is formed
            # from un-rounded conv outputs (W . X(dz) + b sum dz) instead of the stored f16 ones: its dgamma moves by
            # f16 round-off, everything above it not at all
            assert el == 0.0 and eg < 1e-3, sw
        if "Y2_NO_WGRAD_SLAB" in sw:
            assert el == 0.0 and eg < 1e-5, sw                    # float atomics: summation order only (observed 2.6e-7)
        if "Y2_NO_BNBWD_FUSE" in sw:
            assert el == 0.0 and eg < 1e-2, sw                    # other partial sums of the same reduce (observed 1.4e-3)


# ---------------------------------------------------------------- snapshots with optimizer slots
def test_snapshot_restores_adam_slots_and_rejects_shape_mismatch(tmp_path):
    from tensorflow_yolo2_amd import engine as E
    from tensorflow_yolo2_amd.yolo2_nets import net_utils as NU
    spec = list(E.CORE_SPEC) + E.det_head_spec(30)
    net = E.Network(spec, 1, 64, 64, dtype="f32", core_layers=18, training=True)
    net.init_params(3)
    opt = E.AdamOptimizer(net)
    rng = np.random.default_rng(1)
    for _ in range(3):
        net.grads.copy_(torch.as_tensor(rng.standard_normal(net.n_params).astype(np.float32) * 1e-3))
        opt.step()
    path = str(tmp_path / "train_iter_3.npz")
    names = NU.save_variables(net, path, optimizer=opt)
    assert "darknet19/Variable/Adam" in names and "darknet19_detection/output/Variable_1/Adam_1" in names
    assert "beta1_power" in names and "beta2_power" in names
    snap = np.load(path)
    # TF1's Adam holds beta^(t+1) after t applies (it starts at beta and multiplies once per step): ADVICE r2
    assert abs(float(snap["beta1_power"]) - 0.9 ** 4) < 1e-7 and abs(float(snap["beta2_power"]) - 0.999 ** 4) < 1e-7
    # a snapshot converted from a TF checkpoint carries the powers only: the step is recovered from them
    conv = {k: snap[k] for k in snap.files if k != "adam_step"}
    np.savez(str(tmp_path / "train_iter_9.npz"), **conv)
    opt_c = E.AdamOptimizer(net)
    NU.restore_variables(net, str(tmp_path / "train_iter_9.npz"), optimizer=opt_c)
    assert opt_c.t == 3
    net2 = E.Network(spec, 1, 64, 64, dtype="f32", core_layers=18, training=True)
    net2.init_params(4)
    opt2 = E.AdamOptimizer(net2)
    NU.restore_variables(net2, path, optimizer=opt2)
    assert opt2.t == 3 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    assert torch.equal(net2.params, net.params)
    # the resumed run takes the same 4th step as the uninterrupted one
    g = torch.as_tensor(rng.standard_normal(net.n_params).astype(np.float32) * 1e-3)
    net.grads.copy_(g); net2.grads.copy_(g)
    opt.step(); opt2.step()
    assert torch.equal(net2.params, net.params)
    # another head width under the same names: error, not a silent skip
    other = E.Network(list(E.CORE_SPEC) + E.det_head_spec(35), 1, 64, 64, dtype="f32", core_layers=18, training=False)
    other.init_params(0)
    with pytest.raises(ValueError):
        NU.restore_variables(other, path)


# ---------------------------------------------------------------- a-15: the caller scripts through main()
def test_detect_and_train_scripts_run_their_main(tmp_path, capsys):
    """pascal_train_darknet.main(): trains, saves `train_iter_<n>.npz` (variables + Adam slots), a second
    invocation RESUMES at last_iter_num + 1 (pascal_train_darknet.py:83,93-114).
    pascal_detect_darknet.main() on the reference's test image restores that snapshot before running
    (pascal_detect_darknet.py:54-62) and its output equals a forward of the restored variables."""
    from tensorflow_yolo2_amd import engine as E
    from tensorflow_yolo2_amd.pascal import pascal_detect_darknet, pascal_train_darknet
    from tensorflow_yolo2_amd.yolo2_nets import darknet, net_utils
    from tensorflow_yolo2_amd.img_dataset import pascal_voc
    ck = str(tmp_path / "ckpts")
    darknet.reset_default_graph()
    try:
        r1 = pascal_train_darknet.main(["--iters", "3", "--batch", "2", "--size", "64", "--dtype", "f32",
                                        "--ckpt-dir", ck])
        assert r1["first_iter"] == 1 and r1["last_iter"] == 3 and len(r1["losses"]) == 3
        assert all(np.isfinite(r1["losses"]))
        assert os.path.exists(os.path.join(ck, "train_iter_3.npz"))
        trained = r1["network"].export_params()
        darknet.reset_default_graph()
        r2 = pascal_train_darknet.main(["--iters", "2", "--batch", "2", "--size", "64", "--dtype", "f32",
                                        "--ckpt-dir", ck])
        assert r2["first_iter"] == 4 and r2["last_iter"] == 5
        assert os.path.exists(os.path.join(ck, "train_iter_5.npz"))
        out = capsys.readouterr().out
        assert "Model saved in file" in out
        darknet.reset_default_graph()
        img = os.path.join(ROOT, "tests", "golden", "testImg1.jpg")
        d = pascal_detect_darknet.main([img, "--size", "224", "--dtype", "f32", "--ckpt-dir", ck, "--no-show"])
        assert d["restored"] == 5 and tuple(d["predicts"].shape) == (1, 7, 7, 30)
        # the same forward by hand from the snapshot
        net = E.Network(list(E.CORE_SPEC) + E.det_head_spec(30), 1, 224, 224, dtype="f32", core_layers=18,
                        training=False)
        net.init_params(123)
        net_utils.restore_variables(net, os.path.join(ck, "train_iter_5.npz"))
        from PIL import Image
        rgb = np.array(Image.open(img).convert("RGB"), dtype=np.uint8)
        x = pascal_voc.image_read(rgb[:, :, ::-1], 224).reshape((1, 224, 224, 3))
        want = net.forward(dev(x), False, True)
        np.testing.assert_array_equal(d["predicts"].cpu().numpy().reshape(-1), want.cpu().numpy().reshape(-1))
        # and without any snapshot the script still runs (initial values: the C1 plumbing case)
        darknet.reset_default_graph()
        d0 = pascal_detect_darknet.main([img, "--size", "224", "--dtype", "f32", "--no-show"])
        assert d0["restored"] == 0 and torch.isfinite(d0["predicts"]).all()
    finally:
        darknet.reset_default_graph()
        darknet.set_default_dtype("f16")


# ---------------------------------------------------------------- e: GradReducer at world size 2 on the GPU
@pytest.mark.parametrize("strategy,dtype", [("allreduce", "f32"), ("rs_ag", "f32"), ("allreduce", "f16"), ("rs_ag", "f16"),
                                            ("allreduce", "f16x2")])
def test_grad_reducer_two_ranks_on_one_gpu(strategy, dtype):
    """VERDICT r1 weak #8 / ADVICE: backward_marks + comm stream + collective at world > 1, on device tensors.
    Two rank processes share cuda:0 (gloo moves the device tensors; RCCL refuses two ranks on one device).
    f16: the headline type with its loss scaler, including a step that overflows on ONE rank;
    f16x2 (round 5): the split-operand mode through the same sequence (its gradients are fp32, its dY rides the scale)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, Y2_DP_STRATEGY=strategy, Y2_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="2", Y2_TEST_DTYPE=dtype)
    env.pop("Y2_FORCE_DIST", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "dp2 ok" in r.stdout


def test_bench_two_ranks_child_tree_on_one_gpu():
    """`python bench.py --gpus 2` as the driver starts it: the parent spawns the torch.distributed.run
    child tree BEFORE any GPU call (bench.spawn_ranks; a GPU-initialised process is never re-executed), both ranks run the
    sharded detector step with the sliced gradient all-reduce and rank 0 prints the one JSON line.  Two ranks share
    cuda:0 here, so the collective goes through gloo on the device tensors (--dist-backend gloo); everything else of the
    world > 1 branch is the code the 8-GPU run takes."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "Y2_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--all-ranks-on-gpu0", "--dist-backend", "gloo",
           "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-f32-mode", "--no-fast-parity-mode",
           "--sustain-steps", "0", "--fed-steps", "0", "--no-extra-legs"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["grad_allreduce"]["slices"] == 7
    assert np.isfinite(d["value"]) and d["value"] > 0 and np.isfinite(d["ms_per_step"])
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # whole-job rate over all ranks


def test_backward_marks_rejects_out_of_range_layers():
    from tensorflow_yolo2_amd import engine as E, _lib
    net = E.Network([(3, 32, 32, 0), (1, 32, 30, 0)], 1, 8, 8, dtype="f32", training=True)
    net.init_params(0)
    x = torch.zeros((1, 8, 8, 32), device="cuda")
    out = net.forward(x, True, True)
    with pytest.raises(_lib.Y2Error):
        net.backward_marks(torch.ones_like(out), [0, 2])
    net.backward_marks(torch.ones_like(out), [1, 0])


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
