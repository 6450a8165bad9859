"""GPU tests (-m gpu) of the split-operand mode "f16x2" (round 5; include/yolo2_hip.h Y2_F16X2): the reference's
fp32 arithmetic (src/yolo2_nets/darknet.py:10-46, tf.float32 placeholders src/pascal/pascal_train_darknet.py:34-36) on the
f16 matrix pipe -- every MFMA operand is a (hi, lo) pair of halves, a product is hi*hi + lo*hi + hi*lo with fp32
accumulation, everything stored is fp32 wide.  Gates:
  * every distinct layer shape of BASELINE.json configs[3] at batch 64, forward / dgrad / wgrad through y2_conv2d(_backward)
    with GENERAL fp32 inputs (both operand planes in use) against float64: 3e-5 of the tensor's max (observed <= 5e-6;
    the f16 mode's own gate is 1e-3), element-wise 1e-4 where the sum has not cancelled;
  * one whole detector step at 416x416, batch 8 and batch 64, against the PyTorch-CPU restatement: the f32 mode's own
    gates (grid / loss / last-layer gradients 1e-3, object_mask identical, cos(dW) > 0.999), unchanged;
  * stacks against the exact-f32 mode on the same parameters (first-layer forms, pooled / un-pooled, 1x1 / 3x3,
    30-channel output, average-pool tail): outputs and gradients 1e-4 of the max;
  * the loss-scale guard: a forced overflow skips the step on the device and training continues; the fused train op and
    the bit-reproducibility of the backward pass (the three operand-plane pairs of a weight gradient are summed in a
    fixed order)."""
import numpy as np
import pytest
import torch

from oracle import nn_ref as R

pytestmark = pytest.mark.gpu

from _shapes import check_layer_shape, rel_to_max   # noqa: E402
from test_gpu_c4_shapes import C4_SHAPES, _full_detector_step_f32_vs_torch_oracle   # noqa: E402

TOL_LAYER = 3e-5


def dev(a):
    return torch.as_tensor(a).cuda()


@pytest.mark.parametrize("name,k,cin,cout,hw", C4_SHAPES, ids=[s[0] for s in C4_SHAPES])
def test_f16x2_c4_layer_shape_vs_float64(name, k, cin, cout, hw):
    check_layer_shape(64, name, k, cin, cout, hw, "C4", dtype="f16x2", tol=TOL_LAYER, representable=False)


EDGE_SHAPES = [  # (N, name, k, cin, cout, hw): tail tiles, odd maps, one image, the 30-channel output, 32-channel K chunks
    (1, "one-image-7", 3, 1024, 1024, 7), (3, "odd-13", 3, 64, 128, 13), (2, "out30", 1, 256, 30, 7),
    (5, "tail-26", 3, 128, 256, 26), (2, "k64-208", 3, 32, 64, 208), (2, "co32-208", 3, 64, 32, 208),
    (24, "train-shape-7", 3, 512, 1024, 7), (3, "1x1-52", 1, 256, 128, 52), (2, "k32-1x1", 1, 32, 64, 19)]


@pytest.mark.parametrize("N,name,k,cin,cout,hw", EDGE_SHAPES, ids=[s[1] for s in EDGE_SHAPES])
def test_f16x2_edge_shapes_vs_float64(N, name, k, cin, cout, hw):
    check_layer_shape(N, name, k, cin, cout, hw, "edge", dtype="f16x2", tol=TOL_LAYER, representable=False)


def test_f16x2_small_weights_and_large_activations_keep_their_bits():
    """range: weights of 1e-3 (lo plane deep in the f16 subnormals without the filter pre-scale) and activations of a few
    hundred -- the planes still carry ~20 bits of the product sum"""
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(3)
    n, hw, cin, cout = 2, 13, 256, 128
    x = (rng.standard_normal((n, hw, hw, cin)) * 100.0).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 1e-3).astype(np.float32)
    y = E.conv2d(dev(x), dev(w), None, dtype="f16x2").cpu().numpy()
    ref = torch.nn.functional.conv2d(torch.tensor(x).double().permute(0, 3, 1, 2), torch.tensor(w).double().permute(3, 2, 0, 1),
                                     padding=1).permute(0, 2, 3, 1).numpy()
    e = rel_to_max(y, ref)
    print("small weights / large activations: %.2e" % e)
    assert e < TOL_LAYER, e


def _stack_vs_f32(spec, n, hw, tail=None, seed=0, tol=1e-4, dtype="f16x2", tol_b=None):
    """forward: every stored activation and the output, strictly.  Backward: the two modes are the exact gradients of
    functions that may differ in single leaky / arg-max decisions (their conv outputs differ in the 6th digit; a decision
    within that of its boundary falls the other way and moves ONE dy entry by O(1), which the batch-norm sums below then
    spread): dy is compared layer by layer from the top, strictly down to the first layer where entries are off -- there
    the off entries must be few -- and every gradient must agree in direction (cosine) whatever happened.  The
    per-layer backward arithmetic is gated strictly, with that accounting, in test_f16x2_layer_in_network."""
    from tensorflow_yolo2_amd import engine as E, _lib
    rng = np.random.default_rng(seed)
    params = R.init_params(spec, seed=seed)
    x = rng.uniform(-1, 1, (n, hw, hw, spec[0][1])).astype(np.float32)
    res = {}
    tol_b = tol if tol_b is None else tol_b
    for dt in ("f32", dtype):
        kw = {} if tail is None else {"tail": _lib.Y2_TAIL_AVGPOOL, "tail_k": tail}
        net = E.Network(spec, n, hw, hw, dtype=dt, training=True, grad_scale=1.0 if dt == "f32" else 256.0, **kw)
        net.load_params(params)
        out = net.forward(dev(x), True, True).clone()
        g = np.random.default_rng(seed + 1).standard_normal(tuple(out.shape)).astype(np.float32) * 1e-2
        net.backward(dev(g))
        torch.cuda.synchronize()
        dys = {}
        for l in range(len(spec)):
            try:
                dys[l] = net.debug_read(l, 2).cpu().numpy().astype(np.float64) / net.grad_scale
            except Exception:      # the 3-channel layer's dy is fused into its weight gradient
                pass
        res[dt] = (out.cpu().numpy().astype(np.float64), net.export_grads(),
                   [net.debug_read(l, 0).cpu().numpy().astype(np.float64) for l in range(1, len(spec))], dys)
    (o32, g32, a32, d32), (o2, g2, a2, d2) = res["f32"], res[dtype]
    worst = rel_to_max(o2, o32)
    for l in range(len(spec) - 1):
        worst = max(worst, rel_to_max(a2[l], a32[l]))      # stored (split) activations, read back as hi + lo
    assert worst < tol, ("forward", spec, worst)
    flipped_at, worst_b = None, 0.0
    for l in range(len(spec) - 1, -1, -1):
        if l in d32 and l in d2:
            off = np.abs(d2[l] - d32[l]) > max(1e-3, 3 * tol_b) * np.abs(d32[l]).max()
            if off.any():
                assert int(off.sum()) <= 8 + 1e-5 * off.size, ("layer %d: many dy entries off" % l, int(off.sum()))
                flipped_at = l
                break
            worst_b = max(worst_b, rel_to_max(d2[l], d32[l]))
        for k in ("W", "gamma", "beta"):
            worst_b = max(worst_b, rel_to_max(g2[l][k], g32[l][k].astype(np.float64)))
    assert worst_b < tol_b, ("backward above the first decision flip", spec, worst_b)
    cosmin = 1.0
    for l in range(len(spec)):
        a, b = g2[l]["W"].ravel().astype(np.float64), g32[l]["W"].ravel().astype(np.float64)
        cosmin = min(cosmin, float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
    print(dtype + " vs f32 mode %s n=%d hw=%d: forward %.2e  backward %.2e (strict down to layer %s)  min cos(dW) %.6f" %
          (spec, n, hw, worst, worst_b, "0" if flipped_at is None else str(flipped_at), cosmin))
    assert cosmin > 0.99, (spec, cosmin)


STACKS = [
    ("pooled-first-layer", [(3, 3, 32, 1), (3, 32, 64, 1), (3, 64, 128, 0), (1, 128, 64, 0), (3, 64, 128, 1), (1, 128, 30, 0)], 4, 64, None),
    ("unpooled-first-layer-odd", [(3, 3, 32, 0), (3, 32, 64, 1), (1, 64, 30, 0)], 2, 30, None),
    ("no-image-layer", [(3, 32, 64, 1), (3, 64, 128, 0), (1, 128, 64, 0), (3, 64, 32, 0)], 2, 32, None),
    ("wide", [(3, 128, 256, 1), (3, 256, 512, 0), (1, 512, 256, 0), (3, 256, 512, 1), (3, 512, 1024, 0), (1, 1024, 30, 0)], 8, 28, None),
    ("avgpool-tail", [(3, 3, 32, 1), (3, 32, 64, 1), (1, 64, 1000, 0)], 4, 28, 7),
    # a 400-wide image: the first layer's linear-form backward works a row pair in two column segments of 208 pixels (its
    # fp32-wide LDS row images leave room for two workgroups per CU), the second one narrower (conv1_wgrad.hip nseg / ws)
    ("pooled-first-layer-two-segments", [(3, 3, 32, 1), (3, 32, 64, 1), (1, 64, 30, 0)], 1, 400, None),
]


@pytest.mark.parametrize("name,spec,n,hw,tail", STACKS, ids=[s[0] for s in STACKS])
def test_f16x2_stack_vs_exact_f32_mode(name, spec, n, hw, tail):
    _stack_vs_f32(spec, n, hw, tail)


NET_SHAPES = [("conv2+pool", 3, 32, 64, 208, 1), ("conv3", 3, 64, 128, 104, 0), ("conv5+pool", 3, 64, 128, 104, 1),
              ("conv8+pool", 3, 128, 256, 52, 1), ("conv7", 1, 256, 128, 52, 0), ("conv13+pool", 3, 256, 512, 26, 1),
              ("conv14", 3, 512, 1024, 13, 0), ("head1", 3, 1024, 1024, 13, 0)]


@pytest.mark.parametrize("name,k,cin,cout,hw,pool", NET_SHAPES, ids=[s[0] for s in NET_SHAPES])
def test_f16x2_layer_in_network(name, k, cin, cout, hw, pool):
    """conv_bn_layer (darknet.py:32-46) as a single-layer network at a C4 shape, batch 64, split-operand mode: the conv
    output with the epilogue's statistics, BN + leaky (+ pool) into the SPLIT tensor of the consumer, BN backward into the
    split dY tensor, dgamma / dbeta and the in-network weight gradient (three operand-plane launches, one fixed-order
    sum), each against float64 on the values the device stored, decision flips accounted (tests/_shapes.py): 1e-4"""
    from _shapes import check_layer_in_network
    check_layer_in_network(64, name, k, cin, cout, hw, pool, "C4", dtype="f16x2", TOL=1e-4)


def test_f16x2_full_detector_step_416_bs8_vs_torch_oracle():
    _full_detector_step_f32_vs_torch_oracle(8, "f16x2")


def test_f16x2_full_detector_step_416_bs64_vs_torch_oracle():
    """the batch-64 whole-step oracle test of the f32 mode, unchanged, in the split-operand mode"""
    _full_detector_step_f32_vs_torch_oracle(64, "f16x2")


def test_f16x2_train_steps_follow_the_f32_mode():
    """five Adam steps at 128x128 in both modes from the same parameters.  The first loss (same parameters) agrees to
    1e-5; beyond that a train step is a chaotic map of its rounding (Adam's first updates are +-lr whatever the
    gradient's size: every entry whose sign is decided in the 6th digit of the largest moves its parameter by 2 lr,
    observed 1.7e-2 of the loss after ONE update -- the exact-f32 mode against the torch oracle behaves the same), so
    the trajectories are only held to falling and to staying parallel (cosine of the parameter vectors); the gradients
    themselves are gated at 1e-4 / 1e-3 in the tests above"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 4)]
    head = [(3, core[-1][2], 128, 0), (1, 128, 30, 0)]
    n, size, S = 8, 128, 4
    x = dev(synthetic.images(n, size, 1))
    lab = dev(synthetic.det_labels(n, size, S, 2))
    losses, finals = {}, {}
    for dt in ("f32", "f16x2"):
        tr = DetectorTrainer(n, size, dtype=dt, core_spec=core, head_spec=head, seed=7)
        losses[dt] = [float(tr.step(x, lab)[0][4]) for _ in range(5)]
        torch.cuda.synchronize()
        finals[dt] = tr.net.params.double().cpu().numpy()
    rel = [abs(a - b) / abs(a) for a, b in zip(losses["f32"], losses["f16x2"])]
    cos = float(finals["f32"] @ finals["f16x2"] / (np.linalg.norm(finals["f32"]) * np.linalg.norm(finals["f16x2"])))
    print("losses f32 %s\n       f16x2 %s\n  rel %s  cos(params) %.8f" % (losses["f32"], losses["f16x2"], ["%.1e" % v for v in rel], cos))
    assert rel[0] < 1e-5, rel
    assert all(np.isfinite(losses["f16x2"])) and losses["f16x2"][-1] < losses["f16x2"][0]
    assert cos > 0.9999, cos


def test_f16x2_training_survives_a_forced_overflow():
    """the dY planes have f16's exponent range: a loss scale far too large overflows them, the guarded step is skipped on the
    device, the scale backs off and training continues (the f16 mode's guard, engine.LossScaler)"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    n, size, S = 4, 64, 2
    tr = DetectorTrainer(n, size, dtype="f16x2", core_spec=core, head_spec=head, grad_scale=2.0 ** 30)
    assert tr.opt.scaler is not None and tr.opt.scaler.enabled
    x = dev(synthetic.images(n, size, 1))
    lab = dev(synthetic.det_labels(n, size, S, 2))
    for _ in range(40):
        tr.step(x, lab)
    torch.cuda.synchronize()
    found, steps, skipped = tr.opt.scaler.state()
    assert skipped >= 1 and steps >= 1, (found, steps, skipped)
    assert tr.opt.scaler.scale < 2.0 ** 30
    assert torch.isfinite(tr.net.params).all() and torch.isfinite(tr.opt.m).all() and torch.isfinite(tr.opt.v).all()


def test_f16x2_fused_train_op_equals_backward_then_step():
    """y2_backward_adam in the split mode: the fused optimizer + re-pack of the (hi, lo) filter planes leaves the same
    parameters as the separate passes, and a forward after it agrees bit for bit with a context that re-packs from scratch"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    n, size, S = 4, 128, 4
    a = DetectorTrainer(n, size, dtype="f16x2", core_spec=core, head_spec=head, seed=5)
    b = DetectorTrainer(n, size, dtype="f16x2", core_spec=core, head_spec=head, seed=5)
    lab = dev(synthetic.det_labels(n, size, S, 2))
    for it in range(3):
        x = dev(synthetic.images(n, size, 10 + it))
        a.step(x, lab)
        b.net.grads.copy_(a.net.grads)
        b.opt.step()
        torch.cuda.synchronize()
        assert torch.equal(a.net.params, b.net.params), it
    x = dev(synthetic.images(n, size, 3))
    ya = a.net.forward(x, True, True).clone()
    b.net.params_changed()
    yb = b.net.forward(x, True, True)
    assert torch.equal(ya, yb)


def test_f16x2_backward_is_bit_reproducible():
    """the three operand-plane pairs of a weight gradient are summed like split-K partials, in a fixed order"""
    from tensorflow_yolo2_amd import engine as E
    spec = [(3, 3, 32, 1), (3, 32, 64, 1), (3, 64, 128, 0), (1, 128, 64, 0), (3, 64, 128, 1), (1, 128, 30, 0)]
    net = E.Network(spec, 4, 64, 64, dtype="f16x2", training=True)
    net.load_params(R.init_params(spec, seed=2))
    x = dev(np.random.default_rng(0).uniform(-1, 1, (4, 64, 64, 3)).astype(np.float32))
    out = net.forward(x, True, True)
    g = dev(np.random.default_rng(1).standard_normal(tuple(out.shape)).astype(np.float32))
    runs = []
    for _ in range(3):
        net.forward(x, True, True)
        net.backward(g)
        torch.cuda.synchronize()
        runs.append(net.grads.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])


def test_f16_mfma_keeps_subnormal_operands():
    """The lo plane of the split-operand mode lives in f16's subnormal range for |v| < 2^-3 (csrc/common.h hsplit_t): the
    design relies on v_mfma_f32_*_f16 NOT flushing subnormal A / B inputs on gfx950 (first measured with
    scripts/probes/mfma_denorm.hip; ADVICE r5: re-checked here on every run, through the product library).  A 1x1
    convolution of subnormal activations 2^-20 and 3 * 2^-24 with weights of 8 over 32 channels: the exact sums, not 0.
    (Weights stay below 1024 in the split mode: its filter planes hold 64 w in f16, common.h kSplitWScale.)"""
    from tensorflow_yolo2_amd import engine as E
    for a in (2.0 ** -20, 3 * 2.0 ** -24):
        assert 0 < float(np.float16(a)) < 2.0 ** -14                 # subnormal in f16, exactly representable
        x = torch.full((1, 4, 4, 32), a, dtype=torch.float32, device="cuda")
        w = torch.full((1, 1, 32, 32), 8.0, dtype=torch.float32, device="cuda")
        for dt in ("f16", "f16x2"):
            y = E.conv2d(x, w, None, dtype=dt).cpu().numpy()
            want = 32 * a * 8.0
            got = float(y[0, 1, 1, 0])
            assert abs(got - want) <= 2.0 ** -11 * want, (dt, a, got, want)
