"""GPU tests (-m gpu) of "f16x2f" (round 6; include/yolo2_hip.h Y2_F16X2F): the split-operand forward pass of "f16x2"
(reference precision on the f16 matrix pipe: every leaky / arg-max / responsible-box decision is the fp32 reference's,
src/yolo2_nets/darknet.py:10-46) with the two backward contractions of tf.gradients (src/pascal/pascal_train_darknet.py:49-51)
on the HI planes of dY, W and x alone -- one f16 MFMA per product instead of three.  With the forward decisions fixed
the backward pass is linear in dY, so the f16 operand rounding is not amplified.  Gates:
  * every distinct layer shape of BASELINE.json configs[3] at batch 64 through y2_conv2d(_backward) with GENERAL fp32
    inputs (nothing is f16-representable) against float64: forward 3e-5 as f16x2, dgrad / wgrad 1e-3 of the max and
    1e-3 element-wise where the sum has not cancelled;
  * one whole detector step at 416x416, batch 8 and batch 64, against the PyTorch-CPU restatement with the exact-f32 mode's
    gates UNCHANGED (grid / loss / last-layer gradients 1e-3, object_mask identical, 1 - cos(dW) <= 1e-3 on layers 0 / 7 /
    17 / 18);
  * single-layer networks at the C4 shapes (BN backward, in-network weight gradient) against float64 on the hi planes the
    device multiplies: 1e-4, as f16x2;
  * toy stacks against the exact-f32 mode: forward 1e-4 (it IS the f16x2 forward); backward 3e-3 strictly down to the first
    decision flip, cosine below -- contractions of K = 27 ... 64 terms do not average the 2^-11 operand rounding the way the
    network's K >= 288 do (observed 1.6e-3 on the 1 x 400 x 400 three-layer stack; every real layer shape and the whole
    416x416 step hold 1e-3);
  * forward bit-identical to f16x2; overflow guard, fused train op, bit-reproducible backward."""
import numpy as np
import pytest
import torch

from oracle import nn_ref as R

pytestmark = pytest.mark.gpu

from _shapes import check_layer_shape, rel_to_max   # noqa: E402
from test_gpu_c4_shapes import C4_SHAPES, _full_detector_step_f32_vs_torch_oracle   # noqa: E402
from test_gpu_f16x2 import EDGE_SHAPES, STACKS, _stack_vs_f32   # noqa: E402

TOL = 1e-3      # north_star's tolerance, rel. to the tensor's max (tests/_shapes.py)


def dev(a):
    return torch.as_tensor(a).cuda()


@pytest.mark.parametrize("name,k,cin,cout,hw", C4_SHAPES, ids=[s[0] for s in C4_SHAPES])
def test_f16x2f_c4_layer_shape_general_inputs_vs_float64(name, k, cin, cout, hw):
    check_layer_shape(64, name, k, cin, cout, hw, "C4", dtype="f16x2f", tol=TOL, representable=False)


@pytest.mark.parametrize("N,name,k,cin,cout,hw", EDGE_SHAPES, ids=[s[1] for s in EDGE_SHAPES])
def test_f16x2f_edge_shapes_vs_float64(N, name, k, cin, cout, hw):
    check_layer_shape(N, name, k, cin, cout, hw, "edge", dtype="f16x2f", tol=TOL, representable=False)


def test_f16x2f_forward_is_the_f16x2_forward_bit_for_bit():
    from tensorflow_yolo2_amd import engine as E
    spec = [(3, 3, 32, 1), (3, 32, 64, 1), (3, 64, 128, 0), (1, 128, 64, 0), (3, 64, 128, 1), (1, 128, 30, 0)]
    x = dev(np.random.default_rng(0).uniform(-1, 1, (4, 64, 64, 3)).astype(np.float32))
    outs = []
    for dt in ("f16x2", "f16x2f"):
        net = E.Network(spec, 4, 64, 64, dtype=dt, training=True)
        net.load_params(R.init_params(spec, seed=2))
        outs.append(net.forward(x, True, True).clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name,spec,n,hw,tail", STACKS, ids=[s[0] for s in STACKS])
def test_f16x2f_stack_vs_exact_f32_mode(name, spec, n, hw, tail):
    _stack_vs_f32(spec, n, hw, tail, dtype="f16x2f", tol_b=3e-3)


from test_gpu_f16x2 import NET_SHAPES   # noqa: E402


@pytest.mark.parametrize("name,k,cin,cout,hw,pool", NET_SHAPES, ids=[s[0] for s in NET_SHAPES])
def test_f16x2f_layer_in_network(name, k, cin, cout, hw, pool):
    from _shapes import check_layer_in_network
    check_layer_in_network(64, name, k, cin, cout, hw, pool, "C4", dtype="f16x2f", TOL=1e-4)


def test_f16x2f_full_detector_step_416_bs8_vs_torch_oracle():
    _full_detector_step_f32_vs_torch_oracle(8, "f16x2f")


def test_f16x2f_full_detector_step_416_bs64_vs_torch_oracle():
    """the batch-64 whole-step oracle test of the exact-f32 mode, unchanged"""
    _full_detector_step_f32_vs_torch_oracle(64, "f16x2f")


def test_f16x2f_training_survives_a_forced_overflow():
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    n, size, S = 4, 64, 2
    tr = DetectorTrainer(n, size, dtype="f16x2f", core_spec=core, head_spec=head, grad_scale=2.0 ** 30)
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


def test_f16x2f_fused_train_op_equals_backward_then_step_and_is_reproducible():
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 64, 0), (1, 64, 30, 0)]
    n, size, S = 4, 128, 4
    a = DetectorTrainer(n, size, dtype="f16x2f", core_spec=core, head_spec=head, seed=5)
    b = DetectorTrainer(n, size, dtype="f16x2f", core_spec=core, head_spec=head, seed=5)
    lab = dev(synthetic.det_labels(n, size, S, 2))
    for it in range(3):
        x = dev(synthetic.images(n, size, 10 + it))
        a.step(x, lab)
        b.net.grads.copy_(a.net.grads)
        b.opt.step()
        torch.cuda.synchronize()
        assert torch.equal(a.net.params, b.net.params), it
    x = dev(synthetic.images(n, size, 3))
    out = a.net.forward(x, True, True)
    g = dev(np.random.default_rng(1).standard_normal(tuple(out.shape)).astype(np.float32))
    runs = []
    for _ in range(3):
        a.net.forward(x, True, True)
        a.net.backward(g)
        torch.cuda.synchronize()
        runs.append(a.net.grads.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])


def test_f16x2f_train_steps_follow_the_f32_mode():
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 4)]
    head = [(3, core[-1][2], 128, 0), (1, 128, 30, 0)]
    n, size, S = 8, 128, 4
    x = dev(synthetic.images(n, size, 1))
    lab = dev(synthetic.det_labels(n, size, S, 2))
    losses, finals = {}, {}
    for dt in ("f32", "f16x2f"):
        tr = DetectorTrainer(n, size, dtype=dt, core_spec=core, head_spec=head, seed=7)
        losses[dt] = [float(tr.step(x, lab)[0][4]) for _ in range(5)]
        torch.cuda.synchronize()
        finals[dt] = tr.net.params.double().cpu().numpy()
    rel = [abs(a - b) / abs(a) for a, b in zip(losses["f32"], losses["f16x2f"])]
    cos = float(finals["f32"] @ finals["f16x2f"] / (np.linalg.norm(finals["f32"]) * np.linalg.norm(finals["f16x2f"])))
    print("losses f32 %s\n      f16x2f %s\n  rel %s  cos(params) %.8f" % (losses["f32"], losses["f16x2f"], ["%.1e" % v for v in rel], cos))
    assert rel[0] < 1e-5, rel
    assert all(np.isfinite(losses["f16x2f"])) and losses["f16x2f"][-1] < losses["f16x2f"][0]
    assert cos > 0.9999, cos


def test_f16x2f_classifier_and_multi_scale_trainers():
    """the other two train-step drivers in this mode: darknet19 + softmax-CE + Momentum (src/imagenet/imagenet_train_darknet.py:
    46-58; average-pool tail) at reduced width, and the multi-scale detector trainer on two sizes -- the first loss is the
    exact-f32 mode's to 1e-5, the loss falls, everything stays finite"""
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import ClassifierTrainer, MultiScaleDetectorTrainer
    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 4)]
    spec = core + [(1, core[-1][2], 1000, 0)]
    n, size = 8, 128
    x = dev(synthetic.images(n, size, 3))
    lab = torch.as_tensor(np.random.default_rng(5).integers(0, 1000, n).astype(np.int32)).cuda()
    first = {}
    for dt in ("f32", "f16x2f"):
        tr = ClassifierTrainer(n, size, dtype=dt, spec=spec, seed=2)
        losses = [float(tr.step(x, lab)[0]) for _ in range(6)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], (dt, losses)
        assert torch.isfinite(tr.net.params).all()
        first[dt] = losses[0]
    assert abs(first["f16x2f"] - first["f32"]) < 1e-5 * abs(first["f32"]), first
    head = [(3, core[-1][2], 128, 0), (1, 128, 30, 0)]
    ms = MultiScaleDetectorTrainer(4, sizes=(96, 128), period=1, dtype="f16x2f", core_spec=core, head_spec=head, seed=1)
    data = {s: (dev(synthetic.images(4, s, 10 + s)), dev(synthetic.det_labels(4, s, s // 32, 20 + s))) for s in (96, 128)}
    hist = {96: [], 128: []}
    for step in range(8):
        s = (96, 128)[step % 2]
        hist[s].append(float(ms.step(*data[s])[0][4]))
    for s in (96, 128):
        assert all(np.isfinite(hist[s])) and hist[s][-1] < hist[s][0], hist
    assert ms.nets[96].params.data_ptr() == ms.nets[128].params.data_ptr()
