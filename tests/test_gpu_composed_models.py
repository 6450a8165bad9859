"""GPU tests (-m gpu): loss scaling across COMPOSED graphs -- the ResNet-50 swap's f16 loss scale, gradients and overflow
guard (src/pascal/pascal_train_resnet.py:37-50) and the YOLOv2 trainer's one scaler over its three stacks (north star;
not in the reference)."""
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


def test_resnet_f16_loss_scale_gradients_and_overflow_guard():
    """ResNet50Yolo in f16: with the loss scale in front of the backward pass the f16 gradients
    agree with the f32 ones (which tests/test_gpu_resnet.py holds against the float64 oracle) from the head down to
    the root convolution; an overflow skips the update on the device, leaves parameters and Adam slots untouched
    and halves the scale."""
    from oracle import resnet_ref as RR
    from tensorflow_yolo2_amd import engine as E, synthetic
    from tensorflow_yolo2_amd.yolo2_nets import tf_resnet
    n, size, S, div = 2, 64, 2, 8
    blocks = RR.scaled_blocks(div)
    kw = dict(blocks=blocks, root_depth=64 // div, fc_hidden=4096 // div, seed=1)
    x, labels = dev(synthetic.images(n, size, 5)), dev(synthetic.det_labels(n, size, S, 6))
    # The backward pass is linear in the output gradient.  In f16 that holds until values leave the normal range:
    # a SMALL output gradient (x 1e-6: what 50 layers deep looks like late in training) pushed through unscaled
    # underflows in the f16 operands of the dgrad / wgrad kernels; the same gradient times the loss scale does not.
    # (f16 against f32 is not a usable gate at this toy size: 49 batch-norms over 2 x 2 x 2 positions amplify the
    #  storage rounding of the forward pass chaotically, as DESIGN section 4 explains for the Darknet stack.)
    m = tf_resnet.ResNet50Yolo(n, size, dtype="f16", **kw)
    assert m.loss_scale == 1024.0 and m.guard
    f32 = tf_resnet.ResNet50Yolo(n, size, dtype="f32", **kw)
    assert f32.loss_scale == 1.0 and not f32.guard
    del f32
    grid = m.forward(x, True, dropout=False)
    _loss, _, _, dnet = E.yolo_loss(grid, labels, 20, n, size, S, 2)

    def grads_for(factor, scale):
        d = dnet.clone()
        E.check(m_lib().y2_scale(E._ptr(d), d.numel(), float(factor * scale), E._stream()))
        m.backward(d)
        return {k: v.astype(np.float64) / (factor * scale) for k, v in m.export_grads().items()}
    ref = grads_for(1.0, 1.0)                      # O(1) output gradient: in range without any scale
    tiny_unscaled = grads_for(1e-6, 1.0)
    tiny_scaled = grads_for(1e-6, 2.0 ** 20)       # the same tiny gradient behind a loss scale of 2^20 (~1e6)
    names = ("yolo_fc1/weights", "block4/unit_3/bottleneck_v1/conv3/weights", "block3/unit_6/bottleneck_v1/conv2/weights",
             "block2/unit_1/bottleneck_v1/conv1/weights", "block1/unit_1/bottleneck_v1/shortcut/BatchNorm/beta", "conv1/weights")
    assert all(np.isfinite(ref[k]).all() and np.linalg.norm(ref[k]) > 0 for k in names)
    for k in names:
        r = ref[k].ravel()
        e_s = float(np.linalg.norm(tiny_scaled[k].ravel() - r) / np.linalg.norm(r))
        e_u = float(np.linalg.norm(tiny_unscaled[k].ravel() - r) / np.linalg.norm(r))
        print("resnet f16 linearity of the backward pass %-50s scaled %.2e  unscaled %.2e" % (k, e_s, e_u))
        assert e_s < 2e-2, (k, e_s)
    assert max(float(np.linalg.norm(tiny_unscaled[k].ravel() - ref[k].ravel()) / np.linalg.norm(ref[k].ravel()))
               for k in names) > 0.05      # without the scale the small gradient degrades (observed 0.16 vs < 0.02 scaled)
    # ---- overflow: an absurd scale makes f16 gradients inf; the guarded step must skip
    m = tf_resnet.ResNet50Yolo(n, size, dtype="f16", loss_scale=1e9, **kw)
    p0, m0 = m.params.clone(), m.m.clone()
    m.step(x, labels)
    assert m.overflows == 1 and m.loss_scale == 5e8 and m.t == 0
    assert torch.equal(m.params, p0) and torch.equal(m.m, m0)
    # from the default scale the scaler backs off until the steps go through (this toy geometry overflows at 1024:
    # gradients GROW through 49 batch-norms over 8 positions), then training proceeds
    m.loss_scale = 1024.0
    losses = [float(m.step(x, labels)[0][4]) for _ in range(16)]
    print("resnet f16 dynamic scale: %d steps applied of 16, %d overflows, scale %.0f" % (m.t, m.overflows, m.loss_scale))
    assert m.t >= 4 and m.t + m.overflows - 1 == 16 and m.loss_scale < 1024.0
    assert all(np.isfinite(losses)) and torch.isfinite(m.params).all() and torch.isfinite(m.m).all()


def m_lib():
    from tensorflow_yolo2_amd import _lib
    return _lib.load()


def test_yolov2_trainer_one_scaler_for_the_three_stacks():
    """the composed YOLOv2 graph has ONE loss scale, overflow flag and step counter: a train step advances the
    device step counter by one (not by three), an overflow skips the update of ALL three stacks and halves the scale of
    all three; class_argmax (the detector's class choice) equals argmax with first-index ties."""
    from tensorflow_yolo2_amd import engine as E, synthetic
    from tensorflow_yolo2_amd.yolo2_nets.yolov2 import YOLOv2Trainer
    n, size = 2, 64
    tr = YOLOv2Trainer(n, size, dtype="f16", seed=1, width_div=8)
    x = dev(synthetic.images(n, size, 3))
    lab = dev(synthetic.det_labels(n, size, size // 32, 4))
    sc = tr.opts[0].scaler
    assert all(o.scaler is sc for o in tr.opts)
    tr.step(x, lab)
    assert sc.state() == (0, 1, 0)                       # one step counted for the whole graph
    tr.step(x, lab)
    assert sc.state() == (0, 2, 0)
    before = [net.params.clone() for net in tr.nets]
    slots = [o.m.clone() for o in tr.opts]
    sc._apply(1e9)                                       # every f16 gradient overflows
    assert all(net.grad_scale == 1e9 for net in tr.nets)
    tr.step(x, lab)
    assert sc.state() == (1, 2, 1)                       # found_inf, step NOT advanced, one skipped step
    for net, p0 in zip(tr.nets, before):
        assert torch.equal(net.params, p0)
    for o, m0 in zip(tr.opts, slots):
        assert torch.equal(o.m, m0)
    sc._apply(1024.0)
    tr.step(x, lab)                                      # the host sees the overflow one step late and halves the scale
    assert sc.overflows == 1 and all(net.grad_scale == 512.0 for net in tr.nets)
    assert sc.state()[1] == 3 and all(torch.isfinite(net.params).all() for net in tr.nets)
    # class_argmax
    rng = np.random.default_rng(0)
    s = np.round(rng.uniform(0, 1, (3, 50, 20)), 1).astype(np.float32)      # rounded: ties on purpose
    best, cls = E.class_argmax(dev(s))
    np.testing.assert_array_equal(best.cpu().numpy(), s.max(-1))
    np.testing.assert_array_equal(cls.cpu().numpy(), s.argmax(-1).astype(np.int32))
