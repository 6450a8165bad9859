"""Dev: layer-0 gradient error of the f32 detector at 224 vs the torch oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_net import _torch_ref_detector, l2err, dev
from oracle import nn_ref as R
from tensorflow_yolo2_amd import engine as E, synthetic
n, size, S = 2, 224, 7
spec = E.CORE_SPEC + E.det_head_spec(30)
params = R.init_params(spec, seed=0)
x = synthetic.images(n, size, 1234)
labels = synthetic.det_labels(n, size, S, 4321)
ref_net, ref_loss, ref_ious, ref_mask, tp = _torch_ref_detector(params, x, labels, S, size)
net = E.Network(spec, n, size, size, dtype="f32", core_layers=18, training=True)
net.load_params(params)
grid = net.forward(dev(x), True, True)
loss, ious, mask, dnet = E.yolo_loss(grid, dev(labels), 20, n, size, S, 2)
net.backward(dnet)
grads = net.export_grads()
for l in (0, 1, 7, 17, 18, 21):
    print(l, {k: float(l2err(grads[l][k], tp[l][k].grad.numpy())) for k in ("W", "gamma", "beta")})
