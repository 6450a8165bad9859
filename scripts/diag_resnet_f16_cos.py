"""diag: gradient cosines of the f16 fused / operator ResNet paths against the f32 fused path (scripts only)"""
import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_gpu_resnet import _build, dev
from tensorflow_yolo2_amd import engine as E, synthetic
n, size, S = 8, 96, 3
x, lab = dev(synthetic.images(n, size, 9)), dev(synthetic.det_labels(n, size, S, 10))
res = {}
for tag, kw in (("f32", dict(dtype="f32", fused=True)), ("f16 fused", dict(dtype="f16", fused=True, link=False)),
                ("f16 linked", dict(dtype="f16", fused=True)), ("f16 operators", dict(dtype="f16", fused=False))):
    dt = kw.pop("dtype")
    m, _, _ = _build(dt, div=2, size=size, n=n, seed=4, **kw)
    g = m.forward(x, True, dropout=False)
    _l, _i, _m, dnet = E.yolo_loss(g, lab, 20, n, size, S, 2)
    m.grads.zero_()
    m.backward(dnet * (64.0 if dt == "f16" else 1.0))
    res[tag] = m.export_grads()
    del m
names = ("block4/unit_3/bottleneck_v1/conv3/weights", "block3/unit_2/bottleneck_v1/conv2/weights", "block2/unit_4/bottleneck_v1/conv2/weights",
         "block2/unit_1/bottleneck_v1/shortcut/weights", "block1/unit_3/bottleneck_v1/conv2/weights", "block1/unit_1/bottleneck_v1/conv1/BatchNorm/gamma", "conv1/weights")
def cos(u, v):
    u, v = u.ravel().astype(np.float64), v.ravel().astype(np.float64)
    return float(u @ v / (np.linalg.norm(u) * np.linalg.norm(v)))
for nm in names:
    print("%-55s" % nm, " ".join("%s %.4f" % (t, cos(res[t][nm], res["f32"][nm])) for t in ("f16 fused", "f16 linked", "f16 operators")),
          "| fused~operators %.4f" % cos(res["f16 fused"][nm], res["f16 operators"][nm]))
