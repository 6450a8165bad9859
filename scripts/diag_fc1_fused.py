"""diag: which parameters differ between the fused-fc1 and the stored-gradient ResNet models, per step (scripts only)"""
import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_gpu_resnet import _build, dev
from tensorflow_yolo2_amd import synthetic
n, size, S = 8, 96, 3
mode = sys.argv[1] if len(sys.argv) > 1 else "ab"
a, _, _ = _build("f16", div=2, size=size, n=n, fused=True, seed=4, fuse_fc1=(mode[0] == "a"))
b, _, _ = _build("f16", div=2, size=size, n=n, fused=True, seed=4, fuse_fc1=(mode[1] == "a"))
x, lab = dev(synthetic.images(n, size, 9)), dev(synthetic.det_labels(n, size, S, 10))
for it in range(5):
    if it == 2:
        a.loss_scale = b.loss_scale = 2.0 ** 40
    la, lb = a.step(x, lab), b.step(x, lab)
    torch.cuda.synchronize()
    if it == 2:
        a.loss_scale = b.loss_scale = 64.0
    bad = [(name, float((a.p[name] - b.p[name]).abs().max())) for name in a.p if not torch.equal(a.p[name], b.p[name])]
    gbad = [(name, float((a.g[name] - b.g[name]).abs().max())) for name in a.g
            if name != "yolo_fc1/weights" and not ((a.g[name] == b.g[name]) | (a.g[name].isnan() & b.g[name].isnan())).all()]
    print("it", it, "ctrl", a.ctrl[:3].tolist(), b.ctrl[:3].tolist(), "params differing:", len(bad), bad[:3],
          "grads differing:", len(gbad), gbad[:3])
