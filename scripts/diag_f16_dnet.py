import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import engine as E, synthetic
n, size, S = 16, 416, 13
spec = E.CORE_SPEC + E.det_head_spec(30)
x = torch.as_tensor(synthetic.images(n, size, 1234)).cuda()
labels = torch.as_tensor(synthetic.det_labels(n, size, S, 4321)).cuda()
out = {}
for dtype in ("f32", "f16"):
    net = E.Network(spec, n, size, size, dtype=dtype, core_layers=18, training=True)
    net.init_params(0)
    g = net.forward(x, True, True).clone()
    l, ious, mask, d = E.yolo_loss(g, labels, 20, n, size, S, 2)
    net.backward(d)
    out[dtype] = (g, ious.clone(), mask.clone(), d.clone(), net.debug_read(21, 2).clone(), net.debug_read(21, 0).clone(), net.debug_read(21,1).clone(), net.grad_scale)
    del net
def l2(a, b): return float((a - b).norm() / b.norm())
a, b = out["f16"], out["f32"]
print("grid l2", l2(a[0], b[0]), "max|grid|", float(b[0].abs().max()))
print("ious l2", l2(a[1], b[1]), "mask flips", int((a[2] != b[2]).sum()), "of", int(b[2].sum()))
print("dnet l2", l2(a[3], b[3]))
print("dy21 l2", l2(a[4] / a[7], b[4] / b[7]), "x21 l2", l2(a[5], b[5]), "conv21 l2", l2(a[6], b[6]))
