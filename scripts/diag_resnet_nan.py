"""Dev tool (GPU): where does the first non-finite value of the scaled ResNet backward appear; max-pool 3x3/2 backward
(gather form) against torch autograd on inputs WITH ties (ReLU zeros)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from tensorflow_yolo2_amd import engine as E, synthetic
from tensorflow_yolo2_amd.yolo2_nets import tf_resnet
from oracle import resnet_ref as RR

rng = np.random.default_rng(0)
for shape in ((2, 9, 7, 5), (4, 48, 48, 16), (4, 112, 112, 64), (3, 5, 12, 3)):
    x = np.maximum(rng.standard_normal(shape), 0).astype(np.float32)          # ReLU output: ties at 0
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = RR.max_pool_3x3_s2_same(xt.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    dy = rng.standard_normal(tuple(ref.shape)).astype(np.float32)
    ref.backward(torch.tensor(dy, dtype=torch.float64))
    dx = E.max_pool_3x3_s2_backward(torch.as_tensor(x).cuda(), torch.as_tensor(dy).cuda()).cpu().numpy()
    print(shape, "finite", np.isfinite(dx).all(), "sum dx %.6f sum dy %.6f" % (dx.sum(), dy.sum()),
          "max diff vs torch (ties may route differently)", np.abs(dx - xt.grad.numpy()).max())

n, size = 4, 96
m = tf_resnet.ResNet50Yolo(n, size, dtype="f32", seed=1, blocks=RR.scaled_blocks(4), root_depth=16, fc_hidden=512)
x = torch.as_tensor(synthetic.images(n, size, 5)).cuda()
lab = torch.as_tensor(synthetic.det_labels(n, size, size // 32, 6)).cuda()
m.drop_seed = 10
grid = m.forward(x, True, update_moving=False)
print("grid finite", torch.isfinite(grid).all().item())
_l, _i, _m, dnet = E.yolo_loss(grid, lab, 20, n, size, size // 32, 2)
print("loss", _l.cpu().numpy(), "dnet finite", torch.isfinite(dnet).all().item())
m.grads.zero_()
m.backward(dnet)
torch.cuda.synchronize()
bad = []
for (name, shape, tr) in m.vars:
    if not tr:
        continue
    off, k = m.offset[name]
    g = m.grads[off:off + k]
    if not torch.isfinite(g).all():
        bad.append((name, int((~torch.isfinite(g)).sum()), k))
print("non-finite gradient tensors:", bad[:12], "of", len(bad))
