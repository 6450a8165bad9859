"""Dev tool (GPU): the bottleneck-shaped stack of tests/test_gpu_kernel_policies.py, every gradient's error printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch.nn.functional as F
from oracle import nn_ref as R
from tensorflow_yolo2_amd import engine as E
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
for dtype in ("f32", "f16", "bf16"):
  for slopes in ([0.0, 0.0, 1.0], [0.1, 0.1, 0.1], [1.0, 1.0, 1.0], [0.0, 0.1, 0.1], [0.1, 0.0, 0.1]):
    rng = np.random.default_rng(3)
    N, hw, cin, db, depth = 4, 12, 64, 32, 128
    spec = [(1, cin, db, 0), (3, db, db, 0), (1, db, depth, 0)]
    net = E.Network(spec, N, hw, hw, dtype=dtype, training=True, grad_scale=1.0)
    net.set_layer_options(slopes, 1e-5, 0.997, zero_bias_grad=True)
    params = R.init_params(spec, seed=8)
    for p in params:
        p["b"] = np.zeros_like(p["b"])
        p["gamma"] = rng.uniform(0.6, 1.4, p["gamma"].shape).astype(np.float32)
        p["beta"] = rng.uniform(-0.3, 0.3, p["beta"].shape).astype(np.float32)
    net.load_params(params)
    x = rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32)
    out = net.forward(torch.as_tensor(x).cuda(), True, True, update_moving=True).clone()
    dout = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    dx = net.backward_input(torch.as_tensor(dout).cuda())
    g = net.export_grads()
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tp = [{k: torch.tensor(p[k], dtype=torch.float64, requires_grad=True) for k in ("W", "gamma", "beta")} for p in params]
    h = xt.permute(0, 3, 1, 2)
    hs = []
    for (k, _ci, _co, _p), p, s in zip(spec, tp, slopes):
        h = F.conv2d(h, p["W"].permute(3, 2, 0, 1), padding=k // 2)
        mean, var = h.mean((0, 2, 3)), h.var((0, 2, 3), unbiased=False)
        z = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * p["gamma"][None, :, None, None] + p["beta"][None, :, None, None]
        h = torch.maximum(s * z, z)
        h.retain_grad(); hs.append(h)
    ref = h.permute(0, 2, 3, 1)
    ref.backward(torch.tensor(dout, dtype=torch.float64))
    msg = "%s slopes %s: fwd %.1e dx %.1e" % (dtype, slopes, rel(out.cpu().numpy(), ref.detach().numpy()), rel(dx.cpu().numpy(), xt.grad.numpy()))
    for l in range(3):
        msg += " | L%d dW %.1e dg %.1e db %.1e" % (l, rel(g[l]["W"], tp[l]["W"].grad.numpy()), rel(g[l]["gamma"], tp[l]["gamma"].grad.numpy()), rel(g[l]["beta"], tp[l]["beta"].grad.numpy()))
    for l in (1, 2):
        dy = net.debug_read(l, 2).cpu().numpy()
        msg += " | dy%d max %.2e" % (l, np.abs(dy).max())
    print(msg)
