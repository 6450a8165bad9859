import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nn_ref as R
from tensorflow_yolo2_amd import engine as E, synthetic
n, size = 2, 224
spec = E.CORE_SPEC + E.det_head_spec(30)
params = R.init_params(spec, seed=0)
x = torch.as_tensor(synthetic.images(n, size, 1234)).cuda()
net = E.Network(spec, n, size, size, dtype="f32", core_layers=18, training=True)
net.load_params(params)
net.forward(x, True, True)
a = net.debug_read(1, 0).cpu().numpy()
y = net.debug_read(0, 1).cpu().numpy()
# reference BN + leaky + pool from y
m = y.reshape(-1, 32).mean(0); v = y.reshape(-1, 32).var(0)
z = (y - m) / np.sqrt(v + 1e-3)
act = np.maximum(0.1 * z, z)
p = act.reshape(n, size // 2, 2, size // 2, 2, 32).max(axis=(2, 4))
print("max abs diff pooled vs numpy-from-y:", np.abs(a - p).max(), "per-channel max:", np.abs(a - p).reshape(-1, 32).max(0).round(4))
