"""Dev tool (GPU): BASELINE.json configs[2] -- darknet19 classifier fwd+bwd + Momentum, 224x224, batch 128."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.trainer import ClassifierTrainer
bs, size = int(os.environ.get("BATCH", "128")), 224
tr = ClassifierTrainer(bs, size, dtype=os.environ.get("DTYPE", "f16"), seed=0)
x = torch.as_tensor(synthetic.images(bs, size, 1234)).cuda()
lab = torch.as_tensor(np.random.default_rng(5).integers(0, 1000, bs).astype(np.int32)).cuda()
for _ in range(3):
    loss = tr.step(x, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 20
for _ in range(steps):
    loss = tr.step(x, lab)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("C3 classifier train step: %.2f ms, %.0f images/s, loss %.4f, %.0f TFLOP/s whole step" %
      (dt * 1e3, bs / dt, float(loss[0] if hasattr(loss, "__len__") else loss), 2301.3 * bs / 128 / dt / 1e3))
