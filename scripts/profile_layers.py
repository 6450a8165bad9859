"""Dev tool (GPU): per-layer HIP-event time of the detector train step at C4 (416^2, bs 64, f16); MODEL=classifier:
the darknet19() classifier step of configs[2] (BATCH=128 SIZE=224)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import engine as E, synthetic
from tensorflow_yolo2_amd.trainer import DetectorTrainer, ClassifierTrainer
cls = os.environ.get("MODEL", "detector") == "classifier"
bs, size, steps = int(os.environ.get("BATCH", "128" if cls else "64")), int(os.environ.get("SIZE", "224" if cls else "416")), 10
x = torch.as_tensor(synthetic.images(bs, size, 1234)).cuda()
if cls:
    tr = ClassifierTrainer(bs, size, dtype=os.environ.get("DTYPE", "f16"), device="cuda:0", seed=0)
    lab = torch.as_tensor(np.random.default_rng(5).integers(0, 1000, bs).astype(np.int32)).cuda()
else:
    tr = DetectorTrainer(bs, size, dtype=os.environ.get("DTYPE", "f16"), device="cuda:0", seed=0)
    lab = torch.as_tensor(synthetic.det_labels(bs, size, size // 32, 4321)).cuda()
for _ in range(3):
    tr.step(x, lab)
torch.cuda.synchronize()
tr.net.profile_enable(1)
for _ in range(steps):
    tr.step(x, lab)
torch.cuda.synchronize()
nl = tr.net.num_layers
a = tr.net.profile_layers() / steps * 1e3
spec = list(E.CORE_SPEC) + (list(E.CLS_HEAD_SPEC) if cls else E.det_head_spec(30))
h = size
print("layer  k  cin->cout   hw   | fwd us (TF) | dgrad us (TF) | wgrad us (TF) | bn_fwd  bn_bwd")
for l, (k, ci, co, pool) in enumerate(spec):
    fl = 2.0 * bs * h * h * k * k * ci * co
    f = a[l, 0] + a[l, 1]; d = a[l, 2]; w = a[l, 3] + a[l, 4]
    tf = lambda us: fl / (us * 1e-6) / 1e12 if us > 0 else 0
    print("%2d     %d %5d->%-5d %4d | %7.1f %5.0f | %7.1f %5.0f | %7.1f %5.0f | %7.1f %7.1f" %
          (l, k, ci, co, h, f, tf(f), d, tf(d), w, tf(w), a[l, 5], a[l, 6]))
    if pool:
        h = (h + 1) // 2
print("sum", a.sum(0).round(1))
