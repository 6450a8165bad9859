"""Dev tool (GPU): per-layer HIP-event time of the detection FORWARD (core with moving statistics, head with batch
statistics: pascal_detect_darknet.py:41-43) at BATCH x SIZE^2 (default 1 x 224^2: configs[0])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import engine as E, synthetic
bs, size, steps = int(os.environ.get("BATCH", "1")), int(os.environ.get("SIZE", "224")), 20
spec = list(E.CORE_SPEC) + E.det_head_spec(30)
net = E.Network(spec, bs, size, size, dtype=os.environ.get("DTYPE", "f16"), core_layers=18, training=False)
net.init_params(0)
x = torch.as_tensor(synthetic.images(bs, size, 1234)).cuda()
for _ in range(3):
    net.forward(x, False, True)
torch.cuda.synchronize()
net.profile_enable(1)
for _ in range(steps):
    net.forward(x, False, True)
torch.cuda.synchronize()
a = net.profile_layers() / steps * 1e3
h = size
print("layer  k  cin->cout   hw   | conv us  (GB/s of filter) | bn us")
for l, (k, ci, co, pool) in enumerate(spec):
    f = a[l, 0] + a[l, 1]
    wbytes = k * k * ci * co * 2
    print("%2d     %d %5d->%-5d %4d | %7.1f  %8.0f | %7.1f" % (l, k, ci, co, h, f, wbytes / (f * 1e-6) / 1e9 if f > 0 else 0, a[l, 5]))
    if pool:
        h = (h + 1) // 2
print("sum", a.sum(0).round(1), "total us", a.sum().round(1))
