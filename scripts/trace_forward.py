"""Dev tool (GPU): the detection FORWARD alone (no per-launch events) for a kernel trace:
  rocprofv3 --kernel-trace --stats -d gpurun_out/trace_fwd -- python3 scripts/trace_forward.py
BATCH x SIZE^2 from the environment (default 1 x 224^2: configs[0]); 50 forwards after 5 warm-up ones."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorflow_yolo2_amd import engine as E, synthetic
bs, size = int(os.environ.get("BATCH", "1")), int(os.environ.get("SIZE", "224"))
spec = list(E.CORE_SPEC) + E.det_head_spec(30)
net = E.Network(spec, bs, size, size, dtype=os.environ.get("DTYPE", "f16"), core_layers=18, training=False)
net.init_params(0)
x = torch.as_tensor(synthetic.images(bs, size, 1234)).cuda()
for _ in range(5):
    net.forward(x, False, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    net.forward(x, False, True)
torch.cuda.synchronize()
print("ms per forward: %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
