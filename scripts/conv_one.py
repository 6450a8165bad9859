"""Dev tool (GPU): run y2_conv2d (+ backward) at one layer shape a few times -- the target of a rocprofv3 counter pass.
   python scripts/conv_one.py HW CIN COUT K [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorflow_yolo2_amd import engine as E
hw, ci, co, k = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
n = int(os.environ.get("BATCH", "64"))
x = torch.rand(n, hw, hw, ci, device="cuda") * 2 - 1
w = torch.randn(k, k, ci, co, device="cuda") * 0.1
dy = torch.rand(n, hw, hw, co, device="cuda") * 2 - 1
for _ in range(reps):
    E.conv2d(x, w, None, dtype="f16")
    if os.environ.get("BWD", "1") == "1":
        E.conv2d_backward(x, w, dy, dtype="f16")
torch.cuda.synchronize()
