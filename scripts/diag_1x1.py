"""Dev tool (GPU): 1x1 shapes through the op-level y2_conv2d, forward only, REPS times each -- a target for rocprofv3 passes
(Y2_GEMM1=0 / 1 selects conv_igemm / conv_gemm1).  SHAPES="N:hw:cin:cout,..." (default: the three 1x1 shapes of configs[3])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import engine as E
reps = int(os.environ.get("REPS", "10"))
rng = np.random.default_rng(0)
shapes = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("SHAPES", "64:52:256:128,64:26:512:256,64:13:1024:512").split(",")]
for (n, hw, ci, co) in shapes:
    x = torch.as_tensor(rng.uniform(-1, 1, (n, hw, hw, ci)).astype(np.float32)).cuda()
    w = torch.as_tensor(rng.normal(0, 0.05, (1, 1, ci, co)).astype(np.float32)).cuda()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        y = E.conv2d(x, w, None, dtype="f16")
    torch.cuda.synchronize()
    for _ in range(reps):
        y = E.conv2d(x, w, None, dtype="f16")
    torch.cuda.synchronize()
print("done")
