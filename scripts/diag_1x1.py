"""Dev tool (GPU): the three 1x1 shapes of configs[3] through the op-level y2_conv2d, forward only, REPS times each -- a target
for rocprofv3 --pmc passes (Y2_GEMM1=0 / 1 selects conv_igemm / conv_gemm1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import engine as E
reps = int(os.environ.get("REPS", "10"))
rng = np.random.default_rng(0)
for (hw, ci, co) in ((52, 256, 128), (26, 512, 256), (13, 1024, 512)):
    x = torch.as_tensor(rng.uniform(-1, 1, (64, hw, hw, ci)).astype(np.float32)).cuda()
    w = torch.as_tensor(rng.normal(0, 0.05, (1, 1, ci, co)).astype(np.float32)).cuda()
    for _ in range(reps):
        y = E.conv2d(x, w, None, dtype="f16")
    torch.cuda.synchronize()
print("done")
