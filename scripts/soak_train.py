"""Dev tool (GPU): a few hundred detector train steps on one synthetic batch (f16 with the overflow guard, bf16):
the loss must fall and the parameters stay finite -- an end-to-end health check of the kernel set."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.trainer import DetectorTrainer
for dtype, steps in (("f16", 300), ("bf16", 100)):
    tr = DetectorTrainer(64, 416, dtype=dtype, seed=0)
    x = torch.as_tensor(synthetic.images(64, 416, 1)).cuda()
    lab = torch.as_tensor(synthetic.det_labels(64, 416, 13, 2)).cuda()
    ls = []
    for i in range(steps):
        loss, ious, mask = tr.step(x, lab)
        if i % (steps // 10) == 0 or i == steps - 1:
            ls.append(float(loss[4]))
    torch.cuda.synchronize()
    st = tr.opt.scaler.state() if tr.opt.scaler is not None else None
    print(dtype, "loss", " ".join("%.3f" % v for v in ls), "scaler", st, "finite", bool(torch.isfinite(tr.net.params).all()))
