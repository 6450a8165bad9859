"""Dev tool (GPU): full-width detector train steps at several multi-scale sizes (f16), loss must stay finite and fall."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.trainer import MultiScaleDetectorTrainer
bs = int(os.environ.get("BATCH", "16"))
sizes = (320, 416, 544, 608)
ms = MultiScaleDetectorTrainer(bs, sizes=sizes, period=1, dtype="f16", seed=0)
data = {s: (torch.as_tensor(synthetic.images(bs, s, 10 + s)).cuda(), torch.as_tensor(synthetic.det_labels(bs, s, s // 32, 20 + s)).cuda()) for s in sizes}
for rnd in range(3):
    for s in sizes:
        x, lab = data[s]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = ms.step(x, lab)[0]
        torch.cuda.synchronize()
        print("round %d size %d: loss %.4f  %.1f ms" % (rnd, s, float(loss[4]), (time.perf_counter() - t0) * 1e3))
        assert np.isfinite(float(loss[4]))
print("params finite:", bool(torch.isfinite(ms.nets[sizes[0]].params).all()))
