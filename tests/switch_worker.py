"""Helper of tests/test_gpu_r2_host.py::test_ab_switches_select_equivalent_paths (not a test): one detector train
step at batch 16 under whatever Y2_* switches the parent set; saves loss, gradient buffer and updated parameters."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.trainer import DetectorTrainer

out = sys.argv[1]
n, size = 16, 416
tr = DetectorTrainer(n, size, dtype="f16", seed=3)
x = torch.as_tensor(synthetic.images(n, size, 7)).cuda()
lab = torch.as_tensor(synthetic.det_labels(n, size, size // 32, 8)).cuda()
loss, ious, mask = tr.step(x, lab)
torch.cuda.synchronize()
np.savez(out, loss=loss.cpu().numpy(), grads=tr.net.grads.cpu().numpy(), params=tr.net.params.cpu().numpy(),
         ctrl=np.array(tr.opt.scaler.state()))
