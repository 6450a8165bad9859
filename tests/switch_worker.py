"""Helper of tests/test_gpu_callers_and_snapshots.py::test_ab_switches_select_equivalent_paths (not a test): one detector train
step at batch 16 under whatever Y2_* switches the parent set; saves loss, gradient buffer, updated parameters and the
loss's index work (ious, object_mask).  argv[2] (optional): a .npy file with the labels to use instead of the synthetic
ones (the parent drops the object cells whose two IoUs are near-tied in the default run)."""
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
lab_np = np.load(sys.argv[2]) if len(sys.argv) > 2 else synthetic.det_labels(n, size, size // 32, 8)
lab = torch.as_tensor(lab_np).cuda()
loss, ious, mask = tr.step(x, lab)
torch.cuda.synchronize()
np.savez(out, loss=loss.cpu().numpy(), grads=tr.net.grads.cpu().numpy(), params=tr.net.params.cpu().numpy(),
         ctrl=np.array(tr.opt.scaler.state()), ious=ious.cpu().numpy(), mask=mask.cpu().numpy(),
         response=lab[..., 0].cpu().numpy())
