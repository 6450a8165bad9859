"""diag: are the graph-replayed ResNet steps applied or skipped? (ctrl = found_inf, steps applied, steps skipped)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.yolo2_nets.tf_resnet import ResNet50Yolo
bs, size = 32, 224
for graph in (True, False):
    m = ResNet50Yolo(bs, size, dtype="f16", device=torch.device("cuda:0"), seed=0, graph=graph)
    x = torch.as_tensor(synthetic.images(bs, size, 1234)).cuda(); lab = torch.as_tensor(synthetic.det_labels(bs, size, 7, 4321)).cuda()
    for i in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.step(x, lab)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        print("graph" if graph else "eager", i, "ctrl", m.ctrl[:3].tolist(), "loss_scale", m.loss_scale, "ms %.2f" % ms, "loss", float(out[0].sum()))
    del m
