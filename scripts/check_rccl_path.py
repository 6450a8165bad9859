"""Dev tool (GPU): drive the sliced, overlapped all-reduce path of trainer.GradReducer through the REAL nccl
(= RCCL) backend on one GPU (world size 1, Y2_FORCE_DIST=1): same streams, events and API calls as the
multi-GPU run; the result must equal the plain single-process step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.trainer import DetectorTrainer
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
bs, size = int(os.environ.get("BATCH", "16")), 416
x = torch.as_tensor(synthetic.images(bs, size, 1)).cuda(); lab = torch.as_tensor(synthetic.det_labels(bs, size, 13, 2)).cuda()
res = {}
for mode in ("plain", "plain2", "rccl"):
    os.environ["Y2_FORCE_DIST"] = "1" if mode == "rccl" else "0"
    tr = DetectorTrainer(bs, size, dtype="f16", seed=0)
    losses = [float(tr.step(x, lab)[0][4]) for _ in range(4)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        tr.step(x, lab)
    torch.cuda.synchronize()
    res[mode] = (losses, tr.net.params.clone(), (time.perf_counter() - t0) / 10 * 1e3)
    print(mode, "losses", [round(v, 4) for v in losses], "%.2f ms/step" % res[mode][2])
rel = lambda a, b: float((res[a][1] - res[b][1]).norm() / res[b][1].norm())
print("param rel diff plain-plain2 (run-to-run: float atomics + Adam on ~eps gradients):", rel("plain", "plain2"))
print("param rel diff plain-rccl:", rel("plain", "rccl"))
# the first step is deterministic in its forward; later steps drift run to run, and the rccl path must not drift more
assert abs(res["plain"][0][0] - res["rccl"][0][0]) < 1e-4 * abs(res["plain"][0][0])
assert rel("plain", "rccl") < 3 * rel("plain", "plain2") + 1e-3
dist.destroy_process_group()
print("rccl path ok")
