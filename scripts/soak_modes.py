"""Dev tool (GPU): the SAME training run in every arithmetic mode -- STEPS detector train steps at configs[3]'s geometry (416x416,
batch 64) cycling over NB fixed synthetic batches, from the same initial parameters.  Prints the loss every STEPS / 20 steps per
mode, the cosine between the final parameter vectors of each mode and the exact-f32 mode's, and the loss scaler's counters.
What it shows (profiles/r06_soak_modes.txt): the reference-tolerance modes (f16x2, f16x2f) FOLLOW the exact-f32 trajectory --
the first losses agree to 1e-6, the curves stay together and the parameters end at cosine ~1 -- while a train step is a chaotic
map of its rounding (Adam's first updates are +-lr whatever the gradient's size), so bit-for-bit agreement is not the claim;
the headline f16 mode trains to the same loss level along its own trajectory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.trainer import DetectorTrainer

steps, nb, bs, size = int(os.environ.get("STEPS", "200")), int(os.environ.get("NB", "4")), 64, 416
modes = os.environ.get("MODES", "f32,f16x2,f16x2f,f16").split(",")
xs = [torch.as_tensor(synthetic.images(bs, size, 100 + i)).cuda() for i in range(nb)]
labs = [torch.as_tensor(synthetic.det_labels(bs, size, size // 32, 200 + i)).cuda() for i in range(nb)]
final, curves = {}, {}
for dt in modes:
    tr = DetectorTrainer(bs, size, dtype=dt, seed=0)
    ls = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss, ious, mask = tr.step(xs[i % nb], labs[i % nb])
        if i % max(1, steps // 20) == 0 or i == steps - 1:
            ls.append((i, float(loss[4])))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    final[dt] = tr.net.params.double().cpu().numpy()
    curves[dt] = ls
    st = tr.opt.scaler.state() if tr.opt.scaler is not None else None
    print("%-7s %6.2f ms/step (with the loss read-backs)  scaler (found_inf, steps, skipped) %s  finite %s" %
          (dt, el / steps * 1e3, st, bool(torch.isfinite(tr.net.params).all())))
    print("        loss " + " ".join("%d:%.4f" % v for v in ls))
    del tr
    torch.cuda.empty_cache()
ref = final.get("f32")
if ref is not None:
    for dt in modes:
        a = final[dt]
        print("cos(params %s, params f32) = %.8f   relative L2 distance %.3e" %
              (dt, float(a @ ref / (np.linalg.norm(a) * np.linalg.norm(ref))), float(np.linalg.norm(a - ref) / np.linalg.norm(ref))))
    for dt in modes:
        if dt == "f32":
            continue
        d = [abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(curves[dt], curves["f32"])]
        print("%-7s relative loss difference to f32 along the run: first %.2e  max %.2e  last %.2e" % (dt, d[0], max(d), d[-1]))
