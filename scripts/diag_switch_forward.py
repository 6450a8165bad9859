"""Dev tool (GPU): how a change of convolution kernel form (Y2_NO_CONV_RF) propagates through the forward pass of the
randomly initialised detector: relative difference of every layer's conv output between the two settings."""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:          # worker
    sys.path.insert(0, ROOT)
    import torch
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer
    n, size = 16, 416
    tr = DetectorTrainer(n, size, dtype="f16", seed=3)
    x = torch.as_tensor(synthetic.images(n, size, 7)).cuda()
    tr.net.forward(x, True, True)
    rng = np.random.default_rng(0)
    outs = {}
    for l in range(tr.net.num_layers):
        y = tr.net.debug_read(l, 1).cpu().numpy().reshape(-1)
        idx = rng.integers(0, y.size, 200000)
        outs["y%d" % l] = y[idx]
    np.savez(sys.argv[1], **outs)
    sys.exit(0)
def run(env_extra, tag):
    env = dict(os.environ); env.update(env_extra)
    out = "/tmp/fw_%s.npz" % tag
    subprocess.run([sys.executable, os.path.abspath(__file__), out], check=True, env=env, timeout=600, stderr=subprocess.DEVNULL)
    return np.load(out)
a, b = run({}, "on"), run({"Y2_NO_CONV_RF": "1"}, "off")
for l in range(22):
    ya, yb = a["y%d" % l].astype(np.float64), b["y%d" % l].astype(np.float64)
    print("layer %2d  conv output: l2 diff %.2e  max diff / max %.2e  fraction differing %.3f" %
          (l, np.linalg.norm(ya - yb) / np.linalg.norm(yb), np.abs(ya - yb).max() / np.abs(yb).max(), float((ya != yb).mean())))
