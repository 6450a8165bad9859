import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import nn_ref as R
from tensorflow_yolo2_amd import engine as E, synthetic
n, size, S = 2, 224, 7
spec = E.CORE_SPEC + E.det_head_spec(30)
params = R.init_params(spec, seed=0)
x = torch.as_tensor(synthetic.images(n, size, 1234)).cuda()
labels = torch.as_tensor(synthetic.det_labels(n, size, S, 4321)).cuda()
net = E.Network(spec, n, size, size, dtype="f32", core_layers=18, training=True)
net.load_params(params)
grid = net.forward(x, True, True)
loss, ious, mask, dnet = E.yolo_loss(grid, labels, 20, n, size, S, 2)
net.backward(dnet)
out = {"grid": grid.cpu().numpy(), "dnet": dnet.cpu().numpy()}
for l in (21, 20, 19, 18, 10, 1):
    out["dy%d" % l] = net.debug_read(l, 2).cpu().numpy()
    out["x%d" % l] = net.debug_read(l, 0).cpu().numpy()
    out["y%d" % l] = net.debug_read(l, 1).cpu().numpy()
f = sys.argv[1]
if os.path.exists(f):
    ref = np.load(f)
    for k in out:
        d = np.abs(out[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
        print(k, "rel max diff", d)
else:
    np.savez(f, **out)
    print("saved")
