"""diag: where do the bf16 first3 stacks differ from the quantised oracle (tests/test_gpu_ops.py::test_stack_backward)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nn_ref as R
from tensorflow_yolo2_amd import engine as E
from test_gpu_ops import _stack_case, _rand_params
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
spec, shape = _stack_case(True)
rng = np.random.default_rng(12)
params = _rand_params(spec, rng)
x = rng.uniform(-1, 1, shape).astype(np.float32)
net = E.Network(spec, shape[0], shape[1], shape[2], dtype=dtype, training=True)
net.load_params(params)
out = net.forward(torch.as_tensor(x).cuda(), True, True)
q = R.quantizer(dtype)
st = net.layer_statistics(0)
for given in (True, False):
    ref, caches, _ = R.run_stack(x, params, spec, True, np.float64, quant=q, first_stats=(st["mean"], st["var"]) if given else None)
    print("given stats" if given else "oracle's own stats (of the rounded conv output)")
    print("  moments: device mean vs cache mean %.2e, var %.2e" % (np.abs(st["mean"] - caches[0]["mean"]).max(), np.abs(st["var"] - caches[0]["var"]).max() / caches[0]["var"].max()))
    xin = x
    for l in range(1, len(spec)):
        a = net.debug_read(l, 0).cpu().numpy().astype(np.float64)
        r = q(caches[l]["x"]) if q else caches[l]["x"]
        d = np.abs(a - r)
        print("  layer %d input: %d of %d differ, max %.2e (max |ref| %.2e), l2 %.2e" % (l, int((d > 0).sum()), d.size, d.max(), np.abs(r).max(), np.linalg.norm(d) / np.linalg.norm(r)))
    print("  out l2 %.2e" % (np.linalg.norm(out.cpu().numpy() - ref) / np.linalg.norm(ref)))
