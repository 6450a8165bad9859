"""Diagnostic (GPU): per-layer error of the half-precision backward against the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nn_ref as R
from tensorflow_yolo2_amd import engine as E
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import _stack_case, _rand_params, dev

def l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
def mx(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)

for first3 in (True, False):
    spec, shape = _stack_case(first3)
    rng = np.random.default_rng(12)
    params = _rand_params(spec, rng)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    dout = rng.standard_normal((shape[0],) + tuple(R.run_stack(x, params, spec, True, np.float64)[0].shape[1:])).astype(np.float32)
    for dtype in ("f32", "f16", "bf16"):
        for gs in ((1.0,) if dtype != "f16" else (1.0, 1024.0)):
            q = R.quantizer(dtype)
            ref, caches, _ = R.run_stack(x, params, spec, True, np.float64, quant=q)
            _, rgrads = R.run_stack_backward(params, caches, dout.astype(np.float64), np.float64, quant=q, grad_scale=gs)
            net = E.Network(spec, shape[0], shape[1], shape[2], dtype=dtype, training=True, grad_scale=gs)
            net.load_params(params)
            out = net.forward(dev(x), True, True).cpu().numpy()
            net.backward(dev(dout))
            grads = net.export_grads()
            print(f"first3={first3} {dtype} gs={gs}: out max {mx(out, ref):.2e} l2 {l2(out, ref):.2e}")
            for l in range(len(spec)):
                act = net.debug_read(l, 1).cpu().numpy()
                print(f"   L{l} conv l2 {l2(act, caches[l]['h_conv']):.1e} | " + " ".join(
                    f"{k}: max {mx(grads[l][k], rgrads[l][k]):.1e} l2 {l2(grads[l][k], rgrads[l][k]):.1e}" for k in ("W", "gamma", "beta")))
