"""Development check of the split-operand mode (dtype "f16x2"): op-level shapes against float64 and small networks
against the exact-f32 mode.  Usage: python scripts/dev_f16x2_check.py [quick|c4|step8|step64]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _shapes import check_layer_shape  # noqa: E402


def net_vs_f32(spec, n, hw, core_layers=None, seed=0):
    """one forward + backward of a small stack in f16x2 against the exact-f32 mode on the same parameters"""
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(seed)
    params = R.init_params(spec, seed=seed)
    cin = spec[0][1]
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    res = {}
    for dt in ("f32", "f16x2"):
        net = E.Network(spec, n, hw, hw, dtype=dt, core_layers=core_layers, training=True, grad_scale=1024.0 if dt == "f16x2" else 1.0)
        net.load_params(params)
        out = net.forward(torch.as_tensor(x).cuda(), True, True).clone()
        g = np.random.default_rng(seed + 1).standard_normal(tuple(out.shape)).astype(np.float32) * 1e-3
        net.backward(torch.as_tensor(g).cuda())
        torch.cuda.synchronize()
        res[dt] = (out.cpu().numpy().astype(np.float64), net.export_grads())
    o32, g32 = res["f32"]
    o2, g2 = res["f16x2"]
    e_out = np.abs(o2 - o32).max() / np.abs(o32).max()
    worst = 0.0
    for l in range(len(spec)):
        for k in g32[l]:
            if k == "b":
                continue        # analytically zero with batch statistics: round-off noise around 0
            a, b = g2[l][k].astype(np.float64), g32[l][k].astype(np.float64)
            den = max(np.abs(b).max(), 1e-30)
            e = np.abs(a - b).max() / den
            worst = max(worst, e)
            if e > 1e-4:
                print("   layer %d %s: %.2e (max |ref| %.2e)" % (l, k, e, den))
    print("net %s n=%d hw=%d: out %.2e  worst grad %.2e" % (spec, n, hw, e_out, worst))
    return e_out, worst


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    if mode == "quick":
        for (n, name, k, cin, cout, hw) in [(2, "s1", 3, 32, 64, 16), (2, "s2", 3, 64, 128, 13), (3, "s3", 1, 128, 64, 13),
                                            (2, "s4", 3, 128, 256, 26), (2, "s5", 1, 256, 30, 7), (4, "s6", 3, 256, 512, 13),
                                            (2, "s7", 3, 32, 64, 104), (2, "s8", 3, 64, 32, 208), (8, "s9", 3, 512, 1024, 13)]:
            check_layer_shape(n, name, k, cin, cout, hw, "dev", dtype="f16x2", tol=3e-5, representable=False)
        net_vs_f32([(3, 32, 64, 1), (3, 64, 128, 0), (1, 128, 64, 0), (3, 64, 32, 0)], 2, 32)
        net_vs_f32([(3, 3, 32, 1), (3, 32, 64, 1), (3, 64, 128, 0), (1, 128, 64, 0), (3, 64, 128, 1), (1, 128, 30, 0)], 4, 64)
        net_vs_f32([(3, 3, 32, 0), (3, 32, 64, 1), (1, 64, 30, 0)], 2, 30)
    elif mode == "c4":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_c4_shapes import C4_SHAPES
        for (name, k, cin, cout, hw) in C4_SHAPES:
            t0 = time.time()
            check_layer_shape(64, name, k, cin, cout, hw, "C4", dtype="f16x2", tol=3e-5, representable=False)
            print("   %.1f s" % (time.time() - t0))
    elif mode in ("step8", "step64"):
        from test_gpu_c4_shapes import _full_detector_step_f32_vs_torch_oracle
        _full_detector_step_f32_vs_torch_oracle(int(mode[4:]), "f16x2")


def diag(spec, n, hw, seed=0):
    """layer by layer: stored activations, conv outputs, dy and gradients of f16x2 against the f32 mode"""
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(seed)
    params = R.init_params(spec, seed=seed)
    x = rng.uniform(-1, 1, (n, hw, hw, spec[0][1])).astype(np.float32)
    res = {}
    for dt in ("f32", "f16x2"):
        net = E.Network(spec, n, hw, hw, dtype=dt, training=True)
        net.load_params(params)
        out = net.forward(torch.as_tensor(x).cuda(), True, True).clone()
        g = np.random.default_rng(seed + 1).standard_normal(tuple(out.shape)).astype(np.float32) * 1e-2
        net.backward(torch.as_tensor(g).cuda())
        torch.cuda.synchronize()
        d = {"out": out.cpu().numpy(), "grads": net.export_grads()}
        for l in range(len(spec)):
            if l > 0:
                d["a%d" % l] = net.debug_read(l, 0).cpu().numpy()
            d["y%d" % l] = net.debug_read(l, 1).cpu().numpy()
            try:
                d["dy%d" % l] = net.debug_read(l, 2).cpu().numpy() / net.grad_scale
            except Exception as e:
                pass
        res[dt] = d
    r = lambda a, b: np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30)
    for l in range(len(spec)):
        line = "layer %d %s:" % (l, spec[l])
        for key in ("a%d" % l, "y%d" % l, "dy%d" % l):
            if key in res["f32"] and key in res["f16x2"]:
                line += "  %s %.2e" % (key, r(res["f16x2"][key], res["f32"][key].astype(np.float64)))
        for k in ("W", "gamma", "beta"):
            line += "  d%s %.2e" % (k, r(res["f16x2"]["grads"][l][k], res["f32"]["grads"][l][k].astype(np.float64)))
        print(line)
    print("out %.2e" % r(res["f16x2"]["out"], res["f32"]["out"].astype(np.float64)))


if len(sys.argv) > 1 and sys.argv[1] == "diag":
    diag([(3, 128, 256, 1), (3, 256, 512, 0), (1, 512, 256, 0), (3, 256, 512, 1), (3, 512, 1024, 0), (1, 1024, 30, 0)], 8, 28)
