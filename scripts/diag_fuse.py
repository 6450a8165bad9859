"""Dev: compare gradients of a 2-layer net with and without the fused BN-backward reduce."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E
    k, cin, cout, hw, pool = [int(v) for v in sys.argv[2:7]]
    N = 64
    rng = np.random.default_rng(1)
    spec = [(k, cin, cout, pool), (1, cout, 32, 0)]
    net = E.Network(spec, N, hw, hw, dtype="f16", training=True, grad_scale=1.0)
    params = R.init_params(spec, seed=4)
    net.load_params(params)
    x = rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float16).astype(np.float32)
    out = net.forward(torch.as_tensor(x).cuda(), True, True)
    dout = rng.uniform(-1, 1, tuple(out.shape)).astype(np.float32)
    res = []
    worst = 0.0
    for rep in range(int(os.environ.get("REPS", "2"))):
        net.backward(torch.as_tensor(dout).cuda())
        g = net.export_grads()
        if rep < 2:
            res.append(g)
        else:
            d = float(np.abs(g[0]["W"] - res[0][0]["W"]).max() / np.abs(res[0][0]["W"]).max())
            worst = max(worst, d)
            if d > 1e-4:
                print("REP", rep, "dW0 deviates", d, flush=True)
    print("stress", sys.argv[2:7], "worst deviation over reps", worst, flush=True)
    np.savez(sys.argv[7], W0a=res[0][0]["W"], W0b=res[1][0]["W"], W1=res[0][1]["W"], g0=res[0][0]["gamma"], b0=res[0][0]["beta"],
             dy0=net.debug_read(0, 2).cpu().numpy()[:4])
    sys.exit(0)
for shape in ("3 64 128 104 0", "3 128 256 52 1", "3 512 1024 13 0"):
    outs = {}
    for mode in ("fuse", "nofuse"):
        env = dict(os.environ)
        if mode == "nofuse":
            env["Y2_NO_BNBWD_FUSE"] = "1"
        f = "/tmp/diag_%s.npz" % mode
        subprocess.run([sys.executable, __file__, "child"] + shape.split() + [f], env=env, check=True)
        outs[mode] = np.load(f)
    a, b = outs["fuse"], outs["nofuse"]
    rel = lambda u, v: float(np.abs(u - v).max() / max(np.abs(v).max(), 1e-30))
    print(shape, "| fuse vs nofuse: W0 %.2e W1 %.2e gamma0 %.2e beta0 %.2e dy0 %.2e | repeat (fuse) W0 %.2e  repeat (nofuse) W0 %.2e" %
          (rel(a["W0a"], b["W0a"]), rel(a["W1"], b["W1"]), rel(a["g0"], b["g0"]), rel(a["b0"], b["b0"]), rel(a["dy0"], b["dy0"]),
           rel(a["W0b"], a["W0a"]), rel(b["W0b"], b["W0a"])))
