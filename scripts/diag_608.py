"""Dev tool (GPU): why does the f32 directional derivative miss at 608x608?  (1) step-size sweep, (2) op-level f32 checks"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from tensorflow_yolo2_amd import engine as E, synthetic
n = int(os.environ.get("BATCH", "16"))
for size in (608, 416):
    S = size // 32
    spec = list(E.CORE_SPEC) + E.det_head_spec(30)
    x = torch.as_tensor(synthetic.images(n, size, 1234 + size)).cuda()
    labels = torch.as_tensor(synthetic.det_labels(n, size, S, 4321 + size)).cuda()
    f32 = E.Network(spec, n, size, size, dtype="f32", core_layers=18, training=True)
    f32.init_params(0)
    p0 = f32.params.clone(); s0 = f32.state.clone()
    def loss_at(params):
        f32.params.copy_(params); f32.state.copy_(s0); f32.params_changed()
        grid = f32.forward(x, True, True)
        l, ious, mask, d = E.yolo_loss(grid, labels, 20, n, size, S, 2)
        return l[4].item(), d, mask.clone(), grid.clone()
    base, dnet, m0, g0 = loss_at(p0)
    f32.backward(dnet)
    g32 = f32.grads.clone()
    for frac in (2e-3, 1e-3, 5e-4, 2e-4, 1e-4):
        v = g32 * (frac * base / float((g32 * g32).sum()))
        lp, _, mp, gp = loss_at(p0 + v)
        lm, _, mm, gm = loss_at(p0 - v)
        print(size, "frac %.0e base %.4f num %.5f ana %.5f ratio %.3f  fwd %.5f bwd %.5f mask flips +%d -%d grid move %.3e" %
              (frac, base, (lp - lm) / 2, float((g32 * v).sum()), (lp - lm) / 2 / float((g32 * v).sum()), lp - base, base - lm,
               int((mp != m0).sum()), int((mm != m0).sum()), float((gp - g0).abs().max())))
    del f32
    torch.cuda.empty_cache()
