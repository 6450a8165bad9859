"""Worker of tests/test_gpu_resnet_dp.py (started by torch.distributed.run, one process per rank, both on cuda:0, gloo on
the device tensors): data parallelism of the ResNet-50 swap (yolo2_nets/tf_resnet.py; src/pascal/pascal_train_resnet.py:37-50
with slim's clone semantics, src/slim_dir/deployment/model_deploy.py:222-225,436-446).

  1. the replicas' LOCAL gradients (stored-gradient model, no communication) are gathered and summed on the host: the
     reference of the exchange;
  2. one DP step of the fused model -- flat all-reduce of everything in front of yolo_fc1/weights, all-gather of that
     layer's operands, fused product + guarded Adam over the gathered rows -- must leave Adam's first moments equal to
     (1 - beta1) * sum / (world * loss_scale) and its second moments to (1 - beta2) * (...)^2 for EVERY variable including
     yolo_fc1/weights (linear / quadratic in the gradient: no sign chaos), within fp32 summation order;
  3. parameters, both Adam slots and the control words are bit-identical on the two ranks;
  4. an overflow on ONE rank (rank 1's images scaled out of f16's range) skips the step on BOTH ranks: nothing moves,
     skipped += 1 on both; a clean step afterwards is applied on both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensorflow_yolo2_amd import engine as E, synthetic
    from test_gpu_resnet import _build, tf_hidden
    dtype = os.environ.get("Y2_TEST_DTYPE", "f16")
    n, size, S, ls = 4, 96, 3, 16.0
    x = torch.as_tensor(synthetic.images(n, size, 100 + rank)).cuda()
    lab = torch.as_tensor(synthetic.det_labels(n, size, S, 200 + rank)).cuda()
    kw = dict(div=2, size=size, n=n, fused=(dtype != "f32"), seed=4, loss_scale=ls if dtype != "f32" else None)
    a, _, _ = _build(dtype, fuse_fc1=True, **kw)       # the DP model (f32: fuse_fc1 is off by construction)
    b, _, _ = _build(dtype, fuse_fc1=False, **kw)      # stored gradients: the reference
    assert a.drop_seed == b.drop_seed and (a.drop_seed - 4 * 7919 - 1) == rank * 104729
    lsa = a.loss_scale
    # ---- 1. local gradients of this rank, no communication
    grid = b.forward(x, True, update_moving=False)
    _loss, _i, _m, dnet = E.yolo_loss(grid, lab, b.num_class, n, size, S, b.B)
    E.check(E._lib.load().y2_scale(E._ptr(dnet), dnet.numel(), lsa, E._stream()))
    b.backward(dnet)
    torch.cuda.synchronize()
    local = {k: b.g[k].detach().clone() for k in b.g}
    total = {}
    for k in sorted(local):
        rows = [torch.empty_like(local[k]) for _ in range(world)]
        dist.all_gather(rows, local[k].contiguous())
        total[k] = sum(r.double() for r in rows) / (world * lsa)
    p0 = {k: a.p[k].detach().clone() for k in a.g}
    # ---- 2. one DP step of the fused model
    a.step(x, lab)
    torch.cuda.synchronize()
    assert int(a.ctrl[1]) == 1 or not a.guard, a.ctrl[:3].tolist()
    worst_m = worst_v = 0.0
    for k in sorted(total):
        o, c = a.offset[k]
        g = total[k].reshape(-1)
        m_ref, v_ref = 0.1 * g, 0.001 * g * g
        em = float((a.m[o:o + c].double() - m_ref).abs().max() / m_ref.abs().max().clamp_min(1e-30))
        ev = float((a.v[o:o + c].double() - v_ref).abs().max() / v_ref.abs().max().clamp_min(1e-30))
        worst_m, worst_v = max(worst_m, em), max(worst_v, ev)
        assert em < 2e-5 and ev < 4e-5, (k, em, ev)
        moved = float((a.p[k] - p0[k]).abs().max())
        assert 0.0 < moved <= 2.0 * a.lr, (k, moved)             # Adam's first step: at most lr_t ~ lr per element
    # ---- 3. replicas bit-identical

    def same(name, t):
        rows = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(rows, t.contiguous())
        assert all(torch.equal(rows[0], r) for r in rows[1:]), name + " differ between the replicas"

    for name, t in (("params", a.params), ("adam m", a.m), ("adam v", a.v), ("ctrl", a.ctrl[:3])):
        same(name, t)
    if a.guard:
        # ---- 4. overflow on rank 1 only
        before = (a.params.clone(), a.m.clone(), a.v.clone())
        s0, k0 = int(a.ctrl[1]), int(a.ctrl[2])
        a.step(x * (1e7 if rank == 1 else 1.0), lab)
        torch.cuda.synchronize()
        assert int(a.ctrl[1]) == s0 and int(a.ctrl[2]) == k0 + 1, (rank, a.ctrl[:3].tolist())
        for name, t, t0 in (("params", a.params, before[0]), ("m", a.m, before[1]), ("v", a.v, before[2])):
            assert torch.equal(t, t0), name + " moved in a skipped step"
        a.step(x, lab)
        torch.cuda.synchronize()
        assert int(a.ctrl[1]) == s0 + 1, (rank, a.ctrl[:3].tolist())
        for name, t in (("params", a.params), ("adam m", a.m), ("adam v", a.v), ("ctrl", a.ctrl[:3])):
            same(name, t)
        assert a.loss_scale == lsa * 0.5, (a.loss_scale, lsa)
    assert torch.isfinite(a.params).all()
    st = [torch.empty_like(a.state) for _ in range(world)]
    dist.all_gather(st, a.state)
    assert not torch.equal(st[0], st[1])                         # batch-norm moving statistics are per replica
    dist.barrier()
    if rank == 0:
        print("resnet dp2 ok %s fused_fc1=%s worst m %.2e v %.2e" % (dtype, a._fc1_fused_now(), worst_m, worst_v))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
