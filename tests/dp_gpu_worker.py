"""Worker of tests/test_gpu_data_parallel.py::test_grad_reducer_two_ranks_on_one_gpu (started by
torch.distributed.run, one process per rank, both on cuda:0).

Drives trainer.GradReducer.backward_and_reduce -- y2_backward_marks, the per-slice event pairs, the
communication stream and the collective -- at world size 2 and checks, on the device tensors:
  * the reduced gradient buffer == sum over ranks of the gradients each rank computes alone;
  * after step(): parameters (and Adam slots) bit-identical on both ranks;
  * the step == the oracle semantic: Adam on (g0 + g1) / 2 from the common initial values;
  * Y2_TEST_DTYPE=f16 (round 5): the headline arithmetic with its loss scaler -- one step where RANK 1's
    input alone overflows: the non-finite gradients reach every replica through the sum, BOTH ranks skip the step on the
    device, keep bit-identical parameters / Adam slots / loss scale, and train on together at the halved scale.
Backend: nccl (= RCCL) when two ranks may share a device, else gloo on the same device tensors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["Y2_NO_WGRAD_OVERLAP"] = os.environ.get("Y2_NO_WGRAD_OVERLAP", "1")   # two processes time-slice ONE GPU

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("Y2_TEST_BACKEND", "gloo")
    strategy = os.environ.get("Y2_DP_STRATEGY", "allreduce")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from oracle import nn_ref as R, optim_ref as O
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer

    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 128, 0), (3, 128, 128, 0), (3, 128, 128, 0), (1, 128, 30, 0)]
    n, size, S = 2, 64, 2
    x = torch.as_tensor(synthetic.images(n, size, 100 + rank)).cuda()
    lab = torch.as_tensor(synthetic.det_labels(n, size, S, 200 + rank)).cuda()
    dtype = os.environ.get("Y2_TEST_DTYPE", "f32")
    # (f16: this toy geometry -- 8 pixels per channel at the top -- overflows the default loss scale of 1024 on its own;
    #  the clean steps run at 8, the overflow below is forced through rank 1's input)
    tr = DetectorTrainer(n, size, dtype=dtype, core_spec=core, head_spec=head, seed=0,
                         grad_scale=8.0 if dtype in ("f16", "f16x2", "f16x2f") else None)
    assert len(tr.reducer.slices) >= 4, tr.reducer.slices
    p0 = tr.net.params.clone()

    # 1. this rank's own gradients, no communication
    os.environ["Y2_FORCE_DIST"] = "0"
    _, (loss, ious, mask, dnet) = tr.forward_loss(x, lab, True, True)
    tr.net.backward(dnet)
    torch.cuda.synchronize()
    g_local = tr.net.grads.clone()
    both = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(both, g_local)
    g_sum = sum(both)

    # 2. the overlapped path
    _, (loss, ious, mask, dnet) = tr.forward_loss(x, lab, True, True)
    w = tr.reducer.backward_and_reduce(dnet)
    torch.cuda.synchronize()
    assert w == world
    g_red = tr.net.grads.clone()
    err = float((g_red - g_sum).abs().max() / g_sum.abs().max())
    assert err < 1e-5, ("reduced gradients differ from the sum of the replica gradients", err)
    assert not torch.equal(both[0], both[1])                 # the shards really differed

    # 3. optimizer step: replicas stay bit-identical
    tr.opt.step(grad_mult=1.0 / world)
    torch.cuda.synchronize()
    for name, t in (("params", tr.net.params), ("m", tr.opt.m), ("v", tr.opt.v), ("grads", tr.net.grads)):
        rows = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(rows, t.contiguous())
        assert torch.equal(rows[0], rows[1]), name + " differ between the replicas"
    exp, _, _ = O.adam_step(p0.cpu().numpy(), np.zeros(p0.numel(), np.float32), np.zeros(p0.numel(), np.float32),
                            (g_sum / world).cpu().numpy(), 1)
    assert np.abs(tr.net.params.cpu().numpy() - exp).max() < 1e-6

    def same_on_all_ranks(name, t):
        rows = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(rows, t.contiguous())
        assert all(torch.equal(rows[0], r) for r in rows[1:]), name + " differ between the replicas"

    # 4. a second full step through DetectorTrainer.step (the production call sequence)
    tr.step(x, lab)
    torch.cuda.synchronize()
    rows = [torch.empty_like(tr.net.params) for _ in range(world)]
    dist.all_gather(rows, tr.net.params)
    assert torch.equal(rows[0], rows[1])
    assert torch.isfinite(tr.net.params).all()
    if dtype in ("f16", "f16x2", "f16x2f"):
        # 5. an overflow on ONE rank: rank 1's images are scaled until its first conv output leaves f16's range
        sc = tr.opt.scaler
        assert sc is not None and sc.enabled
        torch.cuda.synchronize()
        before = (tr.net.params.clone(), tr.opt.m.clone(), tr.opt.v.clone())
        _, steps0, skipped0 = sc.state()
        scale0 = sc.scale
        x_bad = x * (1e7 if rank == 1 else 1.0)
        tr.step(x_bad, lab)
        torch.cuda.synchronize()
        found, steps1, skipped1 = sc.state()
        assert found == 1 and skipped1 == skipped0 + 1 and steps1 == steps0, (rank, found, steps0, steps1, skipped0, skipped1)
        for name, t, b in (("params", tr.net.params, before[0]), ("m", tr.opt.m, before[1]), ("v", tr.opt.v, before[2])):
            assert torch.equal(t, b), name + " moved in a skipped step"
            same_on_all_ranks(name, t)
        # the host reads the flag one step late: the next (clean) step runs at the old scale and halves it afterwards
        tr.step(x, lab)
        tr.step(x, lab)
        torch.cuda.synchronize()
        found, steps2, skipped2 = sc.state()
        assert found == 0 and steps2 == steps1 + 2 and skipped2 == skipped1, (rank, found, steps2, skipped2)
        assert sc.scale == scale0 * 0.5, (rank, scale0, sc.scale)
        same_on_all_ranks("loss scale", torch.tensor([sc.scale], device="cuda"))
        same_on_all_ranks("ctrl words", sc.ctrl)
        for name, t in (("params", tr.net.params), ("m", tr.opt.m), ("v", tr.opt.v)):
            same_on_all_ranks(name, t)
            assert torch.isfinite(t).all(), name
        assert not torch.equal(tr.net.params, before[0])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("dp2 ok backend=%s strategy=%s dtype=%s slices=%d" % (backend, strategy, dtype, len(tr.reducer.slices)))


if __name__ == "__main__":
    main()
