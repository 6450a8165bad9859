"""Worker of tests/test_gpu_data_parallel.py::test_rccl_backend_runs_the_sliced_allreduce_path (a fresh child process: RCCL
and the package's streams are initialised here and nowhere else).

The only RCCL coverage one GPU allows (RCCL refuses two ranks on one device): init_process_group("nccl", world_size=1)
and Y2_FORCE_DIST=1, so that DetectorTrainer.step takes the multi-GPU call sequence -- y2_backward_marks, one event
pair per slice, the communication stream, dist.all_reduce / reduce_scatter_tensor + all_gather_into_tensor on the real
backend, the optimizer step with grad_mult = 1 / world -- and must give the SAME BITS as the single-process fused
train_op: a SUM over one rank is the identity (slim's clone semantics at num_clones = 1:
src/slim_dir/deployment/model_deploy.py:222-225,436-446)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import synthetic
    from tensorflow_yolo2_amd.trainer import DetectorTrainer, _dist

    core = [(k, ci, co, int(p)) for (k, ci, co, p) in R.scaled_spec(R.CORE_SPEC, 8)]
    head = [(3, core[-1][2], 128, 0), (3, 128, 128, 0), (3, 128, 128, 0), (1, 128, 30, 0)]
    n, size, S = 4, 96, 3
    x = torch.as_tensor(synthetic.images(n, size, 100)).cuda()
    lab = torch.as_tensor(synthetic.det_labels(n, size, S, 200)).cuda()
    for dtype in ("f32", "f16"):
        runs = {}
        for mode in ("single", "allreduce", "rs_ag"):
            os.environ["Y2_FORCE_DIST"] = "0" if mode == "single" else "1"
            os.environ["Y2_DP_STRATEGY"] = "allreduce" if mode == "single" else mode
            assert (_dist() is None) == (mode == "single")
            tr = DetectorTrainer(n, size, dtype=dtype, core_spec=core, head_spec=head, seed=0)
            assert tr.reducer.strategy == os.environ["Y2_DP_STRATEGY"] and len(tr.reducer.slices) >= 4
            losses = []
            for _ in range(2):
                loss, ious, mask = tr.step(x, lab)
                losses.append(loss.clone())
            torch.cuda.synchronize()
            runs[mode] = (tr.net.grads.clone(), tr.net.params.clone(), tr.opt.m.clone(), tr.opt.v.clone(),
                          tr.net.state.clone(), torch.stack(losses), tr.opt.t)
        for mode in ("allreduce", "rs_ag"):
            for name, a, b in zip(("grads", "params", "m", "v", "moving statistics", "losses"), runs[mode], runs["single"]):
                assert torch.equal(a, b), "%s %s: %s differ from the single-process step" % (dtype, mode, name)
            assert runs[mode][6] == runs["single"][6] == 2
        assert torch.isfinite(runs["single"][1]).all()
    dist.destroy_process_group()
    print("rccl world-1 ok")


if __name__ == "__main__":
    main()
