"""Process-group setup of the data-parallel entry points (not in the reference, which is one
process on one device -- src/pascal/pascal_train_darknet.py:30,96-114).

One process per GPU, started by `python -m torch.distributed.run --nproc-per-node N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment).  The rule of this pool: the process group is created here, at the top of
main(), and no program is ever exec'ed from a process that has touched the GPU."""
import os


def init_from_env(all_ranks_on_gpu0=False, backend="nccl"):
    """-> (rank, world, local_rank, dist or None).  world == 1 (no launcher): nothing is initialised.
    all_ranks_on_gpu0: functional test of the N > 1 path on ONE GPU (two processes time-slice it): every rank uses
    cuda:0, the collectives run over gloo on the same device tensors (RCCL refuses two ranks on one device), and the
    weight-gradient side stream is off (two processes with side streams stall on each other's queue slices)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, 1, local_rank, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if all_ranks_on_gpu0:
        local_rank = 0
        os.environ["Y2_NO_WGRAD_OVERLAP"] = "1"
        if backend == "nccl":
            backend = "gloo"
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank, dist
