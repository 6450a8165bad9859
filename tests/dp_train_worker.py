"""Worker of tests/test_dp_train_entry.py (started by torch.distributed.run, one process per rank, both on cuda:0):
runs tensorflow_yolo2_amd.pascal.pascal_train_darknet.main() -- the data-parallel entry point -- for a few iterations
and checks, on the device tensors, that the replicas hold bit-identical variables and Adam slots afterwards while their
batch-norm moving statistics (per-replica batch statistics) differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from tensorflow_yolo2_amd.pascal import pascal_train_darknet
    argv = sys.argv[1:]
    out = pascal_train_darknet.main(argv)          # creates the process group before its first GPU call
    assert dist.is_initialized() and out["world"] == 2, out["world"]
    net, opt = out["network"], out["optimizer"]
    torch.cuda.synchronize()

    def gathered(t):
        rows = [torch.empty_like(t) for _ in range(out["world"])]
        dist.all_gather(rows, t.contiguous())
        return rows

    for name, t in (("params", net.params), ("adam m", opt.m), ("adam v", opt.v)):
        rows = gathered(t)
        assert torch.equal(rows[0], rows[1]), name + " differ between the replicas"
        assert torch.isfinite(rows[0]).all(), name
    st = gathered(net.state)
    assert not torch.equal(st[0], st[1]), "the replicas saw the same shard (moving statistics identical)"
    losses = gathered(torch.tensor(out["losses"], device="cuda"))
    assert not torch.equal(losses[0], losses[1])                 # different shards, different losses
    dist.barrier()
    if out["rank"] == 0:
        print("dp-train ok last_iter=%d losses=%s" % (out["last_iter"], ["%.4f" % v for v in out["losses"]]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
