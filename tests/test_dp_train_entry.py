"""The data-parallel TRAIN entry point (round 6): pascal/pascal_train_darknet.py under
torch.distributed.run -- the counterpart of src/pascal/pascal_train_darknet.py:30,96-114 with one process per GPU.
  * CPU, gloo, world 2: the rank-sharded VOC batcher (src/img_dataset/pascal_voc.py:42-58 with rank / world) -- the ranks'
    epochs are disjoint, cover the list (minus the < world entries at the tail of that epoch's order), are deterministic and
    reshuffle in lockstep;
  * GPU (-m gpu), two rank processes on one GPU: the script's main() for 3 iterations from a devkit; replicas bit-identical
    afterwards; rank 0's snapshot restores into a single-process run that continues at iteration 4."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from test_voc_feed import make_devkit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _epoch_keys(imdb, epochs):
    """[(imname, flipped), ...] per epoch, as the rank's cursor hands them out"""
    out = []
    for _ in range(epochs):
        keys = []
        for _ in range(imdb.per_rank):
            g = imdb._next()
            keys.append((os.path.basename(g["imname"]), bool(g["flipped"])))
        out.append(keys)
    return out


def _batcher_worker(rank, world, port, kit, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tensorflow_yolo2_amd.img_dataset.pascal_voc import pascal_voc
    imdb = pascal_voc("trainval", batch_size=2, devkit_path=kit, image_size=64, cell_size=2, flipped=True, seed=11,
                      rank=rank, world=world)
    mine = _epoch_keys(imdb, 3)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    q.put((rank, everyone, len(imdb.gt_labels)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batcher_two_ranks_gloo(tmp_path, golden_dir):
    import torch.multiprocessing as mp
    from tensorflow_yolo2_amd.img_dataset.pascal_voc import pascal_voc
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir, copies=7)     # 7 images + flips = 14 entries... and 15 below
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batcher_worker, args=(r, world, port, kit, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, everyone, total = q.get(timeout=120)
        res[r] = (everyone, total)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0]                                  # both ranks gathered the same picture
    everyone, total = res[0]
    assert total == 14
    # the one-process batcher with the same seed walks the same shuffled lists: rank r's epoch e is its stride of them
    one = pascal_voc("trainval", batch_size=2, devkit_path=kit, image_size=64, cell_size=2, flipped=True, seed=11)
    for e in range(3):
        full = _epoch_keys(one, 1)[0]
        assert len(full) == total
        a, b = everyone[0][e], everyone[1][e]
        assert len(a) == len(b) == total // world
        assert not (set(a) & set(b)), "the ranks' shards overlap"
        assert set(a) | set(b) == set(full[:(total // world) * world]), "the shards do not cover the epoch"
        assert a == full[0::2][:len(a)] and b == full[1::2][:len(b)]     # stride, deterministic
    assert everyone[0][0] != everyone[0][1]                               # reshuffled between epochs


def test_sharded_batcher_odd_length_and_world_one(tmp_path, golden_dir):
    """an odd list at world 2 skips ONE entry per epoch (a different one each epoch); world 1 is the reference's cursor"""
    from tensorflow_yolo2_amd.img_dataset.pascal_voc import pascal_voc
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir, copies=5)
    kw = dict(batch_size=2, devkit_path=kit, image_size=64, cell_size=2, flipped=False, seed=4)
    r0, r1 = pascal_voc("trainval", rank=0, world=2, **kw), pascal_voc("trainval", rank=1, world=2, **kw)
    one = pascal_voc("trainval", **kw)
    assert r0.per_rank == r1.per_rank == 2 and one.per_rank == 5
    skipped = []
    for _ in range(6):
        full = _epoch_keys(one, 1)[0]
        a, b = _epoch_keys(r0, 1)[0], _epoch_keys(r1, 1)[0]
        assert a == full[0:4:2] and b == full[1:4:2]
        skipped.append(full[4])
    assert len(set(skipped)) > 1
    with pytest.raises(AssertionError):
        pascal_voc("trainval", rank=2, world=2, **kw)
    # get_u8 / get hand out the same images in the sharded order
    r0b = pascal_voc("trainval", rank=0, world=2, **kw)
    im, lab = r0b.get()
    assert im.shape == (2, 64, 64, 3) and lab.shape == (2, 2, 2, 25) and np.isfinite(im).all()


@pytest.mark.gpu
def test_pascal_train_script_two_ranks_on_one_gpu(tmp_path, golden_dir):
    from tensorflow_yolo2_amd.pascal import pascal_train_darknet
    from tensorflow_yolo2_amd.yolo2_nets import darknet
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir, copies=6)
    ck = str(tmp_path / "ckpts")
    common = ["--batch", "4", "--size", "64", "--dtype", "f32", "--ckpt-dir", ck, "--devkit", kit, "--flipped"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "Y2_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_train_worker.py"), "--iters", "3"] + common + \
          ["--all-ranks-on-gpu0", "--dist-backend", "gloo"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "dp-train ok last_iter=3" in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("Model saved in file") == 1              # rank 0 alone writes the snapshot
    assert os.path.exists(os.path.join(ck, "train_iter_3.npz"))
    # the snapshot of the two-replica run restores into ONE process, which trains on from iteration 4
    darknet.reset_default_graph()
    try:
        r2 = pascal_train_darknet.main(["--iters", "2"] + common)
        assert r2["world"] == 1 and r2["first_iter"] == 4 and r2["last_iter"] == 5
        assert len(r2["losses"]) == 2 and all(np.isfinite(r2["losses"]))
        snap = np.load(os.path.join(ck, "train_iter_3.npz"))
        assert int(snap["adam_step"]) == 3
    finally:
        darknet.reset_default_graph()
        darknet.set_default_dtype("f16")
