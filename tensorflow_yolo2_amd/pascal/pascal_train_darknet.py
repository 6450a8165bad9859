"""Counterpart of src/pascal/pascal_train_darknet.py's training loop:
    python -m tensorflow_yolo2_amd.pascal.pascal_train_darknet --iters 20 [--devkit data/VOCdevkit]
With --devkit the batches come from img_dataset.pascal_voc (the reference's imdb.get(), :29,98) through a pinned
double buffer on an upload stream, as uint8 pixels (utils/feeder.py); without it, from synthetic VOC-shaped data
(no dataset ships here).
Graph as the reference (:34-51): core(is_training) -> detection(30) -> reshape -> get_loss -> Adam;
loop as the reference (:83-114): resume from the latest `train_iter_<i>.npz` snapshot of --ckpt-dir
(variables AND Adam slots), run ADD_ITER more iterations, print every 10, save every --save-every.

Data parallel (round 6; SURVEY section 8e -- the reference is one process on one device):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m tensorflow_yolo2_amd.pascal.pascal_train_darknet --iters 20 --batch 64 --size 416 [--devkit ...]
One process per GPU; --batch is PER GPU (weak scaling: the global batch is N x --batch).  Replicas start from identical
parameters (deterministic per-scope seeds, or the same snapshot), every rank reads its stride of ONE shuffled image
list (img_dataset.pascal_voc rank / world), batch-norm statistics are per replica, the train op sums the flat gradient
buffer over RCCL in backward-order slices and divides by N inside the optimizer kernel (slim's clone semantics,
src/slim_dir/deployment/model_deploy.py:222-225,436-446).  Rank 0 prints and writes the snapshots, so the batch-norm
MOVING statistics of a snapshot are rank 0's (slim takes the first clone's update ops); variables and Adam slots are
bit-identical on every rank."""
import argparse
import os

import numpy as np
import torch

from .. import config as cfg, synthetic
from ..utils.timer import Timer
from ..yolo2_nets import darknet, net_utils


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20, help="ADD_ITER (:24; 80000 in the reference)")
    ap.add_argument("--batch", type=int, default=24)        # BATCH_SIZE = 24 (:28)
    ap.add_argument("--size", type=int, default=cfg.IMAGE_SIZE)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--ckpt-dir", default=None, help="snapshot directory (cfg.get_ckpts_dir('darknet19', imdb.name))")
    ap.add_argument("--imagenet-ckpt-dir", default=None, help="classifier snapshots to take the backbone from")
    ap.add_argument("--save-every", type=int, default=40000)   # :111
    ap.add_argument("--ckpt-format", default="npz", choices=("npz", "ckpt"),
                    help="snapshot files: numpy .npz, or TensorFlow V2 checkpoints (.ckpt.index + .data, as the "
                         "reference's tf.train.Saver writes, :111-114)")
    ap.add_argument("--devkit", default=None, help="VOCdevkit directory (cfg.PASCAL_PATH): feed real images")
    ap.add_argument("--image-set", default="trainval")         # pascal_voc('trainval', ...) (:29)
    ap.add_argument("--flipped", action="store_true", help="cfg.FLIPPED: append horizontally flipped copies")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI, default) | gloo")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true", help="functional test of the N > 1 path on one GPU")
    args = ap.parse_args(argv)
    # data parallelism: the process group comes first, before anything touches the GPU
    from ..utils import dist_env
    rank, world, local_rank, dist = dist_env.init_from_env(args.all_ranks_on_gpu0, args.dist_backend)
    torch.cuda.set_device(local_rank)
    S, B, NUM_CLASS = args.size // 32, cfg.B, 20
    darknet.set_default_dtype(args.dtype)
    feeder = None
    if args.devkit:
        from ..img_dataset.pascal_voc import pascal_voc
        from ..utils.feeder import DeviceFeeder
        imdb = pascal_voc(args.image_set, batch_size=args.batch, devkit_path=args.devkit, image_size=args.size,
                          cell_size=S, flipped=args.flipped, rank=rank, world=world)
        feeder = DeviceFeeder(lambda im, lab: imdb.get_u8(im, lab), args.batch, args.size, S)
    # the placeholder (:34): uint8 BGR pixels when fed from images (the conversion x / 255 * 2 - 1 runs on the device)
    input_data = torch.empty((args.batch, args.size, args.size, 3), dtype=torch.uint8 if feeder else torch.float32,
                             device="cuda")
    core_net = darknet.darknet19_core(input_data, is_training=True)
    final_conv_layer = darknet.darknet19_detection(core_net, 5 * B + NUM_CLASS)
    grid_net = final_conv_layer.reshape([-1, S, S, 5 * B + NUM_CLASS])
    optimizer = net_utils.AdamOptimizer()
    network = grid_net.build(training=True)
    last_iter_num = 0
    if args.ckpt_dir:
        if rank == 0:
            os.makedirs(args.ckpt_dir, exist_ok=True)
        if dist is not None:
            dist.barrier()                                    # every rank restores from the same files
        last_iter_num = net_utils.restore_darknet19_variables(
            network, args.ckpt_dir, net_name='darknet19', save_epoch=False,
            imagenet_ckpt_dir=args.imagenet_ckpt_dir, optimizer=optimizer.slots(network))
    TOTAL_ITER = args.iters + last_iter_num
    T = Timer()
    T.tic()
    losses = []
    for i in range(last_iter_num + 1, TOTAL_ITER + 1):
        if feeder:
            image, gt_labels = feeder.get()                   # device tensors; this stream waits for their upload
            input_data.copy_(image)                           # device-to-device (33 MB at 64 x 416^2: ~15 us)
        else:
            # (synthetic shards: rank r of world w draws seed i * w + r -- world 1 keeps the seeds of the one-process run)
            input_data.copy_(torch.as_tensor(synthetic.images(args.batch, args.size, i * world + rank)))
            gt_labels = synthetic.det_labels(args.batch, args.size, S, 1000 + i * world + rank)
        loss, ious, object_mask = net_utils.get_loss(grid_net, gt_labels, num_class=NUM_CLASS,
                                                     batch_size=args.batch, image_size=args.size, S=S, B=B,
                                                     OFFSET=cfg.yolo_grid_offset(S, B))
        optimizer.minimize(loss)()
        if feeder:
            feeder.release()
            if i < TOTAL_ITER:
                feeder.prefetch()                             # batch i+1 is assembled and uploaded while step i runs
        losses.append(float(loss))
        if i % 10 == 0 and rank == 0:
            _time = T.toc(average=False)
            print('iter {:d}/{:d}, total loss: {:.3}, take {:.2}s'.format(i, TOTAL_ITER, losses[-1], _time))
            T.tic()
        if args.ckpt_dir and rank == 0 and (i % args.save_every == 0 or i == TOTAL_ITER):
            save_path = os.path.join(args.ckpt_dir, cfg.TRAIN_SNAPSHOT_PREFIX + '_iter_' + str(i) + '.' + args.ckpt_format)
            net_utils.save_variables(network, save_path, optimizer=optimizer.slots(network))
            print("Model saved in file: %s" % save_path)
    if dist is not None:
        dist.barrier()                                        # the last snapshot is on disk when any rank returns
    return {"losses": losses, "last_iter": TOTAL_ITER, "first_iter": last_iter_num + 1, "network": network,
            "optimizer": optimizer.slots(network), "rank": rank, "world": world}


if __name__ == "__main__":
    main()
