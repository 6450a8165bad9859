"""Counterpart of src/pascal/pascal_train_darknet.py's training loop:
    python -m tensorflow_yolo2_amd.pascal.pascal_train_darknet --iters 20 [--devkit data/VOCdevkit]
With --devkit the batches come from img_dataset.pascal_voc (the reference's imdb.get(), :29,98) through a pinned
double buffer on an upload stream, as uint8 pixels (utils/feeder.py); without it, from synthetic VOC-shaped data
(no dataset ships here).
Graph as the reference (:34-51): core(is_training) -> detection(30) -> reshape -> get_loss -> Adam;
loop as the reference (:83-114): resume from the latest `train_iter_<i>.npz` snapshot of --ckpt-dir
(variables AND Adam slots), run ADD_ITER more iterations, print every 10, save every --save-every."""
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
    args = ap.parse_args(argv)
    S, B, NUM_CLASS = args.size // 32, cfg.B, 20
    darknet.set_default_dtype(args.dtype)
    feeder = None
    if args.devkit:
        from ..img_dataset.pascal_voc import pascal_voc
        from ..utils.feeder import DeviceFeeder
        imdb = pascal_voc(args.image_set, batch_size=args.batch, devkit_path=args.devkit, image_size=args.size,
                          cell_size=S, flipped=args.flipped)
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
        os.makedirs(args.ckpt_dir, exist_ok=True)
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
            input_data.copy_(torch.as_tensor(synthetic.images(args.batch, args.size, i)))
            gt_labels = synthetic.det_labels(args.batch, args.size, S, 1000 + i)
        loss, ious, object_mask = net_utils.get_loss(grid_net, gt_labels, num_class=NUM_CLASS,
                                                     batch_size=args.batch, image_size=args.size, S=S, B=B,
                                                     OFFSET=cfg.yolo_grid_offset(S, B))
        optimizer.minimize(loss)()
        if feeder:
            feeder.release()
            if i < TOTAL_ITER:
                feeder.prefetch()                             # batch i+1 is assembled and uploaded while step i runs
        losses.append(float(loss))
        if i % 10 == 0:
            _time = T.toc(average=False)
            print('iter {:d}/{:d}, total loss: {:.3}, take {:.2}s'.format(i, TOTAL_ITER, losses[-1], _time))
            T.tic()
        if args.ckpt_dir and (i % args.save_every == 0 or i == TOTAL_ITER):
            save_path = os.path.join(args.ckpt_dir, cfg.TRAIN_SNAPSHOT_PREFIX + '_iter_' + str(i) + '.' + args.ckpt_format)
            net_utils.save_variables(network, save_path, optimizer=optimizer.slots(network))
            print("Model saved in file: %s" % save_path)
    return {"losses": losses, "last_iter": TOTAL_ITER, "first_iter": last_iter_num + 1, "network": network}


if __name__ == "__main__":
    main()
