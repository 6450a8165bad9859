"""Counterpart of src/pascal/pascal_train_resnet.py ("use pretrained resnet50 to imitate YOLOv1"):
    python -m tensorflow_yolo2_amd.pascal.pascal_train_resnet --iters 20 [--devkit data/VOCdevkit] [--weights-path DIR]
Graph as the reference (:31-58): resnet_v1_50(input) under resnet_arg_scope -> flatten -> fully_connected(4096) ->
dropout(0.5) -> fully_connected(S*S*(5B+C)) -> reshape -> get_loss -> AdamOptimizer(0.0005), batch 4, 224 x 224.
Loop as the reference (:69-100): restore_resnet_tf_variables (latest `train_iter_<i>` snapshot of --ckpt-dir, else
the convolutional layers of the downloaded slim `resnet_v1_50.ckpt` under --weights-path), ADD_ITER more
iterations, print every 10, save every 40000.  With --devkit the batches come from img_dataset.pascal_voc through
the pinned double buffer (utils/feeder.py); without it from synthetic VOC-shaped data."""
import argparse
import os

import torch

from .. import config as cfg, synthetic
from ..utils.timer import Timer
from ..yolo2_nets import net_utils, tf_resnet


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20, help="ADD_ITER (:23; 200000 in the reference)")
    ap.add_argument("--batch", type=int, default=4)            # BATCH_SIZE = 4 (:28)
    ap.add_argument("--dtype", default="f32", help="f32 (the reference's precision) | f16 | bf16 (dynamic loss scale)")
    ap.add_argument("--ckpt-dir", default=None, help="snapshot directory (cfg.get_ckpts_dir('resnet50', imdb.name))")
    ap.add_argument("--weights-path", default=None, help="directory (or file) of slim's resnet_v1_50.ckpt (cfg.WEIGHTS_PATH)")
    ap.add_argument("--save-every", type=int, default=40000)  # :97
    ap.add_argument("--ckpt-format", default="npz", choices=("npz", "ckpt"))
    ap.add_argument("--devkit", default=None, help="VOCdevkit directory (cfg.PASCAL_PATH): feed real images")
    ap.add_argument("--image-set", default="trainval")         # pascal_voc('trainval', ...) (:30)
    ap.add_argument("--flipped", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from one HIP graph")
    ap.add_argument("--width-div", type=int, default=1, help="divide every channel count (plumbing runs and tests)")
    args = ap.parse_args(argv)
    size, B, NUM_CLASS = cfg.IMAGE_SIZE, cfg.B, 20
    S = size // 32
    kw = {}
    if args.width_div != 1:
        d = args.width_div
        kw = dict(blocks=[(n, [(dep // d, db // d, st) for (dep, db, st) in units]) for n, units in tf_resnet.BLOCKS_50],
                  root_depth=64 // d, fc_hidden=4096 // d)
    model = tf_resnet.ResNet50Yolo(args.batch, size, B=B, num_class=NUM_CLASS, dtype=args.dtype, graph=args.graph, **kw)
    feeder = None
    if args.devkit:
        from ..img_dataset.pascal_voc import pascal_voc
        from ..utils.feeder import DeviceFeeder
        imdb = pascal_voc(args.image_set, batch_size=args.batch, devkit_path=args.devkit, image_size=size, cell_size=S,
                          flipped=args.flipped)
        feeder = DeviceFeeder(lambda im, lab: imdb.get_u8(im, lab), args.batch, size, S)
    last_iter_num = 0
    if args.ckpt_dir:
        os.makedirs(args.ckpt_dir, exist_ok=True)
    last_iter_num = net_utils.restore_resnet_tf_variables(model, args.ckpt_dir, 'resnet50', save_epoch=False,
                                                          weights_path=args.weights_path)
    TOTAL_ITER = args.iters + last_iter_num
    T = Timer()
    T.tic()
    losses = []
    for i in range(last_iter_num + 1, TOTAL_ITER + 1):
        if feeder:
            image_u8, gt_labels = feeder.get()
            image = image_u8.to(torch.float32).div_(255.0).mul_(2.0).sub_(1.0)     # (x / 255) * 2 - 1 (pascal_voc.py:66)
        else:
            image = torch.as_tensor(synthetic.images(args.batch, size, i)).cuda()
            gt_labels = torch.as_tensor(synthetic.det_labels(args.batch, size, S, 1000 + i)).cuda()
        loss, ious, object_mask = model.step(image, gt_labels)
        if feeder:
            feeder.release()
            if i < TOTAL_ITER:
                feeder.prefetch()
        if i % 10 == 0 or i == TOTAL_ITER:
            losses.append(float(loss[4]))          # the only host read of the loop
            if i % 10 == 0:
                _time = T.toc(average=False)
                print('iter {:d}/{:d}, total loss: {:.3}, take {:.2}s'.format(i, TOTAL_ITER, losses[-1], _time))
                T.tic()
        if args.ckpt_dir and (i % args.save_every == 0 or i == TOTAL_ITER):
            save_path = os.path.join(args.ckpt_dir, cfg.TRAIN_SNAPSHOT_PREFIX + '_iter_' + str(i) + '.' + args.ckpt_format)
            net_utils.save_resnet_variables(model, save_path)
            print("Model saved in file: %s" % save_path)
    return {"losses": losses, "last_iter": TOTAL_ITER, "first_iter": last_iter_num + 1, "model": model}


if __name__ == "__main__":
    main()
