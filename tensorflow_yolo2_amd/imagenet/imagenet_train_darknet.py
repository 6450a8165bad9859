"""Counterpart of src/imagenet/imagenet_train_darknet.py:
    python -m tensorflow_yolo2_amd.imagenet.imagenet_train_darknet --iters 20 [--image-list train.txt --val-list val.txt]
Graph as the reference (:46-61): darknet19(input, is_training) -> sparse_softmax_cross_entropy_with_logits ->
reduce_mean -> MomentumOptimizer(0.001, 0.9); accuracy = mean(argmax == label).  Loop as the reference (:87-135): restore
the latest `train_epoch_<e>` snapshot (variables and Momentum slots; the reference requires one -- here a fresh tree
starts from the initial values), print loss / accuracy / time every step, a validation batch with is_training = 0 every
25 steps, a snapshot at the end of every --save-every steps."""
import argparse
import os
import re

import numpy as np
import torch

from .. import config as cfg, engine as E, synthetic
from ..trainer import ClassifierTrainer
from ..utils.timer import Timer
from ..yolo2_nets import net_utils
from . import load_batch, read_image_list


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)           # ilsvrc_cls('train', batch_size=...) of the reference run
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--ckpt-dir", default=None, help="cfg.get_ckpts_dir('darknet19', imdb.name)")
    ap.add_argument("--ckpt-format", default="npz", choices=("npz", "ckpt"))
    ap.add_argument("--save-every", type=int, default=0, help="steps between snapshots (the reference: every 2 epochs)")
    ap.add_argument("--image-list", default=None, help="training images: lines `path label`")
    ap.add_argument("--val-list", default=None)
    args = ap.parse_args(argv)
    size = 224
    tr = ClassifierTrainer(args.batch, size, dtype=args.dtype)
    old_epoch = 0
    if args.ckpt_dir:
        os.makedirs(args.ckpt_dir, exist_ok=True)
        ckpts = net_utils.get_ordered_ckpts(args.ckpt_dir, 'darknet19', save_epoch=True)
        if ckpts:
            print('Restorining model snapshots from {:s}'.format(ckpts[-1]))
            net_utils.restore_variables(tr.net, ckpts[-1], kind="classifier", optimizer=tr.opt)
            print('Restored.')
            old_epoch = int(re.search(r"_(\d+)\.(npz|ckpt)$", ckpts[-1]).group(1))
    train = read_image_list(args.image_list) if args.image_list else None
    val = read_image_list(args.val_list) if args.val_list else None
    epoch = old_epoch + 1
    rng = np.random.default_rng(epoch)
    T = Timer()
    log = []
    for i in range(args.iters):
        T.tic()
        if train:
            pick = [train[j] for j in rng.integers(0, len(train), args.batch)]
            images, labels = load_batch(pick, size)
        else:
            images, labels = synthetic.images(args.batch, size, 10 * epoch + i), synthetic.cls_labels(args.batch, 77 + i)
        images, labels = torch.as_tensor(images).cuda(), torch.as_tensor(labels).cuda()
        loss, logits = tr.step(images, labels)
        loss_value, acc_value = float(loss), float(E.accuracy(logits, labels))
        _time = T.toc(average=False)
        print('epoch {:d}, iter {:d}/{:d}, training loss: {:.3}, training acc: {:.3}, take {:.2}s'
              .format(epoch, i + 1, args.iters, loss_value, acc_value, _time))
        log.append((loss_value, acc_value))
        if (i + 1) % 25 == 0 and val:
            T.tic()
            vi, vl = load_batch([val[j] for j in rng.integers(0, len(val), args.batch)], size)
            vi, vl = torch.as_tensor(vi).cuda(), torch.as_tensor(vl).cuda()
            vlogits = tr.net.forward(vi, False, False)                  # is_training: 0
            vloss, _ = E.softmax_cross_entropy(vlogits, vl, need_grad=False)
            print('###validation loss: {:.3}, validation acc: {:.3}, take {:.2}s'
                  .format(float(vloss), float(E.accuracy(vlogits, vl)), T.toc(average=False)))
        if args.ckpt_dir and ((args.save_every and (i + 1) % args.save_every == 0) or i + 1 == args.iters):
            save_path = os.path.join(args.ckpt_dir, cfg.TRAIN_SNAPSHOT_PREFIX + '_epoch_' + str(epoch) + '.' + args.ckpt_format)
            net_utils.save_variables(tr.net, save_path, kind="classifier", optimizer=tr.opt)
            print("Model saved in file: %s" % save_path)
    return {"log": log, "epoch": epoch, "trainer": tr}


if __name__ == "__main__":
    main()
