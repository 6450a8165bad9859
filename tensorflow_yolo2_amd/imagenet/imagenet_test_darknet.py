"""Counterpart of src/imagenet/imagenet_test_darknet.py: validation accuracy with per-batch timing.
    python -m tensorflow_yolo2_amd.imagenet.imagenet_test_darknet --image-list val.txt --ckpt-dir DIR [--batch 50]
darknet19(is_training = 0) -> accuracy per batch (:30-34), the latest snapshot restored (:47-51), the loop and the two
summary lines of :53-68 (the image count must be a multiple of the batch size, :21)."""
import argparse

import torch

from .. import engine as E, synthetic
from ..utils.timer import Timer
from ..yolo2_nets import net_utils
from . import load_batch, read_image_list


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--image-list", default=None)
    ap.add_argument("--batches", type=int, default=2, help="synthetic batches when no list is given")
    ap.add_argument("--batch", type=int, default=50)           # ilsvrc_cls('val', batch_size=50) (:20)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--ckpt-dir", default=None)
    args = ap.parse_args(argv)
    size = 224
    items = read_image_list(args.image_list) if args.image_list else None
    if items is not None:
        assert 0 == (len(items) % args.batch)
    total_batch = len(items) // args.batch if items is not None else args.batches
    net = E.Network(list(E.CORE_SPEC) + list(E.CLS_HEAD_SPEC), args.batch, size, size, dtype=args.dtype,
                    core_layers=len(E.CORE_SPEC) + len(E.CLS_HEAD_SPEC), tail=E._lib.Y2_TAIL_AVGPOOL, tail_k=size // 32,
                    training=False)
    net.init_params(0)
    if args.ckpt_dir:
        ckpts = net_utils.get_ordered_ckpts(args.ckpt_dir, 'darknet19', save_epoch=True)
        if ckpts:
            print('Restorining model snapshots from {:s}'.format(ckpts[-1]))
            net_utils.restore_variables(net, ckpts[-1], kind="classifier")
            print('Restored.')
    T = Timer()
    accumulated_acc = accumulated_time = 0.0
    for i in range(total_batch):
        if items is not None:
            images, labels = load_batch(items[i * args.batch:(i + 1) * args.batch], size)
        else:
            images, labels = synthetic.images(args.batch, size, i), synthetic.cls_labels(args.batch, i)
        images, labels = torch.as_tensor(images).cuda(), torch.as_tensor(labels).cuda()
        T.tic()
        logits_value = net.forward(images, False, False)
        accuracy_value = float(E.accuracy(logits_value, labels))      # the host read closes the timed region
        _time = T.toc(average=False)
        print("batch {:d}/{:d}, acc: {:3f}, time: {:2f}sec".format(i + 1, total_batch, accuracy_value, _time))
        accumulated_acc += accuracy_value
        accumulated_time += _time
    print("###########validation accuracy:", (accumulated_acc / float(total_batch)))
    print("###########average time per batch:", (accumulated_time / float(total_batch)))
    return {"accuracy": accumulated_acc / float(total_batch), "time_per_batch": accumulated_time / float(total_batch),
            "network": net}


if __name__ == "__main__":
    main()
