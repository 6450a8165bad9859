"""Counterpart of src/pascal/pascal_detect_resnet.py ("use trained resnet50 to detect"):
    python -m tensorflow_yolo2_amd.pascal.pascal_detect_resnet IMAGE [--ckpt-dir DIR | --weights FILE]
Same sequence as the reference (:33-66): read + bilinear resize to 224 + x/255*2-1 on BGR pixels,
resnet_v1_50(is_training=False) -> flatten -> fully_connected(4096) -> fully_connected(S*S*30) (no dropout in the
detection graph) -> reshape, restore_resnet_tf_variables, run, show_yolo_detection."""
import argparse
import os

import numpy as np
import torch

from .. import config as cfg
from ..img_dataset import pascal_voc
from ..yolo2_nets import net_utils, tf_resnet


class _Imdb:
    classes = pascal_voc.CLASSES
    num_class = len(pascal_voc.CLASSES)
    name = 'voc_2007'


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("image")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--weights", default=None, help="snapshot file to restore every variable from")
    ap.add_argument("--ckpt-dir", default=None, help="directory of train_iter_* snapshots (cfg.get_ckpts_dir('resnet50', ...))")
    ap.add_argument("--no-show", action="store_true")
    ap.add_argument("--width-div", type=int, default=1, help="divide every channel count (plumbing runs and tests)")
    args = ap.parse_args(argv)
    from PIL import Image
    imdb = _Imdb()
    size, B = cfg.IMAGE_SIZE, cfg.B
    S = size // 32
    rgb = np.array(Image.open(args.image).convert("RGB"), dtype=np.uint8)
    bgr = rgb[:, :, ::-1]                                   # cv2.imread gives BGR; the reference never swaps
    image = pascal_voc.image_read(bgr, size).reshape((1, size, size, 3))
    kw = {}
    if args.width_div != 1:
        d = args.width_div
        kw = dict(blocks=[(n, [(dep // d, db // d, st) for (dep, db, st) in units]) for n, units in tf_resnet.BLOCKS_50],
                  root_depth=64 // d, fc_hidden=4096 // d)
    model = tf_resnet.ResNet50Yolo(1, size, B=B, num_class=imdb.num_class, dtype=args.dtype, **kw)
    restored = 0
    if args.weights and not (os.path.isfile(args.weights) or os.path.isfile(args.weights + ".index")):
        # the reference always restores before it runs (pascal_detect_resnet.py: saver.restore): a missing file is an error
        raise FileNotFoundError("--weights %s: no such snapshot (.npz, V1 .ckpt file or V2 .ckpt prefix)" % args.weights)
    if args.weights:
        print('Restorining model from weight file {:s}'.format(args.weights))
        names, _ = net_utils.restore_resnet_variables(model, args.weights, with_optimizer=False)
        restored = len(names)
        print('Restored.')
    elif args.ckpt_dir:
        restored = net_utils.restore_resnet_tf_variables(model, args.ckpt_dir, 'resnet50', save_epoch=False)
    if not restored:
        import warnings
        warnings.warn("pascal_detect_resnet: no variables were restored (no --weights and no snapshot in --ckpt-dir): "
                      "the boxes below come from randomly initialised variables", RuntimeWarning)
    input_data = torch.as_tensor(np.ascontiguousarray(image)).cuda()
    predicts = model.forward(input_data, is_training=False, update_moving=False, dropout=False).cpu().numpy()
    cfg.S = S
    dets = net_utils.show_yolo_detection(args.image, predicts, imdb, show=not args.no_show)
    return {"detections": dets, "predicts": predicts, "restored": restored, "model": model}


if __name__ == "__main__":
    main()
