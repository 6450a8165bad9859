"""Counterpart of src/imagenet/imagenet_predict_darknet.py: top-5 classes of one image.
    python -m tensorflow_yolo2_amd.imagenet.imagenet_predict_darknet IMAGE [--ckpt-dir DIR] [--classes synsets.txt]
As the reference (:30-65): darknet19(is_training = 0) -> tf.nn.top_k(logits, 5).  The reference feeds the resized BGR
pixels WITHOUT the loader's x / 255 * 2 - 1 (`image = cv2.resize(...)`, then straight into the placeholder, :52-58) and
prints the top-5 LOGITS as "probabilities"; --raw-pixels reproduces that, the default normalises like the training
loader does."""
import argparse

import numpy as np
import torch

from .. import engine as E
from ..img_dataset import pascal_voc
from ..utils.timer import Timer
from ..yolo2_nets import net_utils


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("image")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--ckpt-dir", default=None)
    ap.add_argument("--classes", default=None, help="one class name per line (imdb.classes)")
    ap.add_argument("--raw-pixels", action="store_true", help="feed 0..255 pixels as the reference script does")
    args = ap.parse_args(argv)
    from PIL import Image
    size = 224
    rgb = np.array(Image.open(args.image).convert("RGB"), dtype=np.uint8)
    image = pascal_voc.image_read(rgb[:, :, ::-1], size)
    if args.raw_pixels:
        image = (image + 1.0) * 0.5 * 255.0
    image = np.ascontiguousarray(image, dtype=np.float32).reshape((1, size, size, 3))
    net = E.Network(list(E.CORE_SPEC) + list(E.CLS_HEAD_SPEC), 1, size, size, dtype=args.dtype,
                    core_layers=len(E.CORE_SPEC) + len(E.CLS_HEAD_SPEC), tail=E._lib.Y2_TAIL_AVGPOOL, tail_k=size // 32,
                    training=False)
    net.init_params(0)
    if args.ckpt_dir:
        ckpts = net_utils.get_ordered_ckpts(args.ckpt_dir, 'darknet19', save_epoch=True)
        if ckpts:
            print('Restorining model snapshots from {:s}'.format(ckpts[-1]))
            net_utils.restore_variables(net, ckpts[-1], kind="classifier")
            print('Restored.')
    classes = [l.strip() for l in open(args.classes)] if args.classes else [str(i) for i in range(1000)]
    T = Timer()
    T.tic()
    logits = net.forward(torch.as_tensor(image).cuda(), False, False)
    values, idxs = torch.topk(logits.float(), 5, dim=1)             # host-side reporting, not the compute path
    probs, preds = values.cpu().numpy(), idxs.cpu().numpy()
    _time = T.toc(average=False)
    print("predictions:", [classes[i] for i in preds[0]])
    print("probabilities:", probs[0])
    return {"predictions": preds[0].tolist(), "values": probs[0], "time": _time, "network": net}


if __name__ == "__main__":
    main()
