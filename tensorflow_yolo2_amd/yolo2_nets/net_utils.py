"""Drop-in counterparts of the hot-path half of src/yolo2_nets/net_utils.py:
    get_iou(boxes1, boxes2, scope='iou')                                           (:222-260)
    get_loss(net, labels, num_class, batch_size, image_size, S, B, OFFSET,
             scope='loss_layer') -> (loss, ious, object_mask)                      (:263-372)
    show_yolo_detection(image_path, predict_output, imdb, object_thresh=0.5)       (:375-439)
All arithmetic runs in the fused HIP kernels of csrc/loss.hip through the C ABI."""
import numpy as np
import torch

from .. import config as cfg
from .. import engine
from .darknet import NetTensor


def get_iou(boxes1, boxes2, scope='iou'):
    """boxes: [..., 4] = (x_center, y_center, w, h) device tensors -> IoU [...]."""
    return engine.get_iou(boxes1, boxes2)


class Loss(torch.Tensor):
    """The scalar `loss` tensor of the reference, remembering what tf.gradients would need."""
    pass


def get_loss(net, labels, num_class, batch_size, image_size, S, B, OFFSET, scope='loss_layer'):
    """Returns (loss, ious, object_mask) like the reference.  `loss` is a 0-d tensor carrying
    `.parts` (class, object, noobject, coord losses), `.dnet` (d loss / d net) and `.source`
    (the NetTensor it came from) so that an optimizer's minimize(loss) can run the backward."""
    expect = cfg.yolo_grid_offset(S, B)
    if np.asarray(OFFSET).shape != expect.shape or not (np.asarray(OFFSET) == expect).all():
        raise ValueError("OFFSET must be the [S,S,B] column-index grid of config.py:40-42")
    source = net if isinstance(net, NetTensor) else None
    value = net.run(training=True) if source is not None else net
    if not torch.is_tensor(labels):
        labels = torch.as_tensor(np.asarray(labels, np.float32))
    labels = labels.to(value.device, torch.float32)
    loss5, ious, mask, dnet = engine.yolo_loss(value, labels, num_class, batch_size, image_size, S, B,
                                               need_grad=True, lambda_coord=float(cfg.LAMBDA_COORD),
                                               lambda_noobj=float(cfg.LAMBDA_NOOBJ))
    loss = loss5[4].as_subclass(Loss)
    loss.parts = dict(class_loss=loss5[0], object_loss=loss5[1], noobject_loss=loss5[2], coord_loss=loss5[3])
    loss.dnet = dnet
    loss.source = source
    return loss, ious, mask


class AdamOptimizer:
    """tf.train.AdamOptimizer().minimize(loss) (pascal_train_darknet.py:51): returns a train op;
    calling it applies one update (backward through the stack that produced `loss`, then Adam)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08):
        self.args = (learning_rate, beta1, beta2, epsilon)
        self._opt = {}

    def minimize(self, loss):
        def train_op(loss=loss):
            src = loss.source
            if src is None or src.network is None or not src.network.training:
                raise RuntimeError("loss was not produced by a training-mode NetTensor")
            net = src.network
            net.backward(loss.dnet.contiguous())
            opt = self._opt.get(id(net))
            if opt is None:
                opt = self._opt[id(net)] = engine.AdamOptimizer(net, *self.args)
            opt.step()
        return train_op


def decode_yolo_detection(predict_output, im_w, im_h, num_class, S=cfg.S, B=cfg.B, object_thresh=0.5):
    """The arithmetic of show_yolo_detection (:393-421) on the GPU; returns the reference's
    (upper_left_x, upper_left_y, w, h, class_index, confidence, c, r, i) tuples in loop order."""
    if not torch.is_tensor(predict_output):
        predict_output = torch.as_tensor(np.asarray(predict_output, np.float32)).cuda()
    return engine.decode_detections(predict_output, S, B, num_class, im_w, im_h, object_thresh)


def show_yolo_detection(image_path, predict_output, imdb, object_thresh=0.5, show=True):
    """Compute bounding boxes from the yolo detection network prediction, print them like the
    reference and (when matplotlib + PIL are importable and show=True) draw them."""
    from PIL import Image
    im = np.array(Image.open(image_path), dtype=np.uint8)
    im_h, im_w = im.shape[0], im.shape[1]
    dets = decode_yolo_detection(predict_output, im_w, im_h, imdb.num_class, cfg.S, cfg.B, object_thresh)
    for (ulx, uly, w, h, cls, conf, _c, _r, _i) in dets:
        print("predicted bounding boxes: ({:d}, {:d}), width:{:d}, height:{:d}".format(ulx, uly, w, h))
    if show:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            import matplotlib.patches as patches
            fig, ax = plt.subplots(1)
            ax.imshow(im)
            for (ulx, uly, w, h, cls, conf, _c, _r, _i) in dets:
                ax.add_patch(patches.Rectangle((ulx, uly), w, h, linewidth=1, edgecolor='r', facecolor='none'))
                ax.text(ulx, uly, imdb.classes[int(cls)] + ":" + str(np.float32(conf)), color='r')
            fig.savefig("yolo_detection.png")
            plt.close(fig)
        except ImportError:
            pass
    return dets


# ---------------------------------------------------------------------------
# snapshots keyed by the reference's TF variable names (the counterpart of
# get_ordered_ckpts / restore_darknet19_variables, src/yolo2_nets/net_utils.py:14-110).
# TF checkpoint FILES are out of scope (no TensorFlow here); the name map is what a converter needs,
# and these .npz snapshots carry exactly the names tf.train.Saver would write.
# ---------------------------------------------------------------------------
import glob
import os
import re

from . import darknet as _darknet


def _kind_of(network):
    n = network.num_layers
    return {18: "core", 19: "classifier", 22: "detector"}.get(n, "detector")


def save_variables(network, path, kind=None):
    """write every parameter and BN moving statistic of `network` under its TF variable name"""
    kind = kind or _kind_of(network)
    names = _darknet.variable_names("classifier" if kind in ("core", "classifier") else kind,
                                    network.spec[-1][2])[:network.num_layers]
    blob = {}
    for layer, nm in zip(network.export_params(), names):
        for k, tfname in nm.items():
            blob[tfname] = layer[k]
    np.savez(path, **blob)
    return sorted(blob)


def restore_variables(network, path, kind=None):
    """load the variables whose names are present in the snapshot, leave the others as they are
    (the reference restores the ImageNet-trained backbone into the detector this way, :83-103).
    Returns (restored names, names left untouched)."""
    kind = kind or _kind_of(network)
    names = _darknet.variable_names("classifier" if kind in ("core", "classifier") else kind,
                                    network.spec[-1][2])[:network.num_layers]
    snap = np.load(path)
    layers = network.export_params()
    restored, kept = [], []
    for layer, nm in zip(layers, names):
        for k, tfname in nm.items():
            if tfname in snap.files and tuple(snap[tfname].shape) == tuple(layer[k].shape):
                layer[k] = snap[tfname]
                restored.append(tfname)
            else:
                kept.append(tfname)
    network.load_params(layers)
    return restored, kept


def get_ordered_ckpts(ckpt_dir, net_name='darknet19', save_epoch=True):
    """snapshot files of `net_name` in `ckpt_dir`, oldest first (reference :14-38 orders by mtime)"""
    tag = "epoch" if save_epoch else "iter"
    files = [f for f in glob.glob(os.path.join(ckpt_dir, "%s_%s_*.npz" % (net_name, tag)))]
    return sorted(files, key=os.path.getmtime)


def restore_darknet19_variables(network, ckpt_dir, net_name='darknet19', save_epoch=True, imagenet_ckpt_dir=None):
    """Reference :64-110: restore the latest snapshot and return its epoch / iteration number; with no
    snapshot, restore what an ImageNet-classifier snapshot holds (the backbone) and return 0."""
    sfiles = get_ordered_ckpts(ckpt_dir, net_name, save_epoch)
    if not sfiles:
        if imagenet_ckpt_dir:
            prior = get_ordered_ckpts(imagenet_ckpt_dir, net_name, True)
            if prior:
                restore_variables(network, prior[-1])
        return 0
    restore_variables(network, sfiles[-1])
    m = re.search(r"_(\d+)\.npz$", sfiles[-1])
    return int(m.group(1)) if m else 0
