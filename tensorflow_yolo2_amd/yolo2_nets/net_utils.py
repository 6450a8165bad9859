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
    # batch statistics, no moving-statistics update yet: the reference's UPDATE_OPS hang off train_op
    # (pascal_train_darknet.py:49-51), so a loss that is only evaluated leaves them alone
    value = net.run(training=True, update_moving=False) if source is not None else net
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


class _Optimizer:
    """tf.train.*Optimizer().minimize(loss): returns a train op; calling it applies one update (the BN
    moving-statistics update of the forward that produced `loss`, backward through that stack, then the
    optimizer).  Slots live with the variable store, so every graph sharing the scopes trains ONE state."""
    _engine_cls = None

    def __init__(self, *args):
        self.args = args
        self._opt = {}
        self._reducers = {}

    def _reducer(self, net):
        """data parallelism (trainer.GradReducer): the sliced, overlapped SUM all-reduce of `net`'s gradient buffer"""
        from ..trainer import GradReducer
        r = self._reducers.get(id(net))
        if r is None:
            r = self._reducers[id(net)] = GradReducer(net)
        return r

    def slots(self, net):
        """the engine optimizer (slot tensors) that train ops of this object apply to `net`'s variables"""
        return self._slots(net)

    def _slots(self, net):
        store = getattr(net, "store", None)
        key = id(store) if store is not None else id(net)
        opt = self._opt.get(key)
        if opt is None:
            opt = self._opt[key] = self._engine_cls(net, *self.args)
            if store is not None:
                store.optimizers[type(self).__name__] = opt
        if opt.net is not net:                 # same flat buffers, another context (batch / size)
            assert opt.net.n_params == net.n_params, "one optimizer per variable chain"
            opt.net = net
            if opt.scaler is not None:
                opt.scaler.attach(net)
        return opt

    def minimize(self, loss):
        def train_op(loss=loss):
            src = loss.source
            if src is None or src.network is None or not src.network.training:
                raise RuntimeError("loss was not produced by a training-mode NetTensor")
            net = src.network
            net.update_moving_stats()          # tf.control_dependencies(update_ops)
            from ..trainer import _dist
            if _dist() is None:
                net.backward(loss.dnet.contiguous())
                self._slots(net).step()
            else:
                # one process per GPU under torch.distributed (pascal_train_darknet.py under torchrun): the replicas'
                # gradients are summed slice by slice behind the backward pass and the update divides by the world
                # size -- slim's clone semantics (src/slim_dir/deployment/model_deploy.py:222-225,436-446)
                world = self._reducer(net).backward_and_reduce(loss.dnet.contiguous())
                self._slots(net).step(grad_mult=1.0 / world)
            store = getattr(net, "store", None)
            if store is not None:
                store.version += 1
                # the fused update re-packed THIS context's filter copies already: only the other graphs on the
                # store need the re-pack the version bump asks for (it re-reads every parameter: 193 MB per step)
                opt = self._slots(net)
                if getattr(opt, "fused_pack", False) and net.training:
                    net._seen_version = store.version
        return train_op


class AdamOptimizer(_Optimizer):
    """tf.train.AdamOptimizer().minimize(loss) (pascal_train_darknet.py:51)"""
    _engine_cls = engine.AdamOptimizer

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08):
        _Optimizer.__init__(self, learning_rate, beta1, beta2, epsilon)


class MomentumOptimizer(_Optimizer):
    """tf.train.MomentumOptimizer(0.001, 0.9).minimize(loss) (imagenet_train_darknet.py:58)"""
    _engine_cls = engine.MomentumOptimizer

    def __init__(self, learning_rate, momentum):
        _Optimizer.__init__(self, learning_rate, momentum)


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
# Two file formats, same names and contents: `.npz` (numpy) and `.ckpt` = TensorFlow's V2 checkpoint
# (`<prefix>.index` + `<prefix>.data-00000-of-00001`, utils/tf_bundle.py: read and written without TensorFlow), so
# that the checkpoints the reference's tf.train.Saver writes -- and the author's published ones (README.md:22-24)
# -- restore into these buffers, and snapshots written here load in the reference.
# ---------------------------------------------------------------------------
import glob
import os
import re

from . import darknet as _darknet
from ..utils import tf_bundle as _bundle


class _BundleSnap:
    """np.load-like view of a TF V2 checkpoint"""

    def __init__(self, prefix):
        self.r = _bundle.BundleReader(prefix)
        self.files = self.r.names()

    def __getitem__(self, name):
        return self.r.get_tensor(name)


def _ckpt_prefix(path):
    """the checkpoint prefix if `path` names a TF V2 checkpoint (prefix, .index or .meta file), else None"""
    for suffix in ("", ".index", ".meta"):
        if suffix and not path.endswith(suffix):
            continue
        prefix = path[:len(path) - len(suffix)] if suffix else path
        if _bundle.is_bundle(prefix):
            return prefix
    return None


class _DictSnap:
    """np.load-like view of a dict of arrays (a V1 checkpoint read whole)"""

    def __init__(self, d):
        self.d = d
        self.files = sorted(d)

    def __getitem__(self, name):
        return self.d[name]


def _open_snapshot(path):
    prefix = _ckpt_prefix(path)
    if prefix:
        return _BundleSnap(prefix)
    if _bundle.is_v1_checkpoint(path):          # single-file tf.train.Saver V1 checkpoint (slim's resnet_v1_50.ckpt)
        return _DictSnap(_bundle.read_checkpoint_v1(path))
    return np.load(path)


def _kind_of(network):
    n = network.num_layers
    return {18: "core", 19: "classifier", 22: "detector"}.get(n, "detector")


def _names(network, kind):
    kind = kind or _kind_of(network)
    return _darknet.variable_names("classifier" if kind in ("core", "classifier") else kind,
                                   network.spec[-1][2])[:network.num_layers]


def _slot_views(network, flat):
    """per-layer dict of numpy copies of a flat optimizer slot laid out like the parameters"""
    host = flat.detach().cpu().numpy()
    out = []
    for l in range(network.num_layers):
        o = network._offsets[l]
        shp = network._shapes(l)
        d = {}
        for i, k in enumerate(engine.PARAM_KEYS):
            n = int(np.prod(shp[k]))
            d[k] = host[o[i]:o[i] + n].reshape(shp[k]).copy()
        out.append(d)
    return out


def save_variables(network, path, kind=None, optimizer=None):
    """write what tf.train.Saver() writes for the reference graph, under the TF variable names: every
    parameter and BN moving statistic and -- with `optimizer` (an engine Adam/Momentum optimizer) -- its slots
    (`<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power`; `<var>/Momentum`), so that a resumed run
    does not restart Adam at t = 0 with empty moments (pascal_train_darknet.py:83,88,111-114)."""
    names = _names(network, kind)
    blob = {}
    for layer, nm in zip(network.export_params(), names):
        for k, tfname in nm.items():
            blob[tfname] = layer[k]
    if optimizer is not None:
        st = optimizer.export_state()
        if "m" in st:
            flat_m = torch.as_tensor(st["m"])
            flat_v = torch.as_tensor(st["v"])
            for lm, lv, nm in zip(_slot_views(network, flat_m), _slot_views(network, flat_v), names):
                for k in engine.PARAM_KEYS:
                    blob[nm[k] + "/Adam"] = lm[k]
                    blob[nm[k] + "/Adam_1"] = lv[k]
            # TF1's Adam creates beta^1 and multiplies once per apply: after t steps the variables hold beta^(t+1)
            blob["beta1_power"] = np.float32(np.float64(optimizer.b1) ** (st["t"] + 1))
            blob["beta2_power"] = np.float32(np.float64(optimizer.b2) ** (st["t"] + 1))
            blob["adam_step"] = np.int64(st["t"])       # authoritative here (not a TF variable)
        else:
            for la, nm in zip(_slot_views(network, torch.as_tensor(st["accum"])), names):
                for k in engine.PARAM_KEYS:
                    blob[nm[k] + "/Momentum"] = la[k]
    if path.endswith(".ckpt"):
        _bundle.write_bundle(path, {k: np.asarray(v) for k, v in blob.items()})
    else:
        np.savez(path, **blob)
    return sorted(blob)


_F32_MIN_NORMAL = 1.1754943508222875e-38


def adam_step_from_powers(beta1_power=None, beta2_power=None, b1=0.9, b2=0.999):
    """Step count t of a TF-written Adam checkpoint, which only holds `beta1_power` = b1^(t+1) and `beta2_power` =
    b2^(t+1) as float32 variables.  b1^(t+1) leaves the normal float32 range after ~830 steps and is 0.0 after ~1000
    (the reference saves at 40000), so the slowly decaying b2 power is used while it is a normal float32 (t < ~87000);
    beyond both, the bias corrections are 1 to float precision and any large t gives the same update."""
    for power, b in ((beta2_power, b2), (beta1_power, b1)):
        if power is None:
            continue
        v = float(np.asarray(power).reshape(-1)[0])
        if np.isfinite(v) and _F32_MIN_NORMAL <= v < 1.0 and 0.0 < b < 1.0:
            return max(0, int(round(np.log(v) / np.log(b))) - 1)
        if np.isfinite(v) and v >= 1.0:
            return 0
    have = [p for p in (beta1_power, beta2_power) if p is not None]
    return 10 ** 6 if have else 0


def restore_variables(network, path, kind=None, optimizer=None):
    """`path`: an .npz snapshot or a TF V2 checkpoint (prefix `x.ckpt`, or its `.index` / `.meta` file name).
    load the variables whose names are present in the snapshot, leave the others as they are
    (the reference restores the ImageNet-trained backbone into the detector this way, :83-103).
    A variable present with another shape raises (tf.train.Saver does too).  With `optimizer`, its slots
    are restored when the snapshot holds them.  Returns (restored names, names left untouched)."""
    names = _names(network, kind)
    snap = _open_snapshot(path)
    layers = network.export_params()
    restored, kept = [], []
    for layer, nm in zip(layers, names):
        for k, tfname in nm.items():
            if tfname in snap.files:
                if tuple(snap[tfname].shape) != tuple(layer[k].shape):
                    raise ValueError("snapshot %s: %s has shape %s, the graph expects %s" %
                                     (path, tfname, tuple(snap[tfname].shape), tuple(layer[k].shape)))
                layer[k] = snap[tfname]
                restored.append(tfname)
            else:
                kept.append(tfname)
    network.load_params(layers)
    if optimizer is not None:
        st = optimizer.export_state()
        if "m" in st and all((nm["W"] + "/Adam") in snap.files for nm in names):
            for slot, suffix in (("m", "/Adam"), ("v", "/Adam_1")):
                flat = st[slot]
                for l, nm in enumerate(names):
                    o = network._offsets[l]
                    for i, k in enumerate(engine.PARAM_KEYS):
                        a = snap[nm[k] + suffix]
                        flat[o[i]:o[i] + a.size] = a.reshape(-1)
            if "adam_step" in snap.files:
                st["t"] = int(snap["adam_step"])
            else:   # a checkpoint written by TF only has the beta powers: beta^(t+1) after t steps
                st["t"] = adam_step_from_powers(snap["beta1_power"] if "beta1_power" in snap.files else None,
                                                snap["beta2_power"] if "beta2_power" in snap.files else None,
                                                optimizer.b1, optimizer.b2)
            optimizer.load_state(st)
        elif "accum" in st and all((nm["W"] + "/Momentum") in snap.files for nm in names):
            flat = st["accum"]
            for l, nm in enumerate(names):
                o = network._offsets[l]
                for i, k in enumerate(engine.PARAM_KEYS):
                    a = snap[nm[k] + "/Momentum"]
                    flat[o[i]:o[i] + a.size] = a.reshape(-1)
            optimizer.load_state(st)
    return restored, kept


def get_ordered_ckpts(ckpt_dir, net_name='darknet19', save_epoch=True):
    """snapshot files of `net_name` in `ckpt_dir`, oldest first (reference :14-38 orders by mtime;
    the number in the name breaks ties of files written within one clock tick)"""
    tag = "epoch" if save_epoch else "iter"
    # reference :27-28: cfg.TRAIN_SNAPSHOT_PREFIX + '_' + save_interval + '_*.ckpt.meta' inside
    # cfg.get_ckpts_dir(net_name, imdb.name); here the caller passes that directory
    stem = os.path.join(ckpt_dir, "%s_%s_*" % (cfg.TRAIN_SNAPSHOT_PREFIX, tag))
    files = glob.glob(stem + ".npz") + [f[:-len(".index")] for f in glob.glob(stem + ".ckpt.index")]

    def key(f):
        m = re.search(r"_(\d+)\.(npz|ckpt)$", f)
        stamp = os.path.getmtime(f if f.endswith(".npz") else f + ".index")
        return (stamp, int(m.group(1)) if m else 0)
    return sorted(files, key=key)


def restore_darknet19_variables(network, ckpt_dir, net_name='darknet19', save_epoch=True, imagenet_ckpt_dir=None,
                                optimizer=None):
    """Reference :64-110: restore the latest snapshot and return its epoch / iteration number; with no
    snapshot, restore what an ImageNet-classifier snapshot holds (the backbone) and return 0."""
    sfiles = get_ordered_ckpts(ckpt_dir, net_name, save_epoch)
    if not sfiles:
        if imagenet_ckpt_dir:
            prior = get_ordered_ckpts(imagenet_ckpt_dir, net_name, True)
            if prior:
                restore_variables(network, prior[-1])
        return 0
    restore_variables(network, sfiles[-1], optimizer=optimizer)
    m = re.search(r"_(\d+)\.(npz|ckpt)$", sfiles[-1])
    return int(m.group(1)) if m else 0


# ---------------------------------------------------------------------------
# ResNet-50 swap (yolo2_nets/tf_resnet.py: ResNet50Yolo): the counterpart of restore_resnet_tf_variables
# (src/yolo2_nets/net_utils.py:137-219).  TF names: the backbone lives under the scope `resnet_v1_50/`, the head
# (`yolo_fc1`, `yolo_fc2`) at the root; Adam slots `<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power`.
# ---------------------------------------------------------------------------
RESNET_SCOPE = "resnet_v1_50/"
RESNET_HEAD_SCOPES = ("yolo_fc1", "yolo_fc2", "loss_layer")        # reference :177-186: excluded from the backbone restore


def resnet_tf_name(name):
    return name if name.split("/")[0] in RESNET_HEAD_SCOPES else RESNET_SCOPE + name


def save_resnet_variables(model, path, with_optimizer=True):
    """what tf.train.Saver() writes for the ResNet graph: every variable (moving statistics included) and the Adam
    slots, `.npz` or a TF V2 checkpoint (`.ckpt`)"""
    blob = {}
    if getattr(model, "graph", False):
        model._follow_ctrl()
    m_host = model.m.detach().cpu().numpy() if with_optimizer else None
    v_host = model.v.detach().cpu().numpy() if with_optimizer else None
    for (name, shape, trainable) in model.vars:
        tfname = resnet_tf_name(name)
        blob[tfname] = model.p[name].detach().cpu().numpy().copy()
        if trainable:
            t_off, n = model.offset[name]                   # the flat layout may hold slots that are not variables
            if with_optimizer:
                blob[tfname + "/Adam"] = m_host[t_off:t_off + n].reshape(shape).copy()
                blob[tfname + "/Adam_1"] = v_host[t_off:t_off + n].reshape(shape).copy()
    if with_optimizer:
        blob["beta1_power"] = np.float32(np.float64(0.9) ** (model.t + 1))       # TF1: beta^(t+1) after t applies
        blob["beta2_power"] = np.float32(np.float64(0.999) ** (model.t + 1))
        blob["adam_step"] = np.int64(model.t)
    if path.endswith(".ckpt"):
        _bundle.write_bundle(path, {k: np.asarray(v) for k, v in blob.items()})
    else:
        np.savez(path, **blob)
    return sorted(blob)


def restore_resnet_variables(model, path, exclude=(), with_optimizer=True):
    """load the variables of `path` (.npz, TF V2 prefix, or a single-file V1 checkpoint such as slim's
    resnet_v1_50.ckpt) that the graph has and whose scope is not in `exclude`; shapes must match.  Adam slots and the
    step count are restored when the file holds all of them.  Returns (restored names, names left as they were)."""
    snap = _open_snapshot(path)
    have = set(snap.files)
    restored, kept = [], []
    for (name, shape, _t) in model.vars:
        tfname = resnet_tf_name(name)
        if name.split("/")[0] in exclude or tfname not in have:
            kept.append(tfname)
            continue
        a = np.asarray(snap[tfname], np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("snapshot %s: %s has shape %s, the graph expects %s" % (path, tfname, tuple(a.shape), tuple(shape)))
        model.p[name].copy_(torch.as_tensor(a).to(model.p[name].device))
        restored.append(tfname)
    trainable = [(n, sh) for (n, sh, t) in model.vars if t]
    if with_optimizer and all(resnet_tf_name(n) + "/Adam" in have and resnet_tf_name(n) + "/Adam_1" in have
                              for n, _ in trainable):
        m_host = np.zeros(model.m.numel(), np.float32)
        v_host = np.zeros(model.v.numel(), np.float32)
        for n, sh in trainable:
            off, k = model.offset[n]
            m_host[off:off + k] = np.asarray(snap[resnet_tf_name(n) + "/Adam"], np.float32).reshape(-1)
            v_host[off:off + k] = np.asarray(snap[resnet_tf_name(n) + "/Adam_1"], np.float32).reshape(-1)
        model.m.copy_(torch.as_tensor(m_host).to(model.m.device))
        model.v.copy_(torch.as_tensor(v_host).to(model.v.device))
        if "adam_step" in have:
            t = int(snap["adam_step"])
        else:
            t = adam_step_from_powers(snap["beta1_power"] if "beta1_power" in have else None,
                                      snap["beta2_power"] if "beta2_power" in have else None)
        model.t = t
        model.ctrl[1] = t                               # the guarded update keeps its step count on the device
    if hasattr(model, "params_changed"):
        model.params_changed()                          # fused stacks re-pack their filters from the restored values
    return restored, kept


def restore_resnet_tf_variables(model, ckpt_dir, net_name='resnet50', retrain=False, detection=True, save_epoch=True,
                                new_optimizer=None, weights_path=None):
    """Reference :137-219.  No snapshot in `ckpt_dir`: the head and the optimizer keep their initial values and the
    convolutional layers are restored from the downloaded `resnet_v1_50.ckpt` under `weights_path` (cfg.WEIGHTS_PATH;
    skipped when the file is absent) -> 0.  Otherwise the latest snapshot is restored (without its optimizer slots
    when `new_optimizer` names a new optimizer) -> its iteration / epoch number.
    Difference from the reference: its `retrain` argument is accepted and IGNORED there (:137-219 never reads it); here
    `retrain=True` skips the snapshots and starts again from the downloaded backbone weights, as the name says.  The
    default (False) is the reference's behaviour."""
    sfiles = get_ordered_ckpts(ckpt_dir, net_name, save_epoch) if ckpt_dir else []
    if not sfiles or retrain:
        if weights_path:
            f = weights_path if os.path.isfile(weights_path) or _ckpt_prefix(weights_path) else \
                os.path.join(weights_path, "resnet_v1_50.ckpt")
            if os.path.isfile(f) or _ckpt_prefix(f):
                print('Initializing new variables to train from downloaded resnet50 weights')
                restore_resnet_variables(model, f, exclude=RESNET_HEAD_SCOPES if detection else (), with_optimizer=False)
        return 0
    print('Restorining model snapshots from {:s}'.format(sfiles[-1]))
    restore_resnet_variables(model, sfiles[-1], with_optimizer=new_optimizer is None)
    print('Restored.')
    m = re.search(r"_(\d+)\.(npz|ckpt)$", sfiles[-1])
    return int(m.group(1)) if m else 0
