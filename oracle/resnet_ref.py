"""CPU restatement (PyTorch autograd, float64) of the reference's ResNet-50 backbone swap -- TEST INFRASTRUCTURE
ONLY (never imported by the product path).

Follows, line by line:
  * src/yolo2_nets/tf_resnet.py:12-32            resnet_v1_50 block table: (depth, bottleneck depth, stride) units
        block1 [(256,64,1)]*2 + [(256,64,2)], block2 [(512,128,1)]*3 + [(512,128,2)],
        block3 [(1024,256,1)]*5 + [(1024,256,2)], block4 [(2048,512,1)]*3, global_pool=False
  * src/slim_dir/nets/resnet_v1.py:68-112          bottleneck: shortcut (subsample or 1x1 stride-s conv, no activation),
        conv1 1x1 + BN + ReLU, conv2 = conv2d_same(3, stride) + BN + ReLU, conv3 1x1 + BN, relu(shortcut + residual)
  * src/slim_dir/nets/resnet_v1.py:185-199         root: conv2d_same(64, 7, stride 2) + BN + ReLU, max_pool2d 3x3/2 'SAME'
  * src/slim_dir/nets/resnet_utils.py:60-122       subsample = 1x1 max pool with stride; conv2d_same = explicit
        padding (k-1)//2 before, the rest after, then a VALID strided convolution
  * src/slim_dir/nets/resnet_utils.py:230-257      resnet_arg_scope: batch_norm decay 0.997, epsilon 1e-5, scale=True;
        convolutions without biases (normalizer_fn set)
  * src/pascal/pascal_train_resnet.py:37-50        flatten -> fully_connected 4096 (ReLU) -> dropout 0.5 ->
        fully_connected S*S*(5B+C) (slim default activation: ReLU) -> reshape [-1,S,S,5B+C] -> get_loss
Pinning: conv2d_same / subsample are the slim KATs this repo already holds (tests/test_oracle.py, from
src/slim_dir/nets/resnet_v1_test.py:58-152); batch norm, max pool and the fully connected layers are unpinned by
the reference (TF semantics restated from documentation), as for the Darknet path.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # resnet_utils.py:232
BN_DECAY = 0.997       # resnet_utils.py:231

BLOCKS_50 = [("block1", [(256, 64, 1)] * 2 + [(256, 64, 2)]),
             ("block2", [(512, 128, 1)] * 3 + [(512, 128, 2)]),
             ("block3", [(1024, 256, 1)] * 5 + [(1024, 256, 2)]),
             ("block4", [(2048, 512, 1)] * 3)]


def scaled_blocks(div):
    """the same topology at 1/div width (tests)"""
    return [(name, [(d // div, b // div, s) for (d, b, s) in units]) for (name, units) in BLOCKS_50]


def param_list(blocks, root_depth=64, fc_hidden=4096, fc_out=1470, feat_hw=7):
    """ordered (name, shape) of every variable, as slim creates them (weights HWIO; BN gamma/beta + moving stats)"""
    out = [("conv1/weights", (7, 7, 3, root_depth))] + _bn_names("conv1", root_depth)
    cin = root_depth
    for bname, units in blocks:
        for i, (depth, db, stride) in enumerate(units):
            p = "%s/unit_%d/bottleneck_v1/" % (bname, i + 1)
            if depth != cin:
                out += [(p + "shortcut/weights", (1, 1, cin, depth))] + _bn_names(p + "shortcut", depth)
            out += [(p + "conv1/weights", (1, 1, cin, db))] + _bn_names(p + "conv1", db)
            out += [(p + "conv2/weights", (3, 3, db, db))] + _bn_names(p + "conv2", db)
            out += [(p + "conv3/weights", (1, 1, db, depth))] + _bn_names(p + "conv3", depth)
            cin = depth
    flat = feat_hw * feat_hw * cin
    out += [("yolo_fc1/weights", (flat, fc_hidden)), ("yolo_fc1/biases", (fc_hidden,)),
            ("yolo_fc2/weights", (fc_hidden, fc_out)), ("yolo_fc2/biases", (fc_out,))]
    return out


def _bn_names(scope, c):
    s = scope.rstrip("/") + "/BatchNorm/"
    return [(s + "gamma", (c,)), (s + "beta", (c,)), (s + "moving_mean", (c,)), (s + "moving_variance", (c,))]


def init_params(blocks, seed=0, **kw):
    """variance_scaling_initializer (truncated normal, stddev sqrt(1.3 * 2 / fan_in)) for convolutions,
    xavier uniform for fully connected weights, zeros for biases, BN 1 / 0 / 0 / 1"""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in param_list(blocks, **kw):
        if name.endswith("weights") and len(shape) == 4:
            fan_in = shape[0] * shape[1] * shape[2]
            std = np.sqrt(1.3 * 2.0 / fan_in)
            w = rng.standard_normal(shape)
            bad = np.abs(w) > 2
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2
            params[name] = (w * std).astype(np.float32)
        elif name.endswith("weights"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            params[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif name.endswith("gamma") or name.endswith("moving_variance"):
            params[name] = np.ones(shape, np.float32)
        else:
            params[name] = np.zeros(shape, np.float32)
    return params


def conv2d_same(x, w, stride):
    """x NCHW, w HWIO (torch tensors) -- resnet_utils.py:77-122"""
    k = w.shape[0]
    wt = w.permute(3, 2, 0, 1)
    if stride == 1:
        return F.conv2d(x, wt, padding=k // 2)
    pad_total = k - 1
    pb = pad_total // 2
    pe = pad_total - pb
    return F.conv2d(F.pad(x, (pb, pe, pb, pe)), wt, stride=stride)


def subsample(x, factor):
    return x if factor == 1 else x[:, :, ::factor, ::factor]


def max_pool_3x3_s2_same(x):
    h, w = x.shape[2], x.shape[3]
    ho, wo = (h + 1) // 2, (w + 1) // 2
    ph = max((ho - 1) * 2 + 3 - h, 0)
    pw = max((wo - 1) * 2 + 3 - w, 0)
    xp = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
    return F.max_pool2d(xp, 3, 2)


def batch_norm(x, p, scope, is_training, movings=None):
    s = scope.rstrip("/") + "/BatchNorm/"
    g, b = p[s + "gamma"], p[s + "beta"]
    if is_training:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        if movings is not None:
            movings[s + "moving_mean"] = BN_DECAY * p[s + "moving_mean"].detach() + (1 - BN_DECAY) * mean.detach()
            movings[s + "moving_variance"] = BN_DECAY * p[s + "moving_variance"].detach() + (1 - BN_DECAY) * var.detach()
    else:
        mean, var = p[s + "moving_mean"], p[s + "moving_variance"]
    return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) * g[None, :, None, None] \
        + b[None, :, None, None]


def bottleneck(x, p, scope, depth, db, stride, is_training, movings):
    cin = x.shape[1]
    if depth == cin:
        shortcut = subsample(x, stride)
    else:
        # slim.conv2d(inputs, depth, [1, 1], stride=stride, activation_fn=None): a strided 1x1 convolution
        shortcut = batch_norm(F.conv2d(x, p[scope + "shortcut/weights"].permute(3, 2, 0, 1), stride=stride),
                              p, scope + "shortcut", is_training, movings)
    r = F.relu(batch_norm(F.conv2d(x, p[scope + "conv1/weights"].permute(3, 2, 0, 1)), p, scope + "conv1", is_training, movings))
    r = F.relu(batch_norm(conv2d_same(r, p[scope + "conv2/weights"], stride), p, scope + "conv2", is_training, movings))
    r = batch_norm(F.conv2d(r, p[scope + "conv3/weights"].permute(3, 2, 0, 1)), p, scope + "conv3", is_training, movings)
    return F.relu(shortcut + r)


def resnet_v1_50(x_nhwc, p, blocks=None, is_training=True, movings=None):
    """-> features NHWC [N, H/32, W/32, depth] (global_pool=False, tf_resnet.py:15)"""
    blocks = blocks or BLOCKS_50
    x = x_nhwc.permute(0, 3, 1, 2)
    x = F.relu(batch_norm(conv2d_same(x, p["conv1/weights"], 2), p, "conv1", is_training, movings))
    x = max_pool_3x3_s2_same(x)
    for bname, units in blocks:
        for i, (depth, db, stride) in enumerate(units):
            x = bottleneck(x, p, "%s/unit_%d/bottleneck_v1/" % (bname, i + 1), depth, db, stride, is_training, movings)
    return x.permute(0, 2, 3, 1)


def yolo_fc_head(feat_nhwc, p, drop_mask=None, keep_prob=0.5):
    """pascal_train_resnet.py:39-48; drop_mask: the 0/1 keep mask of tf.nn.dropout (None: no dropout)"""
    net = feat_nhwc.reshape(feat_nhwc.shape[0], -1)                       # slim.flatten, NHWC order
    fc1 = F.relu(net @ p["yolo_fc1/weights"] + p["yolo_fc1/biases"])
    if drop_mask is not None:
        fc1 = fc1 * drop_mask / keep_prob
    return F.relu(fc1 @ p["yolo_fc2/weights"] + p["yolo_fc2/biases"])


def to_torch(params, dtype=torch.float64, requires_grad=True):
    out = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not ("moving" in k):
            t.requires_grad_(True)
        out[k] = t
    return out
