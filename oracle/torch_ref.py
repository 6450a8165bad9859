"""Independent PyTorch-CPU implementation of the same graph (oneDNN convs +
autograd).  TEST INFRASTRUCTURE ONLY.  Two uses:
  1. cross-checks the numpy restatement (forward and hand-written backward);
  2. `bench.py`'s cpu_baseline leg ("CPU restatement (PyTorch-CPU fp32), not TF1").
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nn_ref
from .loss_ref import LAMBDA_COORD, LAMBDA_NOOBJ


class _TFMax(torch.autograd.Function):
    """tf.maximum(x, y): gradient to x where x >= y, else to y."""
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x >= y)
        return torch.maximum(x, y)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return torch.where(m, g, torch.zeros_like(g)), torch.where(m, torch.zeros_like(g), g)


class _TFMin(torch.autograd.Function):
    """tf.minimum(x, y): gradient to x where x <= y, else to y."""
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x <= y)
        return torch.minimum(x, y)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return torch.where(m, g, torch.zeros_like(g)), torch.where(m, torch.zeros_like(g), g)


def tf_max(x, y):
    x, y = torch.broadcast_tensors(torch.as_tensor(x, dtype=y.dtype) if not torch.is_tensor(x) else x,
                                   torch.as_tensor(y, dtype=x.dtype) if not torch.is_tensor(y) else y)
    return _TFMax.apply(x, y)


def tf_min(x, y):
    x, y = torch.broadcast_tensors(x, torch.as_tensor(y, dtype=x.dtype) if not torch.is_tensor(y) else y)
    return _TFMin.apply(x, y)


def leaky(h):
    return tf_max(nn_ref.ALPHA * h, h)


def conv_bn_layer(x_nchw, p, is_training, pool):
    W = p["W"].permute(3, 2, 0, 1)                      # HWIO -> OIHW
    k = W.shape[-1]
    h = F.conv2d(x_nchw, W, p["b"], padding=k // 2)
    if is_training:
        mean = h.mean(dim=(0, 2, 3), keepdim=True)
        var = ((h - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    else:
        mean = p["moving_mean"].view(1, -1, 1, 1)
        var = p["moving_var"].view(1, -1, 1, 1)
    hb = p["gamma"].view(1, -1, 1, 1) * (h - mean) / torch.sqrt(var + nn_ref.BN_EPS) + p["beta"].view(1, -1, 1, 1)
    a = leaky(hb)
    if pool:
        a = F.max_pool2d(a, 2, 2, ceil_mode=True)
    return a, (mean.flatten(), var.flatten())


def to_torch_params(params, dtype=torch.float32, requires_grad=False):
    out = []
    for p in params:
        q = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}
        if requires_grad:
            for k in ("W", "b", "gamma", "beta"):
                q[k].requires_grad_(True)
        out.append(q)
    return out


def run_stack(x_nhwc, params, spec, is_training):
    x = x_nhwc.permute(0, 3, 1, 2)
    stats = []
    for p, (_k, _ci, _co, pool) in zip(params, spec):
        x, st = conv_bn_layer(x, p, is_training, pool)
        stats.append(st)
    return x.permute(0, 2, 3, 1), stats


def get_iou(b1, b2):
    def corners(b):
        return (b[..., 0] - b[..., 2] / 2.0, b[..., 1] - b[..., 3] / 2.0,
                b[..., 0] + b[..., 2] / 2.0, b[..., 1] + b[..., 3] / 2.0)
    x1a, y1a, x2a, y2a = corners(b1)
    x1b, y1b, x2b, y2b = corners(b2)
    lu_x, lu_y = tf_max(x1a, x1b), tf_max(y1a, y1b)
    rd_x, rd_y = tf_min(x2a, x2b), tf_min(y2a, y2b)
    zero = torch.zeros_like(lu_x)
    ix, iy = tf_max(zero, rd_x - lu_x), tf_max(zero, rd_y - lu_y)
    inter = ix * iy
    sq1 = (x2a - x1a) * (y2a - y1a)
    sq2 = (x2b - x1b) * (y2b - y1b)
    union = tf_max(sq1 + sq2 - inter, torch.full_like(inter, 1e-10))
    r = inter / union
    return tf_max(tf_min(r, torch.ones_like(r)), torch.zeros_like(r))   # clip_by_value


def get_loss(net, labels, num_class, batch_size, image_size, S, B, OFFSET):
    dt = net.dtype
    pc = net[..., :num_class]
    conf = net[..., num_class:num_class + B]
    pb = net[..., num_class + B:].reshape(batch_size, S, S, B, 4)
    resp = labels[..., 0].reshape(batch_size, S, S, 1)
    classes = labels[..., 5:]
    class_loss = ((resp * (pc - classes)) ** 2).sum(dim=(1, 2, 3)).mean()
    gt = labels[..., 1:5].reshape(batch_size, S, S, 1, 4).repeat(1, 1, 1, B, 1) / float(image_size)
    offset = torch.tensor(np.asarray(OFFSET), dtype=dt).reshape(1, S, S, B)
    offset_t = offset.permute(0, 2, 1, 3)
    px = (pb[..., 0] + offset) / float(S)
    py = (pb[..., 1] + offset_t) / float(S)
    pw = pb[..., 2] ** 2
    ph = pb[..., 3] ** 2
    ious = get_iou(torch.stack([px, py, pw, ph], dim=4), gt)
    mask = (ious >= ious.max(dim=3, keepdim=True).values).to(dt).detach() * resp
    tx = gt[..., 0] * S - offset
    ty = gt[..., 1] * S - offset_t
    tw, th = torch.sqrt(gt[..., 2]), torch.sqrt(gt[..., 3])
    delta = torch.stack([pb[..., 0] - tx, pb[..., 1] - ty, pb[..., 2] - tw, pb[..., 3] - th], dim=4)
    coord_loss = ((mask[..., None] * delta) ** 2).sum(dim=(1, 2, 3, 4)).mean() * LAMBDA_COORD
    object_loss = ((mask * (conf - ious)) ** 2).sum(dim=(1, 2, 3)).mean()
    noobject_loss = (((1.0 - mask) * conf) ** 2).sum(dim=(1, 2, 3)).mean() * LAMBDA_NOOBJ
    total = class_loss + object_loss + noobject_loss + coord_loss
    return total, ious, mask, dict(class_loss=class_loss, object_loss=object_loss,
                                   noobject_loss=noobject_loss, coord_loss=coord_loss)


def detector_train_step_fn(core_spec, head_spec, params, S, B, num_class, image_size):
    """Returns f(images, labels) -> (loss, grads list) used by the cpu_baseline leg."""
    spec = list(core_spec) + list(head_spec)

    def step(images, labels):
        for p in params:
            for k in ("W", "b", "gamma", "beta"):
                p[k].grad = None
        net, _ = run_stack(images, params, spec, True)
        n = images.shape[0]
        loss, ious, mask, _ = get_loss(net.reshape(n, S, S, -1), labels, num_class, n, image_size, S, B,
                                       _offset(S, B))
        loss.backward()
        return loss.detach(), ious.detach(), mask
    return step


def _offset(S, B):
    from .loss_ref import yolo_grid_offset
    return yolo_grid_offset(S, B)
