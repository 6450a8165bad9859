"""CPU SPECIFICATION (test infrastructure only) of the YOLOv2 pieces that the north star names but the
reference does NOT contain (SURVEY.md §8 rows a-x1, a-x2): reorg / passthrough concat, anchor-box
decode, per-image greedy NMS.  There is no reference file:line to follow and nothing to pin against --
**parity unpinned**: these functions ARE the definition the HIP kernels (csrc/ext.hip) are tested
against, bit-exactly for the index work (reorg, NMS keep lists).  All arithmetic is float32 in the
written operation order.
"""
import numpy as np

f32 = np.float32


def reorg(x, stride=2):
    """space-to-depth: y[n, h//s, w//s, ((h%s)*s + w%s)*C + c] = x[n, h, w, c]"""
    n, h, w, c = x.shape
    s = stride
    y = x.reshape(n, h // s, s, w // s, s, c).transpose(0, 1, 3, 2, 4, 5)
    return np.ascontiguousarray(y.reshape(n, h // s, w // s, s * s * c))


def reorg_backward(dy, stride=2):
    n, ho, wo, cc = dy.shape
    s = stride
    c = cc // (s * s)
    x = dy.reshape(n, ho, wo, s, s, c).transpose(0, 1, 3, 2, 4, 5)
    return np.ascontiguousarray(x.reshape(n, ho * s, wo * s, c))


def passthrough_concat(fine, coarse):
    """concat(reorg2(fine [N,2H,2W,Cf]), coarse [N,H,W,Cc]) along channels"""
    return np.concatenate([reorg(fine, 2), coarse], axis=3)


def passthrough_concat_backward(dout, cf):
    return reorg_backward(dout[..., :4 * cf], 2), np.ascontiguousarray(dout[..., 4 * cf:])


def decode_anchors(net, anchors):
    """net [N,S,S,B,5+C] -> boxes [N,S*S*B,4] (cx,cy,w,h relative to the image), scores [N,S*S*B,C]
    (exp overflow of an untrained net's tw/th gives inf, as on the device)"""
    with np.errstate(over="ignore", invalid="ignore"):
        return _decode_anchors(net, anchors)


def _decode_anchors(net, anchors):
    net = np.asarray(net, f32)
    n, s, _, b, d = net.shape
    c = d - 5
    col = np.arange(s, dtype=f32)[None, None, :, None]
    row = np.arange(s, dtype=f32)[None, :, None, None]
    sig = lambda v: (f32(1) / (f32(1) + np.exp(-v, dtype=f32))).astype(f32)
    fs = f32(s)
    bx = (sig(net[..., 0]) + col) / fs
    by = (sig(net[..., 1]) + row) / fs
    an = np.asarray(anchors, f32)
    bw = an[None, None, None, :, 0] * np.exp(net[..., 2], dtype=f32) / fs
    bh = an[None, None, None, :, 1] * np.exp(net[..., 3], dtype=f32) / fs
    so = sig(net[..., 4])
    logits = net[..., 5:]
    e = np.exp(logits - logits.max(axis=-1, keepdims=True), dtype=f32)
    sm = e / e.sum(axis=-1, keepdims=True, dtype=f32)
    boxes = np.stack([bx, by, bw, bh], axis=-1).astype(f32).reshape(n, s * s * b, 4)
    scores = (so[..., None] * sm).astype(f32).reshape(n, s * s * b, c)
    return boxes, scores


def nms_iou(a, b):
    """a, b: (cx, cy, w, h) float32; the operation order the kernel follows (inf / nan from absurd boxes
    propagate exactly as IEEE float32 does on the device)"""
    with np.errstate(all="ignore"):
        return _nms_iou(a, b)


def _nms_iou(a, b):
    ax1, ax2 = a[0] - a[2] * f32(0.5), a[0] + a[2] * f32(0.5)
    ay1, ay2 = a[1] - a[3] * f32(0.5), a[1] + a[3] * f32(0.5)
    bx1, bx2 = b[0] - b[2] * f32(0.5), b[0] + b[2] * f32(0.5)
    by1, by2 = b[1] - b[3] * f32(0.5), b[1] + b[3] * f32(0.5)
    iw = max(f32(0), f32(min(ax2, bx2) - max(ax1, bx1)))
    ih = max(f32(0), f32(min(ay2, by2) - max(ay1, by1)))
    inter = f32(iw * ih)
    uni = f32(f32(a[2] * a[3] + b[2] * b[3]) - inter)
    return f32(inter / uni) if uni > 0 else f32(0)


def nms(boxes, scores, classes=None, iou_thresh=0.5, score_thresh=0.0, max_out=100, class_aware=False):
    """one image: boxes [K,4], scores [K] -> kept original indices (greedy, score descending, ties by index)"""
    boxes, scores = np.asarray(boxes, f32), np.asarray(scores, f32)
    order = [i for i in np.lexsort((np.arange(len(scores)), -scores.astype(np.float64))) if scores[i] >= f32(score_thresh)]
    sup = np.zeros(len(order), bool)
    keep = []
    thr = f32(iou_thresh)
    for a in range(len(order)):
        if len(keep) >= max_out:
            break
        if sup[a]:
            continue
        i = order[a]
        keep.append(int(i))
        for b in range(a + 1, len(order)):
            j = order[b]
            if sup[b] or (class_aware and classes[i] != classes[j]):
                continue
            if nms_iou(boxes[i], boxes[j]) > thr:
                sup[b] = True
    return keep


# ---------------------------------------------------------------------------
# YOLOv2 anchor-box ("region") loss, forward + analytic gradient.  NOT in the reference (its get_loss is the
# YOLOv1 grid loss); the north star names a "confidence/class/coord multi-part loss" on the anchor model, so
# this is the specification csrc/ext.hip's yolov2_loss_kernel is tested against (parity unpinned; the gradient
# is cross-checked against torch autograd of the same formula in tests/test_oracle.py).
#
#   net     [N,S,S,B,5+C]  raw outputs (tx, ty, tw, th, to, class logits)
#   labels  [N,S,S,5+C]    the reference's label grid (img_dataset/pascal_voc.py:146-163): resp, cx, cy, w, h in
#                          resized pixels, one-hot class -- one ground-truth box per cell
#   anchors [B,2]          (w, h) in cell units
# With g = ground truth in cell units (gx = cx / image_size * S, ...), (row, col) its cell, b* the anchor whose
# (w, h) has the largest shape IoU with (gw, gh) (first one on ties), p = the decoded prediction
# (px = sigmoid(tx) + col, py = sigmoid(ty) + row, pw = aw exp(tw), ph = ah exp(th)):
#   coord    = coord_scale  * [(sig(tx) - (gx - col))^2 + (sig(ty) - (gy - row))^2 + (tw - ln(gw/aw))^2 + (th - ln(gh/ah))^2]
#   object   = object_scale * (sig(to) - IoU(p, g))^2         IoU is a constant target (no gradient through it)
#   class    = class_scale  * cross_entropy(softmax(logits), class)
#                                                             ... summed over the responsible (cell, b*) pairs
#   noobject = noobject_scale * sig(to)^2   over every other (cell, anchor) whose best IoU with ANY ground truth of
#                                           its image is <= thresh
#   loss = (coord + object + class + noobject) / N            (parts returned in that order, then the total)
# ---------------------------------------------------------------------------
YOLOV2_SCALES = dict(coord_scale=1.0, object_scale=5.0, noobject_scale=1.0, class_scale=1.0, thresh=0.6)


def _box_iou_cwh(ax, ay, aw, ah, bx, by, bw, bh):
    iw = np.maximum(0.0, np.minimum(ax + aw / 2, bx + bw / 2) - np.maximum(ax - aw / 2, bx - bw / 2))
    ih = np.maximum(0.0, np.minimum(ay + ah / 2, by + bh / 2) - np.maximum(ay - ah / 2, by - bh / 2))
    inter = iw * ih
    uni = aw * ah + bw * bh - inter
    return np.where(uni > 0, inter / np.where(uni > 0, uni, 1.0), 0.0)


def yolov2_loss(net, labels, anchors, image_size, coord_scale=1.0, object_scale=5.0, noobject_scale=1.0,
                class_scale=1.0, thresh=0.6, dtype=np.float64):
    """-> (loss[5] = coord, object, noobject, class, total; dnet [N,S,S,B,5+C])"""
    net = np.asarray(net, dtype)
    labels = np.asarray(labels, dtype)
    an = np.asarray(anchors, dtype)
    n, s, _, b, d = net.shape
    c = d - 5
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    dnet = np.zeros_like(net)
    parts = np.zeros(4, dtype)
    col = np.arange(s, dtype=dtype)[None, :, None]
    row = np.arange(s, dtype=dtype)[:, None, None]
    for i in range(n):
        t = net[i]
        sx, sy, so = sig(t[..., 0]), sig(t[..., 1]), sig(t[..., 4])
        px, py = sx + col, sy + row
        pw, ph = an[None, None, :, 0] * np.exp(t[..., 2]), an[None, None, :, 1] * np.exp(t[..., 3])
        cells = np.argwhere(labels[i, :, :, 0] > 0)
        truths = []
        for (r, q) in cells:
            lab = labels[i, r, q]
            truths.append((lab[1] / image_size * s, lab[2] / image_size * s, lab[3] / image_size * s,
                           lab[4] / image_size * s, int(np.argmax(lab[5:])), int(r), int(q)))
        best = np.zeros((s, s, b), dtype)
        for (gx, gy, gw, gh, _k, _r, _q) in truths:
            best = np.maximum(best, _box_iou_cwh(px, py, pw, ph, gx, gy, gw, gh))
        resp = np.zeros((s, s, b), bool)
        for (gx, gy, gw, gh, k, r, q) in truths:
            inter = np.minimum(gw, an[:, 0]) * np.minimum(gh, an[:, 1])
            shape_iou = inter / (gw * gh + an[:, 0] * an[:, 1] - inter)
            bs = int(np.argmax(shape_iou))                       # first maximum
            resp[r, q, bs] = True
            tt = t[r, q, bs]
            ex, ey = sx[r, q, bs] - (gx - q), sy[r, q, bs] - (gy - r)
            ew, eh = tt[2] - np.log(gw / an[bs, 0]), tt[3] - np.log(gh / an[bs, 1])
            parts[0] += coord_scale * (ex * ex + ey * ey + ew * ew + eh * eh)
            dnet[i, r, q, bs, 0] = coord_scale * 2 * ex * sx[r, q, bs] * (1 - sx[r, q, bs])
            dnet[i, r, q, bs, 1] = coord_scale * 2 * ey * sy[r, q, bs] * (1 - sy[r, q, bs])
            dnet[i, r, q, bs, 2] = coord_scale * 2 * ew
            dnet[i, r, q, bs, 3] = coord_scale * 2 * eh
            iou = float(_box_iou_cwh(px[r, q, bs], py[r, q, bs], pw[r, q, bs], ph[r, q, bs], gx, gy, gw, gh))
            eo = so[r, q, bs] - iou
            parts[1] += object_scale * eo * eo
            dnet[i, r, q, bs, 4] = object_scale * 2 * eo * so[r, q, bs] * (1 - so[r, q, bs])
            lg = tt[5:]
            m = lg.max()
            lse = m + np.log(np.exp(lg - m).sum())
            parts[3] += class_scale * (lse - lg[k])
            sm = np.exp(lg - lse)
            sm[k] -= 1.0
            dnet[i, r, q, bs, 5:] = class_scale * sm
        noobj = (~resp) & (best <= thresh)
        parts[2] += noobject_scale * (so[noobj] ** 2).sum()
        dnet[i, ..., 4] += np.where(noobj, noobject_scale * 2 * so * so * (1 - so), 0.0)
    parts = parts / n
    dnet = dnet / n
    return np.concatenate([parts, [parts.sum()]]).astype(dtype), dnet
