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
