"""numpy restatement of src/yolo2_nets/net_utils.py:222-439 (get_iou, get_loss,
show_yolo_detection's decode) and src/config.py:37-45.  TEST INFRASTRUCTURE ONLY.

get_loss_backward is the hand-derived gradient of the restated graph with
TensorFlow's sub-gradient conventions at ties:
  * tf.maximum(x, y): gradient to x where x >= y, else to y
  * tf.minimum(x, y): gradient to x where x <= y, else to y
  * tf.clip_by_value(t, lo, hi) = maximum(minimum(t, hi), lo): passes where lo <= t <= hi
  * tf.cast(bool) and the >= mask carry no gradient; no stop_gradient anywhere,
    so the object loss back-propagates through `ious` (net_utils.py:353).
"""
import numpy as np

LAMBDA_COORD = 5.0    # config.py:44
LAMBDA_NOOBJ = 0.5    # config.py:45


def yolo_grid_offset(S, B):
    """config.py:40-42: array(range(S)*S*B).reshape(B,S,S).transpose(1,2,0)
    -> shape [S(y), S(x), B] with value = column index x."""
    off = np.array(list(range(S)) * S * B)
    off = np.reshape(off, (B, S, S))
    return np.transpose(off, (1, 2, 0))


def _corners(b):
    """net_utils.py:231-241."""
    return (b[..., 0] - b[..., 2] / 2.0, b[..., 1] - b[..., 3] / 2.0,
            b[..., 0] + b[..., 2] / 2.0, b[..., 1] + b[..., 3] / 2.0)


def get_iou(boxes1, boxes2):
    """net_utils.py:222-260.  boxes: [..., 4] = (x_center, y_center, w, h)."""
    x1a, y1a, x2a, y2a = _corners(boxes1)
    x1b, y1b, x2b, y2b = _corners(boxes2)
    lu_x, lu_y = np.maximum(x1a, x1b), np.maximum(y1a, y1b)
    rd_x, rd_y = np.minimum(x2a, x2b), np.minimum(y2a, y2b)
    ix, iy = np.maximum(0.0, rd_x - lu_x), np.maximum(0.0, rd_y - lu_y)
    inter = ix * iy
    sq1 = (x2a - x1a) * (y2a - y1a)
    sq2 = (x2b - x1b) * (y2b - y1b)
    union = np.maximum(sq1 + sq2 - inter, 1e-10)
    return np.clip(inter / union, 0.0, 1.0)


def _split(net, labels, num_class, batch_size, S, B, dtype):
    net = net.astype(dtype)
    labels = labels.astype(dtype)
    pc = net[..., :num_class]                                          # :279
    conf = net[..., num_class:num_class + B]                           # :281
    pb = net[..., num_class + B:].reshape(batch_size, S, S, B, 4)      # :284-285
    resp = labels[..., 0].reshape(batch_size, S, S, 1)                 # :290-291
    classes = labels[..., 5:]                                          # :292
    return net, labels, pc, conf, pb, resp, classes


def get_loss(net, labels, num_class, batch_size, image_size, S, B, OFFSET, dtype=np.float32):
    """net_utils.py:263-372.  Returns (total, ious, object_mask, parts) where
    parts = dict(class_loss, object_loss, noobject_loss, coord_loss)."""
    net, labels, pc, conf, pb, resp, classes = _split(net, labels, num_class, batch_size, S, B, dtype)
    class_delta = resp * (pc - classes)                                # :294-295
    class_loss = np.mean(np.sum(class_delta ** 2, axis=(1, 2, 3)))     # :296-297

    gt = labels[..., 1:5].reshape(batch_size, S, S, 1, 4)              # :302
    gt = np.tile(gt, (1, 1, 1, B, 1)) / dtype(image_size)              # :303
    offset = np.asarray(OFFSET, dtype).reshape(1, S, S, B)             # :307-309
    offset_t = np.transpose(offset, (0, 2, 1, 3))                      # :312
    px = (pb[..., 0] + offset) / dtype(S)                              # :310
    py = (pb[..., 1] + offset_t) / dtype(S)                            # :311-312
    pw = pb[..., 2] ** 2                                               # :313
    ph = pb[..., 3] ** 2                                               # :314
    pred = np.stack([px, py, pw, ph], axis=4)                          # :315-316
    ious = get_iou(pred, gt)                                           # :320

    object_mask = (ious >= ious.max(axis=3, keepdims=True)).astype(dtype) * resp   # :323-324
    noobject_mask = 1.0 - object_mask                                  # :325-326

    tx = gt[..., 0] * S - offset                                       # :330
    ty = gt[..., 1] * S - offset_t                                     # :331-332
    tw = np.sqrt(gt[..., 2])                                           # :333
    th = np.sqrt(gt[..., 3])                                           # :334
    delta = np.stack([pb[..., 0] - tx, pb[..., 1] - ty, pb[..., 2] - tw, pb[..., 3] - th], axis=4)
    delta = object_mask[..., None] * delta                             # :337-345
    coord_loss = np.mean(np.sum(delta ** 2, axis=(1, 2, 3, 4))) * LAMBDA_COORD     # :346-347

    object_delta = object_mask * (conf - ious)                         # :353
    object_loss = np.mean(np.sum(object_delta ** 2, axis=(1, 2, 3)))   # :354-355
    noobject_delta = noobject_mask * conf                              # :357
    noobject_loss = np.mean(np.sum(noobject_delta ** 2, axis=(1, 2, 3))) * LAMBDA_NOOBJ  # :358-359

    total = class_loss + object_loss + noobject_loss + coord_loss      # :372
    parts = dict(class_loss=class_loss, object_loss=object_loss,
                 noobject_loss=noobject_loss, coord_loss=coord_loss)
    return total, ious, object_mask, parts


def get_loss_backward(net, labels, num_class, batch_size, image_size, S, B, OFFSET, dtype=np.float32):
    """d(total loss)/d(net), shape of net, following TF's autodiff of net_utils.py:263-372."""
    net, labels, pc, conf, pb, resp, classes = _split(net, labels, num_class, batch_size, S, B, dtype)
    _tot, ious, mask, _ = get_loss(net, labels, num_class, batch_size, image_size, S, B, OFFSET, dtype)
    n = dtype(batch_size)
    dnet = np.zeros_like(net)
    # class term
    dnet[..., :num_class] = 2.0 * resp * resp * (pc - classes) / n
    # confidence: object + noobject
    dconf = 2.0 * mask * mask * (conf - ious) / n + LAMBDA_NOOBJ * 2.0 * (1.0 - mask) ** 2 * conf / n
    dnet[..., num_class:num_class + B] = dconf
    # gradient into ious from the object term
    diou = -2.0 * mask * mask * (conf - ious) / n

    gt = labels[..., 1:5].reshape(batch_size, S, S, 1, 4)
    gt = np.tile(gt, (1, 1, 1, B, 1)) / dtype(image_size)
    offset = np.asarray(OFFSET, dtype).reshape(1, S, S, B)
    offset_t = np.transpose(offset, (0, 2, 1, 3))
    px = (pb[..., 0] + offset) / dtype(S)
    py = (pb[..., 1] + offset_t) / dtype(S)
    pw = pb[..., 2] ** 2
    ph = pb[..., 3] ** 2

    # ---- forward pieces of get_iou (boxes1 = prediction, boxes2 = ground truth)
    x1a, y1a, x2a, y2a = px - pw / 2.0, py - ph / 2.0, px + pw / 2.0, py + ph / 2.0
    x1b, y1b, x2b, y2b = _corners(gt)
    lu_x, lu_y = np.maximum(x1a, x1b), np.maximum(y1a, y1b)
    rd_x, rd_y = np.minimum(x2a, x2b), np.minimum(y2a, y2b)
    ddx, ddy = rd_x - lu_x, rd_y - lu_y
    ix, iy = np.maximum(0.0, ddx), np.maximum(0.0, ddy)
    inter = ix * iy
    sq1 = (x2a - x1a) * (y2a - y1a)
    sq2 = (x2b - x1b) * (y2b - y1b)
    u_raw = sq1 + sq2 - inter
    union = np.maximum(u_raw, 1e-10)
    ratio = inter / union
    # ---- backward
    g_ratio = np.where((ratio >= 0.0) & (ratio <= 1.0), diou, 0.0)     # clip_by_value
    g_inter = g_ratio / union
    g_union = -g_ratio * inter / (union * union)
    g_uraw = np.where(u_raw >= 1e-10, g_union, 0.0)                    # maximum(expr, 1e-10): expr is x
    g_sq1 = g_uraw
    g_inter = g_inter - g_uraw
    g_ix, g_iy = g_inter * iy, g_inter * ix
    g_ddx = np.where(0.0 >= ddx, 0.0, g_ix)                            # maximum(0.0, d): const is x, wins ties
    g_ddy = np.where(0.0 >= ddy, 0.0, g_iy)
    g_rd_x, g_lu_x, g_rd_y, g_lu_y = g_ddx, -g_ddx, g_ddy, -g_ddy
    g_x1a = np.where(x1a >= x1b, g_lu_x, 0.0)                          # maximum(boxes1, boxes2)
    g_y1a = np.where(y1a >= y1b, g_lu_y, 0.0)
    g_x2a = np.where(x2a <= x2b, g_rd_x, 0.0)                          # minimum(boxes1, boxes2)
    g_y2a = np.where(y2a <= y2b, g_rd_y, 0.0)
    # sq1 = (x2a - x1a) * (y2a - y1a)
    g_x2a = g_x2a + g_sq1 * (y2a - y1a)
    g_x1a = g_x1a - g_sq1 * (y2a - y1a)
    g_y2a = g_y2a + g_sq1 * (x2a - x1a)
    g_y1a = g_y1a - g_sq1 * (x2a - x1a)
    g_px = g_x1a + g_x2a
    g_py = g_y1a + g_y2a
    g_pw = (g_x2a - g_x1a) / 2.0
    g_ph = (g_y2a - g_y1a) / 2.0
    dpb = np.zeros_like(pb)
    dpb[..., 0] = g_px / dtype(S)
    dpb[..., 1] = g_py / dtype(S)
    dpb[..., 2] = g_pw * 2.0 * pb[..., 2]
    dpb[..., 3] = g_ph * 2.0 * pb[..., 3]
    # coordinate term
    tx = gt[..., 0] * S - offset
    ty = gt[..., 1] * S - offset_t
    tw = np.sqrt(gt[..., 2])
    th = np.sqrt(gt[..., 3])
    tgt = np.stack([tx, ty, tw, th], axis=4)
    dpb += LAMBDA_COORD * 2.0 * (mask * mask)[..., None] * (pb - tgt) / n
    dnet[..., num_class + B:] = dpb.reshape(batch_size, S, S, B * 4)
    return dnet


def py2_int_div(a, b):
    """Python-2 `/` on two ints (net_utils.py:420-421): floor division."""
    return a // b


def decode_detections(predict_output, im_w, im_h, num_class, S, B, offset=None, object_thresh=0.5):
    """show_yolo_detection (net_utils.py:375-439) without matplotlib: returns a
    list of (upper_left_x, upper_left_y, w, h, class_index, confidence, cell_row,
    cell_col, box) in the reference's loop order (`for c.. for r.. for i..`)."""
    if offset is None:
        offset = yolo_grid_offset(S, B)
    predicts = np.asarray(predict_output, np.float32).reshape([S, S, num_class + B * 5])   # :393 (sess.run gives float32)
    pcls = predicts[:, :, :num_class]
    pconf = predicts[:, :, num_class:num_class + B]
    pbox = np.reshape(predicts[:, :, num_class + B:], [S, S, B, 4])
    pobj = pconf > object_thresh                                                  # :398
    xs = (pbox[:, :, :, 0] + offset) / float(S)                                   # :403
    ys = (pbox[:, :, :, 1] + np.transpose(offset, (1, 0, 2))) / float(S)          # :404-405
    ws = np.square(pbox[:, :, :, 2])
    hs = np.square(pbox[:, :, :, 3])
    out = []
    for c in range(S):                                                            # :410
        for r in range(S):
            for i in range(B):
                if pobj[c, r, i]:
                    # numpy-1.x scalar promotion (the reference's era): float32 scalar *
                    # python int -> float64, so every product below is a float64 product.
                    x = int(float(xs[c, r, i]) * im_w)                            # :414
                    y = int(float(ys[c, r, i]) * im_h)
                    w = int(float(ws[c, r, i]) * im_w)
                    h = int(float(hs[c, r, i]) * im_h)
                    cls = int(np.argmax(pcls[c, r]))                              # :418
                    ulx = x - py2_int_div(w, 2)                                   # :420
                    uly = y - py2_int_div(h, 2)                                   # :421
                    out.append((ulx, uly, w, h, cls, float(pconf[c, r, i]), c, r, i))
    return out
