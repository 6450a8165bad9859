"""PASCAL VOC detection metric (average precision per class, mAP) for the detector's decoded boxes.

NOT in the reference (it has no evaluation code at all -- BASELINE.md section 1; SURVEY 8 a-x2 lists mAP as an extension):
the specification is the VOC devkit's published protocol -- detections of a class sorted by confidence, each
matched to the still-unmatched ground-truth box of its image with the highest IoU; IoU >= 0.5 is a true positive,
a second detection of an already matched box a false positive, `difficult` boxes neither; IoU on integer pixel
boxes with the devkit's +1 extent.  AP: VOC2007's 11-point interpolation or (VOC2010+) the area under the
monotone precision envelope.  Pure host code (numpy); the boxes come from decode_yolo_detection / NMS on the GPU."""
import numpy as np


def box_iou_voc(box, boxes):
    """IoU of one [xmin, ymin, xmax, ymax] box against an [n, 4] array, inclusive pixel coordinates (+1 extents)"""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
    ixmin = np.maximum(boxes[:, 0], box[0])
    iymin = np.maximum(boxes[:, 1], box[1])
    ixmax = np.minimum(boxes[:, 2], box[2])
    iymax = np.minimum(boxes[:, 3], box[3])
    iw = np.maximum(ixmax - ixmin + 1.0, 0.0)
    ih = np.maximum(iymax - iymin + 1.0, 0.0)
    inter = iw * ih
    union = ((box[2] - box[0] + 1.0) * (box[3] - box[1] + 1.0) +
             (boxes[:, 2] - boxes[:, 0] + 1.0) * (boxes[:, 3] - boxes[:, 1] + 1.0) - inter)
    return inter / union


def average_precision(recall, precision, use_07_metric=False):
    recall, precision = np.asarray(recall, np.float64), np.asarray(precision, np.float64)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = precision[recall >= t].max() if (recall >= t).any() else 0.0
            ap += p / 11.0
        return float(ap)
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return float(((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]).sum())


def eval_class(detections, ground_truth, iou_thresh=0.5, use_07_metric=False):
    """detections: [(image_id, confidence, xmin, ymin, xmax, ymax)] of ONE class;
    ground_truth: {image_id: {"boxes": [n,4], "difficult": [n] bool}} of the same class.
    -> (ap, recall[], precision[])"""
    gt = {k: {"boxes": np.asarray(v["boxes"], np.float64).reshape(-1, 4),
              "difficult": np.asarray(v.get("difficult", np.zeros(len(v["boxes"]))), bool).reshape(-1),
              "taken": np.zeros(len(v["boxes"]), bool)} for k, v in ground_truth.items()}
    npos = int(sum((~g["difficult"]).sum() for g in gt.values()))
    dets = sorted(detections, key=lambda d: -d[1])
    tp, fp = np.zeros(len(dets)), np.zeros(len(dets))
    for i, (img, _conf, x0, y0, x1, y1) in enumerate(dets):
        g = gt.get(img)
        best, j = -1.0, -1
        if g is not None and len(g["boxes"]):
            ious = box_iou_voc((x0, y0, x1, y1), g["boxes"])
            j = int(ious.argmax())
            best = float(ious[j])
        if best >= iou_thresh:
            if g["difficult"][j]:
                continue                     # neither a hit nor a miss
            if not g["taken"][j]:
                tp[i] = 1.0
                g["taken"][j] = True
            else:
                fp[i] = 1.0
        else:
            fp[i] = 1.0
    ctp, cfp = np.cumsum(tp), np.cumsum(fp)
    recall = ctp / max(npos, 1)
    precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
    return average_precision(recall, precision, use_07_metric), recall, precision


def voc_map(detections, ground_truth, num_class=20, iou_thresh=0.5, use_07_metric=True):
    """detections: [(image_id, class, confidence, xmin, ymin, xmax, ymax)];
    ground_truth: [(image_id, class, xmin, ymin, xmax, ymax, difficult)] -> (mAP over the classes that have ground
    truth, {class: AP})"""
    aps = {}
    for c in range(num_class):
        gtc = {}
        for (img, cls, x0, y0, x1, y1, diff) in ground_truth:
            if cls == c:
                e = gtc.setdefault(img, {"boxes": [], "difficult": []})
                e["boxes"].append((x0, y0, x1, y1))
                e["difficult"].append(bool(diff))
        if not gtc:
            continue
        dc = [(img, conf, x0, y0, x1, y1) for (img, cls, conf, x0, y0, x1, y1) in detections if cls == c]
        aps[c] = eval_class(dc, gtc, iou_thresh, use_07_metric)[0]
    return (float(np.mean(list(aps.values()))) if aps else 0.0), aps


def detections_from_decode(image_id, dets):
    """decode_yolo_detection tuples (upper_left_x, upper_left_y, w, h, class, confidence, ...) -> voc_map rows"""
    return [(image_id, int(d[4]), float(d[5]), d[0], d[1], d[0] + d[2] - 1, d[1] + d[3] - 1) for d in dets]
