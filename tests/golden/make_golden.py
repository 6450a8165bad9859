"""Generates the committed golden vectors in this directory from the CPU oracle
(run from the repo root: python tests/golden/make_golden.py).

The reference cannot be imported here (Python 2 + TensorFlow 1.x, neither present), so the
vectors come from the numpy restatement in oracle/, which is itself pinned by the reference's
known-answer tests (tests/test_oracle.py).  Everything is seeded; files are small .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nn_ref as R, loss_ref as L, data_ref as D  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def labels_for(rng, n, S, size, C=20):
    lab = np.zeros((n, S, S, 5 + C), np.float32)
    for i in range(n):
        objs = []
        for _ in range(int(rng.integers(1, 4))):
            x1, y1 = rng.uniform(1, size * 0.7, 2)
            bw, bh = rng.uniform(size * 0.05, size * 0.3, 2)
            objs.append((x1, y1, x1 + bw, y1 + bh, int(rng.integers(0, C))))
        lab[i] = D.encode_boxes(objs, size, size, size, S, C)
    return lab


def main():
    # 1. reference KATs (src/slim_dir/nets/resnet_v1_test.py:72-152)
    np.savez(os.path.join(HERE, "slim_conv_kat.npz"),
             y_even=np.array([[14, 28, 43, 26], [28, 48, 66, 37], [43, 66, 84, 46], [26, 37, 46, 22]], np.float32),
             y_odd=np.array([[14, 28, 43, 58, 34], [28, 48, 66, 84, 46], [43, 66, 84, 102, 55],
                             [58, 84, 102, 120, 64], [34, 46, 55, 64, 30]], np.float32),
             y_odd_sub2=np.array([[14, 43, 34], [43, 84, 55], [34, 55, 30]], np.float32),
             y_even_stride2=np.array([[48, 37], [37, 22]], np.float32))
    # 2. label grids for the reference fixture tests/testImg2Anno.xml
    w, h, objs = D.parse_voc_xml(open(os.path.join(HERE, "testImg2Anno.xml")).read())
    np.savez(os.path.join(HERE, "label_grid_testImg2.npz"),
             grid_224_7=D.encode_boxes(objs, h, w, 224, 7), grid_416_13=D.encode_boxes(objs, h, w, 416, 13))
    # 3. get_loss / get_iou / decode on seeded inputs
    for (S, size, n) in ((7, 224, 2), (13, 416, 2)):
        rng = np.random.default_rng(0)
        net = rng.uniform(-0.5, 1.2, (n, S, S, 30)).astype(np.float32)
        lab = labels_for(rng, n, S, size)
        off = L.yolo_grid_offset(S, 2)
        tot, ious, mask, parts = L.get_loss(net, lab, 20, n, size, S, 2, off, np.float32)
        tot64, _, _, parts64 = L.get_loss(net, lab, 20, n, size, S, 2, off, np.float64)
        dnet = L.get_loss_backward(net, lab, 20, n, size, S, 2, off, np.float64)
        dets = L.decode_detections(net[0], 353, 500, 20, S, 2)
        np.savez(os.path.join(HERE, "loss_S%d.npz" % S), net=net, labels=lab, ious=ious, mask=mask,
                 total=np.float64(tot64), parts=np.array([parts64[k] for k in
                                                          ("class_loss", "object_loss", "noobject_loss", "coord_loss")]),
                 dnet=dnet.astype(np.float32),
                 dets=np.array([d[:5] + d[6:] for d in dets], np.int32).reshape(-1, 8),
                 dets_conf=np.array([d[5] for d in dets], np.float32))
    # 4. a small conv-BN-leaky-pool stack: forward, moving stats and all gradients
    rng = np.random.default_rng(0)
    spec = [(3, 3, 32, True), (3, 32, 64, False), (1, 64, 32, True), (3, 32, 30, False)]
    params = R.init_params(spec, seed=0)
    for p in params:
        p["gamma"] = rng.uniform(0.5, 1.5, p["gamma"].shape).astype(np.float32)
        p["beta"] = rng.uniform(-0.3, 0.3, p["beta"].shape).astype(np.float32)
    x = rng.uniform(-1, 1, (2, 16, 12, 3)).astype(np.float32)
    out, caches, movings = R.run_stack(x, params, spec, True, np.float64)
    dout = rng.standard_normal(out.shape).astype(np.float32)
    _, grads = R.run_stack_backward(params, caches, dout.astype(np.float64), np.float64)
    blob = dict(x=x, out=out.astype(np.float32), dout=dout, spec=np.array(spec, np.int32))
    for l, (p, g, mv) in enumerate(zip(params, grads, movings)):
        for k in ("W", "b", "gamma", "beta"):
            blob["p%d_%s" % (l, k)] = p[k]
            blob["g%d_%s" % (l, k)] = g[k].astype(np.float32)
        blob["mm%d" % l] = mv[0].astype(np.float32)
        blob["mv%d" % l] = mv[1].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "tiny_stack.npz"), **blob)
    # YOLOv2 extension pieces (no reference code exists: vectors of this repo's own specification)
    from oracle import ext_ref as X
    rng = np.random.default_rng(7)
    fine = rng.standard_normal((2, 8, 12, 6)).astype(np.float32)
    coarse = rng.standard_normal((2, 4, 6, 5)).astype(np.float32)
    net5 = rng.standard_normal((2, 5, 5, 3, 5 + 4)).astype(np.float32)
    anchors = np.array([[1.3221, 1.73145], [3.19275, 4.00944], [5.05587, 8.09892]], np.float32)
    boxes, scores = X.decode_anchors(net5, anchors)
    K = 160
    nb = np.concatenate([rng.uniform(0.2, 0.8, (K, 2)), rng.uniform(0.05, 0.4, (K, 2))], axis=1).astype(np.float32)
    ns = np.round(rng.uniform(0, 1, K), 2).astype(np.float32)          # rounded: ties on purpose
    ncls = rng.integers(0, 3, K).astype(np.int32)
    keep_a = np.array(X.nms(nb, ns, None, 0.45, 0.1, 50, False), np.int32)
    keep_c = np.array(X.nms(nb, ns, ncls, 0.45, 0.1, 50, True), np.int32)
    np.savez_compressed(os.path.join(HERE, "yolo2_ext_kat.npz"), fine=fine, coarse=coarse,
                        concat=X.passthrough_concat(fine, coarse), net5=net5, anchors=anchors, boxes=boxes,
                        scores=scores, nms_boxes=nb, nms_scores=ns, nms_classes=ncls, keep_agnostic=keep_a,
                        keep_class_aware=keep_c)
    # C1 input: the reference's tests/testImg1.jpg (352x240, a data file) decoded with PIL,
    # BGR order, cv2-compatible bilinear resize to 224x224 and x/255*2-1 (pascal_detect_darknet.py:31-36)
    from PIL import Image
    rgb = np.array(Image.open(os.path.join(HERE, "testImg1.jpg")).convert("RGB"), dtype=np.uint8)
    bgr = rgb[:, :, ::-1]
    pre = D.normalise(D.resize_bilinear_u8(bgr, 224, 224)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "testImg1_input224.npz"), image=pre, shape=np.array(rgb.shape))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
