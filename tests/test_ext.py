"""YOLOv2 pieces beyond the reference (SURVEY §8 a-x1 / a-x2): reorg / passthrough concat, anchor decode,
per-image NMS.  No reference code exists for them, so the specification is oracle/ext_ref.py; the CPU tests
pin that specification (golden vectors, round trips, a brute-force NMS property), the GPU tests require the
HIP kernels to reproduce it: bit-exact for the index work, 1e-5 relative for the exp/sigmoid decode."""
import os

import numpy as np
import pytest

from oracle import ext_ref as X


def _kat(golden_dir):
    return np.load(os.path.join(golden_dir, "yolo2_ext_kat.npz"))


# ---------------------------------------------------------------- CPU: the specification itself
def test_spec_reorg_roundtrip_and_golden(golden_dir):
    g = _kat(golden_dir)
    np.testing.assert_array_equal(X.passthrough_concat(g["fine"], g["coarse"]), g["concat"])
    x = np.arange(2 * 4 * 6 * 3, dtype=np.float32).reshape(2, 4, 6, 3)
    y = X.reorg(x, 2)
    assert y.shape == (2, 2, 3, 12)
    # channel order: (h%2)*2 + w%2 major, original channel minor
    assert y[0, 0, 0, 0] == x[0, 0, 0, 0] and y[0, 0, 0, 3] == x[0, 0, 1, 0]
    assert y[0, 0, 0, 6] == x[0, 1, 0, 0] and y[0, 0, 0, 11] == x[0, 1, 1, 2]
    np.testing.assert_array_equal(X.reorg_backward(y, 2), x)
    df, dc = X.passthrough_concat_backward(g["concat"], g["fine"].shape[3])
    np.testing.assert_array_equal(df, g["fine"])
    np.testing.assert_array_equal(dc, g["coarse"])


def test_spec_decode_anchors_golden_and_ranges(golden_dir):
    g = _kat(golden_dir)
    boxes, scores = X.decode_anchors(g["net5"], g["anchors"])
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(scores, g["scores"])
    assert (boxes[..., :2] > 0).all() and (boxes[..., :2] < 1).all() and (boxes[..., 2:] > 0).all()
    # per anchor the class scores sum to the objectness
    so = 1.0 / (1.0 + np.exp(-g["net5"][..., 4].astype(np.float64)))
    np.testing.assert_allclose(scores.sum(-1).reshape(so.shape), so, rtol=1e-5)
    # zero logits sit at the cell centre with the anchor's size
    b0, _ = X.decode_anchors(np.zeros((1, 2, 2, 1, 7), np.float32), [[1.0, 2.0]])
    np.testing.assert_allclose(b0[0, 3], [0.75, 0.75, 0.5, 1.0], rtol=1e-6)


def test_spec_nms_golden_and_bruteforce_property(golden_dir):
    g = _kat(golden_dir)
    ka = X.nms(g["nms_boxes"], g["nms_scores"], None, 0.45, 0.1, 50, False)
    kc = X.nms(g["nms_boxes"], g["nms_scores"], g["nms_classes"], 0.45, 0.1, 50, True)
    np.testing.assert_array_equal(ka, g["keep_agnostic"])
    np.testing.assert_array_equal(kc, g["keep_class_aware"])
    b, s = g["nms_boxes"], g["nms_scores"]
    # kept boxes: sorted by (score desc, index asc), above the threshold, pairwise IoU <= thresh;
    # every dropped candidate is suppressed by an earlier kept one
    assert all((s[i] > s[j]) or (s[i] == s[j] and i < j) for i, j in zip(ka[:-1], ka[1:]))
    assert all(s[i] >= np.float32(0.1) for i in ka)
    for x in range(len(ka)):
        for y in range(x + 1, len(ka)):
            assert X.nms_iou(b[ka[x]], b[ka[y]]) <= np.float32(0.45)
    if len(ka) < 50:
        for j in range(len(s)):
            if s[j] >= np.float32(0.1) and j not in ka:
                assert any(X.nms_iou(b[i], b[j]) > np.float32(0.45) and ((s[i] > s[j]) or (s[i] == s[j] and i < j))
                           for i in ka)
    assert set(ka) <= set(kc) or len(kc) == 50   # class-aware suppresses less


# ---------------------------------------------------------------- GPU: the HIP kernels vs the specification
def dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,stride", [((2, 8, 12, 6), 2), ((3, 6, 6, 5), 2), ((1, 9, 6, 4), 3), ((4, 26, 26, 64), 2)])
def test_gpu_reorg_bit_exact(shape, stride):
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    y = E.reorg(dev(x), stride)
    np.testing.assert_array_equal(y.cpu().numpy(), X.reorg(x, stride))
    np.testing.assert_array_equal(E.reorg(y, stride, inverse=True).cpu().numpy(), x)


@pytest.mark.gpu
def test_gpu_passthrough_concat_golden_and_yolo2_shape(golden_dir):
    from tensorflow_yolo2_amd import engine as E
    g = _kat(golden_dir)
    out = E.passthrough_concat(dev(g["fine"]), dev(g["coarse"]))
    np.testing.assert_array_equal(out.cpu().numpy(), g["concat"])
    df, dc = E.passthrough_concat_backward(out, g["fine"].shape[3])
    np.testing.assert_array_equal(df.cpu().numpy(), g["fine"])
    np.testing.assert_array_equal(dc.cpu().numpy(), g["coarse"])
    # the YOLOv2 route: 26x26x512 -> 13x13x2048, concatenated with 13x13x1024 (batch 8)
    rng = np.random.default_rng(2)
    fine = rng.standard_normal((8, 26, 26, 512)).astype(np.float32)
    coarse = rng.standard_normal((8, 13, 13, 1024)).astype(np.float32)
    out = E.passthrough_concat(dev(fine), dev(coarse))
    assert tuple(out.shape) == (8, 13, 13, 3072)
    np.testing.assert_array_equal(out.cpu().numpy(), X.passthrough_concat(fine, coarse))


@pytest.mark.gpu
@pytest.mark.parametrize("S,B,C,N", [(5, 3, 4, 2), (13, 5, 20, 4), (19, 5, 20, 2)])
def test_gpu_decode_anchors(golden_dir, S, B, C, N):
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(S)
    net = (rng.standard_normal((N, S, S, B, 5 + C)) * 1.5).astype(np.float32)
    anchors = rng.uniform(0.5, 9.0, (B, 2)).astype(np.float32)
    rb, rs = X.decode_anchors(net, anchors)
    boxes, scores = E.decode_anchors(dev(net), anchors)
    # tolerance: device expf vs numpy exp differ in the last ulps
    np.testing.assert_allclose(boxes.cpu().numpy(), rb, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(scores.cpu().numpy(), rs, rtol=1e-5, atol=1e-8)


def _random_candidates(rng, K, clustered):
    if clustered:
        centres = rng.uniform(0.2, 0.8, (12, 2))
        c = centres[rng.integers(0, 12, K)] + rng.normal(0, 0.02, (K, 2))
        wh = rng.uniform(0.15, 0.3, (K, 2))
    else:
        c = rng.uniform(0.1, 0.9, (K, 2))
        wh = rng.uniform(0.03, 0.35, (K, 2))
    boxes = np.concatenate([c, wh], axis=1).astype(np.float32)
    scores = np.round(rng.uniform(0, 1, K), 3).astype(np.float32)   # rounded: ties occur
    return boxes, scores, rng.integers(0, 4, K).astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("K,clustered,class_aware,max_out", [(160, False, False, 50), (845, True, False, 100),
                                                               (845, False, True, 200), (1805, True, True, 100),
                                                               (3000, True, False, 64)])
def test_gpu_nms_bit_exact(golden_dir, K, clustered, class_aware, max_out):
    import torch
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(K + 7 * class_aware)
    N = 3
    data = [_random_candidates(rng, K, clustered) for _ in range(N)]
    boxes = np.stack([d[0] for d in data]); scores = np.stack([d[1] for d in data]); cls = np.stack([d[2] for d in data])
    keep, count = E.nms(dev(boxes), dev(scores), dev(cls), 0.45, 0.25, max_out, class_aware)
    torch.cuda.synchronize()
    keep, count = keep.cpu().numpy(), count.cpu().numpy()
    for n in range(N):
        ref = X.nms(boxes[n], scores[n], cls[n], 0.45, 0.25, max_out, class_aware)
        assert count[n] == len(ref)
        np.testing.assert_array_equal(keep[n, :count[n]], np.array(ref, np.int32))
        assert (keep[n, count[n]:] == -1).all()


@pytest.mark.gpu
def test_gpu_nms_golden(golden_dir):
    from tensorflow_yolo2_amd import engine as E
    g = _kat(golden_dir)
    b, s, c = dev(g["nms_boxes"][None]), dev(g["nms_scores"][None]), dev(g["nms_classes"][None])
    keep, count = E.nms(b, s, None, 0.45, 0.1, 50, False)
    np.testing.assert_array_equal(keep[0, :int(count[0])].cpu().numpy(), g["keep_agnostic"])
    keep, count = E.nms(b, s, c, 0.45, 0.1, 50, True)
    np.testing.assert_array_equal(keep[0, :int(count[0])].cpu().numpy(), g["keep_class_aware"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 12, 6), (1, 7, 9, 5), (3, 26, 26, 64)])
def test_gpu_maxpool_forward_backward(shape):
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(sum(shape))
    x = np.round(rng.standard_normal(shape), 1).astype(np.float32)        # rounded: ties inside windows
    y = E.max_pool_2x2(dev(x))
    np.testing.assert_array_equal(y.cpu().numpy(), R.max_pool_2x2(x))
    dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    dx = E.max_pool_2x2_backward(dev(x), dev(dy))
    np.testing.assert_array_equal(dx.cpu().numpy(), R.max_pool_2x2_backward(x, dy))


@pytest.mark.gpu
def test_gpu_yolov2_composed_detector_matches_composed_oracle():
    """The north star's model (passthrough + anchors + NMS), composed from three conv-BN-leaky stacks and the
    ext ops, against the same composition of the oracle's pieces (fp32, inference-mode BN, 160x160 input)."""
    import torch
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd.yolo2_nets import yolov2
    n, size = 2, 160
    det = yolov2.YOLOv2Detector(n, size, dtype="f32", seed=4)
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (n, size, size, 3)).astype(np.float32)
    grid = det.forward(dev(x))
    sa, sb, sc = yolov2.yolov2_specs(20, 5)
    pa, pb, pc = det.stem.export_params(), det.deep.export_params(), det.head.export_params()
    fine, _, _ = R.run_stack(x, pa, sa, False, np.float64)
    coarse, _, _ = R.run_stack(R.max_pool_2x2(fine), pb, sb, False, np.float64)
    ref, _, _ = R.run_stack(X.passthrough_concat(fine, coarse), pc, sc, False, np.float64)
    ref = ref.reshape(n, 5, 5, 5, 25)
    assert tuple(grid.shape) == (n, 5, 5, 5, 25)
    assert np.abs(grid.cpu().numpy() - ref).max() < 1e-3 * np.abs(ref).max()
    boxes, best, cls, keep, count = det.detect(dev(x), score_thresh=0.02, iou_thresh=0.45, max_out=50)
    torch.cuda.synchronize()
    g = grid.cpu().numpy()
    rb, rs = X.decode_anchors(g, det.anchors)
    np.testing.assert_allclose(boxes.cpu().numpy(), rb, rtol=1e-5, atol=1e-7)
    # NMS on the device's own scores: index work, bit-exact against the specification
    b, s, c = boxes.cpu().numpy(), best.cpu().numpy(), cls.cpu().numpy()
    for i in range(n):
        refk = X.nms(b[i], s[i], c[i], 0.45, 0.02, 50, True)
        assert int(count[i]) == len(refk) > 0
        np.testing.assert_array_equal(keep[i, :len(refk)].cpu().numpy(), np.array(refk, np.int32))


# ---------------------------------------------------------------- trainable YOLOv2 (anchor loss + composed backward)
@pytest.mark.gpu
@pytest.mark.parametrize("n,S,size", [(3, 5, 160), (8, 13, 416), (1, 19, 608)])
def test_gpu_yolov2_loss_matches_specification(n, S, size):
    from tensorflow_yolo2_amd import engine as E, synthetic
    from tensorflow_yolo2_amd.yolo2_nets.yolov2 import ANCHORS_VOC
    rng = np.random.default_rng(n * 100 + S)
    net = (rng.standard_normal((n, S, S, 5, 25)) * 0.7).astype(np.float32)
    lab = synthetic.det_labels(n, size, S, 5 + n)
    if n > 2:
        lab[1] = 0                                        # an image without objects
    ref_loss, ref_d = X.yolov2_loss(net, lab, ANCHORS_VOC, size, dtype=np.float64)
    loss, dnet = E.yolov2_loss(dev(net), dev(lab), ANCHORS_VOC, size)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=2e-5)
    err = np.abs(dnet.cpu().numpy() - ref_d).max() / np.abs(ref_d).max()
    assert err < 2e-5, err
    # forward only, and other scales
    sc = dict(coord_scale=2.0, object_scale=3.0, noobject_scale=0.5, class_scale=1.5, thresh=0.4)
    l2, d2 = E.yolov2_loss(dev(net), dev(lab), ANCHORS_VOC, size, need_grad=False, scales=sc)
    r2, _ = X.yolov2_loss(net, lab, ANCHORS_VOC, size, dtype=np.float64, **sc)
    assert d2 is None
    np.testing.assert_allclose(l2.cpu().numpy(), r2, rtol=2e-5)


@pytest.mark.gpu
def test_gpu_yolov2_train_step_gradients_match_composed_oracle():
    """The north star's TRAINABLE model at reduced width (f32): forward, anchor loss and the backward pass through
    head -> passthrough concat -> 13x13 stack -> 2x2 pool (+ the passthrough branch) -> stem, against the same
    composition of the oracle's pieces; then optimizer steps in f32 and f16 keep the loss falling and finite."""
    import torch
    from oracle import nn_ref as R
    from tensorflow_yolo2_amd import engine as E, synthetic
    from tensorflow_yolo2_amd.yolo2_nets import yolov2
    n, size = 2, 96
    S = size // 32
    tr = yolov2.YOLOv2Trainer(n, size, dtype="f32", seed=3, width_div=8)
    sa, sb, sc = tr.specs
    x = synthetic.images(n, size, 21)
    lab = synthetic.det_labels(n, size, S, 22)
    stem, deep, head = tr.nets
    pa, pb, pc = stem.export_params(), deep.export_params(), head.export_params()
    # ---- oracle composition, float64
    fine, ca, _ = R.run_stack(x, pa, sa, True, np.float64)
    pooled = R.max_pool_2x2(fine)
    coarse, cb, _ = R.run_stack(pooled, pb, sb, True, np.float64)
    cat = X.passthrough_concat(fine, coarse)
    out, cc, _ = R.run_stack(cat, pc, sc, True, np.float64)
    ref_loss, dnet = X.yolov2_loss(out.reshape(n, S, S, 5, 25), lab, tr.anchors, size)
    dcat, gc = R.run_stack_backward(pc, cc, dnet.reshape(out.shape), np.float64, need_input_grad=True)
    dfine_a, dcoarse = X.passthrough_concat_backward(dcat, tr.cf)
    dpooled, gb = R.run_stack_backward(pb, cb, dcoarse, np.float64, need_input_grad=True)
    dfine = dfine_a + R.max_pool_2x2_backward(fine, dpooled)
    _, ga = R.run_stack_backward(pa, ca, dfine, np.float64)
    # ---- device: the same step without the optimizer
    grid, fine_d = tr.forward(dev(x), True)
    assert np.abs(grid.cpu().numpy().reshape(out.shape) - out).max() < 1e-3 * np.abs(out).max()
    loss, dn = E.yolov2_loss(grid.contiguous(), dev(lab), tr.anchors, size)
    assert abs(loss[4].item() - ref_loss[4]) < 1e-3 * ref_loss[4]
    dcat_d = head.backward_input(dn.view(n, S, S, -1))
    assert np.abs(dcat_d.cpu().numpy() - dcat).max() < 2e-3 * np.abs(dcat).max()
    df_d, dc_d = E.passthrough_concat_backward(dcat_d, tr.cf)
    dp_d = deep.backward_input(dc_d)
    E.accumulate(df_d, E.max_pool_2x2_backward(fine_d, dp_d))
    assert np.abs(df_d.cpu().numpy() - dfine).max() < 5e-3 * np.abs(dfine).max()
    stem.backward(df_d)
    for net, ref, name in ((head, gc, "head"), (deep, gb, "deep"), (stem, ga, "stem")):
        g = net.export_grads()
        for l in (0, len(g) - 1):
            e = np.linalg.norm(g[l]["W"] - ref[l]["W"]) / np.linalg.norm(ref[l]["W"])
            assert e < 2e-2, (name, l, e)
    # ---- whole steps: the loss falls, everything stays finite (f32 and the benchmarked f16 mode)
    # (round 6: and the reference-tolerance modes of the f16 pipe -- the composed graph takes its input gradients through
    #  y2_backward_input, whose lowest dgrad stays fp32 in f16x2f; their first loss is the f32 mode's to 1e-5)
    first = {}
    for dtype in ("f32", "f16", "f16x2", "f16x2f"):
        t2 = yolov2.YOLOv2Trainer(n, size, dtype=dtype, seed=3, width_div=8)
        losses = [float(t2.step(dev(x), dev(lab))[4]) for _ in range(8)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], (dtype, losses)
        first[dtype] = losses[0]
        for net in t2.nets:
            assert torch.isfinite(net.params).all()
    for dtype in ("f16x2", "f16x2f"):
        assert abs(first[dtype] - first["f32"]) < 1e-5 * abs(first["f32"]), (dtype, first)
    # multi-scale: another input size on the same three parameter sets
    x2 = synthetic.images(n, 128, 5)
    l2 = t2.step(dev(x2), dev(synthetic.det_labels(n, 128, 4, 6)))
    assert np.isfinite(float(l2[4])) and len(t2.ctx) == 2
    assert t2.ctx[128][0].params.data_ptr() == t2.ctx[96][0].params.data_ptr()
