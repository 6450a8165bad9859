"""utils/voc_eval.py (SURVEY 8 a-x2: mAP, not in the reference) against hand-worked cases of the VOC protocol."""
import numpy as np

from tensorflow_yolo2_amd.utils import voc_eval as V


def test_iou_uses_inclusive_pixel_extents():
    assert V.box_iou_voc((0, 0, 9, 9), [(0, 0, 9, 9)])[0] == 1.0
    # 10x10 boxes shifted by 5: intersection 5x10 = 50, union 150
    np.testing.assert_allclose(V.box_iou_voc((0, 0, 9, 9), [(5, 0, 14, 9)]), [50.0 / 150.0])
    assert V.box_iou_voc((0, 0, 9, 9), [(10, 0, 19, 9)])[0] == 0.0


def test_average_precision_both_conventions():
    # three detections: TP, FP, TP with 2 positives -> recall .5 .5 1, precision 1 .5 2/3
    rec, prec = [0.5, 0.5, 1.0], [1.0, 0.5, 2.0 / 3.0]
    np.testing.assert_allclose(V.average_precision(rec, prec, False), 0.5 * 1.0 + 0.5 * (2.0 / 3.0))
    # 11-point: thresholds 0..0.5 see max precision 1 (6 points), 0.6..1.0 see 2/3 (5 points)
    np.testing.assert_allclose(V.average_precision(rec, prec, True), (6 * 1.0 + 5 * 2.0 / 3.0) / 11.0)


def test_matching_rules_duplicates_difficult_and_order():
    gt = {"a": {"boxes": [(10, 10, 50, 50), (100, 100, 150, 150)], "difficult": [False, True]},
          "b": {"boxes": [(0, 0, 20, 20)], "difficult": [False]}}
    dets = [("a", 0.9, 10, 10, 50, 50),        # TP
            ("a", 0.8, 12, 12, 50, 50),        # duplicate of a matched box -> FP
            ("a", 0.7, 100, 100, 150, 150),    # hits a difficult box -> ignored
            ("b", 0.6, 200, 200, 220, 220),    # no overlap -> FP
            ("b", 0.5, 1, 1, 20, 20)]          # TP (IoU 400/441)
    ap, rec, prec = V.eval_class(dets, gt)
    np.testing.assert_allclose(rec, [0.5, 0.5, 0.5, 0.5, 1.0])
    np.testing.assert_allclose(prec, [1.0, 0.5, 0.5, 1.0 / 3.0, 0.5])
    np.testing.assert_allclose(ap, 0.5 * 1.0 + 0.5 * 0.5)
    # the same detections in another order give the same result (sorting is by confidence)
    ap2, _, _ = V.eval_class(dets[::-1], gt)
    assert ap2 == ap


def test_map_over_classes_and_decode_rows():
    gts = [("i1", 11, 48, 240, 195, 371, 0), ("i1", 14, 8, 12, 352, 498, 0), ("i2", 14, 5, 5, 60, 90, 0)]
    perfect = [("i1", 11, 0.9, 48, 240, 195, 371), ("i1", 14, 0.8, 8, 12, 352, 498), ("i2", 14, 0.7, 5, 5, 60, 90)]
    m, aps = V.voc_map(perfect, gts)
    assert abs(m - 1.0) < 1e-12 and set(aps) == {11, 14}
    m2, aps2 = V.voc_map(perfect[:2], gts)                 # one person missed: recall 0.5 for class 14
    np.testing.assert_allclose(aps2[14], 6.0 / 11.0)         # 11-point: precision 1 up to recall 0.5
    np.testing.assert_allclose(m2, (1.0 + 6.0 / 11.0) / 2)
    rows = V.detections_from_decode("i1", [(48, 240, 148, 132, 11, 0.9, 4, 2, 0)])
    assert rows == [("i1", 11, 0.9, 48, 240, 195, 371)]
