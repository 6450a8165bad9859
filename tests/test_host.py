"""CPU tests of the host-side logic and of the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import nn_ref as R, loss_ref as L, data_ref as D


def test_golden_label_grid_matches_oracle_and_host_encoder(golden_dir):
    from tensorflow_yolo2_amd.img_dataset import pascal_voc
    g = np.load(os.path.join(golden_dir, "label_grid_testImg2.npz"))
    xml = open(os.path.join(golden_dir, "testImg2Anno.xml")).read()
    w, h, objs = D.parse_voc_xml(xml)
    for (size, S, key) in ((224, 7, "grid_224_7"), (416, 13, "grid_416_13")):
        np.testing.assert_array_equal(D.encode_boxes(objs, h, w, size, S), g[key])
        lab, n = pascal_voc.load_pascal_annotation(xml, size, S)
        assert n == 2
        np.testing.assert_array_equal(lab, g[key])
    assert g["grid_224_7"][4, 2, 5 + 11] == 1 and g["grid_224_7"][3, 3, 5 + 14] == 1


def test_golden_slim_kats_match_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "slim_conv_kat.npz"))
    i = np.arange(5)
    m5 = (i[:, None] + i[None, :]).astype(np.float32)
    w = m5[:3, :3].reshape(3, 3, 1, 1)
    np.testing.assert_array_equal(R.conv2d_same(m5[:4, :4].reshape(1, 4, 4, 1), w)[0, :, :, 0], g["y_even"])
    np.testing.assert_array_equal(R.conv2d_same(m5.reshape(1, 5, 5, 1), w)[0, :, :, 0], g["y_odd"])
    np.testing.assert_array_equal(R.conv2d_same(m5[:4, :4].reshape(1, 4, 4, 1), w, 2)[0, :, :, 0], g["y_even_stride2"])


@pytest.mark.parametrize("S,size", [(7, 224), (13, 416)])
def test_golden_loss_matches_oracle(golden_dir, S, size):
    g = np.load(os.path.join(golden_dir, "loss_S%d.npz" % S))
    n = g["net"].shape[0]
    off = L.yolo_grid_offset(S, 2)
    tot, ious, mask, parts = L.get_loss(g["net"], g["labels"], 20, n, size, S, 2, off, np.float32)
    np.testing.assert_array_equal(ious, g["ious"])
    np.testing.assert_array_equal(mask, g["mask"])
    assert abs(tot - g["total"]) < 1e-5 * abs(g["total"])
    dets = L.decode_detections(g["net"][0], 353, 500, 20, S, 2)
    np.testing.assert_array_equal(np.array([d[:5] + d[6:] for d in dets], np.int32).reshape(-1, 8), g["dets"])


def test_golden_tiny_stack_matches_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_stack.npz"))
    spec = [tuple(int(v) for v in s) for s in g["spec"]]
    params = []
    for l, (k, ci, co, _p) in enumerate(spec):
        p = R.init_layer(np.random.default_rng(0), k, ci, co)
        for key in ("W", "b", "gamma", "beta"):
            p[key] = g["p%d_%s" % (l, key)]
        params.append(p)
    out, caches, _ = R.run_stack(g["x"], params, spec, True, np.float64)
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-6)
    _, grads = R.run_stack_backward(params, caches, g["dout"].astype(np.float64), np.float64)
    for l in range(len(spec)):
        np.testing.assert_allclose(grads[l]["W"], g["g%d_W" % l], rtol=1e-4, atol=1e-6)


def test_config_offsets_and_variable_names():
    from tensorflow_yolo2_amd import config as cfg
    from tensorflow_yolo2_amd.yolo2_nets import darknet
    assert (cfg.YOLO_GRID_OFFSET == L.yolo_grid_offset(7, 2)).all()
    assert (cfg.yolo_grid_offset(13, 2) == L.yolo_grid_offset(13, 2)).all()
    names = darknet.variable_names("detector")
    assert len(names) == 22 and names[0]["W"] == "darknet19/Variable" and names[0]["b"] == "darknet19/Variable_1"
    assert names[1]["W"] == "darknet19/Variable_2" and names[1]["gamma"] == "darknet19/batch_normalization_1/gamma"
    assert names[17]["moving_var"] == "darknet19/batch_normalization_17/moving_variance"
    assert names[21]["W"] == "darknet19_detection/output/Variable"
    assert len(darknet.variable_names("classifier")) == 19
    # 88 trainable tensors in the detector graph
    assert sum(4 for _ in names) == 88


def test_synthetic_inputs_are_deterministic_and_encoded_like_the_reference():
    from tensorflow_yolo2_amd import synthetic
    a, b = synthetic.images(2, 32, 7), synthetic.images(2, 32, 7)
    assert a.dtype == np.float32 and (a == b).all() and a.min() >= -1 and a.max() < 1
    lab = synthetic.det_labels(4, 416, 13, 5)
    assert lab.shape == (4, 13, 13, 25)
    resp = lab[..., 0]
    assert set(np.unique(resp)) <= {0.0, 1.0} and resp.sum() >= 4
    cells = np.argwhere(resp == 1)
    for (n, y, x) in cells:
        cx, cy = lab[n, y, x, 1], lab[n, y, x, 2]
        assert int(cx * 13 / 416) == x and int(cy * 13 / 416) == y
        assert lab[n, y, x, 5:].sum() == 1


def test_layer_slices_cover_the_flat_buffer_in_backward_order():
    from tensorflow_yolo2_amd import trainer
    sl = trainer.layer_slices(22, cuts=(0, 13, 18))
    assert sl == [(18, 22), (13, 18), (0, 13)]
    offs = list(range(0, 220, 10))
    ranges = [trainer.slice_range(offs, 1000, 22, lo, hi) for (lo, hi) in sl]
    assert ranges == [(180, 1000), (130, 180), (0, 130)]
    assert trainer.layer_slices(3, cuts=(0, 13, 18)) == [(0, 3)]
    # default: head filters one by one, a 3.4 MB tail; contiguous cover of [0, n) in backward order
    d = trainer.layer_slices(22)
    assert d[0] == (20, 22) and d[-1] == (0, 8)
    assert [lo for (lo, hi) in d] == [hi for (lo, hi) in d[1:]] + [0] and d[0][1] == 22
    assert trainer.layer_slices(19)[0][1] == 19 and trainer.layer_slices(19)[-1][0] == 0


def test_multi_scale_schedule_is_deterministic_and_in_range():
    """{320..608} step 32, redrawn every 10 steps, a pure function of (seed, step): every rank agrees."""
    from tensorflow_yolo2_amd.trainer import multi_scale_size, MULTI_SCALE_SIZES
    assert MULTI_SCALE_SIZES == (320, 352, 384, 416, 448, 480, 512, 544, 576, 608)
    sizes = [multi_scale_size(s, seed=5) for s in range(200)]
    assert all(s in MULTI_SCALE_SIZES for s in sizes)
    assert all(len(set(sizes[i:i + 10])) == 1 for i in range(0, 200, 10))
    assert len(set(sizes)) >= 5
    assert sizes == [multi_scale_size(s, seed=5) for s in range(200)]
    assert sizes != [multi_scale_size(s, seed=6) for s in range(200)]


def test_c1_image_preprocessing_matches_golden(golden_dir):
    """BASELINE.json configs[0] input: the reference's tests/testImg1.jpg (352x240) -> BGR, cv2-compatible
    bilinear resize to 224x224, x/255*2-1 (pascal_detect_darknet.py:31-36).  The product's image_read must
    reproduce the oracle's restatement bit for bit (JPEG-decoder / cv2 bit equality itself is unpinned)."""
    from PIL import Image
    from tensorflow_yolo2_amd.img_dataset import pascal_voc
    g = np.load(os.path.join(golden_dir, "testImg1_input224.npz"))
    rgb = np.array(Image.open(os.path.join(golden_dir, "testImg1.jpg")).convert("RGB"), dtype=np.uint8)
    assert tuple(rgb.shape) == tuple(g["shape"]) == (240, 352, 3)
    img = pascal_voc.image_read(rgb[:, :, ::-1], 224)
    assert img.shape == (224, 224, 3) and img.dtype == np.float32
    np.testing.assert_array_equal(img, g["image"])
    assert -1.0 <= img.min() and img.max() <= 1.0
    np.testing.assert_array_equal(D.normalise(D.resize_bilinear_u8(rgb[:, :, ::-1], 224, 224)), g["image"])


def test_adam_step_recovered_from_tf_beta_powers_survives_underflow():
    """A TF-written Adam checkpoint only holds float32 beta1_power = 0.9^(t+1) and beta2_power = 0.999^(t+1).  The first
    is denormal after ~830 steps and 0.0 after ~1000 (the reference saves at 40000: pascal_train_darknet.py:104-109),
    so the step must come from beta2_power; with neither usable any large t is equivalent."""
    from tensorflow_yolo2_amd.yolo2_nets.net_utils import adam_step_from_powers as f
    for t in (0, 1, 3, 50, 985, 1100, 40000, 80000):
        b1p = np.float32(np.float64(0.9) ** (t + 1))
        b2p = np.float32(np.float64(0.999) ** (t + 1))
        assert f(b1p, b2p) == t, t
    for t in (0, 3, 50, 400):                                  # beta1_power alone while it is a normal float32
        assert f(np.float32(np.float64(0.9) ** (t + 1)), None) == t
    assert f(np.float32(0.0), None) >= 10 ** 5                 # underflowed, no second power: bias corrections are 1
    assert f(np.float32(0.0), np.float32(0.0)) >= 10 ** 5
    assert f(np.float32(1e-42), None) >= 10 ** 5               # denormal: log of it is not a step count
    assert f(np.float32(np.nan), np.float32(np.inf)) >= 10 ** 5
    assert f(None, None) == 0
    assert f(np.float32(1.0), np.float32(1.0)) == 0
