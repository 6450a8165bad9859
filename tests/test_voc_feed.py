"""The fed training loop (SURVEY 8 f-1 + a-15): img_dataset.pascal_voc batcher (reference
src/img_dataset/pascal_voc.py:13-86) on a one-image VOC devkit assembled from the reference's own fixtures
(tests/golden/testImg2.jpg + testImg2Anno.xml: data files), the uint8 input path (y2_forward_u8), the pinned
double-buffer feeder and a detector train step against the oracle."""
import os
import shutil

import numpy as np
import pytest

from oracle import data_ref as D


def make_devkit(root, golden_dir, copies=1):
    voc = os.path.join(root, "VOC2007")
    for d in ("JPEGImages", "Annotations", os.path.join("ImageSets", "Main")):
        os.makedirs(os.path.join(voc, d), exist_ok=True)
    names = []
    for i in range(copies):
        name = "%06d" % (i + 1)
        shutil.copy(os.path.join(golden_dir, "testImg2.jpg"), os.path.join(voc, "JPEGImages", name + ".jpg"))
        shutil.copy(os.path.join(golden_dir, "testImg2Anno.xml"), os.path.join(voc, "Annotations", name + ".xml"))
        names.append(name)
    with open(os.path.join(voc, "ImageSets", "Main", "trainval.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    return root


def test_batcher_matches_the_reference_semantics(tmp_path, golden_dir):
    from tensorflow_yolo2_amd.img_dataset.pascal_voc import pascal_voc, imread_bgr, flip_label
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir)
    g = np.load(os.path.join(golden_dir, "label_grid_testImg2.npz"))
    for size, S, key in ((224, 7, "grid_224_7"), (416, 13, "grid_416_13")):
        imdb = pascal_voc("trainval", batch_size=2, devkit_path=kit, image_size=size, cell_size=S, flipped=True, seed=1)
        assert len(imdb.gt_labels) == 2 and sorted(x["flipped"] for x in imdb.gt_labels) == [False, True]
        images, labels = imdb.get()                    # one pass over {image, flipped image}; the cursor wraps
        assert images.shape == (2, size, size, 3) and images.dtype == np.float32 and labels.shape == (2, S, S, 25)
        assert imdb.cursor == 0
        order = [x["flipped"] for x in imdb.gt_labels]  # (re-shuffled at the wrap: look the pair up by content)
        plain = int(np.argmax([np.array_equal(labels[i], g[key].astype(np.float32)) for i in range(2)]))
        np.testing.assert_array_equal(labels[plain], g[key].astype(np.float32))            # golden label grid
        np.testing.assert_array_equal(labels[1 - plain], flip_label(g[key], size).astype(np.float32))
        # image_read: BGR, cv2-compatible resize, x / 255 * 2 - 1 (oracle/data_ref.py), and its mirror image
        bgr = imread_bgr(os.path.join(kit, "VOC2007", "JPEGImages", "000001.jpg"))
        assert bgr.shape == (500, 353, 3)
        ref = D.normalise(D.resize_bilinear_u8(bgr, size, size)).astype(np.float32)
        np.testing.assert_array_equal(images[plain], ref)
        np.testing.assert_array_equal(images[1 - plain], ref[:, ::-1, :])
        # get_u8: the same batches before the float conversion
        a = pascal_voc("trainval", batch_size=4, devkit_path=kit, image_size=size, cell_size=S, flipped=True, seed=5)
        b = pascal_voc("trainval", batch_size=4, devkit_path=kit, image_size=size, cell_size=S, flipped=True, seed=5)
        fa, la = a.get()
        ub, lb = b.get_u8()
        assert ub.dtype == np.uint8
        np.testing.assert_array_equal(fa, D.normalise(ub).astype(np.float32))
        np.testing.assert_array_equal(la, lb)
        assert len(order) == 2
    # a flipped label: responsible cells mirrored, x -> size - 1 - x (pascal_voc.py:74-84)
    lab = g["grid_224_7"]
    fl = flip_label(lab, 224)
    assert fl[4, 4, 0] == 1 and fl[4, 2, 0] == 0 and abs(fl[4, 4, 1] - (223 - lab[4, 2, 1])) < 1e-12
    np.testing.assert_array_equal(fl[4, 4, 2:], lab[4, 2, 2:])


@pytest.mark.gpu
def test_uint8_input_equals_float_input_and_train_step_vs_oracle(tmp_path, golden_dir):
    """XML + JPEG -> image_read -> label grid -> one detector train step (f32 mode, 224x224, S = 7: the reference's
    own shape, pascal_train_darknet.py:26-42) fed with uint8 pixels, against the float64 oracle fed with the
    float image; and y2_forward_u8 == y2_forward on the converted image, bit for bit."""
    import torch
    from oracle import nn_ref as R, loss_ref as L, torch_ref as T
    from tensorflow_yolo2_amd import engine as E
    from tensorflow_yolo2_amd.img_dataset.pascal_voc import pascal_voc
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir)
    n, size, S = 2, 224, 7
    imdb = pascal_voc("trainval", batch_size=n, devkit_path=kit, image_size=size, cell_size=S, flipped=True, seed=0)
    u8, labels = imdb.get_u8()
    xf = D.normalise(u8).astype(np.float32)
    spec = E.CORE_SPEC + E.det_head_spec(30)
    params = R.init_params(spec, seed=0)
    for dtype in ("f32", "f16"):
        net = E.Network(spec, n, size, size, dtype=dtype, core_layers=18, training=True)
        net.load_params(params)
        g_u8 = net.forward(torch.as_tensor(u8).cuda(), True, True).clone()
        g_f = net.forward(torch.as_tensor(xf).cuda(), True, True).clone()
        assert torch.equal(g_u8, g_f), dtype
    # f32 step from the uint8 batch against the float64 oracle
    tp = T.to_torch_params(params, torch.float64, requires_grad=True)
    rnet, _ = T.run_stack(torch.tensor(xf, dtype=torch.float64), tp, R.CORE_SPEC + R.det_head_spec(30), True)
    rloss, rious, rmask, _ = T.get_loss(rnet.reshape(n, S, S, 30), torch.tensor(labels, dtype=torch.float64), 20, n,
                                        size, S, 2, L.yolo_grid_offset(S, 2))
    rloss.backward()
    net = E.Network(spec, n, size, size, dtype="f32", core_layers=18, training=True)
    net.load_params(params)
    grid = net.forward(torch.as_tensor(u8).cuda(), True, True)
    loss, ious, mask, dnet = E.yolo_loss(grid, torch.as_tensor(labels).cuda(), 20, n, size, S, 2)
    ref = rnet.detach().numpy()
    assert np.abs(grid.cpu().numpy() - ref).max() < 1e-3 * np.abs(ref).max()
    assert abs(loss[4].item() - rloss.item()) < 1e-3 * abs(rloss.item())
    np.testing.assert_array_equal(mask.cpu().numpy(), rmask.numpy())
    assert int(mask.sum().item()) == 4                      # dog + person in each of the two images
    net.backward(dnet)
    g = net.export_grads()
    for k in ("W", "gamma", "beta"):
        r = tp[21][k].grad.numpy()
        assert np.linalg.norm(g[21][k] - r) < 1e-3 * np.linalg.norm(r), k
    opt = E.AdamOptimizer(net)
    opt.step()
    assert torch.isfinite(net.params).all()


@pytest.mark.gpu
def test_feeder_double_buffer_and_fed_train_script(tmp_path, golden_dir):
    """utils/feeder.DeviceFeeder hands over exactly the batches the batcher produces, in order, while the next one
    uploads; pascal_train_darknet.main(--devkit) trains from them, saves a TF V2 checkpoint and resumes from it."""
    import torch
    from tensorflow_yolo2_amd.img_dataset.pascal_voc import pascal_voc
    from tensorflow_yolo2_amd.utils.feeder import DeviceFeeder
    from tensorflow_yolo2_amd.utils import tf_bundle
    from tensorflow_yolo2_amd.pascal import pascal_train_darknet
    from tensorflow_yolo2_amd.yolo2_nets import darknet
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir, copies=3)
    a = pascal_voc("trainval", batch_size=4, devkit_path=kit, image_size=64, cell_size=2, flipped=True, seed=3)
    b = pascal_voc("trainval", batch_size=4, devkit_path=kit, image_size=64, cell_size=2, flipped=True, seed=3)
    feeder = DeviceFeeder(lambda im, lab: a.get_u8(im, lab), 4, 64, 2)
    for step in range(5):
        img, lab = feeder.get()
        got_i, got_l = img.clone(), lab.clone()
        feeder.release()
        feeder.prefetch()
        eu8, el = b.get_u8()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got_i.cpu().numpy(), eu8)
        np.testing.assert_array_equal(got_l.cpu().numpy(), el)
    darknet.reset_default_graph()
    ck = str(tmp_path / "ckpts")
    try:
        r1 = pascal_train_darknet.main(["--iters", "3", "--batch", "4", "--size", "64", "--dtype", "f32", "--ckpt-dir", ck,
                                        "--devkit", kit, "--flipped", "--ckpt-format", "ckpt"])
        assert len(r1["losses"]) == 3 and all(np.isfinite(r1["losses"]))
        prefix = os.path.join(ck, "train_iter_3.ckpt")
        assert tf_bundle.is_bundle(prefix)
        names = tf_bundle.BundleReader(prefix).names()
        assert "darknet19/Variable" in names and "darknet19_detection/output/batch_normalization/moving_variance" in names
        assert "darknet19/Variable/Adam_1" in names and "beta1_power" in names
        darknet.reset_default_graph()
        r2 = pascal_train_darknet.main(["--iters", "2", "--batch", "4", "--size", "64", "--dtype", "f32", "--ckpt-dir", ck,
                                        "--devkit", kit, "--flipped", "--ckpt-format", "ckpt"])
        assert r2["first_iter"] == 4 and r2["last_iter"] == 5
        assert tf_bundle.is_bundle(os.path.join(ck, "train_iter_5.ckpt"))
    finally:
        darknet.reset_default_graph()
        darknet.set_default_dtype("f16")


@pytest.mark.gpu
def test_tf_checkpoint_round_trip_through_the_network(tmp_path):
    """save_variables(.ckpt) writes a TensorFlow V2 checkpoint under the reference's variable names (with the Adam slots
    and TF's beta powers beta^(t+1)); restore_darknet19_variables restores it -- variables, slots and step -- and a
    classifier checkpoint restores the backbone of a detector (net_utils.py:83-103)."""
    import torch
    from tensorflow_yolo2_amd import engine as E
    from tensorflow_yolo2_amd.utils import tf_bundle
    from tensorflow_yolo2_amd.yolo2_nets import net_utils as NU
    spec = list(E.CORE_SPEC) + E.det_head_spec(30)
    det = E.Network(spec, 1, 64, 64, dtype="f32", core_layers=18, training=True)
    det.init_params(5)
    opt = E.AdamOptimizer(det)
    det.grads.normal_()
    for _ in range(3):
        opt.step()
    os.makedirs(tmp_path / "voc", exist_ok=True)
    prefix = str(tmp_path / "voc" / "train_iter_3.ckpt")
    NU.save_variables(det, prefix, optimizer=opt)
    r = tf_bundle.BundleReader(prefix)
    assert abs(float(r.get_tensor("beta1_power")) - 0.9 ** 4) < 1e-7 and r.get_tensor("beta1_power").shape == ()
    assert r.get_tensor("darknet19_detection/conv1/Variable").shape == (3, 3, 1024, 1024)
    want_p, want_m, want_s = det.params.clone(), opt.m.clone(), det.state.clone()
    det2 = E.Network(spec, 1, 64, 64, dtype="f32", core_layers=18, training=True)
    det2.init_params(11)
    opt2 = E.AdamOptimizer(det2)
    assert NU.restore_darknet19_variables(det2, str(tmp_path / "voc"), save_epoch=False, optimizer=opt2) == 3
    assert torch.equal(det2.params, want_p) and torch.equal(det2.state, want_s) and torch.equal(opt2.m, want_m)
    assert opt2.t == 3
    # a checkpoint TF itself wrote has no adam_step entry: the step comes from beta1_power = beta1^(t+1)
    blob = {k: r.get_tensor(k) for k in r.names() if k != "adam_step"}
    tf_bundle.write_bundle(str(tmp_path / "voc" / "train_iter_4.ckpt"), blob)
    opt3 = E.AdamOptimizer(det2)
    NU.restore_variables(det2, str(tmp_path / "voc" / "train_iter_4.ckpt.index"), optimizer=opt3)
    assert opt3.t == 3
    # after 40000 steps TF's float32 beta1_power has underflowed to 0.0 (log(0) used to raise here)
    blob["beta1_power"] = np.float32(0.0)
    blob["beta2_power"] = np.float32(np.float64(0.999) ** 40001)
    tf_bundle.write_bundle(str(tmp_path / "voc" / "train_iter_40000.ckpt"), blob)
    opt4 = E.AdamOptimizer(det2)
    NU.restore_variables(det2, str(tmp_path / "voc" / "train_iter_40000.ckpt"), optimizer=opt4)
    assert opt4.t == 40000
    det2.grads.normal_()
    opt4.step()
    assert torch.isfinite(det2.params).all()
    # classifier checkpoint -> detector backbone
    cls = E.Network(list(E.CORE_SPEC) + list(E.CLS_HEAD_SPEC), 1, 64, 64, dtype="f32", core_layers=19, training=False)
    cls.init_params(7)
    os.makedirs(tmp_path / "imagenet", exist_ok=True)
    NU.save_variables(cls, str(tmp_path / "imagenet" / "train_epoch_98.ckpt"), "classifier")
    os.makedirs(tmp_path / "empty", exist_ok=True)
    before = det2.export_params()
    assert NU.restore_darknet19_variables(det2, str(tmp_path / "empty"), imagenet_ckpt_dir=str(tmp_path / "imagenet")) == 0
    after, src = det2.export_params(), cls.export_params()
    for l in range(18):
        np.testing.assert_array_equal(after[l]["W"], src[l]["W"])
        np.testing.assert_array_equal(after[l]["moving_var"], src[l]["moving_var"])
    for l in range(18, 22):
        np.testing.assert_array_equal(after[l]["W"], before[l]["W"])


@pytest.mark.gpu
def test_resnet_callers_train_resume_backbone_restore_and_detect(tmp_path, golden_dir):
    """Counterparts of src/pascal/pascal_train_resnet.py / pascal_detect_resnet.py and restore_resnet_tf_variables
    (net_utils.py:137-219) at 1/8 width: a fed training run from a V1 `resnet_v1_50.ckpt` (the convolutional layers
    restored, the head and Adam left at their initial values), snapshots in both formats, a resumed run that continues
    the step count with the Adam slots, and the detection script on a snapshot."""
    import torch
    from tensorflow_yolo2_amd.pascal import pascal_train_resnet, pascal_detect_resnet
    from tensorflow_yolo2_amd.utils import tf_bundle as B
    from tensorflow_yolo2_amd.yolo2_nets import net_utils, tf_resnet
    kit = make_devkit(str(tmp_path / "VOCdevkit"), golden_dir, copies=3)
    # a "downloaded" backbone: a V1 checkpoint with the backbone's variables under resnet_v1_50/ plus slim's logits
    # layer, which the graph does not have
    d = 8
    kw = dict(blocks=[(n, [(dep // d, db // d, st) for (dep, db, st) in units]) for n, units in tf_resnet.BLOCKS_50],
              root_depth=64 // d, fc_hidden=4096 // d)
    donor = tf_resnet.ResNet50Yolo(2, 224, seed=11, **kw)
    rng = np.random.default_rng(0)
    pre = {}
    for (name, shape, _t) in donor.vars:
        if name.split("/")[0] not in net_utils.RESNET_HEAD_SCOPES:
            pre["resnet_v1_50/" + name] = (rng.standard_normal(shape) * 0.05 + (1.0 if name.endswith(("gamma", "moving_variance")) else 0.0)).astype(np.float32)
    pre["resnet_v1_50/logits/weights"] = np.zeros((1, 1, 256, 10), np.float32)
    wdir = tmp_path / "weights"; wdir.mkdir()
    B.write_checkpoint_v1(str(wdir / "resnet_v1_50.ckpt"), pre)
    del donor
    ck = str(tmp_path / "ckpts")
    common = ["--batch", "2", "--width-div", "8", "--devkit", kit, "--ckpt-dir", ck, "--weights-path", str(wdir)]
    r1 = pascal_train_resnet.main(common + ["--iters", "3", "--ckpt-format", "ckpt"])
    m1 = r1["model"]
    assert r1["first_iter"] == 1 and r1["last_iter"] == 3 and m1.t == 3 and np.isfinite(r1["losses"]).all()
    assert os.path.isfile(os.path.join(ck, "train_iter_3.ckpt.index"))
    snap = B.BundleReader(os.path.join(ck, "train_iter_3.ckpt"))
    assert snap.has_tensor("resnet_v1_50/block4/unit_3/bottleneck_v1/conv3/BatchNorm/moving_variance")
    assert snap.has_tensor("yolo_fc1/weights/Adam_1") and snap.has_tensor("beta2_power")
    np.testing.assert_allclose(snap.get_tensor("beta1_power"), 0.9 ** 4, rtol=1e-6)
    # the backbone came from the V1 file (three Adam steps of 5e-4 away), the head did not
    fresh = tf_resnet.ResNet50Yolo(2, 224, **kw)
    w = m1.p["block1/unit_1/bottleneck_v1/conv1/weights"].cpu().numpy()
    assert np.abs(w - pre["resnet_v1_50/block1/unit_1/bottleneck_v1/conv1/weights"]).max() < 3 * 5e-4 * 1.01
    assert np.abs(w - fresh.p["block1/unit_1/bottleneck_v1/conv1/weights"].cpu().numpy()).max() > 0.05
    restored, kept = net_utils.restore_resnet_variables(fresh, str(wdir / "resnet_v1_50.ckpt"),
                                                        exclude=net_utils.RESNET_HEAD_SCOPES, with_optimizer=False)
    assert len(restored) == len(pre) - 1 and sorted(kept) == ["yolo_fc1/biases", "yolo_fc1/weights", "yolo_fc2/biases", "yolo_fc2/weights"]
    # resume: the latest snapshot, Adam slots and step count included; the .npz format beside it
    r2 = pascal_train_resnet.main(common + ["--iters", "2", "--ckpt-format", "npz", "--graph"])
    m2 = r2["model"]
    assert r2["first_iter"] == 4 and r2["last_iter"] == 5
    m2._follow_ctrl()
    assert m2.t == 5 and os.path.isfile(os.path.join(ck, "train_iter_5.npz"))
    z = np.load(os.path.join(ck, "train_iter_5.npz"))
    assert int(z["adam_step"]) == 5 and float(np.abs(z["yolo_fc2/weights/Adam"]).max()) > 0
    again = tf_resnet.ResNet50Yolo(2, 224, **kw)
    assert net_utils.restore_resnet_tf_variables(again, ck, 'resnet50', save_epoch=False) == 5
    assert torch.equal(again.params, m2.params) and torch.equal(again.m, m2.m) and again.t == 5
    # detection on the snapshot: moving statistics, no dropout, decode
    out = pascal_detect_resnet.main([os.path.join(golden_dir, "testImg2.jpg"), "--ckpt-dir", ck, "--width-div", "8", "--no-show"])
    assert out["restored"] == 5 and out["predicts"].shape == (1, 7, 7, 30) and np.isfinite(out["predicts"]).all()
    ref = again.forward(torch.as_tensor(np.zeros((2, 224, 224, 3), np.float32)).cuda(), is_training=False, dropout=False)
    assert ref.shape == (2, 7, 7, 30)


@pytest.mark.gpu
def test_imagenet_callers_train_test_predict(tmp_path, golden_dir):
    """Counterparts of src/imagenet/imagenet_{train,test,predict}_darknet.py on the classifier path: two training
    steps from an image list (loss / accuracy lines, snapshot with Momentum slots under the TF names), a resumed run
    that bumps the epoch, the validation loop over the same list and the top-5 script on one image.  f32: an untrained
    network in INFERENCE mode (moving statistics still 0 / 1) grows to 2e6 at the logits, beyond the f16 range."""
    from tensorflow_yolo2_amd.imagenet import (imagenet_train_darknet, imagenet_test_darknet, imagenet_predict_darknet,
                                               read_image_list, load_batch)
    img = os.path.join(golden_dir, "testImg2.jpg")
    lst = tmp_path / "train.txt"
    lst.write_text("".join("%s %d\n" % (img, (7 * i) % 1000) for i in range(4)) + "# comment\n")
    items = read_image_list(str(lst))
    assert len(items) == 4 and items[1] == (img, 7)
    ims, labs = load_batch(items[:2], 224)
    assert ims.shape == (2, 224, 224, 3) and ims.dtype == np.float32 and labs.tolist() == [0, 7] and -1 <= ims.min() < ims.max() <= 1
    ck = str(tmp_path / "ck")
    r1 = imagenet_train_darknet.main(["--iters", "2", "--batch", "4", "--image-list", str(lst), "--ckpt-dir", ck, "--dtype", "f32"])
    assert r1["epoch"] == 1 and len(r1["log"]) == 2 and np.isfinite(r1["log"]).all()
    z = np.load(os.path.join(ck, "train_epoch_1.npz"))
    assert "darknet19/Variable/Momentum" in z.files and z["darknet19/Variable_36"].shape == (1, 1, 1024, 1000)
    r2 = imagenet_train_darknet.main(["--iters", "1", "--batch", "4", "--ckpt-dir", ck, "--ckpt-format", "ckpt", "--dtype", "f32"])
    assert r2["epoch"] == 2 and os.path.isfile(os.path.join(ck, "train_epoch_2.ckpt.index"))
    t = imagenet_test_darknet.main(["--image-list", str(lst), "--batch", "2", "--ckpt-dir", ck, "--dtype", "f32"])
    assert 0.0 <= t["accuracy"] <= 1.0 and t["time_per_batch"] > 0
    p = imagenet_predict_darknet.main([img, "--ckpt-dir", ck, "--dtype", "f32"])
    assert len(p["predictions"]) == 5 and len(set(p["predictions"])) == 5 and np.all(np.diff(p["values"]) <= 0)
    raw = imagenet_predict_darknet.main([img, "--ckpt-dir", ck, "--raw-pixels", "--dtype", "f32"])
    assert len(raw["predictions"]) == 5 and np.isfinite(raw["values"]).all()
