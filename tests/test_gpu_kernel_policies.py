"""GPU tests (-m gpu) of round 4's kernels: the first layer's batch-norm statistics from the Gram matrix of the input
patches (csrc/conv1_wgrad.hip conv1_gram_kernel / conv1_gram_stats_kernel), the tile choice of conv_haloq, the
inference batch norm folded into the conv epilogue at odd geometries."""
import numpy as np
import pytest
import torch

from oracle import nn_ref as R

pytestmark = pytest.mark.gpu

from _shapes import TOL, f16_representable, rel_to_max     # noqa: E402
import _obs                                                 # noqa: E402


def _conv1_moments_float64(x, W, b, chunk=8):
    """per-channel mean and biased variance of y = conv3x3_same(x, W) + b over all pixels, float64 (torch CPU)"""
    import torch.nn.functional as F
    wt = torch.as_tensor(W).double().permute(3, 2, 0, 1)
    s1 = torch.zeros(W.shape[3], dtype=torch.float64)
    s2 = torch.zeros(W.shape[3], dtype=torch.float64)
    n = 0
    for i in range(0, x.shape[0], chunk):
        xi = torch.as_tensor(x[i:i + chunk]).double().permute(0, 3, 1, 2)
        y = F.conv2d(xi, wt, torch.as_tensor(b).double(), padding=1)
        s1 += y.sum((0, 2, 3))
        s2 += (y * y).sum((0, 2, 3))
        n += y.shape[0] * y.shape[2] * y.shape[3]
    mean = s1 / n
    return mean.numpy(), (s2 / n - mean * mean).numpy()


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("N,hw,kind", [(4, 64, "uniform"), (3, 50, "image"), (64, 416, "image"), (128, 224, "uniform")])
def test_first_layer_statistics_from_the_gram_matrix(N, hw, kind, dtype):
    """conv_bn_layer of the 3 -> 32 layer (darknet.py:150, 39-46) in training mode: the batch mean / variance the device
    normalises with -- read back through the moving averages of ONE update from (0, 1): moving = 0.99 moving + 0.01 batch
    -- against float64 moments of the exact conv output.  `image`: a smooth, strongly correlated input with a non-zero
    mean (neighbouring pixels nearly equal, as in photographs): the case where sum y^2 - (sum y)^2 / M cancels and the
    centred Gram form is needed.  Gate 1e-4 of the largest moment (observed ~1e-6; the round-3 statistics-only conv pass
    it replaces is held to the same gate through Y2_NO_CONV1_GRAM in test_ab_switches_select_equivalent_paths)."""
    if N * hw * hw > 3e6 and dtype != "f16":
        pytest.skip("the large geometries run in the benchmarked type only")
    # (f32: the parity mode keeps the statistics-only convolution pass -- net.hip -- and is held to the same gate)
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(N + hw)
    spec = [(3, 3, 32, 1), (1, 32, 32, 0)]
    if kind == "uniform":
        x = rng.uniform(-1, 1, (N, hw, hw, 3))
    else:
        base = rng.uniform(-0.2, 0.9, (N, 1, 1, 3))
        gy = np.linspace(0, 1, hw)[None, :, None, None]
        gx = np.linspace(0, 1, hw)[None, None, :, None]
        x = np.clip(base + 0.3 * np.sin(3 * gy + base) * np.cos(2 * gx) + 0.02 * rng.standard_normal((N, hw, hw, 3)), -1, 1)
    x = f16_representable(x.astype(np.float32)) if dtype != "bf16" else \
        torch.as_tensor(x.astype(np.float32)).bfloat16().float().numpy()
    params = R.init_params(spec, seed=6)
    q = (lambda a: f16_representable(a)) if dtype != "bf16" else (lambda a: torch.as_tensor(a).bfloat16().float().numpy())
    params[0]["W"] = q(params[0]["W"])
    params[0]["b"] = rng.uniform(-0.5, 0.5, 32).astype(np.float32)
    net = E.Network(spec, N, hw, hw, dtype=dtype, training=True, grad_scale=1.0)
    net.load_params(params)
    out = net.forward(torch.as_tensor(x).cuda(), True, True, update_moving=True)
    assert torch.isfinite(out).all()
    st = net.export_params()[0]
    mean_dev = st["moving_mean"].astype(np.float64) / 0.01
    var_dev = (st["moving_var"].astype(np.float64) - 0.99) / 0.01
    mean, var = _conv1_moments_float64(x, params[0]["W"], params[0]["b"])
    e_m = np.abs(mean_dev - mean).max() / max(np.abs(mean).max(), np.sqrt(var.max()))
    # the read-back itself costs ~6e-6 of the variance (0.99 + 0.01 var in float32)
    e_v = np.abs(var_dev - var).max() / var.max()
    print("conv1 statistics %s N=%d %dx%d %s: mean %.2e  var %.2e  (var range %.3g .. %.3g)" %
          (kind, N, hw, hw, dtype, e_m, e_v, var.min(), var.max()))
    how = "gram" if dtype != "f32" else "conv pass"
    _obs.gate("conv1 statistics (%s) mean %s %s" % (how, kind, dtype), e_m, 1e-4)
    _obs.gate("conv1 statistics (%s) var %s %s" % (how, kind, dtype), e_v, 1e-4)


@pytest.mark.parametrize("N,hw,cin,cout,k", [(128, 14, 256, 512, 3), (128, 7, 512, 1024, 3), (128, 7, 1024, 512, 3),
                                             (128, 28, 256, 128, 3), (128, 14, 512, 256, 3), (24, 20, 256, 512, 3),
                                             (48, 10, 512, 1024, 3),
                                             # K split over workgroups (fewer than 3072 pixels): single images, batch 24
                                             (1, 7, 1024, 1024, 3), (1, 14, 256, 512, 3), (24, 7, 512, 1024, 3),
                                             (3, 13, 1024, 1024, 3)])
def test_tile_choice_shapes_in_network_inference_fold(N, hw, cin, cout, k):
    """The tiles the cost model of conv_haloq picks away from the 416x416 batch-64 defaults (512 x 128, 256 x 128,
    512 x 64, 256 x 64: configs[2] at batch 128, the 320 / 608 maps of configs[4]) as LAYERS: a two-layer stack in
    inference mode folds scale / shift / leaky into the first layer's epilogue (bordered output addressing for every
    tile shape); the same stack on a training binding runs conv -> y -> bn_act: the same bits.  The per-shape float64
    gates of these tiles are in test_gpu_shapes_c2_c3_c5.py (C3) and test_gpu_resnet.py."""
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(hw * 7 + cin)
    spec = [(k, cin, cout, 0), (1, cout, 32, 0)]
    x = torch.as_tensor(f16_representable(rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32))).cuda()
    params = R.init_params(spec, seed=2)
    for p in params:
        p["moving_mean"] = rng.uniform(-0.5, 0.5, p["moving_mean"].shape).astype(np.float32)
        p["moving_var"] = rng.uniform(0.5, 2.0, p["moving_var"].shape).astype(np.float32)
        p["gamma"] = rng.uniform(0.5, 1.5, p["gamma"].shape).astype(np.float32)
        p["beta"] = rng.uniform(-0.3, 0.3, p["beta"].shape).astype(np.float32)
    outs = []
    for training in (False, True):
        net = E.Network(spec, N, hw, hw, dtype="f16", core_layers=2, training=training)
        net.load_params(params)
        outs.append((net.forward(x, False, False).clone(), net.debug_read(1, 0).clone()))
        del net
    assert torch.equal(outs[0][1], outs[1][1])          # the first layer's activation as the second layer reads it
    assert torch.equal(outs[0][0], outs[1][0])
    # and against float64 on the stored input: conv -> affine -> leaky at every pixel of a few images
    W = f16_representable(params[0]["W"]).reshape(k * k * cin, cout).astype(np.float64)
    sc = params[0]["gamma"].astype(np.float64) / np.sqrt(params[0]["moving_var"].astype(np.float64) + 1e-3)
    sh = params[0]["beta"].astype(np.float64) - params[0]["moving_mean"].astype(np.float64) * sc
    from _shapes import gather_patches, sample_pixels
    pts = sample_pixels(N, hw, rng, 300)
    # the device rounds conv + bias to f16 and applies the affine to THAT value (the two-pass form's stored y)
    yq = (gather_patches(x, pts, hw, k) @ W + params[0]["b"].astype(np.float64)).astype(np.float16).astype(np.float64)
    z = yq * sc + sh
    ref = np.maximum(0.1 * z, z)
    got = outs[0][1].reshape(-1, cout)[torch.as_tensor(pts).cuda()].double().cpu().numpy()
    _obs.gate("folded inference layer %dx%d %d->%d N=%d" % (hw, hw, cin, cout, N), rel_to_max(got, ref), TOL)


SMALL_M = [(1, 7, 1024, 1024), (1, 14, 256, 512), (24, 7, 512, 1024), (24, 7, 1024, 1024), (4, 14, 512, 256), (2, 26, 256, 512),
           (1, 13, 1024, 1024), (16, 7, 128, 256),
           (2, 10, 640, 256), (5, 9, 384, 128), (3, 11, 1152, 384)]     # K chunks that do not divide by the split depth


@pytest.mark.parametrize("dtype,tol", [("f16", 1e-3), ("f32", 1e-5)])
@pytest.mark.parametrize("N,hw,cin,cout", SMALL_M, ids=["%dx%d^2_%d-%d" % s for s in SMALL_M])
def test_small_launches_k_split_vs_float64(N, hw, cin, cout, dtype, tol):
    """3x3 layers with fewer than 3072 pixels (single-image detection, the reference's training batch 24 at 224x224:
    src/pascal/pascal_train_darknet.py:26) run conv_haloq with the K range split over workgroups, fp32 partial tiles added
    in split order (conv_haloq.hip haloq_ks): forward, dgrad and wgrad against float64 (tests/_shapes.py)"""
    from _shapes import check_layer_shape
    check_layer_shape(N, "small-M", 3, cin, cout, hw, "K-split", dtype=dtype, tol=tol)


SMALL_M_1X1 = [(1, 7, 1024, 512), (1, 14, 512, 256), (24, 7, 1024, 512), (32, 7, 2048, 512), (32, 7, 512, 2048), (4, 14, 512, 256),
               (1, 13, 1024, 512), (2, 10, 640, 256), (3, 11, 1152, 384), (1, 7, 1024, 1000)]


@pytest.mark.parametrize("dtype,tol", [("f16", 1e-3), ("f32", 1e-5)])
@pytest.mark.parametrize("N,hw,cin,cout", SMALL_M_1X1, ids=["%dx%d^2_%d-%d" % s for s in SMALL_M_1X1])
def test_small_1x1_launches_k_split_vs_float64(N, hw, cin, cout, dtype, tol):
    """round 6: 1x1 layers with fewer than 3072 pixels (single-image detection, batch 24 at 7x7, the ResNet swap's 7x7 units:
    src/yolo2_nets/darknet.py:174,176; src/slim_dir/nets/resnet_v1.py:99-112) run conv_igemm with its K steps split over
    workgroups (conv_igemm.hip launch_ks), fp32 partial tiles added in split order: forward, dgrad, wgrad against float64"""
    from _shapes import check_layer_shape
    check_layer_shape(N, "small-M-1x1", 1, cin, cout, hw, "K-split-1x1", dtype=dtype, tol=tol)


@pytest.mark.parametrize("N,hw,cin,cout", [(24, 7, 1024, 512), (1, 14, 512, 256), (32, 7, 512, 2048)])
def test_small_1x1_launches_k_split_in_network(N, hw, cin, cout):
    """... and in training mode inside a network: statistics from the stored values (conv_ks_stats_kernel), BN + leaky, the
    backward pass with the standalone BN-backward reduce (a K-split dgrad does not carry the fused one)"""
    from _shapes import check_layer_in_network
    check_layer_in_network(N, "small-M-1x1", 1, cin, cout, hw, 0, "K-split-1x1")


@pytest.mark.parametrize("N,hw,cin,cout,pool", [(24, 7, 512, 1024, 0), (1, 14, 256, 512, 1), (2, 13, 1024, 1024, 0)])
def test_small_launches_k_split_in_network(N, hw, cin, cout, pool):
    """the same as LAYERS in training mode: the batch statistics of a K-split launch come from one record over the stored
    values (conv_ks_stats_kernel), then BN + leaky (+ pool) and the backward pass (tests/_shapes.py)"""
    from _shapes import check_layer_in_network
    check_layer_in_network(N, "small-M", 3, cin, cout, hw, pool, "K-split")


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("f16", 2e-3), ("f16x2", 3e-4)])
def test_layer_options_relu_and_linear_stack_vs_torch(dtype, tol, stride):
    """y2_set_layer_options: a bottleneck-shaped stack (1x1 ReLU, 3x3 ReLU, 1x1 without activation; batch norm with slim's
    eps 1e-5 / decay 0.997; the conv-bias slots held at zero and given no gradient) against float64 autograd: output,
    input gradient, every parameter gradient, moving statistics (src/slim_dir/nets/resnet_v1.py:99-112).
    stride 2 (round 5): the 3x3 layer is a SUBSAMPLING layer of the executor (pool = 2) -- slim's conv2d_same(stride=2),
    the stride-1 output at even rows / columns (src/slim_dir/nets/resnet_utils.py:77-122), its batch norm over the kept
    positions: the bottleneck's conv2 in the last unit of blocks 1-3."""
    import torch.nn.functional as F
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(3)
    N, hw, cin, db, depth = 4, 12, 64, 32, 128
    spec = [(1, cin, db, 0), (3, db, db, 2 if stride == 2 else 0), (1, db, depth, 0)]
    slopes = [0.0, 0.0, 1.0]
    net = E.Network(spec, N, hw, hw, dtype=dtype, training=True, grad_scale=1.0)
    net.set_layer_options(slopes, 1e-5, 0.997, zero_bias_grad=True)
    params = R.init_params(spec, seed=8)
    for p in params:
        p["b"] = np.zeros_like(p["b"])
        p["gamma"] = rng.uniform(0.6, 1.4, p["gamma"].shape).astype(np.float32)
        p["beta"] = rng.uniform(-0.3, 0.3, p["beta"].shape).astype(np.float32)
    net.load_params(params)
    x = rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32)
    out = net.forward(torch.as_tensor(x).cuda(), True, True, update_moving=True).clone()
    dout = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    dx = net.backward_input(torch.as_tensor(dout).cuda())
    g = net.export_grads()
    # float64 reference.  f16: the activations' DECISIONS are inputs of the reference -- the device's own ReLU masks, read
    # from what it stored (the consumer's input is > 0 exactly where the ReLU passed) -- because the f16 run is the exact
    # gradient of a function that differs from the float64 one where a y, STORED in half precision, sits within its
    # rounding of the ReLU boundary and falls the other way.  Those decisions are counted and must (a) be few and (b) all
    # sit at a near-zero z of the reference; with the same decisions on both sides every gradient is a smooth function of
    # the stored values and is gated element-wise (this replaces the 0.15 l2 gate).
    masks = None
    if dtype != "f32":
        masks = [net.debug_read(l + 1, 0).cpu().numpy() > 0 for l in range(2)]
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tp = [{k: torch.tensor(p[k], dtype=torch.float64, requires_grad=True) for k in ("W", "gamma", "beta")} for p in params]
    h = xt.permute(0, 3, 1, 2)
    mv = []
    flips = 0
    for l, ((k, _ci, _co, _p), p, s) in enumerate(zip(spec, tp, slopes)):
        h = F.conv2d(h, p["W"].permute(3, 2, 0, 1), padding=k // 2)
        if _p == 2:
            h = h[:, :, ::2, ::2]          # conv2d_same(stride=2) = the stride-1 output at even positions
        mean, var = h.mean((0, 2, 3)), h.var((0, 2, 3), unbiased=False)
        mv.append((0.003 * mean.detach().numpy(), 0.997 + 0.003 * var.detach().numpy()))
        z = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * p["gamma"][None, :, None, None] \
            + p["beta"][None, :, None, None]
        if masks is not None and l < 2:
            m = torch.tensor(masks[l]).permute(0, 3, 1, 2)
            other = m != (z.detach() > 0)
            flips += int(other.sum())
            # a decision may only differ where the reference's z is within the f16 rounding of the stored conv output
            # (2^-11 of |y| times the scale: z is O(1), a few 1e-3 at most)
            assert float(z.detach().abs()[other].max()) < 5e-3 if other.any() else True, "a ReLU decision differs away from z = 0"
            h = torch.where(m, z, s * z)
        else:
            h = torch.maximum(s * z, z)
    ref = h.permute(0, 2, 3, 1)
    ref.backward(torch.tensor(dout, dtype=torch.float64))
    if masks is not None:
        n_dec = sum(int(np.prod(mk.shape)) for mk in masks)
        print("layer-options stack %s: %d of %d ReLU decisions differ from float64 (all at |z| < 5e-3)" % (dtype, flips, n_dec))
        assert flips <= 0.01 * n_dec, (flips, n_dec)
    gtol = tol if dtype != "f16" else 5e-3       # f16: dy, dA and the conv outputs are stored in half precision
    _obs.gate("layer-options stack forward %s" % dtype, rel_to_max(out.cpu().numpy(), ref.detach().numpy()), tol)
    _obs.gate("layer-options stack input gradient %s" % dtype, rel_to_max(dx.cpu().numpy(), xt.grad.numpy()), gtol)
    for l in range(3):
        for k in ("W", "gamma", "beta"):
            _obs.gate("layer-options stack d%s %s" % (k, dtype), rel_to_max(g[l][k], tp[l][k].grad.numpy()), gtol)
        assert float(np.abs(g[l]["b"]).max()) == 0.0                    # not a variable of this graph
    st = net.export_params()
    for l in range(3):
        assert rel_to_max(st[l]["moving_mean"], mv[l][0]) < max(tol, 1e-3) and rel_to_max(st[l]["moving_var"], mv[l][1]) < max(tol, 1e-3)


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
def test_forward_join_equals_forward_then_add_relu(dtype):
    """y2_forward_join writes relu(join + stack(x)) from the last layer's apply pass (the join of a bottleneck unit,
    slim_dir/nets/resnet_v1.py:112): the bits of y2_forward followed by y2_add_relu, in training and in inference mode,
    and the backward pass from y2_add_relu_backward's gradient is the same afterwards"""
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(11)
    N, hw, cin, db, depth = 3, 14, 256, 64, 256
    spec = [(1, cin, db, 0), (3, db, db, 0), (1, db, depth, 0)]
    params = R.init_params(spec, seed=5)
    x = torch.as_tensor(f16_representable(rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32))).cuda()
    join = torch.as_tensor(rng.uniform(-1, 1, (N, hw, hw, depth)).astype(np.float32)).cuda()
    net = E.Network(spec, N, hw, hw, dtype=dtype, core_layers=3, training=True, grad_scale=1.0)
    net.load_params(params)
    net.set_layer_options([0.0, 0.0, 1.0], 1e-5, 0.997, zero_bias_grad=True)
    for training in (True, False):
        two = E.add_relu(net.forward(x, training, training).clone(), join)
        one = net.forward(x, training, training, join=join).clone()
        assert torch.isfinite(one).all() and (one == 0).any() and (one > 0).any()
        assert torch.equal(one, two), training
    # backward after the joined forward (training statistics): the same gradients as after the two-call form
    dout = torch.as_tensor(rng.uniform(-1, 1, tuple(join.shape)).astype(np.float32)).cuda()
    res = []
    for joined in (False, True):
        out = net.forward(x, True, True, join=join) if joined else E.add_relu(net.forward(x, True, True).clone(), join)
        g = E.add_relu_backward(dout, out.clone())
        dx = net.backward_input(g).clone()
        res.append((dx, net.grads.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    with pytest.raises(Exception):
        net.forward(x, True, True, join=join[:1].contiguous())
