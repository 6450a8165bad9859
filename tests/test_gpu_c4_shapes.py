"""GPU tests (-m gpu): the benchmarked arithmetic (f16 operands, fp32 accumulate) at the benchmarked
shapes -- every distinct layer shape of BASELINE.json configs[3], batch 64 --
against a float64 oracle with oracle-provided, f16-representable inputs, full-K dot products and NO storage
quantiser in the oracle.  Gate: 1e-3 relative to the tensor's max (north_star's tolerance; SURVEY section 7
"per layer with oracle-provided inputs at 1e-3").  Reference op: tf.nn.conv2d(x, W, [1,1,1,1], 'SAME') + bias
(src/yolo2_nets/darknet.py:20-21,32-36) and its two gradients (Conv2DBackpropInput / Conv2DBackpropFilter
behind minimize(), src/pascal/pascal_train_darknet.py:49-51).

The launches are the network's own: y2_conv2d / y2_conv2d_backward call the same policy (launch_conv,
launch_wgrad_auto) with the same (N, H, W, Cin, Cout, k), so the 512-pixel tiles of the 208x208 / 104x104
layers, the ring-form weight gradient with its real split-K depth, tail tiles and the 64-image border wrap
all run here.  A second group runs single-layer NETWORKS at the same shapes (BN statistics in the conv
epilogue, the BN passes, the in-network weight gradient) against float64 restatements of
tf.layers.batch_normalization + leaky + max_pool evaluated on the values as stored."""
import numpy as np
import pytest
import torch

from oracle import nn_ref as R

pytestmark = pytest.mark.gpu

N = 64
# (name, k, cin, cout, hw): SURVEY.md 8(d), conv1 (3 -> 32 at 416) has its own kernels and its own test below
C4_SHAPES = [
    ("conv2", 3, 32, 64, 208),
    ("conv3/5", 3, 64, 128, 104),
    ("conv4", 3, 128, 64, 104),
    ("conv6/8", 3, 128, 256, 52),
    ("conv7", 1, 256, 128, 52),
    ("conv9/11/13", 3, 256, 512, 26),
    ("conv10/12", 1, 512, 256, 26),
    ("conv14/16/18", 3, 512, 1024, 13),
    ("conv15/17", 1, 1024, 512, 13),
    ("head1-3", 3, 1024, 1024, 13),
    ("head_out", 1, 1024, 30, 13),
]
TOL = 1e-3


def f16_representable(a):
    return a.astype(np.float16).astype(np.float32)


def sample_pixels(n, hw, rng, extra=400):
    """linear pixel indices m = (n*H + h)*W + w: image corners and edges of the first / last image, the
    pixels either side of every plausible tile boundary (128..512-pixel tiles), the very last pixels, random"""
    M = n * hw * hw
    pts = {0, 1, hw - 1, hw, hw * hw - 1, hw * hw, M - 1, M - 2, M - hw, M - hw * hw, M - hw * hw - 1}
    for t in (128, 256, 384, 512):
        for q in (1, 2, 3, M // t // 2, M // t - 1, M // t):
            for d in (-1, 0, 1):
                pts.add(q * t + d)
    pts |= set(int(v) for v in rng.integers(0, M, extra))
    pts = np.array(sorted(p for p in pts if 0 <= p < M), dtype=np.int64)
    return pts


def gather_patches(t, pts, hw, k):
    """t [N,H,W,C] float32 -> float64 [len(pts), k*k*C] SAME-padded patches, taps row-major"""
    n = pts // (hw * hw)
    h = (pts // hw) % hw
    w = pts % hw
    r = k // 2
    C = t.shape[3]
    out = np.zeros((len(pts), k * k, C), np.float64)
    for dh in range(k):
        for dw in range(k):
            hh, ww = h + dh - r, w + dw - r
            ok = (hh >= 0) & (hh < hw) & (ww >= 0) & (ww < hw)
            out[ok, dh * k + dw, :] = t[n[ok], hh[ok], ww[ok], :]
    return out.reshape(len(pts), k * k * C)


def rel_to_max(got, ref):
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("name,k,cin,cout,hw", C4_SHAPES, ids=[s[0] for s in C4_SHAPES])
def test_c4_layer_shape_f16_vs_float64(name, k, cin, cout, hw):
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(k * 1000003 + cin * 1009 + cout * 31 + hw)
    x = f16_representable(rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32))
    w = f16_representable(np.clip(rng.normal(0, 0.1, (k, k, cin, cout)), -0.2, 0.2).astype(np.float32))
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    dy = f16_representable(rng.uniform(-1, 1, (N, hw, hw, cout)).astype(np.float32))
    xd, wd, dyd = torch.as_tensor(x).cuda(), torch.as_tensor(w).cuda(), torch.as_tensor(dy).cuda()
    pts = sample_pixels(N, hw, rng)

    # ---- forward: y = conv(x, W) + b at the sampled pixels, every cout, K = k*k*cin in float64
    y = E.conv2d(xd, wd, torch.as_tensor(b).cuda(), dtype="f16").cpu().numpy().reshape(-1, cout)
    ref = gather_patches(x, pts, hw, k) @ w.reshape(k * k * cin, cout).astype(np.float64) + b.astype(np.float64)
    e_fwd = rel_to_max(y[pts], ref)

    # ---- dgrad: dx = conv(dy, flip(W)^T) at the sampled pixels, every cin
    dx, dw = E.conv2d_backward(xd, wd, dyd, dtype="f16")
    dx = dx.cpu().numpy().reshape(-1, cin)
    wflip = w[::-1, ::-1].transpose(0, 1, 3, 2).reshape(k * k * cout, cin).astype(np.float64)
    ref = gather_patches(dy, pts, hw, k) @ wflip
    e_dx = rel_to_max(dx[pts], ref)

    # ---- wgrad: dW[t, ci, co] = sum over ALL N*H*W pixels, for a sample of (ci, co) pairs
    ci_s = np.unique(np.r_[0, 1, 31, 32 % cin, 63 % cin, cin - 1, rng.integers(0, cin, 6)])
    co_s = np.unique(np.r_[0, 1, 31 % cout, 32 % cout, cout - 1, rng.integers(0, cout, 6)])
    dw = dw.cpu().numpy()
    ref = np.zeros((k, k, len(ci_s), len(co_s)), np.float64)
    dys = dy[..., co_s].astype(np.float64)
    r = k // 2
    for dh in range(k):
        for dwi in range(k):
            h0, h1 = max(0, r - dh), min(hw, hw + r - dh)          # output rows whose tap (dh, dw) is inside
            w0, w1 = max(0, r - dwi), min(hw, hw + r - dwi)
            xs = x[:, h0 + dh - r:h1 + dh - r, w0 + dwi - r:w1 + dwi - r, :][..., ci_s].astype(np.float64)
            ref[dh, dwi] = np.einsum("nhwi,nhwo->io", xs, dys[:, h0:h1, w0:w1, :], optimize=True)
    e_dw = rel_to_max(dw[:, :, ci_s][:, :, :, co_s], ref)
    print("C4 %-13s f16 vs float64 (rel. to max): forward %.2e  dgrad %.2e  wgrad %.2e" % (name, e_fwd, e_dx, e_dw))
    assert e_fwd < TOL and e_dx < TOL and e_dw < TOL, (name, e_fwd, e_dx, e_dw)


NET_SHAPES = [("conv2+pool", 3, 32, 64, 208, 1), ("conv3", 3, 64, 128, 104, 0), ("conv5+pool", 3, 64, 128, 104, 1),
              ("conv8+pool", 3, 128, 256, 52, 1), ("conv7", 1, 256, 128, 52, 0), ("conv13+pool", 3, 256, 512, 26, 1),
              ("conv14", 3, 512, 1024, 13, 0), ("head1", 3, 1024, 1024, 13, 0)]


@pytest.mark.parametrize("name,k,cin,cout,hw,pool", NET_SHAPES, ids=[s[0] for s in NET_SHAPES])
def test_c4_layer_in_network_f16_bn_passes(name, k, cin, cout, hw, pool):
    """conv_bn_layer (darknet.py:32-46) as a single-layer network at a C4 shape, f16, batch 64, followed by a
    1x1 layer so that the layer under test also runs its dgrad-side passes: conv output (sampled float64), the
    epilogue's batch statistics, BN + leaky (+ pool) forward, BN backward (dy, dgamma, dbeta) and the in-network
    weight gradient -- each against float64 arithmetic on the values the device stored."""
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(k * 1000003 + cin * 1009 + cout * 31 + hw + 17 * pool + 5)
    spec = [(k, cin, cout, pool), (1, cout, 32, 0)]
    net = E.Network(spec, N, hw, hw, dtype="f16", training=True, grad_scale=1.0)
    params = R.init_params(spec, seed=4)
    for p in params:
        p["W"] = f16_representable(p["W"])
        p["gamma"] = rng.uniform(0.5, 1.5, p["gamma"].shape).astype(np.float32)
        p["beta"] = rng.uniform(-0.3, 0.3, p["beta"].shape).astype(np.float32)
        p["b"] = rng.uniform(-0.2, 0.2, p["b"].shape).astype(np.float32)
    net.load_params(params)
    x = f16_representable(rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32))
    out = net.forward(torch.as_tensor(x).cuda(), True, True)
    y = net.debug_read(0, 1).cpu().numpy()                       # conv output + bias as stored (f16 values)
    pts = sample_pixels(N, hw, rng, 200)
    ref = gather_patches(x, pts, hw, k) @ params[0]["W"].reshape(k * k * cin, cout).astype(np.float64) \
        + params[0]["b"].astype(np.float64)
    e_conv = rel_to_max(y.reshape(-1, cout)[pts], ref)
    # BN(train) + leaky (+ pool) of the stored y, float64
    y64 = y.astype(np.float64)
    mean, var = y64.mean((0, 1, 2)), y64.var((0, 1, 2))
    inv = 1.0 / np.sqrt(var + 1e-3)
    z = (y64 - mean) * inv * params[0]["gamma"] + params[0]["beta"]
    act = np.maximum(0.1 * z, z)
    a_ref = act.reshape(N, hw // 2, 2, hw // 2, 2, cout).max((2, 4)) if pool else act
    a = net.debug_read(1, 0).cpu().numpy()                       # input of layer 1 = output of the layer under test
    e_act = rel_to_max(a, a_ref)
    # backward: seed a gradient at the output, read the layer's dy and parameter gradients
    dout = rng.uniform(-1, 1, tuple(out.shape)).astype(np.float32)   # O(1): grad_scale is 1 here, keep f16 dy normal
    net.backward(torch.as_tensor(dout).cuda())
    g = net.export_grads()
    dy = net.debug_read(0, 2).cpu().numpy().astype(np.float64)   # d loss / d y as stored
    # what dy must be, from the stored dA (= dgrad of layer 1, checked at op level above) ...
    # recompute dA from layer 1's own stored dy and filter in float64 (1x1 conv: a plain matmul)
    dy1 = net.debug_read(1, 2).cpu().numpy().astype(np.float64)
    dA = (dy1.reshape(-1, 32) @ params[1]["W"].reshape(cout, 32).astype(np.float64).T)
    Ho = hw // 2 if pool else hw
    dA = f16_representable(dA.astype(np.float32)).astype(np.float64).reshape(N, Ho, Ho, cout)   # stored as f16
    if pool:
        zz = act.reshape(N, Ho, 2, Ho, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(N, Ho, Ho, 4, cout)
        first = zz.argmax(3)                                       # first maximum in row-major window order
        onehot = (np.arange(4)[None, None, None, :, None] == first[:, :, :, None, :])
        dact = (onehot * dA[:, :, :, None, :]).reshape(N, Ho, Ho, 2, 2, cout).transpose(0, 1, 3, 2, 4, 5)
        dact = dact.reshape(N, hw, hw, cout)
    else:
        dact = dA
    dz = dact * np.where(0.1 * z >= z, 0.1, 1.0)
    M = N * hw * hw
    xhat = (y64 - mean) * inv
    dbeta, dgamma = dz.sum((0, 1, 2)), (dz * xhat).sum((0, 1, 2))
    dy_ref = params[0]["gamma"] * inv * (dz - dbeta / M - xhat * dgamma / M)
    # Decision points: leaky'(z) at z ~ 0 and the arg-max of a 2x2 window with two near-equal maxima are
    # decided in fp32 on the device and in float64 here; of ~1e8 elements a handful sit within fp32 round-off of
    # the boundary and may legitimately fall the other way (each moves ONE dy entry by up to 0.9 |dA| scale).
    # Such entries must (a) be few and (b) all sit at a near-tie; everything else is gated at 1e-3 of the max.
    near = np.abs(z) < 1e-4
    if pool:
        zz_sorted = np.sort(zz, axis=3)
        tie = (zz_sorted[:, :, :, 3, :] - zz_sorted[:, :, :, 2, :]) < 1e-4
        near_w = near.reshape(N, Ho, 2, Ho, 2, cout).any((2, 4)) | tie
        near = np.repeat(np.repeat(near_w, 2, axis=1), 2, axis=2)
    bad = np.abs(dy - dy_ref) > TOL * np.abs(dy_ref).max()
    nbad = int(bad.sum())
    assert nbad <= 8 + 2e-7 * dy.size, ("too many dy entries off", nbad)
    assert not (bad & ~near).any(), "a dy entry is off away from any decision boundary"
    e_dy = rel_to_max(np.where(bad, dy_ref, dy), dy_ref)
    flip = nbad * 2.0 * np.abs(dA).max()                       # what the flipped entries can move a channel sum by
    e_dg = max(0.0, float(np.abs(g[0]["gamma"] - dgamma).max() - flip * np.abs(xhat).max())) / np.abs(dgamma).max()
    e_db = max(0.0, float(np.abs(g[0]["beta"] - dbeta).max() - flip)) / np.abs(dbeta).max()
    # in-network weight gradient from the stored x and stored dy, sampled (ci, co) pairs over all pixels
    ci_s = np.unique(np.r_[0, cin - 1, rng.integers(0, cin, 4)])
    co_s = np.unique(np.r_[0, cout - 1, rng.integers(0, cout, 4)])
    ref = np.zeros((k, k, len(ci_s), len(co_s)), np.float64)
    dys = dy[..., co_s]
    r = k // 2
    for dh in range(k):
        for dwi in range(k):
            h0, h1 = max(0, r - dh), min(hw, hw + r - dh)
            w0, w1 = max(0, r - dwi), min(hw, hw + r - dwi)
            xs = x[:, h0 + dh - r:h1 + dh - r, w0 + dwi - r:w1 + dwi - r, :][..., ci_s].astype(np.float64)
            ref[dh, dwi] = np.einsum("nhwi,nhwo->io", xs, dys[:, h0:h1, w0:w1, :], optimize=True)
    e_dw = rel_to_max(g[0]["W"][:, :, ci_s][:, :, :, co_s], ref)
    if e_dw > TOL:
        got = g[0]["W"][:, :, ci_s][:, :, :, co_s]
        print("DBG dW mismatch: per-tap max err", np.abs(got - ref).max((2, 3)), "ref max", np.abs(ref).max(),
              "ci_s", ci_s, "co_s", co_s, "got/ref sample", got[1, 1, :2, :2], ref[1, 1, :2, :2])
    print("C4 net %-12s f16: conv %.2e  bn+act %.2e  dy %.2e (%d near-tie flips)  dgamma %.2e  dbeta %.2e  dW %.2e" %
          (name, e_conv, e_act, e_dy, nbad, e_dg, e_db, e_dw))
    assert e_conv < TOL and e_act < TOL, (e_conv, e_act)
    # dy / dgamma / dbeta / dW are functions of the f16-stored dA and dy: their own storage rounding (2^-11 of
    # each value) stays inside 1e-3 of the max
    assert e_dy < TOL and e_dg < TOL and e_db < TOL and e_dw < TOL, (e_dy, e_dg, e_db, e_dw)


def test_c4_first_layer_f16_vs_float64():
    """conv1 (3 -> 32 at 416x416, batch 64: its own kernels, input stored with 4 channels, K 27 -> 48) + BN +
    leaky + pool forward, against float64 at sampled pixels / on the stored values."""
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(5)
    hw, cout = 416, 32
    spec = [(3, 3, 32, 1), (1, 32, 32, 0)]
    net = E.Network(spec, N, hw, hw, dtype="f16", training=True, grad_scale=1.0)
    params = R.init_params(spec, seed=6)
    params[0]["W"] = f16_representable(params[0]["W"])
    params[0]["gamma"] = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    params[0]["beta"] = rng.uniform(-0.3, 0.3, 32).astype(np.float32)
    net.load_params(params)
    x = f16_representable(rng.uniform(-1, 1, (N, hw, hw, 3)).astype(np.float32))
    net.forward(torch.as_tensor(x).cuda(), True, True)
    y = net.debug_read(0, 1).cpu().numpy()
    pts = sample_pixels(N, hw, rng, 600)
    ref = gather_patches(x, pts, hw, 3) @ params[0]["W"].reshape(27, cout).astype(np.float64) + params[0]["b"].astype(np.float64)
    e_conv = rel_to_max(y.reshape(-1, cout)[pts], ref)
    a = net.debug_read(1, 0).cpu().numpy()
    mean = np.zeros(cout); m2 = np.zeros(cout)
    for i in range(N):                                             # float64 statistics image by image (memory)
        yi = y[i].astype(np.float64).reshape(-1, cout)
        mean += yi.sum(0); m2 += (yi * yi).sum(0)
    M = N * hw * hw
    mean /= M
    var = m2 / M - mean * mean
    inv = 1.0 / np.sqrt(var + 1e-3)
    worst = 0.0
    amax = 0.0
    for i in range(0, N, 7):
        z = (y[i].astype(np.float64) - mean) * inv * params[0]["gamma"] + params[0]["beta"]
        act = np.maximum(0.1 * z, z).reshape(hw // 2, 2, hw // 2, 2, cout).max((1, 3))
        worst = max(worst, float(np.abs(a[i] - act).max()))
        amax = max(amax, float(np.abs(act).max()))
    print("C4 conv1 f16: conv %.2e  bn+act+pool %.2e" % (e_conv, worst / amax))
    assert e_conv < TOL and worst / amax < TOL


def test_full_detector_step_f32_416_bs8_vs_torch_oracle():
    """One whole detector step in the parity-grade mode at BASELINE.json configs[3]'s geometry (416x416, S=13;
    batch 8 to bound the host time of the oracle): grid_net, loss, ious, object_mask, and gradients against the
    PyTorch-CPU restatement (oracle/torch_ref.py, fp32 autograd).  The batch-64 run of the benchmarked f16 mode is
    covered by the per-shape tests above and the property test in test_gpu_net.py."""
    from oracle import torch_ref as T, loss_ref as L
    from tensorflow_yolo2_amd import engine as E, synthetic
    n, size, S = 8, 416, 13
    spec = E.CORE_SPEC + E.det_head_spec(30)
    params = R.init_params(spec, seed=0)
    x = synthetic.images(n, size, 1234)
    labels = synthetic.det_labels(n, size, S, 4321)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    tp = T.to_torch_params(params, torch.float32, requires_grad=True)
    rnet, _ = T.run_stack(torch.tensor(x), tp, R.CORE_SPEC + R.det_head_spec(30), True)
    rloss, rious, rmask, _ = T.get_loss(rnet.reshape(n, S, S, 30), torch.tensor(labels), 20, n, size, S, 2,
                                        L.yolo_grid_offset(S, 2))
    rloss.backward()
    net = E.Network(spec, n, size, size, dtype="f32", core_layers=18, training=True)
    net.load_params(params)
    grid = net.forward(torch.as_tensor(x).cuda(), True, True)
    e_grid = rel_to_max(grid.cpu().numpy(), rnet.detach().numpy().astype(np.float64))
    loss, ious, mask, dnet = E.yolo_loss(grid, torch.as_tensor(labels).cuda(), 20, n, size, S, 2)
    e_loss = abs(loss[4].item() - rloss.item()) / abs(rloss.item())
    mism = int((mask.cpu().numpy() != rmask.numpy()).sum())
    net.backward(dnet)
    g = net.export_grads()
    e_last = max(float(np.linalg.norm(g[21][k] - tp[21][k].grad.numpy()) / np.linalg.norm(tp[21][k].grad.numpy()))
                 for k in ("W", "gamma", "beta"))
    cosines = {}
    for l in (0, 7, 17, 18):
        a = g[l]["W"].ravel().astype(np.float64)
        b = tp[l]["W"].grad.numpy().ravel().astype(np.float64)
        cosines[l] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    print("C4 f32 step bs8: grid %.2e  loss %.2e  mask mismatches %d  last-layer grads %.2e  cos(dW) %s" %
          (e_grid, e_loss, mism, e_last, cosines))
    assert e_grid < TOL and e_loss < TOL and e_last < TOL
    # object_mask is index work: it may only differ where two IoUs tie to within fp32 round-off of the two sides
    assert mism == 0 or rel_to_max(ious.cpu().numpy(), rious.detach().numpy().astype(np.float64)) < 3e-3
    assert all(c > 0.999 for c in cosines.values()), cosines
