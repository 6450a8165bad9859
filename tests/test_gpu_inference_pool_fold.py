"""GPU tests (-m gpu) of round 5's pooled inference fold: conv + inference batch norm + leaky + 2x2 max pool in ONE launch of
the conv_haloq kernels (csrc/conv_haloq.hip, conv_epilogue.h: ConvArgs::aff_pool -- the tile's pixels in window-major
order, four consecutive patch rows = one window).  Reference: conv_bn_layer with pool (src/yolo2_nets/darknet.py:24-25,
39-46) at is_training=False (src/pascal/pascal_detect_darknet.py:41)."""
import numpy as np
import pytest
import torch

from oracle import nn_ref as R

pytestmark = pytest.mark.gpu

from _shapes import TOL, f16_representable, rel_to_max, gather_patches     # noqa: E402
import _obs                                                                # noqa: E402

# (N, hw, cin, cout, dtype): the two pooled conv_haloq layers of configs[1] (416x416, batch 32), the same layers at
# configs[2]'s sizes (224x224, batch 128), a few images (tiles that cross image boundaries and end inside a row pair),
# widths whose row pairs do not divide the tile, the exact-f32 mode
CASES = [(32, 52, 128, 256, "f16"), (32, 26, 256, 512, "f16"), (128, 28, 128, 256, "f16"), (128, 14, 256, 512, "f16"),
         (3, 52, 128, 256, "f16"), (7, 22, 256, 512, "f16"), (5, 36, 64, 128, "f16"), (9, 20, 128, 256, "bf16"),
         (4, 40, 128, 128, "f32")]


@pytest.mark.parametrize("N,hw,cin,cout,dtype", CASES)
def test_pooled_layer_in_the_inference_fold(N, hw, cin, cout, dtype):
    """A pooled 3x3 layer followed by a 1x1 layer, inference mode.  An inference binding folds the first layer's batch
    norm, activation and pool into its convolution launch; a training binding runs conv -> y -> bn_act(pool): the
    consumer's input tensor and the stack's output must be the same bits.  Then float64 on the stored input: conv,
    rounding to the storage type, affine, maximum over the window, leaky at a few hundred pooled pixels."""
    from tensorflow_yolo2_amd import engine as E
    rng = np.random.default_rng(hw * 11 + cin)
    spec = [(3, cin, cout, 1), (1, cout, 32, 0)]
    x32 = rng.uniform(-1, 1, (N, hw, hw, cin)).astype(np.float32)
    if dtype == "bf16":
        x32 = torch.as_tensor(x32).bfloat16().float().numpy()
    elif dtype == "f16":
        x32 = f16_representable(x32)
    x = torch.as_tensor(x32).cuda()
    params = R.init_params(spec, seed=4)
    for p in params:
        p["moving_mean"] = rng.uniform(-0.5, 0.5, p["moving_mean"].shape).astype(np.float32)
        p["moving_var"] = rng.uniform(0.5, 2.0, p["moving_var"].shape).astype(np.float32)
        # both signs of the scale: the maximum is taken on z = y * scale + shift, not on y
        p["gamma"] = (rng.uniform(0.5, 1.5, p["gamma"].shape) * rng.choice([-1.0, 1.0], p["gamma"].shape)).astype(np.float32)
        p["beta"] = rng.uniform(-0.3, 0.3, p["beta"].shape).astype(np.float32)
    outs = []
    for training in (False, True):
        net = E.Network(spec, N, hw, hw, dtype=dtype, core_layers=2, training=training)
        net.load_params(params)
        out = net.forward(x, False, False).clone()
        folded = False
        try:
            net.debug_read(0, 1)
        except E._lib.Y2Error as e:
            folded = "folded" in str(e)
        outs.append((out, net.debug_read(1, 0).clone(), folded))
        del net
    assert outs[0][2] and not outs[1][2], "the inference binding must take the folded form (and the training one must not)"
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0])
    # float64 at sampled pooled pixels
    q = {"f16": lambda a: f16_representable(a), "bf16": lambda a: torch.as_tensor(a).bfloat16().float().numpy(),
         "f32": lambda a: a}[dtype]
    W = q(params[0]["W"]).reshape(9 * cin, cout).astype(np.float64)
    sc = params[0]["gamma"].astype(np.float64) / np.sqrt(params[0]["moving_var"].astype(np.float64) + 1e-3)
    sh = params[0]["beta"].astype(np.float64) - params[0]["moving_mean"].astype(np.float64) * sc
    ho = hw // 2
    npts = 200
    n_i = rng.integers(0, N, npts)
    h_i = rng.integers(0, ho, npts)
    w_i = rng.integers(0, ho, npts)
    # corners of the first / last image and of row pairs: where a window-major tile begins and ends
    n_i[:4], h_i[:4], w_i[:4] = [0, 0, N - 1, N - 1], [0, ho - 1, 0, ho - 1], [0, ho - 1, ho - 1, 0]
    zs = []
    for dh in (0, 1):
        for dw in (0, 1):
            pts = (n_i * hw + 2 * h_i + dh) * hw + 2 * w_i + dw
            y = gather_patches(x, pts, hw, 3) @ W + params[0]["b"].astype(np.float64)
            yq = torch.as_tensor(y).to({"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]).double().numpy()
            zs.append(yq * sc + sh)
    zm = np.maximum(np.maximum(zs[0], zs[1]), np.maximum(zs[2], zs[3]))
    ref = np.maximum(0.1 * zm, zm)
    got = outs[0][1].reshape(N, ho, ho, cout)[torch.as_tensor(n_i).cuda(), torch.as_tensor(h_i).cuda(),
                                               torch.as_tensor(w_i).cuda()].double().cpu().numpy()
    tol = {"f16": TOL, "bf16": 8e-3, "f32": 1e-5}[dtype]
    _obs.gate("pooled folded inference layer %dx%d %d->%d N=%d %s" % (hw, hw, cin, cout, N, dtype), rel_to_max(got, ref), tol)
