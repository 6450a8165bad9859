"""numpy restatement of src/yolo2_nets/darknet.py (layers, Darknet-19 backbone,
detection head, classifier) forward AND backward.  TEST INFRASTRUCTURE ONLY.

Tensors are NHWC, filters HWIO, exactly as the reference hands them to TF
(`tf.nn.conv2d(x, W, strides=[1,s,s,1], padding='SAME')`, darknet.py:20-21).
`dtype` selects the arithmetic type: float32 mirrors the reference, float64 is
used by the parity tests as the "exact" answer.
"""
import numpy as np

ALPHA = 0.1          # darknet.py:5
BN_EPS = 1e-3        # tf.layers.batch_normalization default (darknet.py:42-44 passes none)
BN_MOMENTUM = 0.99   # tf.layers.batch_normalization default

# (filter_size, in_chl, out_chl, maxpool_after) -- darknet.py:150-177
CORE_SPEC = [
    (3, 3, 32, True),
    (3, 32, 64, True),
    (3, 64, 128, False), (3, 128, 64, False), (3, 64, 128, True),
    (3, 128, 256, False), (1, 256, 128, False), (3, 128, 256, True),
    (3, 256, 512, False), (1, 512, 256, False), (3, 256, 512, False),
    (1, 512, 256, False), (3, 256, 512, True),
    (3, 512, 1024, False), (1, 1024, 512, False), (3, 512, 1024, False),
    (1, 1024, 512, False), (3, 512, 1024, False),
]


def det_head_spec(output_filter):
    """darknet.py:189-200 -- 3 x (3x3, 1024->1024) + 1x1 1024->output_filter."""
    return [(3, 1024, 1024, False)] * 3 + [(1, 1024, output_filter, False)]


CLS_HEAD_SPEC = [(1, 1024, 1000, False)]  # darknet.py:115


def scaled_spec(spec, width_div):
    """Channel-reduced variant used only to keep CPU tests small."""
    if width_div == 1:
        return list(spec)
    out = []
    for (k, ci, co, p) in spec:
        ci2 = ci if ci == 3 else max(ci // width_div, 32)
        co2 = max(co // width_div, 32)
        out.append((k, ci2, co2, p))
    return out


# --------------------------------------------------------------------------
# initialisers -- darknet.py:10-17
# --------------------------------------------------------------------------
def truncated_normal(rng, shape, stddev=0.1):
    """tf.truncated_normal: N(0, stddev), values beyond 2 sigma re-drawn."""
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def init_layer(rng, k, cin, cout):
    return {
        "W": truncated_normal(rng, (k, k, cin, cout)),        # darknet.py:10-12
        "b": np.full((cout,), 0.1, np.float32),               # darknet.py:15-17
        "gamma": np.ones((cout,), np.float32),                # BN defaults
        "beta": np.zeros((cout,), np.float32),
        "moving_mean": np.zeros((cout,), np.float32),
        "moving_var": np.ones((cout,), np.float32),
    }


def init_params(spec, seed=0):
    rng = np.random.default_rng(seed)
    return [init_layer(rng, k, ci, co) for (k, ci, co, _p) in spec]


# --------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------
def _same_pad(size, k, stride):
    """TF 'SAME' padding: (pad_before, pad_after)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


def conv2d_same(x, W, stride=1):
    """tf.nn.conv2d(..., padding='SAME') -- darknet.py:20-21."""
    kh, kw, ci, co = W.shape
    n, h, w, c = x.shape
    assert c == ci
    pt, pb = _same_pad(h, kh, stride)
    pl, pr = _same_pad(w, kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))
    win = win[:, ::stride, ::stride]              # [n, ho, wo, c, kh, kw]
    return np.einsum("nhwcij,ijco->nhwo", win, W, optimize=True)


def conv2d_same_backward(x, W, dy):
    """Gradients of conv2d_same (stride 1) wrt x and W
    (TF: Conv2DBackpropInput / Conv2DBackpropFilter)."""
    kh, kw, ci, co = W.shape
    n, h, w, _ = x.shape
    pt, pb = _same_pad(h, kh, 1)
    pl, pr = _same_pad(w, kw, 1)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))
    dW = np.einsum("nhwcij,nhwo->ijco", win, dy, optimize=True)
    # dx = full correlation of dy with the flipped, io-swapped filter
    dyp = np.pad(dy, ((0, 0), (kh - 1 - pt, kh - 1 - pb), (kw - 1 - pl, kw - 1 - pr), (0, 0)))
    winy = np.lib.stride_tricks.sliding_window_view(dyp, (kh, kw), axis=(1, 2))
    Wf = W[::-1, ::-1]                             # flip taps
    dx = np.einsum("nhwoij,ijco->nhwc", winy, Wf, optimize=True)
    return dx, dW


def subsample(x, factor):
    """slim resnet_utils.subsample: max_pool2d([1,1], stride=factor)
    (src/slim_dir/nets/resnet_utils.py:60-77)."""
    return x if factor == 1 else x[:, ::factor, ::factor, :]


def leaky(h):
    """tf.maximum(alpha*h, h) -- darknet.py:45."""
    return np.maximum(ALPHA * h, h)


def leaky_backward(h, dout):
    # TF maximum: gradient to the first argument (alpha*h) where alpha*h >= h,
    # i.e. h <= 0 (tie at h == 0 -> slope alpha), else to h.
    return np.where(ALPHA * h >= h, ALPHA * dout, dout)


def max_pool_2x2(x):
    """tf.nn.max_pool ksize 2 stride 2 'SAME' -- darknet.py:24-25.
    Odd sizes are padded bottom/right (padding never wins the max)."""
    n, h, w, c = x.shape
    ho, wo = -(-h // 2), -(-w // 2)
    xp = np.full((n, ho * 2, wo * 2, c), -np.inf, x.dtype)
    xp[:, :h, :w] = x
    xr = xp.reshape(n, ho, 2, wo, 2, c)
    return xr.max(axis=(2, 4))


def max_pool_2x2_backward(x, dout):
    """MaxPoolGrad: gradient goes to the first (row-major) max in each window."""
    n, h, w, c = x.shape
    ho, wo = -(-h // 2), -(-w // 2)
    xp = np.full((n, ho * 2, wo * 2, c), -np.inf, x.dtype)
    xp[:, :h, :w] = x
    xr = xp.reshape(n, ho, 2, wo, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, ho, wo, c, 4)
    arg = xr.argmax(axis=-1)                       # first max in (dy,dx) row-major order
    g = np.zeros((n, ho, wo, c, 4), dout.dtype)
    np.put_along_axis(g, arg[..., None], dout[..., None], axis=-1)
    g = g.reshape(n, ho, wo, c, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, ho * 2, wo * 2, c)
    return g[:, :h, :w]


def avg_pool_valid(x, k):
    """tf.layers.average_pooling2d(x, [k,k], [k,k]) (VALID) -- darknet.py:116."""
    n, h, w, c = x.shape
    ho, wo = h // k, w // k
    return x[:, :ho * k, :wo * k].reshape(n, ho, k, wo, k, c).mean(axis=(2, 4))


def avg_pool_valid_backward(x_shape, k, dout):
    n, h, w, c = x_shape
    ho, wo = h // k, w // k
    g = np.zeros(x_shape, dout.dtype)
    g[:, :ho * k, :wo * k] = np.repeat(np.repeat(dout, k, axis=1), k, axis=2) / (k * k)
    return g


def batch_norm_train(x, gamma, beta, eps=BN_EPS):
    """tf.layers.batch_normalization(training=True): biased batch moments over
    N,H,W (tf.nn.moments), then gamma*(x-mean)/sqrt(var+eps)+beta."""
    mean = x.mean(axis=(0, 1, 2))
    var = ((x - mean) ** 2).mean(axis=(0, 1, 2))
    inv = 1.0 / np.sqrt(var + eps)
    return gamma * (x - mean) * inv + beta, mean, var


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var, eps=BN_EPS):
    inv = 1.0 / np.sqrt(moving_var + eps)
    return gamma * (x - moving_mean) * inv + beta


def batch_norm_train_backward(x, gamma, mean, var, dz, eps=BN_EPS):
    m = x.shape[0] * x.shape[1] * x.shape[2]
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * inv
    dbeta = dz.sum(axis=(0, 1, 2))
    dgamma = (dz * xhat).sum(axis=(0, 1, 2))
    dx = gamma * inv * (dz - dbeta / m - xhat * dgamma / m)
    return dx, dgamma, dbeta


def moving_update(moving, batch, momentum=BN_MOMENTUM):
    """assign_moving_average: moving -= (moving - batch) * (1 - momentum)."""
    return moving - (moving - batch) * (1.0 - momentum)


# --------------------------------------------------------------------------
# conv_bn_layer -- darknet.py:32-46
# --------------------------------------------------------------------------
def quantizer(kind):
    """Storage-precision model of the fast modes: rounds to f16 / bf16 (round to
    nearest even) and returns float64.  None / 'f32' -> identity."""
    if kind in (None, "f32"):
        return None
    if kind == "f16":
        return lambda a: np.asarray(a).astype(np.float16).astype(np.float64)
    if kind == "bf16":
        def q(a):
            u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
            u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
            return u.astype(np.uint32).view(np.float32).astype(np.float64)
        return q
    raise ValueError(kind)


def conv_bn_layer(x, p, is_training, pool, dtype=np.float32, bessel=False, quant=None, given_stats=None):
    """Returns (out, cache, new_moving).  cache holds what backward needs.
    quant (see quantizer) models the half-precision STORAGE points of the HIP fast
    modes: filter, conv output, layer output (arithmetic in between stays wide).
    given_stats = (mean, var), training mode only: the batch moments the layer normalises with are INPUTS instead of
    being formed here.  The GPU tests hand over what the device normalised with (Network.layer_statistics) where the
    device forms its moments in another way than from the stored conv output -- the HIP path's pooled 3-channel first
    layer takes them from the Gram matrix of the input patches, csrc/conv1_wgrad.hip -- and gate those moments
    separately against float64 (tests/test_gpu_kernel_policies.py); the oracle itself knows nothing about that form."""
    W = p["W"].astype(dtype)
    if quant is not None:
        W = quant(W).astype(dtype)
    h_conv = conv2d_same(x, W) + p["b"].astype(dtype)                 # darknet.py:35
    if quant is not None:
        h_conv = quant(h_conv).astype(dtype)
    gamma, beta = p["gamma"].astype(dtype), p["beta"].astype(dtype)
    new_moving = None
    if is_training and given_stats is not None:
        mean, var = (np.asarray(v, dtype) for v in given_stats)
        h_bn = gamma * (h_conv - mean) / np.sqrt(var + BN_EPS) + beta
        m = h_conv.shape[0] * h_conv.shape[1] * h_conv.shape[2]
        var_upd = var * (m / max(m - 1, 1)) if bessel else var
        new_moving = (moving_update(p["moving_mean"].astype(dtype), mean),
                      moving_update(p["moving_var"].astype(dtype), var_upd))
    elif is_training:
        h_bn, mean, var = batch_norm_train(h_conv, gamma, beta)
        m = h_conv.shape[0] * h_conv.shape[1] * h_conv.shape[2]
        var_upd = var * (m / max(m - 1, 1)) if bessel else var
        new_moving = (moving_update(p["moving_mean"].astype(dtype), mean),
                      moving_update(p["moving_var"].astype(dtype), var_upd))
    else:
        mean, var = p["moving_mean"].astype(dtype), p["moving_var"].astype(dtype)
        h_bn = batch_norm_infer(h_conv, gamma, beta, mean, var)
    act = leaky(h_bn)                                                  # darknet.py:45
    out = max_pool_2x2(act) if pool else act
    cache = dict(x=x, h_conv=h_conv, h_bn=h_bn, act=act, mean=mean, var=var,
                 is_training=is_training, pool=pool)
    return out, cache, new_moving


def conv_bn_layer_backward(p, cache, dout, dtype=np.float32, need_dx=True, quant=None, grad_scale=1.0):
    W = p["W"].astype(dtype)
    if quant is not None:
        W = quant(W).astype(dtype)
        dout = quant(dout * grad_scale).astype(dtype) / grad_scale
    gamma = p["gamma"].astype(dtype)
    d_act = max_pool_2x2_backward(cache["act"], dout) if cache["pool"] else dout
    d_bn = leaky_backward(cache["h_bn"], d_act)
    if cache["is_training"]:
        d_conv, dgamma, dbeta = batch_norm_train_backward(
            cache["h_conv"], gamma, cache["mean"], cache["var"], d_bn)
    else:
        inv = 1.0 / np.sqrt(cache["var"] + BN_EPS)
        xhat = (cache["h_conv"] - cache["mean"]) * inv
        dgamma = (d_bn * xhat).sum(axis=(0, 1, 2))
        dbeta = d_bn.sum(axis=(0, 1, 2))
        d_conv = d_bn * gamma * inv
    db = d_conv.sum(axis=(0, 1, 2))
    if quant is not None:
        d_conv = quant(d_conv * grad_scale).astype(dtype) / grad_scale
    dx, dW = conv2d_same_backward(cache["x"], W, d_conv)
    grads = dict(W=dW, b=db, gamma=dgamma, beta=dbeta)
    return (dx if need_dx else None), grads


# --------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------
def run_stack(x, params, spec, is_training, dtype=np.float32, bessel=False, quant=None, first_stats=None):
    """is_training: bool, or a per-layer list of bools.  The LAST layer's output is
    never quantised (the HIP path emits it in fp32).
    first_stats = (mean, var): batch moments of layer 0 handed in as inputs (conv_bn_layer given_stats)."""
    caches, movings = [], []
    x = x.astype(dtype)
    if quant is not None:
        x = quant(x).astype(dtype)
    n = len(params)
    for i, (p, (_k, _ci, _co, pool)) in enumerate(zip(params, spec)):
        tr = is_training[i] if isinstance(is_training, (list, tuple)) else is_training
        x, cache, mv = conv_bn_layer(x, p, tr, pool, dtype, bessel, quant, given_stats=(first_stats if i == 0 else None))
        if quant is not None and i + 1 < n:
            x = quant(x).astype(dtype)
        caches.append(cache)
        movings.append(mv)
    return x, caches, movings


def run_stack_backward(params, caches, dout, dtype=np.float32, need_input_grad=False, quant=None, grad_scale=1.0):
    grads = [None] * len(params)
    for i in range(len(params) - 1, -1, -1):
        need_dx = need_input_grad or i > 0
        dout, grads[i] = conv_bn_layer_backward(params[i], caches[i], dout, dtype, need_dx, quant, grad_scale)
    return dout, grads


def darknet19_core(inputs, params, is_training=True, dtype=np.float32, spec=None):
    """darknet.py:126-179."""
    return run_stack(inputs, params, spec or CORE_SPEC, is_training, dtype)


def darknet19_detection(net, params, output_filter, is_training=True, dtype=np.float32, spec=None):
    """darknet.py:182-201 (callers never pass is_training -> batch statistics)."""
    return run_stack(net, params, spec or det_head_spec(output_filter), is_training, dtype)


def darknet19(inputs, params, is_training=True, dtype=np.float32, spec=None, pool_k=7):
    """darknet.py:61-123: core + 1x1 1024->1000 conv_bn + avgpool(7,7) + reshape."""
    spec = spec or (CORE_SPEC + CLS_HEAD_SPEC)
    h, caches, movings = run_stack(inputs, params, spec, is_training, dtype)
    pooled = avg_pool_valid(h, pool_k)
    logits = pooled.reshape(-1, spec[-1][2])
    return logits, (caches, h.shape, pool_k), movings


def darknet19_backward(params, ctx, dlogits, dtype=np.float32):
    caches, hshape, pool_k = ctx
    n = hshape[0]
    dpooled = dlogits.reshape(n, hshape[1] // pool_k, hshape[2] // pool_k, hshape[3])
    dh = avg_pool_valid_backward(hshape, pool_k, dpooled)
    return run_stack_backward(params, caches, dh, dtype)


def sparse_softmax_cross_entropy_mean(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits + reduce_mean
    (src/imagenet/imagenet_train_darknet.py:51-53).  Returns (loss, dlogits)."""
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    n = logits.shape[0]
    loss = -logp[np.arange(n), labels].mean()
    dl = np.exp(logp)
    dl[np.arange(n), labels] -= 1.0
    return loss, dl / n
