"""Drop-in counterparts of src/yolo2_nets/darknet.py (reference, TF1 graph builders).

Same function names, argument names and defaults:
    darknet19_core(inputs, num_classes=None, is_training=True, global_pool=True,
                   output_stride=None, reuse=None, scope='darknet19')       (darknet.py:126-179)
    darknet19_detection(net, output_filter, is_training=True,
                        scope='darknet19_detection', reuse=None)             (darknet.py:182-201)
    darknet19(inputs, ...)                                                   (darknet.py:61-123)
The reference returns symbolic tf.Tensors that a Session evaluates later; here the
functions return a lazy `NetTensor` for the same reason the reference does -- the
detection head extends the backbone's graph -- and `.eval()` (or passing it to
net_utils.get_loss / torch-like `.tensor()`) runs ONE fused stack on the MI355X through
libyolo2_hip.so.  `inputs` is a float32 NHWC torch tensor on the GPU (the reference
feeds NHWC float32 through a placeholder, pascal_train_darknet.py:34).

Variables: like tf.variable_scope, each `scope` owns its variables; calling a function
again with the same scope needs reuse=True (a second un-reused call raises, where TF
would silently create `Variable_22...` -- the reference never does that).
"""
import numpy as np
import torch

from .. import _lib
from ..engine import CORE_SPEC, CLS_HEAD_SPEC, det_head_spec, Network

alpha = 0.1   # darknet.py:5 (leaky slope, fixed in the kernels)

_DEFAULT_DTYPE = "f16"
_SCOPES = {}      # scope name -> {"spec": [...], "layers": list of dict(name -> torch tensor) or None}
_NETWORKS = {}    # (scope chain, shape, dtype, training) -> Network


def set_default_dtype(dtype):
    """'f32' (exact-f32 MFMA parity mode), 'f16' (default) or 'bf16'."""
    global _DEFAULT_DTYPE
    assert dtype in ("f32", "f16", "bf16")
    _DEFAULT_DTYPE = dtype


def reset_default_graph():
    """tf.reset_default_graph() counterpart: drops every scope, variable and cached network."""
    _SCOPES.clear()
    _NETWORKS.clear()


def _declare(scope, spec, reuse):
    if scope in _SCOPES:
        if not reuse:
            raise ValueError("variable scope %r already exists; pass reuse=True to share its variables" % scope)
        if _SCOPES[scope]["spec"] != spec:
            raise ValueError("scope %r was created with a different layer list" % scope)
    else:
        if reuse:
            raise ValueError("reuse=True but variable scope %r does not exist" % scope)
        _SCOPES[scope] = {"spec": spec, "seed": len(_SCOPES)}


class NetTensor:
    """Lazy output of a conv-BN-leaky stack (the counterpart of a symbolic tf.Tensor)."""

    def __init__(self, inputs, segments, tail=_lib.Y2_TAIL_NONE, tail_k=7):
        self.inputs = inputs
        self.segments = segments            # list of (scope, spec, is_training)
        self.tail, self.tail_k = tail, tail_k
        self._value = None
        self.network = None
        self.shape_override = None

    # --- graph plumbing ---------------------------------------------------
    def _network(self, training):
        x = self.inputs
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise TypeError("inputs must be a float32 NHWC torch tensor on the GPU")
        chain = tuple(s for (s, _sp, _t) in self.segments)
        key = (chain, tuple(x.shape), _DEFAULT_DTYPE, bool(training), self.tail)
        net = _NETWORKS.get(key)
        if net is None:
            spec = [l for (_s, sp, _t) in self.segments for l in sp]
            n_core = len(self.segments[0][1])
            net = Network(spec, x.shape[0], x.shape[1], x.shape[2], dtype=_DEFAULT_DTYPE, core_layers=n_core,
                          tail=self.tail, tail_k=self.tail_k, training=training, device=str(x.device))
            net.init_params(seed=0)
            # share variables with networks that already hold one of these scopes
            lo = 0
            for (scope, sp, _t) in self.segments:
                src = _SCOPES[scope].get("owner")
                if src is not None and src[0] is not net:
                    onet, olo = src
                    for i in range(len(sp)):
                        dst, srcv = net.layer_views(lo + i), onet.layer_views(olo + i)
                        for k in dst:
                            dst[k].copy_(srcv[k])
                    net.params_changed()
                else:
                    _SCOPES[scope]["owner"] = (net, lo)
                lo += len(sp)
            _NETWORKS[key] = net
        return net

    def run(self, training=False):
        """Evaluate (the `sess.run(tensor, feed)` of the reference scripts)."""
        net = self._network(training)
        self.network = net
        flags = [bool(t) for (_s, _sp, t) in self.segments]
        out = net.forward(self.inputs.contiguous(), flags[0], flags[-1] if len(flags) > 1 else flags[0])
        if self.shape_override is not None:
            out = out.view(self.shape_override)
        self._value = out
        return out

    def eval(self):
        return self.run(training=False)

    def tensor(self):
        return self._value if self._value is not None else self.eval()

    def reshape(self, shape):
        """tf.reshape(final_conv_layer, [-1, S, S, 5*B + NUM_CLASS]) (pascal_train_darknet.py:42)."""
        t = NetTensor(self.inputs, self.segments, self.tail, self.tail_k)
        n = self.inputs.shape[0]
        shape = [n if s == -1 else s for s in shape]
        t.shape_override = tuple(shape)
        return t


def _as_flag(is_training):
    if torch.is_tensor(is_training):
        return bool(is_training.item())
    return bool(is_training)


def darknet19_core(inputs, num_classes=None, is_training=True, global_pool=True, output_stride=None,
                   reuse=None, scope='darknet19'):
    """Darknet-19 backbone, 18 conv-BN-leaky layers + 5 max-pools -> [N, H/32, W/32, 1024]."""
    _declare(scope, list(CORE_SPEC), reuse)
    return NetTensor(inputs, [(scope, list(CORE_SPEC), _as_flag(is_training))])


def darknet19_detection(net, output_filter, is_training=True, scope='darknet19_detection', reuse=None):
    """3 x (3x3, 1024->1024) + 1x1 1024->output_filter, each conv-BN-leaky (the callers never pass
    is_training, so the head normalises with batch statistics even at detect time)."""
    if not isinstance(net, NetTensor):
        raise TypeError("net must be the output of darknet19_core")
    spec = det_head_spec(int(output_filter))
    _declare(scope, spec, reuse)
    return NetTensor(net.inputs, net.segments + [(scope, spec, _as_flag(is_training))])


def darknet19(inputs, num_classes=None, is_training=True, global_pool=True, output_stride=None, reuse=None,
              scope='darknet19'):
    """ImageNet classifier: backbone + 1x1 1024->1000 conv-BN-leaky + 7x7 average pool -> logits [N, 1000]."""
    spec = list(CORE_SPEC) + list(CLS_HEAD_SPEC)
    _declare(scope + "/cls", spec, reuse)
    k = inputs.shape[1] // 32
    return NetTensor(inputs, [(scope + "/cls", spec, _as_flag(is_training))], tail=_lib.Y2_TAIL_AVGPOOL, tail_k=k)


def variable_names(kind="detector", output_filter=30):
    """TF1 auto-generated variable names of the reference graph, in creation order
    (SURVEY.md section 5): darknet19/Variable, Variable_1, ..., batch_normalization_k/{gamma,beta,
    moving_mean,moving_variance}; head under darknet19_detection/conv{1,2,3}/..., output/...
    Returned per layer as dict(W, b, gamma, beta, moving_mean, moving_var) -> name."""
    out = []
    for i in range(len(CORE_SPEC) + (1 if kind == "classifier" else 0)):
        v = lambda j: "darknet19/Variable" + ("_%d" % j if j else "")
        bn = "darknet19/batch_normalization" + ("_%d" % i if i else "")
        out.append({"W": v(2 * i), "b": v(2 * i + 1), "gamma": bn + "/gamma", "beta": bn + "/beta",
                    "moving_mean": bn + "/moving_mean", "moving_var": bn + "/moving_variance"})
    if kind == "detector":
        for sub in ("conv1", "conv2", "conv3", "output"):
            p = "darknet19_detection/%s/" % sub
            out.append({"W": p + "Variable", "b": p + "Variable_1", "gamma": p + "batch_normalization/gamma",
                        "beta": p + "batch_normalization/beta", "moving_mean": p + "batch_normalization/moving_mean",
                        "moving_var": p + "batch_normalization/moving_variance"})
    return out
