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
_SCOPES = {}      # scope name -> {"spec": [...]}
_STORES = []      # variable stores: one flat parameter / state (/ gradient) buffer per chain of scopes
_NETWORKS = {}    # (scope chain, shape, dtype, training, tail) -> Network


def set_default_dtype(dtype):
    """'f32' (exact-f32 MFMA parity mode), 'f16' (default), 'bf16', 'f16x2' (reference tolerance on the f16 matrix pipe,
    split operands) or 'f16x2f' (f16x2 forward, single-product backward contractions)."""
    global _DEFAULT_DTYPE
    assert dtype in ("f32", "f16", "bf16", "f16x2", "f16x2f")
    _DEFAULT_DTYPE = dtype


def reset_default_graph():
    """tf.reset_default_graph() counterpart: drops every scope, variable and cached network."""
    _SCOPES.clear()
    _NETWORKS.clear()
    del _STORES[:]


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


class _VarStore:
    """The variables of a chain of scopes (darknet19 [+ darknet19_detection]) in ONE flat buffer, in the
    reference's creation order.  Every Network over that chain -- or over a leading part of it (the
    backbone alone), at any batch / input size, training or not -- is bound to views of these tensors,
    which is what tf.variable_scope(..., reuse=True) gives the reference (darknet.py:144,187): a train
    step is visible to every graph that shares the scope at once."""

    def __init__(self, chain, counts, device):
        self.chain = tuple(chain)
        self.counts = list(counts)            # per scope: (n_params, n_state)
        n_p = sum(c[0] for c in counts)
        n_s = sum(c[1] for c in counts)
        self.params = torch.zeros(n_p, dtype=torch.float32, device=device)
        self.state = torch.zeros(n_s, dtype=torch.float32, device=device)
        self.grads = None
        self.nets = []                        # (Network, number of leading scopes it covers)
        self.optimizers = {}                  # optimizer slots live with the variables (net_utils)
        self.version = 0                      # bumped whenever the variables change (optimizer step, restore)

    def sizes(self, nscopes):
        return (sum(c[0] for c in self.counts[:nscopes]), sum(c[1] for c in self.counts[:nscopes]))

    def views(self, nscopes, training):
        n_p, n_s = self.sizes(nscopes)
        if training and self.grads is None:
            self.grads = torch.zeros_like(self.params)
        return self.params[:n_p], (self.grads[:n_p] if training else None), self.state[:n_s]


def _find_store(chain):
    """(store, nscopes) whose chain starts with `chain`; None if `chain` is new or extends a store"""
    for st in _STORES:
        if st.chain[:len(chain)] == tuple(chain):
            return st
    return None


def _store_for(chain, net_factory):
    """Return the store holding `chain`'s variables, creating or extending one.
    net_factory(buffers) builds the Network for the full `chain` (needed to learn the sizes and to draw
    the initial values of new scopes on the device)."""
    chain = tuple(chain)
    st = _find_store(chain)
    if st is not None:
        return st, None
    # a store whose chain is a leading part of the requested one: extend it (the head joins the backbone)
    base = None
    for cand in _STORES:
        if chain[:len(cand.chain)] == cand.chain:
            base = cand
    for cand in _STORES:
        if cand is not base and set(cand.chain) & set(chain):
            raise NotImplementedError("scope(s) %s already live in the chain %s; one scope cannot be the prefix "
                                      "of two different stacks" % (sorted(set(cand.chain) & set(chain)), cand.chain))
    probe = net_factory(None)                 # own buffers: sizes + initial values for every scope of the chain
    probe.init_params(seed=0)
    counts, lo = [], 0
    for scope in chain:
        nl = len(_SCOPES[scope]["spec"])
        o0 = probe._offsets[lo]
        end_p = probe.n_params if lo + nl == probe.num_layers else probe._offsets[lo + nl][0]
        end_s = probe.n_state if lo + nl == probe.num_layers else probe._offsets[lo + nl][4]
        counts.append((end_p - o0[0], end_s - o0[4]))
        lo += nl
    st = _VarStore(chain, counts, probe.device)
    st.params.copy_(probe.params)
    st.state.copy_(probe.state)
    if base is not None:
        n_p, n_s = base.sizes(len(base.chain))
        st.params[:n_p].copy_(base.params)
        st.state[:n_s].copy_(base.state)
        if base.grads is not None:
            st.grads = torch.zeros_like(st.params)
        for (net, k) in base.nets:            # move the existing graphs onto the longer buffer
            net.rebind(*st.views(k, net.training))
            st.nets.append((net, k))
        _STORES.remove(base)
    _STORES.append(st)
    probe.rebind(*st.views(len(chain), probe.training))
    return st, probe


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
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.float32, torch.uint8) and x.dim() == 4):
            raise TypeError("inputs must be a float32 (or uint8 BGR pixel) NHWC torch tensor on the GPU")
        chain = tuple(s for (s, _sp, _t) in self.segments)
        key = (chain, tuple(x.shape), _DEFAULT_DTYPE, bool(training), self.tail)   # (float32 / uint8 inputs share a network)
        net = _NETWORKS.get(key)
        if net is None:
            spec = [l for (_s, sp, _t) in self.segments for l in sp]
            n_core = len(self.segments[0][1])

            def factory(buffers):
                return Network(spec, x.shape[0], x.shape[1], x.shape[2], dtype=_DEFAULT_DTYPE, core_layers=n_core,
                               tail=self.tail, tail_k=self.tail_k, training=training, device=str(x.device),
                               buffers=buffers)
            store, net = _store_for(chain, factory)
            if net is None:
                net = factory(store.views(len(chain), training))
                owner = store.nets[0][0].scale_owner if store.nets else None
                if owner is not None:
                    net.scale_owner = owner
                    net.set_grad_scale(owner.grad_scale)
            store.nets.append((net, len(chain)))
            net.store = store
            _NETWORKS[key] = net
        return net

    def build(self, training=False):
        """create (or find) the Network behind this tensor without running it -- the point at which the
        reference's variables exist (tf.Session + initializer), so that a checkpoint can be restored into
        them before the first run (pascal_detect_darknet.py:54-60)"""
        self.network = self._network(training)
        return self.network

    def run(self, training=False, update_moving=False):
        """Evaluate (the `sess.run(tensor, feed)` of the reference scripts)."""
        net = self._network(training)
        self.network = net
        flags = [bool(t) for (_s, _sp, t) in self.segments]
        if getattr(net, "_seen_version", None) != net.store.version:
            net.params_changed()             # the scope's variables moved under another graph: re-pack the filters
            net._seen_version = net.store.version
        out = net.forward(self.inputs.contiguous(), flags[0], flags[-1] if len(flags) > 1 else flags[0],
                          update_moving=update_moving)
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
