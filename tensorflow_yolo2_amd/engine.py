"""Host-side driver of one network context of libyolo2_hip.so.

PyTorch is plumbing only: it owns device memory (flat parameter / gradient /
BN-state buffers, one workspace blob) and the HIP stream handle.  Every
computation is a hand-written gfx950 kernel behind the C ABI; nothing here
falls back to torch ops.

Reference counterpart: the TF1 graph + variables that
src/yolo2_nets/darknet.py builds, and the autodiff graph of
src/pascal/pascal_train_darknet.py:49-51.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check

# (filter_size, in_chl, out_chl, maxpool_after) -- src/yolo2_nets/darknet.py:150-177
CORE_SPEC = [
    (3, 3, 32, 1),
    (3, 32, 64, 1),
    (3, 64, 128, 0), (3, 128, 64, 0), (3, 64, 128, 1),
    (3, 128, 256, 0), (1, 256, 128, 0), (3, 128, 256, 1),
    (3, 256, 512, 0), (1, 512, 256, 0), (3, 256, 512, 0), (1, 512, 256, 0), (3, 256, 512, 1),
    (3, 512, 1024, 0), (1, 1024, 512, 0), (3, 512, 1024, 0), (1, 1024, 512, 0), (3, 512, 1024, 0),
]


def det_head_spec(output_filter):
    """src/yolo2_nets/darknet.py:189-200."""
    return [(3, 1024, 1024, 0)] * 3 + [(1, 1024, output_filter, 0)]


CLS_HEAD_SPEC = [(1, 1024, 1000, 0)]   # src/yolo2_nets/darknet.py:115

PARAM_KEYS = ("W", "b", "gamma", "beta")
STATE_KEYS = ("moving_mean", "moving_var")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ForwardGraph:
    """Network.forward captured into a HIP graph (Network.forward_graph)."""

    def __init__(self, net, is_training_core, is_training_head, uint8):
        self.net = net
        shape = (net.batch, net.height, net.width, net.spec[0][1])
        self.input = torch.zeros(shape, dtype=torch.uint8 if uint8 else torch.float32, device=net.device)
        self.output = torch.empty(net.out_shape, dtype=torch.float32, device=net.device)
        self._stream = torch.cuda.Stream(device=net.device)
        self._stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):
            for _ in range(2):
                net.forward(self.input, is_training_core, is_training_head, out=self.output, update_moving=False)
        self._stream.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=self._stream):
            net.forward(self.input, is_training_core, is_training_head, out=self.output, update_moving=False)
        self._modes = (is_training_core, is_training_head)
        self._gen = net._host_param_gen

    def replay(self):
        # load_params / init_params / params_changed re-pack the filters at the NEXT eager forward, outside any graph
        # (a replay silently ran the packs baked in at capture time).  One eager forward on the capture
        # stream brings the packed copies the graph reads up to date; the fused optimizer steps re-pack on the device
        # into those same buffers and need nothing.
        if self.net._host_param_gen != self._gen:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self.net.forward(self.input, self._modes[0], self._modes[1], out=self.output, update_moving=False)
            torch.cuda.current_stream().wait_stream(self._stream)
            self._gen = self.net._host_param_gen
        self._graph.replay()
        return self.output

    def __call__(self, images):
        self.input.copy_(images, non_blocking=True)
        return self.replay()


class Network:
    """One conv-BN-leaky(-pool) stack bound to device buffers."""

    def __init__(self, spec, batch, height, width, dtype="f16", core_layers=None, tail=_lib.Y2_TAIL_NONE,
                 tail_k=7, training=True, device="cuda:0", grad_scale=None, bessel=False, share_with=None,
                 buffers=None):
        self.lib = _lib.load()
        self._host_param_gen = 0        # bumped whenever the parameters change on a path that re-packs at the next forward
        if not torch.cuda.is_available():
            raise _lib.Y2Error("no MI355X visible: tensorflow_yolo2_amd has no CPU path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.spec = [tuple(int(v) for v in s) for s in spec]
        self.dtype = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        self.training = bool(training)
        self.batch, self.height, self.width = batch, height, width
        self.core_layers = len(spec) if core_layers is None else core_layers
        flat = (C.c_int * (4 * len(spec)))(*[v for s in self.spec for v in s])
        h = C.c_void_p()
        check(self.lib.y2_ctx_create(C.byref(h), flat, len(spec), self.core_layers, tail, tail_k,
                                     batch, height, width, self.dtype))
        self.h = h
        self.num_layers = len(spec)
        self.n_params = self.lib.y2_param_count(h)
        self.n_state = self.lib.y2_state_count(h)
        shp = (C.c_int * 4)()
        check(self.lib.y2_output_shape(h, shp))
        self.out_shape = tuple(shp)
        if tail == _lib.Y2_TAIL_AVGPOOL:
            self.out_shape = (shp[0], shp[3]) if shp[1] == 1 and shp[2] == 1 else tuple(shp)
        if share_with is not None:
            # same layers at another input size (the net is fully convolutional): ONE set of parameters,
            # gradients and BN state, one workspace per size (multi-scale training, 288 GB of HBM to spend)
            assert share_with.n_params == self.n_params and share_with.n_state == self.n_state
            self.params, self.grads, self.state = share_with.params, share_with.grads, share_with.state
            assert not training or self.grads is not None
        elif buffers is not None:
            # caller-owned flat buffers (a variable scope's store, yolo2_nets/darknet.py): exact-size views
            self.params, self.grads, self.state = self._check_buffers(*buffers)
        else:
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=self.device) if training else None
            self.state = torch.zeros(self.n_state, dtype=torch.float32, device=self.device)
        self.ws_bytes = self.lib.y2_workspace_bytes(h, int(training))
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        check(self.lib.y2_bind(h, _ptr(self.params), _ptr(self.grads), _ptr(self.state), _ptr(self.workspace),
                               self.ws_bytes, int(training), _stream()))
        self.scale_owner = share_with.scale_owner if share_with is not None else self
        if grad_scale is None:
            # f16 and the split-operand mode (f16 planes): gradients ride on a loss scale, dY's exponent range is f16's
            grad_scale = 1024.0 if self.dtype in _lib.LOSS_SCALED else 1.0
        self.grad_scale = float(grad_scale)
        self.bn_eps, self.bn_momentum, self.zero_bias_grad = 1e-3, 0.99, False     # y2_ctx defaults (darknet.py:39-44)
        self._bessel = bool(bessel)
        check(self.lib.y2_set_options(h, self.grad_scale, int(bessel)))
        self._offsets = []
        off = (C.c_size_t * 6)()
        for l in range(self.num_layers):
            check(self.lib.y2_param_offsets(h, l, off))
            self._offsets.append(tuple(off))
        self._out = torch.empty(self.out_shape, dtype=torch.float32, device=self.device)
        self.store = None          # yolo2_nets.darknet._VarStore when a variable scope owns the buffers

    def _check_buffers(self, params, grads, state):
        for t, n, need in ((params, self.n_params, True), (grads, self.n_params, self.training),
                           (state, self.n_state, True)):
            if t is None:
                if need:
                    raise ValueError("missing flat buffer")
                continue
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
                raise ValueError("flat buffer must be a contiguous float32 device tensor of %d elements" % n)
        return params, (grads if self.training else None), state

    def rebind(self, params, grads, state):
        """move this context onto other flat buffers (same sizes); the workspace is re-zeroed"""
        self.params, self.grads, self.state = self._check_buffers(params, grads, state)
        check(self.lib.y2_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.state), _ptr(self.workspace),
                               self.ws_bytes, int(self.training), _stream()))

    def set_layer_options(self, slopes=None, bn_eps=None, bn_momentum=None, zero_bias_grad=None):
        """per-layer activation slopes (0.1 leaky = the reference, 0 ReLU, 1 none) and the stack's batch-norm constants
        (y2_set_layer_options).  Every argument left at None KEEPS the context's current value (a fresh context holds
        tf.layers.batch_normalization's defaults: eps 1e-3, momentum 0.99, conv biases trainable) -- a call that only
        changes the slopes does not reset a ResNet stack's eps 1e-5 / momentum 0.997."""
        arr = None
        if slopes is not None:
            assert len(slopes) == self.num_layers
            arr = (C.c_float * self.num_layers)(*[float(v) for v in slopes])
        if bn_eps is not None:
            self.bn_eps = float(bn_eps)
        if bn_momentum is not None:
            self.bn_momentum = float(bn_momentum)
        if zero_bias_grad is not None:
            self.zero_bias_grad = bool(zero_bias_grad)
        check(self.lib.y2_set_layer_options(self.h, arr, self.num_layers, self.bn_eps, self.bn_momentum,
                                            int(self.zero_bias_grad)))

    def set_grad_scale(self, grad_scale):
        self.grad_scale = float(grad_scale)
        check(self.lib.y2_set_options(self.h, self.grad_scale, int(self._bessel)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.y2_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameters ------------------------------------------------------
    def _shapes(self, l):
        k, ci, co, _ = self.spec[l]
        return {"W": (k, k, ci, co), "b": (co,), "gamma": (co,), "beta": (co,),
                "moving_mean": (co,), "moving_var": (co,)}

    def layer_views(self, l, grads=False):
        """dict of views into the flat buffers (reference creation order W, b, gamma, beta)."""
        o = self._offsets[l]
        shp = self._shapes(l)
        src = self.grads if grads else self.params
        out = {}
        for i, k in enumerate(PARAM_KEYS):
            n = int(np.prod(shp[k]))
            out[k] = src[o[i]:o[i] + n].view(shp[k])
        if not grads:
            for i, k in enumerate(STATE_KEYS):
                n = int(np.prod(shp[k]))
                out[k] = self.state[o[4 + i]:o[4 + i] + n].view(shp[k])
        return out

    def init_params(self, seed=0):
        """darknet.py:10-17 initial values, generated on the device."""
        check(self.lib.y2_init_params(self.h, seed, _stream()))
        self._host_param_gen += 1

    def load_params(self, layers):
        """layers: list of dicts with numpy arrays W, b, gamma, beta, moving_mean, moving_var."""
        assert len(layers) == self.num_layers
        for l, p in enumerate(layers):
            v = self.layer_views(l)
            for k in PARAM_KEYS + STATE_KEYS:
                v[k].copy_(torch.as_tensor(np.asarray(p[k], np.float32)).to(self.device))
        check(self.lib.y2_params_changed(self.h))
        self._host_param_gen += 1
        if self.store is not None:
            self.store.version += 1

    def export_params(self):
        out = []
        for l in range(self.num_layers):
            out.append({k: v.detach().cpu().numpy().copy() for k, v in self.layer_views(l).items()})
        return out

    def export_grads(self):
        return [{k: v.detach().cpu().numpy().copy() for k, v in self.layer_views(l, grads=True).items()}
                for l in range(self.num_layers)]

    def params_changed(self):
        check(self.lib.y2_params_changed(self.h))
        self._host_param_gen += 1

    # ---- execution -------------------------------------------------------
    def forward(self, images, is_training_core=True, is_training_head=True, out=None, update_moving=False, join=None):
        """images: float32 NHWC in [-1, 1) (the reference's placeholder), or uint8 NHWC BGR pixels.
        join (fp32, the output's shape): the stack is the residual branch of a bottleneck unit and the result is
        relu(join + stack(images)) (slim_dir/nets/resnet_v1.py:112), written by the last layer's apply pass.
        update_moving: fold the BN moving-statistics update of the training-mode layers into this pass
        (the train step: UPDATE_OPS under train_op, pascal_train_darknet.py:49-51); False leaves them alone,
        as a sess.run that fetches only the output or the loss does (update_moving_stats() applies it later)."""
        assert images.is_cuda and images.is_contiguous() and images.dtype in (torch.float32, torch.uint8)
        assert tuple(images.shape[:3]) == (self.batch, self.height, self.width), images.shape
        out = self._out if out is None else out
        # uint8 BGR pixels (what cv2.imread + cv2.resize hand to image_read, pascal_voc.py:60-62): the conversion
        # x / 255 * 2 - 1 runs in the input pack kernel
        if join is not None:
            assert images.dtype == torch.float32 and join.is_cuda and join.dtype == torch.float32 and join.is_contiguous()
            assert join.numel() == out.numel() and join.data_ptr() != out.data_ptr()
            check(self.lib.y2_forward_join(self.h, _ptr(images), _ptr(join), int(bool(is_training_core)),
                                           int(bool(is_training_head)), int(bool(update_moving)), _ptr(out), _stream()))
            return out
        fwd = self.lib.y2_forward_u8 if images.dtype == torch.uint8 else self.lib.y2_forward
        check(fwd(self.h, _ptr(images), int(bool(is_training_core)), int(bool(is_training_head)),
                  int(bool(update_moving)), _ptr(out), _stream()))
        return out

    def forward_graph(self, is_training_core=False, is_training_head=True, uint8=False):
        """The forward pass as ONE HIP graph (serving: single-image detection is launch-bound -- ~40 kernels of 3-8 us
        behind ~4 us of host work each; a replay is one host call).  Returns a ForwardGraph: write the batch into
        `.input` (float32 NHWC, or uint8 BGR with uint8=True), call `.replay()`, read `.output` ([N,S,S,C] fp32; valid
        after a synchronisation of the current stream).  Two eager passes run first on the capture stream (filter packs,
        kernel attributes: neither may happen inside a capture).  The moving statistics are never updated by a replay.
        After a later load_params / init_params / params_changed the next replay first refreshes the packed filters
        with one eager forward (ForwardGraph.replay)."""
        return ForwardGraph(self, is_training_core, is_training_head, uint8)

    def update_moving_stats(self):
        check(self.lib.y2_update_moving_stats(self.h, _stream()))

    def backward(self, dout, layer_lo=0, layer_hi=None):
        layer_hi = self.num_layers if layer_hi is None else layer_hi
        if dout is not None:
            assert dout.is_cuda and dout.dtype == torch.float32 and dout.is_contiguous()
        check(self.lib.y2_backward(self.h, _ptr(dout), layer_lo, layer_hi, _stream()))

    # ---- linked stacks (y2_link: bottleneck units chained in the arithmetic type) -----------------------------
    def link(self, x=None, out=None, join=None, join_self=False, dout=None, dx=None):
        """tensors of the next forward / backward calls of this context: x / out / join are Bordered tensors, dout / dx
        [M][C] tensors of the arithmetic type (include/yolo2_hip.h y2_link); all None restores the fp32 interface"""
        cell = lambda b: None if b is None else C.c_void_p(b.cell0)
        raw = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._linked = (x, out, join, dout, dx)           # keep the tensors alive
        check(self.lib.y2_link(self.h, cell(x), cell(out), cell(join), int(bool(join_self)), raw(dout), raw(dx)))

    def forward_linked(self, is_training=True, update_moving=False, out=None, images=None):
        """forward with the linked tensors: `images` only when layer 0 is not linked, `out` (fp32) only when the output
        is not"""
        check(self.lib.y2_forward(self.h, _ptr(images) if images is not None else None, int(bool(is_training)),
                                  int(bool(is_training)), int(bool(update_moving)),
                                  _ptr(out) if out is not None else None, _stream()))
        return out

    def backward_linked(self, dout=None, dinput=None):
        """backward; dout (fp32) only when no typed output gradient is linked; dinput: fp32 [N,H,W,cin] to receive the
        input gradient as well (y2_backward_input)"""
        if dinput is not None:
            check(self.lib.y2_backward_input(self.h, _ptr(dout) if dout is not None else None, _ptr(dinput), _stream()))
        else:
            check(self.lib.y2_backward(self.h, _ptr(dout) if dout is not None else None, 0, self.num_layers, _stream()))

    def backward_input(self, dout):
        """full backward; also returns d loss / d input of the stack, fp32 [N,H,W,cin] (composed graphs)"""
        assert dout.is_cuda and dout.dtype == torch.float32 and dout.is_contiguous()
        dx = torch.empty((self.batch, self.height, self.width, self.spec[0][1]), dtype=torch.float32, device=self.device)
        check(self.lib.y2_backward_input(self.h, _ptr(dout), _ptr(dx), _stream()))
        return dx

    PROFILE_CATEGORIES = ("conv_fwd", "conv1_fwd", "dgrad", "wgrad", "conv1_wgrad", "bn_fwd", "bn_bwd", "misc")

    def backward_marks(self, dout, mark_layers):
        """full backward; mark k fires when every layer >= mark_layers[k] is complete (see wait_mark)"""
        assert dout.is_cuda and dout.dtype == torch.float32 and dout.is_contiguous()
        marks = (C.c_int * len(mark_layers))(*[int(v) for v in mark_layers])
        check(self.lib.y2_backward_marks(self.h, _ptr(dout), len(mark_layers), marks, _stream()))

    def wait_mark(self, k, stream):
        """make `stream` (a torch.cuda.Stream) wait until mark k of the last backward_marks has fired"""
        check(self.lib.y2_wait_mark(self.h, int(k), C.c_void_p(stream.cuda_stream)))

    def profile_enable(self, on=1):
        """0 off; 1 bracket every launch with HIP events (serialised: no side stream); 2 only the MFMA convolution
        launches (forward, dgrad, wgrad), streams as in production"""
        check(self.lib.y2_profile_enable(self.h, int(on)))

    def profile_busy(self, categories=("conv_fwd", "dgrad", "wgrad")):
        """(busy milliseconds, launches): union of the intervals of those launch categories (call before collect)"""
        mask = sum(1 << self.PROFILE_CATEGORIES.index(k) for k in categories)
        ms, n = C.c_double(), C.c_int()
        check(self.lib.y2_profile_busy(self.h, mask, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_layers(self):
        """[num_layers][8] milliseconds per layer and category of the records so far (call before collect)"""
        buf = (C.c_double * (self.num_layers * len(self.PROFILE_CATEGORIES)))()
        check(self.lib.y2_profile_layers(self.h, buf))
        return np.array(buf).reshape(self.num_layers, len(self.PROFILE_CATEGORIES))

    def profile_collect(self):
        n = len(self.PROFILE_CATEGORIES)
        ms = (C.c_double * n)()
        cnt = (C.c_int * n)()
        check(self.lib.y2_profile_collect(self.h, ms, cnt, n))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.PROFILE_CATEGORIES)}

    def layer_statistics(self, layer):
        """the per-channel constants the last forward normalised `layer` with (y2_debug_read selector 3): dict of float64
        arrays mean, invstd, scale, shift and var = 1 / invstd^2 - eps (the biased batch variance, or the moving one)"""
        co = self.spec[layer][2]
        t = torch.empty((4, co), dtype=torch.float32, device=self.device)
        check(self.lib.y2_debug_read(self.h, layer, 3, _ptr(t), _stream()))
        a = t.double().cpu().numpy()
        eps = self.bn_eps
        return {"mean": a[0], "invstd": a[1], "scale": a[2], "shift": a[3], "var": 1.0 / (a[1] * a[1]) - eps}

    def debug_read(self, layer, what):
        k, ci, co, _ = self.spec[layer]
        info = (C.c_int * 8)()
        check(self.lib.y2_layer_info(self.h, layer, info))
        c = ci if what == 0 else co
        t = torch.empty((self.batch, info[4], info[5], c), dtype=torch.float32, device=self.device)
        check(self.lib.y2_debug_read(self.h, layer, what, _ptr(t), _stream()))
        return t


class Bordered:
    """a zero-bordered activation tensor [N][H+1][W+1][C] of the arithmetic type with its guard bands (csrc/common.h
    bpix), allocated zeroed once: what two linked stacks share (Network.link)"""

    def __init__(self, n, h, w, c, dtype, device):
        lib = _lib.load()
        dt = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        if dt not in (_lib.Y2_F32, _lib.Y2_F16, _lib.Y2_BF16):
            # (the type map below has no entry for the split-operand modes, whose cells are two half planes;
            #  y2_link refuses them too)
            raise ValueError("Bordered: linked tensors exist in f32 / f16 / bf16 (not the split-operand modes f16x2 / f16x2f)")
        off = C.c_size_t(0)
        nbytes = lib.y2_bordered_bytes(n, h, w, c, dt, C.byref(off))
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.cell0 = self.buf.data_ptr() + off.value
        self.shape = (n, h, w, c)
        self._off, self._tdtype = off.value, {_lib.Y2_F32: torch.float32, _lib.Y2_F16: torch.float16, _lib.Y2_BF16: torch.bfloat16}[dt]

    def _interior(self):
        """view of the image cells [N, H, W, C] in the tensor's own type: cell (n, h, w) sits at row n (H + 1) + h + 1,
        column w + 1 of a (W + 1)-wide pixel grid that starts at cell 0"""
        n, h, w, c = self.shape
        esz = torch.empty(0, dtype=self._tdtype).element_size()
        rows = n * (h + 1)
        body = self.buf[self._off:self._off + rows * (w + 1) * c * esz].view(self._tdtype).view(n, h + 1, w + 1, c)
        return body[:, 1:, 1:, :]

    def write(self, x):
        """tests: fill the image cells from an fp32 NHWC tensor (borders stay zero)"""
        self._interior().copy_(x.to(self._tdtype))

    def read(self):
        return self._interior().float()


def join_backward(dtype, out_bordered, d1, d2, g, stride=1):
    """g = (d1 + d2) * [out > 0] between linked stacks (y2_join_backward); d2 fp32 or of the arithmetic type.
    stride 2: the unit above is a stride-2 unit, d2 lives on ITS output grid (y2_join_backward_s2)"""
    lib = _lib.load()
    dt = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    n, h, w, c = out_bordered.shape
    fn = lib.y2_join_backward if stride == 1 else lib.y2_join_backward_s2
    check(fn(dt, C.c_void_p(out_bordered.cell0), C.c_void_p(d1.data_ptr()), C.c_void_p(d2.data_ptr()),
             int(d2.dtype == torch.float32), C.c_void_p(g.data_ptr()), n, h, w, c, _stream()))
    return g


def subsample_bordered(dtype, src, dst):
    """resnet_utils.subsample(x, 2) between two Bordered tensors of the arithmetic type (y2_subsample_bordered)"""
    lib = _lib.load()
    dt = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    n, h, w, c = src.shape
    assert tuple(dst.shape) == (n, h // 2, w // 2, c)
    check(lib.y2_subsample_bordered(dt, C.c_void_p(src.cell0), C.c_void_p(dst.cell0), n, h, w, c, _stream()))
    return dst


# ---------------------------------------------------------------------------
# flat-buffer optimizers
# ---------------------------------------------------------------------------
class LossScaler:
    """Dynamic loss scale for the half-precision modes (the fp32 reference has no counterpart).

    The guarded optimizer kernels skip a step whose gradient buffer holds an inf / NaN on the device, without a
    host synchronisation; this object reads the control words ONE STEP LATE (asynchronous copy into pinned
    memory) and halves grad_scale after an overflow, doubling it again after `growth_interval` clean steps up
    to the initial scale's ceiling.  ctrl words: found_inf, step, skipped, ticket, lr_t."""

    def __init__(self, net, growth_interval=2000, max_scale=65536.0, min_scale=1.0):
        self.net = net.scale_owner
        self.nets = [net]
        self.ctrl = torch.zeros(8, dtype=torch.int32, device=net.device)
        self._host = torch.zeros(8, dtype=torch.int32).pin_memory()
        self._event = None
        self.growth_interval, self.max_scale, self.min_scale = growth_interval, max_scale, min_scale
        self.enabled = net.dtype in _lib.LOSS_SCALED
        self._clean = 0
        self.overflows = 0

    @property
    def scale(self):
        return self.net.grad_scale

    def _apply(self, scale):
        seen = set()
        for n in [self.net] + self.nets:
            if id(n) not in seen:
                seen.add(id(n))
                n.set_grad_scale(scale)

    def attach(self, net):
        """another context on the same flat buffers (multi-scale): keep its scale in step"""
        if all(n is not net for n in self.nets):
            self.nets.append(net)
            net.set_grad_scale(self.scale)

    def scan(self, net, full=False, more=False):
        """set ctrl.found_inf from net's gradient buffer: sentinel ranges (default) or every element;
        more: OR into the flag another stack's scan of this step left (composed graphs: one all-or-nothing step)"""
        if full:
            check(net.lib.y2_grad_check_full(_ptr(net.grads), net.n_params, _ptr(self.ctrl), _stream()))
        elif more:
            check(net.lib.y2_grad_check_more(net.h, _ptr(self.ctrl), _stream()))
        else:
            check(net.lib.y2_grad_check(net.h, _ptr(self.ctrl), _stream()))

    def after_step(self):
        """call right after the guarded optimizer launch of a step"""
        if self._event is not None:          # the copy issued after the PREVIOUS step has long finished
            self._event.synchronize()
            if int(self._host[0]) != 0:
                self.overflows += 1
                self._clean = 0
                if self.enabled:
                    self._apply(max(self.scale * 0.5, self.min_scale))
            else:
                self._clean += 1
                if self.enabled and self._clean >= self.growth_interval and self.scale < self.max_scale:
                    self._clean = 0
                    self._apply(min(self.scale * 2.0, self.max_scale))
        self._host.copy_(self.ctrl, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def state(self):
        """(found_inf, device step counter, skipped steps) -- synchronises"""
        c = self.ctrl.cpu().numpy()
        return int(c[0]), int(c[1]), int(c[2])


class AdamOptimizer:
    """tf.train.AdamOptimizer defaults (src/pascal/pascal_train_darknet.py:51).
    guard=True (default for the f16 mode): the overflow-safe kernels + LossScaler; the step counter then lives
    on the device (`t` mirrors it assuming no skipped step; `scaler.state()` is authoritative)."""

    def __init__(self, net, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, guard=None, fused_pack=True):
        self.net, self.lr, self.b1, self.b2, self.eps = net, learning_rate, beta1, beta2, epsilon
        self.fused_pack = bool(fused_pack)
        self.m = torch.zeros_like(net.params)
        self.v = torch.zeros_like(net.params)
        self.t = 0
        self.guard = (net.dtype in _lib.LOSS_SCALED) if guard is None else bool(guard)
        self.scaler = LossScaler(net) if self.guard else None

    def step(self, grad_mult=1.0, full_check=False, joint=None):
        """full_check: scan every gradient element instead of the sentinel ranges (a gradient buffer that was
        not produced by this context's backward pass).
        joint: None, or this optimizer's place in a group that shares ONE LossScaler over a composed graph
        ("first": the caller has scanned every stack into the shared flag -- advance the step counter once;
        "next": use the counter / lr_t the first stack's call advanced)."""
        self.t += 1
        n = self.net
        if self.guard and joint is None:
            self.scaler.scan(n, full_check)
        ctrl = _ptr(self.scaler.ctrl) if self.guard else C.c_void_p(0)
        if joint is not None:
            assert self.guard and self.fused_pack and n.training
            check(n.lib.y2_adam_step_packed(n.h, _ptr(self.m), _ptr(self.v), ctrl, self.t if joint == "first" else -1,
                                            self.lr, self.b1, self.b2, self.eps, grad_mult, _stream()))
            if joint == "first":
                self.scaler.after_step()
            return
        if self.fused_pack and n.training:
            # update + filter re-pack in one pass over the parameters (the context's packed copies stay current)
            check(n.lib.y2_adam_step_packed(n.h, _ptr(self.m), _ptr(self.v), ctrl, self.t, self.lr, self.b1, self.b2,
                                            self.eps, grad_mult, _stream()))
        elif self.guard:
            check(n.lib.y2_adam_step_guarded(_ptr(n.params), _ptr(self.m), _ptr(self.v), _ptr(n.grads), n.n_params,
                                             ctrl, self.lr, self.b1, self.b2, self.eps, grad_mult, _stream()))
            n.params_changed()
        else:
            check(n.lib.y2_adam_step(_ptr(n.params), _ptr(self.m), _ptr(self.v), _ptr(n.grads), n.n_params, self.t,
                                     self.lr, self.b1, self.b2, self.eps, grad_mult, _stream()))
            n.params_changed()
        if self.guard:
            self.scaler.after_step()

    def backward_step(self, dout, grad_mult=1.0):
        """train_op in one library call: net.backward(dout) + the (guarded) step + filter re-pack, with the update
        of the layers above the first one overlapped with the first layer's gradient kernel (single-process
        training; the data-parallel trainers reduce between backward() and step())."""
        n = self.net
        if not (self.fused_pack and n.training) or os.environ.get("Y2_NO_FUSED_TRAIN_OP"):
            n.backward(dout)
            return self.step(grad_mult)
        self.t += 1
        ctrl = _ptr(self.scaler.ctrl) if self.guard else C.c_void_p(0)
        dout = dout.contiguous()
        check(n.lib.y2_backward_adam(n.h, _ptr(dout), _ptr(self.m), _ptr(self.v), ctrl, self.t, self.lr, self.b1,
                                     self.b2, self.eps, grad_mult, _stream()))
        if self.guard:
            self.scaler.after_step()

    # ---- tf.train.Saver slots (a resumed run must not restart Adam at t = 0)
    def export_state(self):
        t = self.scaler.state()[1] if self.guard else self.t
        return {"m": self.m.detach().cpu().numpy().copy(), "v": self.v.detach().cpu().numpy().copy(), "t": int(t)}

    def load_state(self, st):
        self.m.copy_(torch.as_tensor(np.asarray(st["m"], np.float32)).to(self.m.device))
        self.v.copy_(torch.as_tensor(np.asarray(st["v"], np.float32)).to(self.v.device))
        self.t = int(st["t"])
        if self.guard:
            c = self.scaler.ctrl.cpu()
            c[1] = self.t
            self.scaler.ctrl.copy_(c)


class MomentumOptimizer:
    """tf.train.MomentumOptimizer(0.001, 0.9) (src/imagenet/imagenet_train_darknet.py:58)."""

    def __init__(self, net, learning_rate=1e-3, momentum=0.9, guard=None, fused_pack=True):
        self.net, self.lr, self.mom = net, learning_rate, momentum
        self.fused_pack = bool(fused_pack)
        self.accum = torch.zeros_like(net.params)
        self.guard = (net.dtype in _lib.LOSS_SCALED) if guard is None else bool(guard)
        self.scaler = LossScaler(net) if self.guard else None

    def step(self, grad_mult=1.0, full_check=False):
        n = self.net
        if self.guard:
            self.scaler.scan(n, full_check)
        ctrl = _ptr(self.scaler.ctrl) if self.guard else C.c_void_p(0)
        if self.fused_pack and n.training:
            check(n.lib.y2_momentum_step_packed(n.h, _ptr(self.accum), ctrl, self.lr, self.mom, grad_mult, _stream()))
        elif self.guard:
            check(n.lib.y2_momentum_step_guarded(_ptr(n.params), _ptr(self.accum), _ptr(n.grads), n.n_params,
                                                 ctrl, self.lr, self.mom, grad_mult, _stream()))
            n.params_changed()
        else:
            check(n.lib.y2_momentum_step(_ptr(n.params), _ptr(self.accum), _ptr(n.grads), n.n_params, self.lr,
                                         self.mom, grad_mult, _stream()))
            n.params_changed()
        if self.guard:
            self.scaler.after_step()

    def backward_step(self, dout, grad_mult=1.0):
        """see AdamOptimizer.backward_step"""
        n = self.net
        if not (self.fused_pack and n.training) or os.environ.get("Y2_NO_FUSED_TRAIN_OP"):
            n.backward(dout)
            return self.step(grad_mult)
        ctrl = _ptr(self.scaler.ctrl) if self.guard else C.c_void_p(0)
        dout = dout.contiguous()
        check(n.lib.y2_backward_momentum(n.h, _ptr(dout), _ptr(self.accum), ctrl, self.lr, self.mom, grad_mult,
                                         _stream()))
        if self.guard:
            self.scaler.after_step()

    def export_state(self):
        return {"accum": self.accum.detach().cpu().numpy().copy()}

    def load_state(self, st):
        self.accum.copy_(torch.as_tensor(np.asarray(st["accum"], np.float32)).to(self.accum.device))


# ---------------------------------------------------------------------------
# loss / decode ops on device tensors
# ---------------------------------------------------------------------------
LAMBDA_COORD = 5.0   # src/config.py:44
LAMBDA_NOOBJ = 0.5   # src/config.py:45


def yolo_loss(net, labels, num_class, batch_size, image_size, S, B, need_grad=True,
              lambda_coord=LAMBDA_COORD, lambda_noobj=LAMBDA_NOOBJ):
    """get_loss (src/yolo2_nets/net_utils.py:263-372) forward (+ d loss / d net).
    Returns (loss[5] = class, object, noobject, coord, total; ious; object_mask; dnet or None)."""
    lib = _lib.load()
    assert net.is_cuda and labels.is_cuda
    net = net.contiguous().float()
    labels = labels.contiguous().float()
    assert net.numel() == batch_size * S * S * (num_class + 5 * B), "net shape does not match batch_size/S/B"
    assert labels.numel() == batch_size * S * S * (5 + num_class)
    dev = net.device
    loss = torch.empty(5, dtype=torch.float32, device=dev)
    ious = torch.empty((batch_size, S, S, B), dtype=torch.float32, device=dev)
    mask = torch.empty((batch_size, S, S, B), dtype=torch.float32, device=dev)
    dnet = torch.empty((batch_size, S, S, num_class + 5 * B), dtype=torch.float32, device=dev) if need_grad else None
    ws = torch.empty(lib.y2_yolo_loss_workspace_bytes(batch_size, S), dtype=torch.uint8, device=dev)
    check(lib.y2_yolo_loss(_ptr(net), _ptr(labels), num_class, batch_size, float(image_size), S, B,
                           lambda_coord, lambda_noobj, _ptr(loss), _ptr(ious), _ptr(mask), _ptr(dnet), _ptr(ws),
                           _stream()))
    return loss, ious, mask, dnet


def get_iou(boxes1, boxes2):
    """src/yolo2_nets/net_utils.py:222-260 on [..., 4] device tensors."""
    lib = _lib.load()
    b1 = boxes1.contiguous().float()
    b2 = boxes2.contiguous().float()
    assert b1.shape == b2.shape and b1.shape[-1] == 4
    out = torch.empty(b1.shape[:-1], dtype=torch.float32, device=b1.device)
    check(lib.y2_get_iou(_ptr(b1), _ptr(b2), _ptr(out), out.numel(), _stream()))
    return out


def decode_detections(predict, S, B, num_class, im_w, im_h, object_thresh=0.5):
    """Arithmetic of show_yolo_detection (src/yolo2_nets/net_utils.py:393-421).
    Returns the reference's tuples (ulx, uly, w, h, cls, conf, c, r, i) in its loop order."""
    lib = _lib.load()
    p = predict.contiguous().float().view(-1)
    assert p.numel() == S * S * (num_class + 5 * B)
    det = torch.empty((S * S * B, 8), dtype=torch.int32, device=p.device)
    conf = torch.empty(S * S * B, dtype=torch.float32, device=p.device)
    check(lib.y2_decode_detections(_ptr(p), S, B, num_class, im_w, im_h, float(object_thresh), _ptr(det),
                                   _ptr(conf), _stream()))
    det = det.cpu().numpy()
    conf = conf.cpu().numpy()
    out = []
    for i in range(S * S * B):
        if det[i, 0]:
            out.append((int(det[i, 1]), int(det[i, 2]), int(det[i, 3]), int(det[i, 4]), int(det[i, 5]),
                        float(conf[i]), int(det[i, 6]), int(det[i, 7]), i % B))
    return out


# ---- YOLOv2 pieces beyond the reference (SURVEY §8 a-x1/a-x2; specification: oracle/ext_ref.py) ----------
def max_pool_2x2(x):
    """tf.nn.max_pool(x, 2, 2, 'SAME') on [N,H,W,C] fp32 (darknet.py:24-25)"""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=x.device)
    check(lib.y2_maxpool2x2(_ptr(x), _ptr(y), n, h, w, c, _stream()))
    return y


def max_pool_2x2_backward(x, dy):
    lib = _lib.load()
    n, h, w, c = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and tuple(dy.shape) == (n, (h + 1) // 2, (w + 1) // 2, c)
    dx = torch.empty_like(x)
    check(lib.y2_maxpool2x2_backward(_ptr(x), _ptr(dy), _ptr(dx), n, h, w, c, _stream()))
    return dx


def reorg(x, stride=2, inverse=False):
    """space-to-depth [N,H,W,C] -> [N,H/s,W/s,s*s*C]; inverse=True: the gradient (depth-to-space)"""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    s = int(stride)
    if not inverse:
        n, h, w, c = x.shape
        y = torch.empty((n, h // s, w // s, s * s * c), dtype=torch.float32, device=x.device)
        check(lib.y2_reorg(_ptr(x), _ptr(y), n, h, w, c, s, 1, _stream()))
    else:
        n, ho, wo, cc = x.shape
        c = cc // (s * s)
        y = torch.empty((n, ho * s, wo * s, c), dtype=torch.float32, device=x.device)
        check(lib.y2_reorg(_ptr(x), _ptr(y), n, ho * s, wo * s, c, s, 0, _stream()))
    return y


def passthrough_concat(fine, coarse):
    """concat(reorg2(fine [N,2H,2W,Cf]), coarse [N,H,W,Cc]) -> [N,H,W,4*Cf+Cc]"""
    lib = _lib.load()
    n, h, w, cc = coarse.shape
    cf = fine.shape[3]
    assert tuple(fine.shape[:3]) == (n, 2 * h, 2 * w) and fine.is_contiguous() and coarse.is_contiguous()
    out = torch.empty((n, h, w, 4 * cf + cc), dtype=torch.float32, device=coarse.device)
    check(lib.y2_passthrough_concat(_ptr(fine), _ptr(coarse), _ptr(out), n, h, w, cf, cc, _stream()))
    return out


def passthrough_concat_backward(dout, cf):
    lib = _lib.load()
    n, h, w, ct = dout.shape
    cc = ct - 4 * cf
    assert dout.is_contiguous()
    dfine = torch.empty((n, 2 * h, 2 * w, cf), dtype=torch.float32, device=dout.device)
    dcoarse = torch.empty((n, h, w, cc), dtype=torch.float32, device=dout.device)
    check(lib.y2_passthrough_concat_backward(_ptr(dout), _ptr(dfine), _ptr(dcoarse), n, h, w, cf, cc, _stream()))
    return dfine, dcoarse


def add_relu(a, b):
    """relu(a + b): the join of a ResNet bottleneck unit (slim_dir/nets/resnet_v1.py:112)"""
    lib = _lib.load()
    assert a.is_cuda and a.dtype == b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    check(lib.y2_add_relu(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()))
    return out


def add_relu_backward(dout, out, dout2=None):
    """(dout + dout2) * [out > 0]: the gradient of both addends of add_relu (dout2: a second part of the incoming gradient)"""
    lib = _lib.load()
    assert dout.is_cuda and dout.dtype == out.dtype == torch.float32 and dout.is_contiguous() and out.is_contiguous()
    assert dout2 is None or (dout2.is_contiguous() and dout2.dtype == torch.float32 and dout2.shape == dout.shape)
    g = torch.empty_like(dout)
    check(lib.y2_add_relu_backward(_ptr(dout), _ptr(dout2), _ptr(out), _ptr(g), dout.numel(), _stream()))
    return g


def accumulate(dst, src):
    """dst += src (fp32 device tensors of equal size)"""
    lib = _lib.load()
    assert dst.is_cuda and dst.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous()
    assert dst.numel() == src.numel() and src.dtype == torch.float32
    check(lib.y2_accumulate(_ptr(dst), _ptr(src), dst.numel(), _stream()))
    return dst


def class_argmax(scores):
    """scores [..., C] -> (best [...], class index [...] int32); ties: the smallest index"""
    lib = _lib.load()
    assert scores.is_cuda and scores.dtype == torch.float32 and scores.is_contiguous()
    c = scores.shape[-1]
    rows = scores.numel() // c
    best = torch.empty(scores.shape[:-1], dtype=torch.float32, device=scores.device)
    cls = torch.empty(scores.shape[:-1], dtype=torch.int32, device=scores.device)
    check(lib.y2_class_argmax(_ptr(scores), _ptr(best), _ptr(cls), rows, c, _stream()))
    return best, cls


def decode_anchors(net, anchors):
    """net [N,S,S,B,5+C], anchors [B,2] -> boxes [N,S*S*B,4], scores [N,S*S*B,C]"""
    lib = _lib.load()
    n, s, _, b, d = net.shape
    c = d - 5
    assert net.is_cuda and net.dtype == torch.float32 and net.is_contiguous()
    an = torch.as_tensor(np.asarray(anchors, np.float32)).to(net.device).contiguous()
    boxes = torch.empty((n, s * s * b, 4), dtype=torch.float32, device=net.device)
    scores = torch.empty((n, s * s * b, c), dtype=torch.float32, device=net.device)
    check(lib.y2_decode_anchors(_ptr(net), _ptr(an), _ptr(boxes), _ptr(scores), n, s, b, c, _stream()))
    return boxes, scores


def yolov2_loss(net, labels, anchors, image_size, need_grad=True, scales=None):
    """anchor-box loss of the YOLOv2 head (oracle/ext_ref.py yolov2_loss): net [N,S,S,B,5+C], labels [N,S,S,5+C]
    -> (loss[5] = coord, object, noobject, class, total; dnet or None)"""
    lib = _lib.load()
    assert net.is_cuda and net.dtype == torch.float32 and net.is_contiguous() and net.dim() == 5
    n, s, _, b, d = net.shape
    labels = labels.contiguous().float()
    assert tuple(labels.shape) == (n, s, s, d), labels.shape
    an = torch.as_tensor(np.asarray(anchors, np.float32)).to(net.device).contiguous()
    sc = None
    if scales is not None:
        sc = torch.as_tensor(np.asarray([scales[k] for k in ("coord_scale", "object_scale", "noobject_scale",
                                                              "class_scale", "thresh")], np.float32)).to(net.device)
    loss = torch.empty(5, dtype=torch.float32, device=net.device)
    dnet = torch.empty_like(net) if need_grad else None
    ws = torch.empty(lib.y2_yolov2_loss_workspace_bytes(n), dtype=torch.uint8, device=net.device)
    check(lib.y2_yolov2_loss(_ptr(net), _ptr(labels), _ptr(an), n, s, b, d - 5, float(image_size), _ptr(sc), _ptr(loss),
                             _ptr(dnet), _ptr(ws), _stream()))
    return loss, dnet


def nms(boxes, scores, classes=None, iou_thresh=0.5, score_thresh=0.0, max_out=100, class_aware=False):
    """boxes [N,K,4], scores [N,K] (, classes [N,K] int32) -> keep [N,max_out] int32 (-1 padded), count [N]"""
    lib = _lib.load()
    n, k, _ = boxes.shape
    assert boxes.is_contiguous() and scores.is_contiguous() and scores.dtype == torch.float32
    keep = torch.empty((n, max_out), dtype=torch.int32, device=boxes.device)
    count = torch.empty((n,), dtype=torch.int32, device=boxes.device)
    cl = None
    if classes is not None:
        cl = classes.to(torch.int32).contiguous()
    check(lib.y2_nms(_ptr(boxes), _ptr(scores), _ptr(cl), n, k, float(iou_thresh), float(score_thresh), int(max_out),
                     int(bool(class_aware)), _ptr(keep), _ptr(count), _stream()))
    return keep, count


def softmax_cross_entropy(logits, labels, need_grad=True):
    """sparse_softmax_cross_entropy_with_logits + reduce_mean (imagenet_train_darknet.py:51-53)."""
    lib = _lib.load()
    logits = logits.contiguous().float()
    labels = labels.contiguous().to(torch.int32)
    n, c = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if need_grad else None
    check(lib.y2_softmax_cross_entropy(_ptr(logits), _ptr(labels), n, c, _ptr(loss), _ptr(dl), _stream()))
    return loss, dl


def accuracy(logits, labels):
    """reduce_mean(cast(equal(argmax(logits, 1), labels))) (imagenet_train_darknet.py:60-61) -> 0-d tensor"""
    lib = _lib.load()
    logits = logits.contiguous().float()
    labels = labels.contiguous().to(torch.int32)
    n, c = logits.shape
    acc = torch.empty(1, dtype=torch.float32, device=logits.device)
    check(lib.y2_accuracy(_ptr(logits), _ptr(labels), n, c, _ptr(acc), _stream()))
    return acc[0]


def conv2d(x, w, bias=None, dtype="f32"):
    """tf.nn.conv2d(x, W, [1,1,1,1], 'SAME') (+ bias) on fp32 NHWC / HWIO device tensors."""
    lib = _lib.load()
    dt = _lib.DTYPES[dtype]
    n, h, wd, ci = x.shape
    k, _, _, co = w.shape
    x = x.contiguous().float()
    w = w.contiguous().float()
    y = torch.empty((n, h, wd, co), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.y2_conv2d_workspace_bytes(n, h, wd, ci, co, k, dt), dtype=torch.uint8, device=x.device)
    check(lib.y2_conv2d(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, h, wd, ci, co, k, dt, _ptr(ws), _stream()))
    return y


def conv2d_backward(x, w, dy, dtype="f32", dw_out=None):
    """-> (dx, dw); dw_out: a contiguous fp32 tensor of w's size to receive dw (a view of a flat gradient buffer)"""
    lib = _lib.load()
    dt = _lib.DTYPES[dtype]
    n, h, wd, ci = x.shape
    k, _, _, co = w.shape
    x, w, dy = x.contiguous().float(), w.contiguous().float(), dy.contiguous().float()
    dx = torch.empty_like(x)
    if dw_out is not None:
        assert dw_out.is_cuda and dw_out.dtype == torch.float32 and dw_out.is_contiguous() and dw_out.numel() == w.numel()
        dw = dw_out
    else:
        dw = torch.empty_like(w)
    ws = torch.empty(lib.y2_conv2d_workspace_bytes(n, h, wd, ci, co, k, dt), dtype=torch.uint8, device=x.device)
    check(lib.y2_conv2d_backward(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), n, h, wd, ci, co, k, dt,
                                 _ptr(ws), _stream()))
    return dx, dw


# ---- operators of the ResNet-50 backbone swap (csrc/resnet_ops.hip; reference: slim_dir/nets/resnet_v1.py,
# resnet_utils.py, pascal/pascal_train_resnet.py:37-50) on fp32 NHWC device tensors -----------------------------
def _chk(*ts):
    for t in ts:
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())


def batch_norm_forward(x, gamma, beta, moving_mean, moving_var, residual=None, is_training=True, update_moving=False,
                       relu=True, eps=1e-5, decay=0.997):
    """slim.batch_norm (+ residual add, + ReLU) on [..., C] -> (y, save_mean, save_var)"""
    lib = _lib.load()
    _chk(x, gamma, beta, moving_mean, moving_var, residual)
    c = x.shape[-1]
    rows = x.numel() // c
    y = torch.empty_like(x)
    sm = torch.empty(c, dtype=torch.float32, device=x.device)
    sv = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib.y2_batch_norm_forward(_ptr(x), _ptr(residual), _ptr(y), rows, c, _ptr(gamma), _ptr(beta), _ptr(moving_mean),
                                    _ptr(moving_var), _ptr(sm), _ptr(sv), float(eps), float(decay), int(is_training),
                                    int(update_moving), int(relu), _stream()))
    return y, sm, sv


def batch_norm_backward(dy, y, x, gamma, save_mean, save_var, is_training=True, relu=True, want_residual=False, eps=1e-5,
                        dgamma_out=None, dbeta_out=None):
    """-> (dx, dresidual or None, dgamma, dbeta); dgamma_out / dbeta_out: write the parameter gradients there (views
    of a flat gradient buffer)"""
    lib = _lib.load()
    _chk(dy, y, x, gamma, save_mean, save_var, dgamma_out, dbeta_out)
    c = x.shape[-1]
    rows = x.numel() // c
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_residual else None
    dg = dgamma_out if dgamma_out is not None else torch.empty(c, dtype=torch.float32, device=x.device)
    db = dbeta_out if dbeta_out is not None else torch.empty(c, dtype=torch.float32, device=x.device)
    assert dg.numel() == c and db.numel() == c
    check(lib.y2_batch_norm_backward(_ptr(dy), _ptr(y), _ptr(x), _ptr(dx), _ptr(dres), rows, c, _ptr(gamma),
                                     _ptr(save_mean), _ptr(save_var), float(eps), int(is_training), int(relu), _ptr(dg),
                                     _ptr(db), _stream()))
    return dx, dres, dg, db


def subsample(x, factor, out_hw=None):
    """resnet_utils.subsample; with out_hw=(H, W): the gradient (x at the coarse grid -> fine grid [N,H,W,C])"""
    lib = _lib.load()
    _chk(x)
    if out_hw is None:
        n, h, w, c = x.shape
        if factor == 1:
            return x
        y = torch.empty((n, (h + factor - 1) // factor, (w + factor - 1) // factor, c), dtype=torch.float32, device=x.device)
        check(lib.y2_subsample(_ptr(x), _ptr(y), n, h, w, c, int(factor), 1, _stream()))
        return y
    if factor == 1:
        return x
    n, _, _, c = x.shape
    h, w = out_hw
    y = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    check(lib.y2_subsample(_ptr(x), _ptr(y), n, h, w, c, int(factor), 0, _stream()))
    return y


def max_pool_3x3_s2(x):
    lib = _lib.load()
    _chk(x)
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=x.device)
    check(lib.y2_maxpool3x3s2(_ptr(x), _ptr(y), n, h, w, c, _stream()))
    return y


def max_pool_3x3_s2_backward(x, dy):
    lib = _lib.load()
    _chk(x, dy)
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    check(lib.y2_maxpool3x3s2_backward(_ptr(x), _ptr(dy), _ptr(dx), n, h, w, c, _stream()))
    return dx


def conv7x7_s2(x, w, dtype="f32"):
    """the root block's conv2d_same(net, 64, 7, stride=2) (resnet_v1.py:197); dtype f16 / bf16: on the matrix pipe"""
    lib = _lib.load()
    _chk(x, w)
    n, h, wd, _ = x.shape
    co = w.shape[3]
    y = torch.empty((n, (h + 1) // 2, (wd + 1) // 2, co), dtype=torch.float32, device=x.device)
    check(lib.y2_conv7x7s2_t(_ptr(x), _ptr(w), _ptr(y), n, h, wd, co, _lib.DTYPES[dtype], _stream()))
    return y


def conv7x7_s2_backward_filter(x, dy, dtype="f32"):
    lib = _lib.load()
    _chk(x, dy)
    n, h, wd, _ = x.shape
    co = dy.shape[3]
    dw = torch.empty((7, 7, 3, co), dtype=torch.float32, device=x.device)
    check(lib.y2_conv7x7s2_backward_filter_t(_ptr(x), _ptr(dy), _ptr(dw), n, h, wd, co, _lib.DTYPES[dtype], _stream()))
    return dw


def fully_connected(x, w, bias=None, relu=True, dtype="f32"):
    """slim.fully_connected on [rows, in] (rows = the batch, at most 128): act(x w + bias) -> [rows, out] fp32
    (pascal_train_resnet.py:41-46); one pass over the fp32 weights (csrc/fc.hip)"""
    lib = _lib.load()
    _chk(x, w, bias)
    m, k = x.shape
    assert w.shape[0] == k
    n = w.shape[1]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib.y2_fully_connected(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), m, k, n, int(relu), _lib.DTYPES[dtype], _stream()))
    return y


def fully_connected_backward(x, w, dz, dtype="f32", want_dx=True, dw_out=None, want_dw=True):
    """dz = gradient at the pre-activation [rows, out] -> (dx [rows, in] or None, dw [in, out] or None)"""
    lib = _lib.load()
    _chk(x, w, dz)
    m, k = x.shape
    n = w.shape[1]
    assert tuple(dz.shape) == (m, n)
    dx = torch.empty_like(x) if want_dx else None
    if not want_dw:
        check(lib.y2_fully_connected_backward(_ptr(x), _ptr(w), _ptr(dz), _ptr(dx), None, m, k, n, _lib.DTYPES[dtype], _stream()))
        return dx, None
    if dw_out is not None:
        assert dw_out.is_cuda and dw_out.dtype == torch.float32 and dw_out.is_contiguous() and dw_out.numel() == w.numel()
        dw = dw_out
    else:
        dw = torch.empty_like(w)
    check(lib.y2_fully_connected_backward(_ptr(x), _ptr(w), _ptr(dz), _ptr(dx), _ptr(dw), m, k, n, _lib.DTYPES[dtype],
                                          _stream()))
    return dx, dw


def bias_relu_(y, bias, relu=True):
    lib = _lib.load()
    _chk(y, bias)
    c = y.shape[-1]
    check(lib.y2_bias_relu(_ptr(y), _ptr(bias), y.numel() // c, c, int(relu), _stream()))
    return y


def bias_relu_backward(dy, y, relu=True, dbias_out=None):
    lib = _lib.load()
    _chk(dy, y, dbias_out)
    c = y.shape[-1]
    dz = torch.empty_like(dy)
    db = dbias_out if dbias_out is not None else torch.empty(c, dtype=torch.float32, device=dy.device)
    assert db.numel() == c
    check(lib.y2_bias_relu_backward(_ptr(dy), _ptr(y), _ptr(dz), _ptr(db), y.numel() // c, c, int(relu), _stream()))
    return dz, db


def dropout(x, keep_prob, seed):
    """tf.nn.dropout: kept elements scaled by 1/keep_prob; the same (seed) regenerates the mask for the gradient"""
    lib = _lib.load()
    _chk(x)
    y = torch.empty_like(x)
    if isinstance(seed, torch.Tensor):          # int64 [1] on the device: read by the kernel (HIP-graph replay)
        assert seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1
        check(lib.y2_dropout_dev(_ptr(x), _ptr(y), x.numel(), float(keep_prob), _ptr(seed), _stream()))
        return y
    check(lib.y2_dropout(_ptr(x), _ptr(y), x.numel(), float(keep_prob), int(seed), _stream()))
    return y
