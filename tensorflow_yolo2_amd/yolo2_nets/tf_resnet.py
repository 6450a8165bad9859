"""Counterpart of src/yolo2_nets/tf_resnet.py + the graph of src/pascal/pascal_train_resnet.py:37-50: slim's
resnet_v1_50 (global_pool=False) in front of the YOLO grid head
    flatten -> fully_connected 4096 (ReLU) -> dropout 0.5 -> fully_connected S*S*(5B+C) (ReLU) -> [-1,S,S,5B+C]
trained with get_loss and AdamOptimizer(0.0005) -- BASELINE.json configs[4]'s "ResNet-50 slim backbone swap".

The backbone is a DAG (shortcuts), so it is not an engine.Network stack: it is composed here from graph-level
operators, every one a HIP kernel behind the C ABI -- the bottleneck units' 1x1 / 3x3 convolutions are the MFMA
implicit-GEMM kernels (y2_conv2d / y2_conv2d_backward; a fully connected layer is a 1x1 convolution on a 1x1 map),
csrc/resnet_ops.hip has slim.batch_norm (+ residual add + ReLU), subsample, the 3x3/2 max pool, the 7x7/2 root
convolution, bias + ReLU and dropout.  Variable names are slim's (checkpoints of the reference's resnet graph map
one to one).  The reference trains this model at batch 4 and 224x224 (the FC head fixes the input size): the
operator-level composition is written for parity, the Darknet path is the tuned one.
"""
import os

import numpy as np
import torch

from .. import engine as E

BN_EPS, BN_DECAY = 1e-5, 0.997          # resnet_utils.resnet_arg_scope (slim_dir/nets/resnet_utils.py:230-233)

# tf_resnet.py:19-28: (depth, depth_bottleneck, stride) per unit
BLOCKS_50 = [("block1", [(256, 64, 1)] * 2 + [(256, 64, 2)]),
             ("block2", [(512, 128, 1)] * 3 + [(512, 128, 2)]),
             ("block3", [(1024, 256, 1)] * 5 + [(1024, 256, 2)]),
             ("block4", [(2048, 512, 1)] * 3)]


HIDDEN = "/_unused_bias"      # flat-layout slot behind a bottleneck filter (fused stacks: see ResNet50Yolo), never a variable


def variable_list(blocks=BLOCKS_50, root_depth=64, fc_hidden=4096, fc_out=1470, feat_hw=7, hidden_bias=False):
    """(name, shape, trainable) in slim's creation order.  hidden_bias: every bottleneck convolution is followed by a
    slot of `cout` floats named <conv>/_unused_bias -- the conv-bias position of the library's [W, b, gamma, beta] layer
    layout, kept at zero (slim's conv2d has no bias under batch_norm); such entries are layout only, not variables"""
    def bn(scope, c):
        s = scope.rstrip("/") + "/BatchNorm/"
        return [(s + "gamma", (c,), True), (s + "beta", (c,), True), (s + "moving_mean", (c,), False),
                (s + "moving_variance", (c,), False)]
    out = [("conv1/weights", (7, 7, 3, root_depth), True)] + bn("conv1", root_depth)
    cin = root_depth
    for bname, units in blocks:
        for i, (depth, db, _stride) in enumerate(units):
            p = "%s/unit_%d/bottleneck_v1/" % (bname, i + 1)
            def conv(scope, k, ci, co):
                hb = [(p + scope + HIDDEN, (co,), True)] if hidden_bias else []
                return [(p + scope + "/weights", (k, k, ci, co), True)] + hb + bn(p + scope, co)
            if depth != cin:
                out += conv("shortcut", 1, cin, depth)
            out += conv("conv1", 1, cin, db) + conv("conv2", 3, db, db) + conv("conv3", 1, db, depth)
            cin = depth
    flat = feat_hw * feat_hw * cin
    out += [("yolo_fc1/weights", (flat, fc_hidden), True), ("yolo_fc1/biases", (fc_hidden,), True),
            ("yolo_fc2/weights", (fc_hidden, fc_out), True), ("yolo_fc2/biases", (fc_out,), True)]
    return out


_TWO_CALL_JOIN = bool(os.environ.get("Y2_RESNET_TWO_CALL_JOIN"))
_NO_STRIDED_STACKS = bool(os.environ.get("Y2_RESNET_NO_STRIDED_STACKS"))   # A/B: the stride-2 units on fp32 operators
_NO_PACK_GROUP = bool(os.environ.get("Y2_RESNET_NO_PACK_GROUP"))           # A/B: every stack re-packs its filters itself
_NO_STRIDED_LINK = bool(os.environ.get("Y2_RESNET_NO_STRIDED_LINK"))       # A/B: ... on the executor, but outside the linked runs
_NO_LINK = bool(os.environ.get("Y2_RESNET_NO_LINK"))       # A/B: round 4's fp32 hand-over between the fused units


class ResNet50Yolo:
    """resnet_v1_50 + the YOLO fully connected head, forward / backward / Adam(0.0005)"""

    def __init__(self, batch, image_size=224, B=2, num_class=20, dtype="f32", blocks=None, root_depth=64,
                 fc_hidden=4096, seed=0, device="cuda:0", learning_rate=0.0005, keep_prob=0.5, loss_scale=None,
                 graph=False, graph_check_every=16, fused=None, link=None, fuse_fc1=None):
        """dtype: arithmetic of the convolution / FC contractions.  "f32" (default: the reference's precision).  With
        "f16" the gradient of the loss is multiplied by a dynamic loss scale before the backward pass (activation
        gradients 50 layers deep at batch 4 fall below f16's normal range otherwise), the scale is divided out inside
        the optimizer, and the update is the overflow-guarded Adam (an inf / NaN anywhere skips the step on the
        device and halves the scale) -- the policy of the Darknet path (engine.LossScaler)."""
        assert image_size % 32 == 0
        if dtype not in ("f32", "f16", "bf16"):
            # (the FC head and the root convolution have no split-operand form; "f16x2" used to die at the head
            #  with 'bad dtype')
            raise ValueError("ResNet50Yolo: dtype must be 'f32', 'f16' or 'bf16' (got %r); the split-operand modes "
                             "'f16x2' / 'f16x2f' exist for the Darknet-19 stacks only" % (dtype,))
        self.batch, self.size, self.S, self.B, self.num_class = batch, image_size, image_size // 32, B, num_class
        self.dtype, self.device, self.keep_prob = dtype, torch.device(device), keep_prob
        self.blocks = blocks or BLOCKS_50
        self.out_c = 5 * B + num_class
        # fused (round 4; default in the half-precision modes when every channel count fits the stack executor): the
        # stride-1 bottleneck units run as native conv-BN-activation STACKS of the library (engine.Network on views of
        # the flat buffers: conv1 1x1 ReLU, conv2 3x3 ReLU, conv3 1x1 without activation; the projection shortcut a
        # one-layer stack) -- half-precision bordered activations between the three convolutions, batch statistics
        # from the conv epilogues, BN-backward sums in the dgrad epilogues, split-K slab weight gradients -- joined by
        # add + ReLU.  The three stride-2 units, the root and the FC head stay on the graph-level operators.
        if fused is None:
            fused = dtype != "f32" and self._stacks_fit(self.blocks, root_depth)
        self.fused = bool(fused)
        # link (round 5; default on, Y2_RESNET_NO_LINK=1 / link=False: round 4's fp32 hand-over): runs of fused units
        # exchange activations and gradients in the arithmetic type (_linked_units)
        self.link = (not _NO_LINK) if link is None else bool(link)
        self.vars = variable_list(self.blocks, root_depth, fc_hidden, self.S * self.S * self.out_c, self.S)
        self.layout = variable_list(self.blocks, root_depth, fc_hidden, self.S * self.S * self.out_c, self.S,
                                    hidden_bias=self.fused)
        # fuse_fc1 (round 5; the guarded half-precision modes): yolo_fc1/weights (1.64 GB at full width) sits LAST in the
        # flat buffers and its gradient is never stored -- the train step updates it with y2_fc_adam_apply_guarded from
        # the layer's input and dz after the overflow scan + guarded Adam of everything in front of it
        self.fuse_fc1 = (dtype != "f32" and not os.environ.get("Y2_RESNET_NO_FC1_FUSE")) if fuse_fc1 is None else \
            (bool(fuse_fc1) and dtype != "f32")
        if self.fuse_fc1:
            rest = [e for e in self.layout if e[0] != "yolo_fc1/weights"]
            # the 1.64 GB matrix starts on a 256-byte boundary of the flat buffer: fc.hip takes 16-byte loads only from
            # 16-byte-aligned rows (round 5: behind yolo_fc2/biases' 1470 floats it sat at offset 2 mod 4 and the forward
            # and dx products ran their scalar-load forms: dx 666 us instead of 440).  A layout-only slot of zeros: its
            # gradient stays zero, Adam leaves it at zero.
            pad = (-sum(int(np.prod(sh)) for (_n, sh, t) in rest if t)) % 64
            if pad:
                rest.append(("yolo_fc1/_align" + HIDDEN, (pad,), True))
            self.layout = rest + [e for e in self.layout if e[0] == "yolo_fc1/weights"]
        n_train = sum(int(np.prod(s)) for (_n, s, t) in self.layout if t)
        n_state = sum(int(np.prod(s)) for (_n, s, t) in self.layout if not t)
        self.params = torch.zeros(n_train, dtype=torch.float32, device=self.device)      # ONE flat buffer: one Adam
        self.grads = torch.zeros(n_train, dtype=torch.float32, device=self.device)
        self.state = torch.zeros(n_state, dtype=torch.float32, device=self.device)
        self.p, self.g = {}, {}
        self.offset, self.state_offset = {}, {}          # name -> (start, count) in params / grads / m / v, in state
        ot = os_ = 0
        for (name, shape, trainable) in self.layout:
            n = int(np.prod(shape))
            if trainable:
                self.offset[name] = (ot, n)
                if not name.endswith(HIDDEN):
                    self.p[name] = self.params[ot:ot + n].view(shape)
                    self.g[name] = self.grads[ot:ot + n].view(shape)
                ot += n
            else:
                self.state_offset[name] = (os_, n)
                self.p[name] = self.state[os_:os_ + n].view(shape)
                os_ += n
        self._stacks = {}                                 # unit scope -> (main stack, projection stack or None)
        self._stacks_meta = {}
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.t = 0
        self.lr = learning_rate
        # data parallelism (round 6): replicas share `seed` (identical parameters) and draw different dropout masks
        from ..trainer import _dist
        d = _dist()
        self.drop_seed = seed * 7919 + 1 + (d.get_rank() * 104729 if d is not None else 0)
        self.dp_strategy = os.environ.get("Y2_DP_STRATEGY", "allreduce")
        self.init_params(seed)
        self.tape = None
        self.loss_scale = float(loss_scale) if loss_scale is not None else (1024.0 if dtype == "f16" else 1.0)
        self.guard = dtype != "f32"
        self.ctrl = torch.zeros(8, dtype=torch.int32, device=self.device)     # found_inf, step, skipped, -, lr_t
        self.overflows, self._clean, self.growth_interval = 0, 0, 1000
        # graph=True: step() replays ONE HIP graph of forward + loss + backward + guarded Adam (the ~1500 launches of
        # the operator-level step are issued from Python in ~17 ms at batch 4; the replay is bound by the GPU).  The
        # step counter (ctrl), the dropout seed and the overflow flag live in device memory; the host reads ctrl every
        # `graph_check_every` steps to follow the step count and adapt the loss scale (a changed scale re-captures).
        self.graph = bool(graph)
        self._graph, self._gstream, self._gin, self._gout = None, None, None, None
        self._eager_on_gstream, self._since_check, self._skipped_seen = 0, 0, 0
        self.graph_check_every = int(graph_check_every)
        self._seed_dev = torch.tensor([self.drop_seed], dtype=torch.int64, device=self.device) if self.graph else None

    # ---- fused bottleneck units -----------------------------------------------
    @staticmethod
    def _stacks_fit(blocks, root_depth):
        """channel rules of the stack executor (csrc/net.hip y2_ctx_create): inputs in multiples of 32, above 128 in
        multiples of 128"""
        ok = lambda c: c % 32 == 0 and (c <= 128 or c % 128 == 0)
        cin = root_depth
        for _b, units in blocks:
            for (depth, db, _st) in units:
                if not (ok(cin) and ok(db) and ok(depth) and depth <= 2048):
                    return False
                cin = depth
        return True

    def _stack_for(self, scope, hw, cin, depth, db, stride=1):
        """(main stack, projection stack) of a unit at hw x hw, created on first use.  stride 2 (round 5; the last unit of
        blocks 1-3, identity shortcut): conv2 is a SUBSAMPLING layer of the executor (pool = 2: the stride-1 3x3 output at
        even rows / columns, batch norm over the kept positions = slim's conv2d_same(stride=2), resnet_utils.py:77-122)"""
        key = (scope, hw)
        if key not in self._stacks:
            def make(first, last, spec, slopes):
                p0 = self.offset[scope + first + "/weights"][0]
                pe = sum(self.offset[scope + last + "/BatchNorm/beta"])
                s0 = self.state_offset[scope + first + "/BatchNorm/moving_mean"][0]
                se = sum(self.state_offset[scope + last + "/BatchNorm/moving_variance"])
                net = E.Network(spec, self.batch, hw, hw, dtype=self.dtype, core_layers=len(spec), training=True,
                                device=str(self.device), grad_scale=1.0,
                                buffers=(self.params[p0:pe], self.grads[p0:pe], self.state[s0:se]))
                net.set_layer_options(slopes, BN_EPS, BN_DECAY, zero_bias_grad=True)
                return net
            main = make("conv1", "conv3", [(1, cin, db, 0), (3, db, db, 2 if stride == 2 else 0), (1, db, depth, 0)],
                        [0.0, 0.0, 1.0])
            assert stride == 1 or depth == cin, "a strided unit with a projection shortcut is not wired"
            proj = make("shortcut", "shortcut", [(1, cin, depth, 0)], [1.0]) if depth != cin else None
            self._stacks[key] = (main, proj)
        return self._stacks[key]

    def _linked_units(self, hw0, cin0):
        """Round 5: runs of consecutive stride-1 units exchange their activations and gradients in the arithmetic type
        (engine.Network.link; include/yolo2_hip.h y2_link) -- bordered half-precision tensors forward, [M][C] tensors
        backward -- instead of fp32 NHWC tensors with a cast / pack / convert pass on both sides of every join.  A run
        is entered from and left to the fp32 graph-level operators (root, stride-2 units, head); its first unit must have a
        projection shortcut (its input gradient leaves as two fp32 addends), which every first unit of a block has.
        Returns {unit scope: descriptor}; built once per input size."""
        key = ("units", hw0)
        if key in self._stacks_meta:
            return self._stacks_meta[key]
        units, run = {}, []
        hw, cin = hw0, cin0

        def close(run):
            for i, u in enumerate(run):
                u["bottom"], u["top"] = i == 0, i == len(run) - 1
                u["below"], u["above"] = (run[i - 1] if i > 0 else None), (run[i + 1] if i + 1 < len(run) else None)
        for bname, blk in self.blocks:
            for i, (depth, db, stride) in enumerate(blk):
                sc = "%s/unit_%d/bottleneck_v1/" % (bname, i + 1)
                # (round 5: a stride-2 unit with an identity shortcut continues the run it sits on -- its shortcut is the typed
                #  subsample of the run's activation, its shortcut gradient reaches the unit below through y2_join_backward_s2)
                strided_ok = (stride == 2 and depth == cin and hw % 2 == 0 and bool(run) and not _NO_STRIDED_STACKS
                              and not _NO_STRIDED_LINK)
                if self.fused and (stride == 1 or strided_ok):
                    u = {"scope": sc, "hw": hw, "cin": cin, "depth": depth, "db": db, "proj": depth != cin, "stride": stride,
                         "hwo": (hw + stride - 1) // stride}
                    if not run and not u["proj"]:
                        close([u])                      # an identity unit cannot open a run: it stays on the fp32 interface
                    else:
                        run.append(u)
                    units[sc] = u
                else:
                    close(run)
                    run = []
                hw, cin = (hw + stride - 1) // stride, depth
        close(run)
        if not self.link or self.dtype not in ("f16", "bf16"):       # (the typed hand-over exists for the 16-bit types)
            for u in units.values():
                u["bottom"] = u["top"] = True
                u["below"] = u["above"] = None
        dev, dt, n = str(self.device), self.dtype, self.batch
        tdt = torch.float16 if dt == "f16" else torch.bfloat16
        for u in list(units.values()):
            if u["stride"] == 2 and u["bottom"] and u["top"]:
                del units[u["scope"]]          # an unlinked strided unit: the fp32-interface path of forward() ("fused_s")
                continue
            main, proj = self._stack_for(u["scope"], u["hw"], u["cin"], u["depth"], u["db"], stride=u["stride"])
            u["main"], u["pstack"] = main, proj
            m = n * u["hw"] * u["hw"]
            mo = n * u["hwo"] * u["hwo"]
            if not u["top"]:      # my output is the next unit's bordered input
                u["out_b"] = E.Bordered(n, u["hwo"], u["hwo"], u["depth"], dt, dev)
                u["g"] = torch.zeros((mo, u["depth"]), dtype=tdt, device=dev)       # my join's gradient (T), made by join_backward
            if u["proj"] and not (u["bottom"] and u["top"]):
                u["short_b"] = E.Bordered(n, u["hw"], u["hw"], u["depth"], dt, dev)  # the projection's output, joined in T
            if u["stride"] == 2:
                u["short_b"] = E.Bordered(n, u["hwo"], u["hwo"], u["depth"], dt, dev)   # subsample(x), joined in T
            if not u["bottom"]:
                u["dxm"] = torch.zeros((m, u["cin"]), dtype=tdt, device=dev)
                if u["proj"]:
                    u["dxp"] = torch.zeros((m, u["cin"]), dtype=tdt, device=dev)
        for u in units.values():
            if u["bottom"] and u["top"]:
                continue                                  # unlinked: round 4's calls
            xin = u["below"]["out_b"] if not u["bottom"] else None
            out_b = u.get("out_b")
            g_in = u.get("g")                               # None at the top: fp32 dout through the convert pass
            if u["proj"]:
                u["pstack"].link(x=xin, out=u["short_b"], dout=g_in, dx=u.get("dxp"))
                u["main"].link(x=xin, out=out_b, join=u["short_b"], dout=g_in, dx=u.get("dxm"))
            elif u["stride"] == 2:
                u["main"].link(x=xin, out=out_b, join=u["short_b"], dout=g_in, dx=u.get("dxm"))
            else:
                u["main"].link(x=xin, out=out_b, join_self=True, dout=g_in, dx=u.get("dxm"))
        self._stacks_meta[key] = units
        return units

    def params_changed(self):
        """the flat parameter buffer moved (optimizer step, load, restore): the stacks re-pack their filters"""
        for main, proj in self._stacks.values():
            main.params_changed()
            if proj is not None:
                proj.params_changed()
        self._packs_dirty = True

    def _pack_group(self):
        """round 5: ONE filter-pack launch for all stacks (y2_pack_group_run) instead of one per stack at its next forward
        (20 launches of ~10 us per step); the table is built once the stacks exist (after the first forward)"""
        if not getattr(self, "pack_group", not _NO_PACK_GROUP) or not self._stacks or not getattr(self, "_packs_dirty", True):
            return
        import ctypes as C
        nets = [n for pair in self._stacks.values() for n in pair if n is not None]
        key = tuple(id(n) for n in nets)
        lib = E._lib.load()
        if getattr(self, "_pack_key", None) != key:
            if torch.cuda.is_current_stream_capturing():
                return                                   # (built outside captures; this step packs per stack)
            arr = (C.c_void_p * len(nets))(*[n.h.value for n in nets])
            tab = torch.zeros(256 * 3 * len(nets), dtype=torch.uint8, device=self.device)
            nl, nb = C.c_int(0), C.c_int(0)
            E.check(lib.y2_pack_group_table(arr, len(nets), E._ptr(tab), tab.numel(), C.byref(nl), C.byref(nb)))
            self._pack_key, self._pack_arr, self._pack_tab, self._pack_counts = key, arr, tab, (nl.value, nb.value)
        E.check(lib.y2_pack_group_run(self._pack_arr, len(nets), E._ptr(self._pack_tab), self._pack_counts[0],
                                      self._pack_counts[1], E._stream()))
        self._packs_dirty = False

    # ---- variables ---------------------------------------------------------
    def init_params(self, seed=0):
        """slim defaults: variance_scaling_initializer (truncated normal, stddev sqrt(1.3 * 2 / fan_in)) for the
        convolutions, xavier_initializer for fully connected weights, zeros for biases, BN 1 / 0 / 0 / 1"""
        rng = np.random.default_rng(seed)
        for (name, shape, _t) in self.vars:
            if name.endswith("weights") and len(shape) == 4:
                std = np.sqrt(1.3 * 2.0 / (shape[0] * shape[1] * shape[2]))
                w = rng.standard_normal(shape)
                bad = np.abs(w) > 2
                while bad.any():
                    w[bad] = rng.standard_normal(int(bad.sum()))
                    bad = np.abs(w) > 2
                val = (w * std).astype(np.float32)
            elif name.endswith("weights"):
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                val = rng.uniform(-lim, lim, shape).astype(np.float32)
            elif name.endswith("gamma") or name.endswith("moving_variance"):
                val = np.ones(shape, np.float32)
            else:
                val = np.zeros(shape, np.float32)
            self.p[name].copy_(torch.as_tensor(val))
        self.params_changed()

    def load_params(self, params):
        for name, val in params.items():
            self.p[name].copy_(torch.as_tensor(np.asarray(val, np.float32)).to(self.device))
        self.params_changed()

    def export_params(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.p.items()}

    def export_grads(self):
        """gradients of the last backward pass.  After a train step of a fuse_fc1 model (backward(skip_fc1_dw=True)) the
        gradient of yolo_fc1/weights was never formed -- the fused update consumed x and dz -- so that key is LEFT OUT rather
        than returned stale; its slot in `grads` stays allocated because backward() without the skip (tests, the
        f32 mode, gathered batches beyond the fused kernel's 256 rows) stores it there."""
        skip = {"yolo_fc1/weights"} if getattr(self, "_fc1_dw_skipped", False) else set()
        return {k: v.detach().cpu().numpy().copy() for k, v in self.g.items() if k not in skip}

    # ---- building blocks ---------------------------------------------------
    def _conv(self, x, name, stride=1):
        """slim.conv2d without bias; stride 2: 1x1 -> subsample then convolve, 3x3 -> conv2d_same = convolve at stride 1
        then subsample (resnet_utils.py:77-122)"""
        w = self.p[name]
        k = w.shape[0]
        if stride == 1:
            return E.conv2d(x, w, None, self.dtype), ("conv", name, x, 1)
        if k == 1:
            xs = E.subsample(x, stride)
            return E.conv2d(xs, w, None, self.dtype), ("conv1s", name, x, xs, stride)
        full = E.conv2d(x, w, None, self.dtype)
        return E.subsample(full, stride), ("conv3s", name, x, tuple(full.shape[1:3]), stride)

    def _conv_backward(self, rec, dy):
        kind, name = rec[0], rec[1]
        w = self.p[name]
        gw = self.g[name]            # every variable has exactly one consumer: dW lands in the flat gradient buffer
        if kind == "conv":
            dx, _ = E.conv2d_backward(rec[2], w, dy.contiguous(), self.dtype, dw_out=gw)
        elif kind == "conv1s":
            x, xs, stride = rec[2], rec[3], rec[4]
            dxs, _ = E.conv2d_backward(xs, w, dy.contiguous(), self.dtype, dw_out=gw)
            dx = E.subsample(dxs, stride, out_hw=tuple(x.shape[1:3]))
        else:
            x, hw, stride = rec[2], rec[3], rec[4]
            dfull = E.subsample(dy.contiguous(), stride, out_hw=hw)
            dx, _ = E.conv2d_backward(x, w, dfull, self.dtype, dw_out=gw)
        return dx

    def _bn(self, x, scope, relu, residual=None):
        s = scope.rstrip("/") + "/BatchNorm/"
        y, sm, sv = E.batch_norm_forward(x, self.p[s + "gamma"], self.p[s + "beta"], self.p[s + "moving_mean"],
                                         self.p[s + "moving_variance"], residual, self._training, self._update_moving,
                                         relu, BN_EPS, BN_DECAY)
        return y, ("bn", s, x, y, sm, sv, relu, residual is not None)

    def _bn_backward(self, rec, dy):
        _k, s, x, y, sm, sv, relu, has_res = rec
        dx, dres, _dg, _db = E.batch_norm_backward(dy.contiguous(), y, x, self.p[s + "gamma"], sm, sv, self._training, relu,
                                                   has_res, BN_EPS, dgamma_out=self.g[s + "gamma"],
                                                   dbeta_out=self.g[s + "beta"])
        return dx, dres

    # ---- graph ---------------------------------------------------------------
    def forward(self, images, is_training=True, update_moving=False, dropout=True):
        """images [N,size,size,3] fp32 -> grid_net [N,S,S,5B+C] (pascal_train_resnet.py:37-50)"""
        assert tuple(images.shape) == (self.batch, self.size, self.size, 3) and images.is_cuda
        self._training, self._update_moving = bool(is_training), bool(update_moving)
        self._pack_group()
        tape = []
        x = E.conv7x7_s2(images.contiguous(), self.p["conv1/weights"], self.dtype)         # resnet_v1.py:197
        tape.append(("conv7", images))
        x, r = self._bn(x, "conv1", True); tape.append(r)
        pooled = E.max_pool_3x3_s2(x); tape.append(("pool", x)); x = pooled                # resnet_v1.py:198
        units = self._linked_units(int(x.shape[1]), int(x.shape[3])) if self.fused else {}
        for bname, blk in self.blocks:
            for i, (depth, db, stride) in enumerate(blk):                                    # resnet_v1.py:99-112
                sc = "%s/unit_%d/bottleneck_v1/" % (bname, i + 1)
                if self.fused and sc in units and not (units[sc]["bottom"] and units[sc]["top"]):
                    # linked run: typed tensors between the units, fp32 only where the run meets the graph-level operators
                    u = units[sc]
                    xin = x.contiguous() if u["bottom"] else None
                    out = None
                    if u["top"]:
                        out = torch.empty((self.batch, u["hwo"], u["hwo"], u["depth"]), dtype=torch.float32, device=self.device)
                    if u["stride"] == 2:       # the identity shortcut subsample(x), in the arithmetic type
                        E.subsample_bordered(self.dtype, u["below"]["out_b"], u["short_b"])
                    if u["proj"]:
                        u["pstack"].forward_linked(is_training, update_moving, images=xin)
                    u["main"].forward_linked(is_training, update_moving, out=out, images=xin)
                    tape.append(("linked", u, out))
                    x = out
                    continue
                if self.fused and stride == 2 and x is not None and depth == x.shape[3] and x.shape[1] % 2 == 0 and not _NO_STRIDED_STACKS:
                    # round 5: the strided unit on the stack executor too (conv2 a subsampling layer); the identity
                    # shortcut subsample(x) (resnet_v1.py:99-101) joins in the main stack's last apply pass
                    main, _ = self._stack_for(sc, int(x.shape[1]), int(x.shape[3]), depth, db, stride=2)
                    xin = x.contiguous()
                    short = E.subsample(xin, stride)
                    out = main.forward(xin, is_training, is_training, update_moving=update_moving, join=short)
                    tape.append(("fused_s", main, out, stride, tuple(xin.shape[1:3])))
                    x = out
                    continue
                if self.fused and stride == 1:
                    main, proj = self._stack_for(sc, int(x.shape[1]), int(x.shape[3]), depth, db)
                    xin = x.contiguous()
                    short = proj.forward(xin, is_training, is_training, update_moving=update_moving) if proj is not None else xin
                    # relu(shortcut + residual) in the last layer's apply pass of the main stack (its own output buffer)
                    if _TWO_CALL_JOIN:      # A/B: the branch output stored, then y2_add_relu
                        out = E.add_relu(main.forward(xin, is_training, is_training, update_moving=update_moving), short)
                    else:
                        out = main.forward(xin, is_training, is_training, update_moving=update_moving, join=short)
                    tape.append(("fused", main, proj, out))
                    x = out
                    continue
                unit = {"in": x, "stride": stride}
                if depth == x.shape[3]:
                    shortcut = E.subsample(x, stride)
                    unit["short"] = None
                else:
                    sconv, r1 = self._conv(x, sc + "shortcut/weights", stride)
                    shortcut, r2 = self._bn(sconv, sc + "shortcut", False)
                    unit["short"] = (r1, r2)
                a, c1 = self._conv(x, sc + "conv1/weights"); a, b1 = self._bn(a, sc + "conv1", True)
                a, c2 = self._conv(a, sc + "conv2/weights", stride); a, b2 = self._bn(a, sc + "conv2", True)
                a, c3 = self._conv(a, sc + "conv3/weights")
                out, b3 = self._bn(a, sc + "conv3", True, residual=shortcut.contiguous())   # relu(shortcut + BN(conv3))
                unit["res"] = (c1, b1, c2, b2, c3, b3)
                tape.append(("unit", unit))
                x = out
        feat = x                                                                              # [N,S,S,depth]
        n = self.batch
        flat = feat.reshape(n, -1).contiguous()                                               # slim.flatten (NHWC order)
        fc1 = E.fully_connected(flat, self.p["yolo_fc1/weights"], self.p["yolo_fc1/biases"], True, self.dtype)
        use_drop = bool(dropout and is_training)
        if self._seed_dev is not None:
            self._seed_dev.add_(1)                  # captured with the step: every replay draws a new mask
            seed_now = self._seed_dev
        else:
            self.drop_seed += 1
            seed_now = self.drop_seed
        h = E.dropout(fc1, self.keep_prob, seed_now) if use_drop else fc1
        fc2 = E.fully_connected(h, self.p["yolo_fc2/weights"], self.p["yolo_fc2/biases"], True, self.dtype)
        tape.append(("head", feat, flat, fc1, h, fc2, use_drop, seed_now))
        self.tape = tape
        return fc2.view(n, self.S, self.S, self.out_c)

    def backward(self, dgrid, skip_fc1_dw=False):
        """gradients of every trainable variable into self.grads (skip_fc1_dw: all but yolo_fc1/weights, whose update is
        fused with its gradient in the train step: _fc1_operands keeps the layer's input and dz)"""
        assert self.tape is not None
        n = self.batch
        _k, feat, flat, fc1, h, fc2, use_drop, seed = self.tape[-1]
        dz2, _ = E.bias_relu_backward(dgrid.reshape(n, -1).contiguous(), fc2, True, dbias_out=self.g["yolo_fc2/biases"])
        dh, _ = E.fully_connected_backward(h, self.p["yolo_fc2/weights"], dz2, self.dtype,
                                           dw_out=self.g["yolo_fc2/weights"])
        dfc1 = E.dropout(dh, self.keep_prob, seed) if use_drop else dh                      # same mask, same 1/keep scale
        dz1, _ = E.bias_relu_backward(dfc1, fc1, True, dbias_out=self.g["yolo_fc1/biases"])
        self._fc1_dw_skipped = bool(skip_fc1_dw)
        if skip_fc1_dw:
            dflat, _ = E.fully_connected_backward(flat, self.p["yolo_fc1/weights"], dz1, self.dtype, want_dw=False)
            self._fc1_operands = (flat, dz1)
        else:
            dflat, _ = E.fully_connected_backward(flat, self.p["yolo_fc1/weights"], dz1, self.dtype,
                                                  dw_out=self.g["yolo_fc1/weights"])
        dx = dflat.reshape(feat.shape).contiguous()
        dx2 = None          # a fused unit leaves its input gradient as two addends (main branch, shortcut): the unit below
        for rec in reversed(self.tape[:-1]):    # folds their sum into its own join's backward, anything else adds them first
            if rec[0] not in ("fused", "fused_s", "linked") and dx2 is not None:
                dx = E.accumulate(dx, dx2)
                dx2 = None
            if rec[0] == "linked":
                _k, u, out = rec
                main, proj = u["main"], u["pstack"]
                if u["top"]:
                    # the run's top: the incoming gradient is fp32; g = d(relu(r + s)) to both branches, fp32
                    g32 = E.add_relu_backward(dx.contiguous(), out, dx2)
                    dx2 = None
                    main.backward_linked(dout=g32)                        # input gradient -> u["dxm"] (T)
                    if proj is not None:
                        proj.backward_linked(dout=g32)                    # -> u["dxp"] (T)
                    u["_short_grad"] = g32 if proj is None else u["dxp"]
                else:
                    a = u["above"]
                    # g = (main branch of the unit above + its shortcut branch) * [my output > 0], all in T
                    E.join_backward(self.dtype, u["out_b"], a["dxm"], a["_short_grad"], u["g"], stride=a["stride"])
                    if u["bottom"]:
                        # the run's bottom: the input gradient leaves as two fp32 addends (main branch, projection)
                        dx = torch.empty((self.batch, u["hw"], u["hw"], u["cin"]), dtype=torch.float32, device=self.device)
                        dx2 = torch.empty_like(dx)
                        main.backward_linked(dinput=dx)
                        proj.backward_linked(dinput=dx2)
                    else:
                        main.backward_linked()
                        if proj is not None:
                            proj.backward_linked()
                        u["_short_grad"] = u["g"] if proj is None else u["dxp"]
                if u["top"] and u["bottom"]:
                    raise AssertionError("unlinked units take the fused path")
            elif rec[0] == "fused_s":
                _k, main, out, stride, in_hw = rec
                g = E.add_relu_backward(dx.contiguous(), out, dx2)     # at the unit's output resolution
                dx = main.backward_input(g)                            # main branch, at the input resolution
                dx2 = E.subsample(g, stride, out_hw=in_hw)             # identity shortcut: g back at the even positions
            elif rec[0] == "fused":
                _k, main, proj, out = rec
                g = E.add_relu_backward(dx.contiguous(), out, dx2)     # d(relu(r + s)) = dout * [out > 0], to both branches
                dx = main.backward_input(g)
                dx2 = proj.backward_input(g) if proj is not None else g
            elif rec[0] == "unit":
                u = rec[1]
                c1, b1, c2, b2, c3, b3 = u["res"]
                da, dshort = self._bn_backward(b3, dx)
                da = self._conv_backward(c3, da)
                da, _ = self._bn_backward(b2, da)
                da = self._conv_backward(c2, da)
                da, _ = self._bn_backward(b1, da)
                dxin = self._conv_backward(c1, da)
                if u["short"] is None:
                    dsx = E.subsample(dshort, u["stride"], out_hw=tuple(u["in"].shape[1:3]))
                else:
                    r1, r2 = u["short"]
                    ds, _ = self._bn_backward(r2, dshort)
                    dsx = self._conv_backward(r1, ds)
                dx = E.accumulate(dxin.contiguous(), dsx.contiguous())
            elif rec[0] == "pool":
                dx = E.max_pool_3x3_s2_backward(rec[1], dx.contiguous())
            elif rec[0] == "bn":
                dx, _ = self._bn_backward(rec, dx)
            elif rec[0] == "conv7":
                dw = E.conv7x7_s2_backward_filter(rec[1].contiguous(), dx.contiguous(), self.dtype)
                self.g["conv1/weights"].copy_(dw)

    def flops_per_step(self):
        """algorithmic FLOPs of one train step: 3 x the forward multiply-adds of every convolution / FC layer at the
        size it is DEFINED at (a stride-2 3x3 counts its strided output; this composition computes it at stride 1 and
        subsamples -- 4x the listed work for those three layers)"""
        tot, h, cin = 0.0, self.size // 2, 3
        n = self.batch
        tot += 2.0 * n * h * h * 49 * 3 * self.vars[0][1][3]
        cin = self.vars[0][1][3]
        h //= 2
        for _b, units in self.blocks:
            for (depth, db, stride) in units:
                ho = (h + stride - 1) // stride
                if depth != cin:
                    tot += 2.0 * n * ho * ho * cin * depth
                tot += 2.0 * n * h * h * cin * db + 2.0 * n * ho * ho * 9 * db * db + 2.0 * n * ho * ho * db * depth
                cin, h = depth, ho
        flat = h * h * cin
        fc_hidden = self.p["yolo_fc1/biases"].numel()
        tot += 2.0 * n * flat * fc_hidden + 2.0 * n * fc_hidden * self.p["yolo_fc2/biases"].numel()
        return 3.0 * tot

    def _step_device(self, images, labels):
        """the device work of one iteration with every piece of state in device memory (what the graph captures):
        forward, loss, scaled backward, full overflow scan, guarded Adam"""
        grid = self.forward(images, True, update_moving=True)
        loss, ious, mask, dnet = E.yolo_loss(grid, labels, self.num_class, self.batch, self.size, self.S, self.B)
        lib = E._lib.load()
        if self.loss_scale != 1.0:
            E.check(lib.y2_scale(E._ptr(dnet), dnet.numel(), self.loss_scale, E._stream()))
        self._guarded_update(lib, dnet)
        self.params_changed()       # the fused stacks re-pack their filters at the top of the next forward (captured with it)
        self.tape = None
        return loss, ious, mask

    # ---- data parallelism (round 6; SURVEY section 8e, slim's clone semantics: model_deploy.py:222-225,436-446) ----------
    # One process per GPU, identically seeded replicas, batch-norm statistics per replica.  Everything in FRONT of
    # yolo_fc1/weights in the flat gradient buffer (25.6 M floats at full width) is SUM all-reduced as one range and divided
    # by the world size inside the optimizer kernel.  yolo_fc1/weights itself (1.64 GB) is designed for xGMI's point-to-point
    # links instead: its gradient is never materialised -- the replicas ALL-GATHER the two operands of dW = x^T dz, the layer's
    # input x [batch, 100352] (12.8 MB per rank at batch 32) and dz [batch, 4096] (0.5 MB), and every rank runs the fused
    # product + guarded Adam update (y2_fc_adam_apply_guarded) over the batch * world gathered rows with grad_mult / world:
    # 107 MB on the wire at 8 GPUs instead of a 1.64 GB all-reduce, the replicas stay bit-identical (same rows, same order,
    # same kernel) and the update keeps its fused form.  The overflow guard sees the reduced gradients (an inf on one rank
    # reaches every rank through the sum) and the gathered dz (identical on every rank): all replicas skip or none.
    def _dp(self):
        from ..trainer import _dist
        d = _dist()
        return (d, d.get_world_size()) if d is not None else (None, 1)

    def _fc1_fused_now(self):
        """the fused yolo_fc1 update applies when the gathered batch fits the kernel's row fragments (256 rows in the 16-bit
        types); beyond that the gradient is stored and all-reduced with the rest"""
        _d, world = self._dp()
        return self.fuse_fc1 and self.batch * world <= 256

    def _guarded_update(self, lib, dnet):
        """overflow scan + guarded Adam of one step (dnet: run the backward pass first).  fuse_fc1: everything in front of
        yolo_fc1/weights is scanned and updated from the stored gradients, then the weight is updated from the layer's input
        and dz without its gradient ever being stored; dz is range-checked against the arithmetic type's largest finite
        value first (the kernel rounds it to that type: ADVICE r5 -- a |dz| above 65504 is finite in the fp32 bias gradient
        that used to stand guard alone)"""
        fused = self._fc1_fused_now()
        if dnet is not None:
            self.backward(dnet, skip_fc1_dw=fused)
        dist, world = self._dp()
        n = self.offset["yolo_fc1/weights"][0] if fused else self.params.numel()
        if dist is not None:
            from ..trainer import reduce_flat
            reduce_flat(self.grads, [(0, n)], dist, self.dp_strategy)
        gmult = 1.0 / (self.loss_scale * world)
        E.check(lib.y2_grad_check_full(E._ptr(self.grads), n, E._ptr(self.ctrl), E._stream()))
        if fused:
            flat, dz1 = self._fc1_operands
            if dist is not None:
                flat_all = torch.empty((world * flat.shape[0], flat.shape[1]), dtype=flat.dtype, device=flat.device)
                dz_all = torch.empty((world * dz1.shape[0], dz1.shape[1]), dtype=dz1.dtype, device=dz1.device)
                dist.all_gather_into_tensor(flat_all, flat.contiguous())
                dist.all_gather_into_tensor(dz_all, dz1.contiguous())
                flat, dz1 = flat_all, dz_all
            limit = 65504.0 if self.dtype == "f16" else 3.0e38
            E.check(lib.y2_range_check(E._ptr(dz1), dz1.numel(), limit, E._ptr(self.ctrl), E._stream()))
            E.check(lib.y2_range_check(E._ptr(flat), flat.numel(), limit, E._ptr(self.ctrl), E._stream()))
        E.check(lib.y2_adam_step_guarded(E._ptr(self.params), E._ptr(self.m), E._ptr(self.v), E._ptr(self.grads), n,
                                         E._ptr(self.ctrl), self.lr, 0.9, 0.999, 1e-8, gmult, E._stream()))
        if fused:
            o, cnt = self.offset["yolo_fc1/weights"]
            E.check(lib.y2_fc_adam_apply_guarded(E._ptr(flat), E._ptr(dz1), E._ptr(self.params[o:o + cnt]), E._ptr(self.m[o:o + cnt]),
                                                 E._ptr(self.v[o:o + cnt]), flat.shape[0], flat.shape[1], dz1.shape[1],
                                                 E._lib.DTYPES[self.dtype], E._ptr(self.ctrl), 0.9, 0.999, 1e-8,
                                                 gmult, E._stream()))
            self._fc1_operands = None

    def _follow_ctrl(self):
        """host read of the device control block: step count, skipped steps -> loss scale (graph mode)"""
        c = self.ctrl.cpu()
        self.t = int(c[1])
        skipped = int(c[2])
        changed = False
        if skipped > self._skipped_seen:
            self.overflows += skipped - self._skipped_seen
            self._skipped_seen = skipped
            self._clean = 0
            if self.loss_scale > 1.0:
                self.loss_scale = max(self.loss_scale * 0.5, 1.0)
                changed = True
        else:
            self._clean += self._since_check
            if self.guard and self._clean >= self.growth_interval and self.loss_scale < 65536.0:
                self._clean = 0
                self.loss_scale *= 2.0
                changed = True
        self._since_check = 0
        if changed:
            self._graph = None          # the scale is an argument of two captured launches: capture again

    def _step_graph(self, images, labels):
        cur = torch.cuda.current_stream()
        if self._gstream is None:
            self._gstream = torch.cuda.Stream(device=self.device)
            self._gin = (torch.empty_like(images), torch.empty_like(labels))
        gs = self._gstream
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            self._gin[0].copy_(images)
            self._gin[1].copy_(labels)
            if self._eager_on_gstream < 2:
                # two ordinary steps on the graph's stream first: they size the per-stream scratch buffers and set
                # the kernels' attributes (neither is allowed inside a capture); they are real steps
                self._eager_on_gstream += 1
                out = self._step_device(*self._gin)
            else:
                if self._graph is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=gs):
                        self._gout = self._step_device(*self._gin)
                    self._graph = g
                self._graph.replay()
                out = self._gout
        cur.wait_stream(gs)
        self._since_check += 1
        if self._since_check >= self.graph_check_every:
            self._follow_ctrl()
        return out

    def step(self, images, labels):
        """one iteration of pascal_train_resnet.py:49-62: get_loss + AdamOptimizer(0.0005).minimize"""
        if self.graph and self._dp()[0] is None:      # (replicas: the collectives are issued from the host, step by step)
            return self._step_graph(images, labels)
        grid = self.forward(images, True, update_moving=True)
        loss, ious, mask, dnet = E.yolo_loss(grid, labels, self.num_class, self.batch, self.size, self.S, self.B)
        lib = E._lib.load()
        if self.loss_scale != 1.0:
            E.check(lib.y2_scale(E._ptr(dnet), dnet.numel(), self.loss_scale, E._stream()))
        self.backward(dnet, skip_fc1_dw=self.guard and self._fc1_fused_now())
        n = self.params.numel()
        if not self.guard:
            dist, world = self._dp()
            if dist is not None:                         # f32: every gradient is stored; one SUM over the flat buffer
                from ..trainer import reduce_flat
                reduce_flat(self.grads, [(0, n)], dist, self.dp_strategy)
            self.t += 1
            E.check(lib.y2_adam_step(E._ptr(self.params), E._ptr(self.m), E._ptr(self.v), E._ptr(self.grads), n, self.t,
                                     self.lr, 0.9, 0.999, 1e-8, 1.0 / (self.loss_scale * world), E._stream()))
            self.params_changed()
            return loss, ious, mask
        # half precision: full overflow scan, then the guarded update (skipped as a whole on the device when any
        # gradient is inf / NaN; the step counter and TF's lr_t live in ctrl)
        self._guarded_update(lib, None)
        self.params_changed()
        c = self.ctrl.cpu()                      # this untuned batch-4 path can afford the host read every step
        self.t = int(c[1])
        if int(c[0]):
            self.overflows += 1
            self._clean = 0
            self.loss_scale = max(self.loss_scale * 0.5, 1.0)
        else:
            self._clean += 1
            if self._clean >= self.growth_interval and self.loss_scale < 65536.0:
                self._clean = 0
                self.loss_scale *= 2.0
        return loss, ious, mask


def resnet_v1_50(inputs, num_classes=None, is_training=True, global_pool=False, output_stride=None, reuse=None,
                 scope='resnet_v1_50', **kw):
    """tf_resnet.py:12-32 signature: returns the model object whose .forward(inputs) evaluates the graph"""
    if num_classes is not None or global_pool or output_stride is not None:
        raise NotImplementedError("the reference's swap uses the dense features only (global_pool=False, no logits)")
    return ResNet50Yolo(inputs.shape[0], inputs.shape[1], **kw)
