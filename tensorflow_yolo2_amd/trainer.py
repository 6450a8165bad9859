"""Train-step drivers: the counterpart of the `sess.run([loss, train_op, ...])` loop of
src/pascal/pascal_train_darknet.py:96-102 and src/imagenet/imagenet_train_darknet.py:106-135.

Data parallelism (not in the reference): one process per GPU,
identically seeded replicas, batch sharded by rank, BN statistics per replica, ONE
flat fp32 gradient buffer all-reduced (SUM) over RCCL/xGMI, then scaled by 1/world
inside the optimizer kernel -- slim's clone semantics (model_deploy.py:222-225,
436-446: clone loss / num_clones, add_n over clones).  The all-reduce is issued in
layer slices on a side stream as soon as backward has finished them (head first:
its three 37.7 MB filters are 59 % of the payload), overlapping the remaining
dgrad/wgrad kernels.
"""
import numpy as np
import torch

from . import _lib, engine
from .engine import (CORE_SPEC, CLS_HEAD_SPEC, det_head_spec, Network, AdamOptimizer, MomentumOptimizer,
                     yolo_loss, softmax_cross_entropy)


def _dist():
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    # Y2_FORCE_DIST=1: take the sliced all-reduce path even at world size 1 (single-GPU check of the RCCL calls)
    return dist if (dist.get_world_size() > 1 or os.environ.get("Y2_FORCE_DIST") == "1") else None


DEFAULT_CUTS = (0, 8, 13, 16, 18, 19, 20)


def layer_slices(num_layers, cuts=None):
    """Backward-order layer slices for the overlapped all-reduce.  Default cuts for the 22-layer detector:
    the three 37.7 MB head filters one by one (the first collective starts after ~10 % of the backward pass
    instead of ~30 %), the 13x13 block in two, then 26x26 (15 MB) and everything below it (3.4 MB): what is
    still to be reduced when the last weight gradient lands is 3.4 MB, not 18.7.
    Y2_DP_CUTS="0,13,18" overrides (round 1's three slices)."""
    import os
    if cuts is None:
        env = os.environ.get("Y2_DP_CUTS")
        cuts = tuple(int(v) for v in env.split(",")) if env else DEFAULT_CUTS
    cuts = sorted({0, num_layers} | {min(max(int(c), 0), num_layers) for c in cuts})
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)][::-1]


def slice_range(w_offsets, n_params, num_layers, lo, hi):
    """[start, end) of layers [lo, hi) in the flat buffer (w_offsets[l] = offset of layer l's filter)."""
    start = w_offsets[lo]
    end = n_params if hi == num_layers else w_offsets[hi]
    return start, end


def reduce_flat(flat, ranges, dist, strategy="allreduce"):
    """SUM-reduce the given [start, end) ranges of a flat tensor over all ranks (any device/backend).
    strategy "allreduce": one dist.all_reduce per range (RCCL picks ring / tree by size);
    strategy "rs_ag": explicit reduce_scatter + all_gather of each range (the direct algorithm over the
    point-to-point xGMI links SURVEY section 5 prices at ~0.3 ms for 193 MB against ~2.2 ms for one ring);
    the < world trailing elements of a range that do not divide go through a small all_reduce."""
    import os
    world = dist.get_world_size()
    # world 1 only under Y2_FORCE_DIST=1 (tests/rccl_worker.py: the real backend's calls on one GPU)
    multi = world > 1 or os.environ.get("Y2_FORCE_DIST") == "1"
    for (s, e) in ranges:
        if strategy == "rs_ag" and multi and (e - s) >= world * 1024:
            n = (e - s) // world * world
            body = flat[s:s + n]
            rank = dist.get_rank()
            shard = body[rank * (n // world):(rank + 1) * (n // world)]
            dist.reduce_scatter_tensor(shard, body, op=dist.ReduceOp.SUM)   # in place: my shard of the sums
            dist.all_gather_into_tensor(body, shard)
            if s + n < e:
                dist.all_reduce(flat[s + n:e], op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM)
    return flat


class GradReducer:
    """Sliced, overlapped all-reduce of a Network's flat gradient buffer."""

    def __init__(self, net, slices=None, strategy=None):
        import os
        self.net = net
        n = net.num_layers
        if slices is None:
            slices = layer_slices(n)
        self.slices = slices
        self.strategy = strategy or os.environ.get("Y2_DP_STRATEGY", "allreduce")
        assert self.strategy in ("allreduce", "rs_ag"), self.strategy
        self.comm_stream = torch.cuda.Stream(device=net.device) if net.device.type == "cuda" else None
        self._pending = []

    def describe(self):
        import os
        return {"collective": "rccl sum", "strategy": self.strategy, "slices": len(self.slices),
                "slice_layers": [list(s) for s in self.slices],
                "slice_MB": [round((self._range(lo, hi)[1] - self._range(lo, hi)[0]) * 4 / 1e6, 1)
                             for (lo, hi) in self.slices],
                "NCCL_ALGO": os.environ.get("NCCL_ALGO", "default"),
                "NCCL_PROTO": os.environ.get("NCCL_PROTO", "default")}

    def _range(self, lo, hi):
        return slice_range([o[0] for o in self.net._offsets], self.net.n_params, self.net.num_layers, lo, hi)

    def reduce_all_async(self):
        """SUM-reduce the whole gradient buffer on the communication stream once the work queued so far on the
        current stream is done (a stack of a composed graph whose backward just ran); join() before the update"""
        dist = _dist()
        if dist is None:
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            reduce_flat(self.net.grads, [(0, self.net.n_params)], dist, self.strategy)

    def join(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def backward_reduce_step(self, dout, opt):
        """backward + gradient mean over the replicas + optimizer step.  One process: the library's fused train_op
        (AdamOptimizer.backward_step); replicas: backward_and_reduce, then the step on the reduced gradients."""
        if _dist() is None:
            opt.backward_step(dout)
            return 1
        world = self.backward_and_reduce(dout)
        opt.step(grad_mult=1.0 / world)
        return world

    def backward_and_reduce(self, dout):
        dist = _dist()
        net = self.net
        if dist is None:
            net.backward(dout)
            return 1
        # ONE backward pass; the library records an event pair per slice boundary, the all-reduce of a slice
        # waits for its pair on the communication stream -- the compute stream never joins in between
        net.backward_marks(dout, [lo for (lo, hi) in self.slices])
        with torch.cuda.stream(self.comm_stream):
            for k, (lo, hi) in enumerate(self.slices):
                net.wait_mark(k, self.comm_stream)
                s, e = self._range(lo, hi)
                reduce_flat(net.grads, [(s, e)], dist, self.strategy)
        torch.cuda.current_stream().wait_stream(self.comm_stream)
        return dist.get_world_size()


class DetectorTrainer:
    """darknet19_core + darknet19_detection + get_loss + AdamOptimizer().minimize
    (src/pascal/pascal_train_darknet.py:39-51) as one object."""

    def __init__(self, batch, image_size=224, S=None, B=2, num_class=20, dtype="f16", device="cuda:0",
                 grad_scale=None, seed=0, core_spec=None, head_spec=None):
        core_spec = list(core_spec or CORE_SPEC)
        head_spec = list(head_spec or det_head_spec(5 * B + num_class))
        self.net = Network(core_spec + head_spec, batch, image_size, image_size, dtype=dtype,
                           core_layers=len(core_spec), training=True, device=device, grad_scale=grad_scale)
        self.batch, self.image_size, self.B, self.num_class = batch, image_size, B, num_class
        self.S = self.net.out_shape[1] if S is None else S
        assert self.net.out_shape[1] == self.S == self.net.out_shape[2], (self.net.out_shape, self.S)
        self.net.init_params(seed)
        self.opt = AdamOptimizer(self.net)
        self.reducer = GradReducer(self.net)
        self.last = None

    def forward_loss(self, images, labels, is_training=True, need_grad=True, update_moving=False):
        # head BN always batch stats (darknet.py:184); moving statistics move only inside a train step
        grid_net = self.net.forward(images, is_training, True, update_moving=update_moving)
        return grid_net, yolo_loss(grid_net, labels, self.num_class, self.batch, self.image_size, self.S, self.B,
                                   need_grad=need_grad)

    def step(self, images, labels):
        grid_net, (loss, ious, mask, dnet) = self.forward_loss(images, labels, True, True, update_moving=True)
        self.reducer.backward_reduce_step(dnet, self.opt)
        self.last = (loss, ious, mask)
        return loss, ious, mask


MULTI_SCALE_SIZES = tuple(range(320, 609, 32))   # YOLOv2 multi-scale schedule (BASELINE.json configs[4])


def multi_scale_size(step, sizes=MULTI_SCALE_SIZES, period=10, seed=0):
    """input size of training step `step`: redrawn every `period` steps, identical on every rank"""
    rng = np.random.default_rng([seed, step // period])
    return int(sizes[int(rng.integers(0, len(sizes)))])


class MultiScaleDetectorTrainer:
    """Detector train step at a size that changes every `period` batches ({320..608}, step 32).
    NOT in the reference (SURVEY §8 a-x2).  One context + workspace per size, created on first use, all
    bound to the same flat parameter / gradient / BN-state buffers and one Adam state."""

    def __init__(self, batch, sizes=MULTI_SCALE_SIZES, period=10, B=2, num_class=20, dtype="f16", device="cuda:0",
                 grad_scale=None, seed=0, core_spec=None, head_spec=None):
        self.batch, self.sizes, self.period, self.seed = batch, tuple(sizes), period, seed
        self.B, self.num_class, self.dtype, self.device, self.grad_scale = B, num_class, dtype, device, grad_scale
        self.core_spec = list(core_spec or CORE_SPEC)
        self.head_spec = list(head_spec or det_head_spec(5 * B + num_class))
        self.nets, self.reducers = {}, {}
        self.opt = None
        self.steps = 0

    def size_for_step(self, step=None):
        return multi_scale_size(self.steps if step is None else step, self.sizes, self.period, self.seed)

    def _net(self, size):
        if size not in self.nets:
            first = next(iter(self.nets.values())) if self.nets else None
            net = Network(self.core_spec + self.head_spec, self.batch, size, size, dtype=self.dtype,
                          core_layers=len(self.core_spec), training=True, device=self.device,
                          grad_scale=self.grad_scale, share_with=first)
            if first is None:
                net.init_params(self.seed)
                self.opt = AdamOptimizer(net)
            self.nets[size] = net
            self.reducers[size] = GradReducer(net)
        return self.nets[size]

    def step(self, images, labels):
        size = int(images.shape[1])
        assert size in self.sizes and size == int(images.shape[2]), images.shape
        net = self._net(size)
        net.params_changed()                       # the shared parameters moved since this context last ran
        S = net.out_shape[1]
        grid_net = net.forward(images, True, True, update_moving=True)
        loss, ious, mask, dnet = yolo_loss(grid_net, labels, self.num_class, self.batch, size, S, self.B)
        self.opt.net = net                         # same flat buffers; keeps params_changed() on the live context
        if self.opt.scaler is not None:
            self.opt.scaler.attach(net)            # one loss scale for every size
        self.reducers[size].backward_reduce_step(dnet, self.opt)
        self.steps += 1
        return loss, ious, mask


class ClassifierTrainer:
    """darknet19 + sparse softmax CE + MomentumOptimizer(0.001, 0.9)
    (src/imagenet/imagenet_train_darknet.py:46-58)."""

    def __init__(self, batch, image_size=224, dtype="f16", device="cuda:0", grad_scale=None, seed=0, spec=None):
        spec = list(spec or (CORE_SPEC + CLS_HEAD_SPEC))
        self.net = Network(spec, batch, image_size, image_size, dtype=dtype, core_layers=len(spec),
                           tail=_lib.Y2_TAIL_AVGPOOL, tail_k=image_size // 32, training=True, device=device,
                           grad_scale=grad_scale)
        self.net.init_params(seed)
        self.opt = MomentumOptimizer(self.net, 1e-3, 0.9)
        self.reducer = GradReducer(self.net)

    def step(self, images, labels):
        logits = self.net.forward(images, True, True, update_moving=True)
        loss, dlogits = softmax_cross_entropy(logits, labels)
        self.reducer.backward_reduce_step(dlogits, self.opt)
        return loss, logits
