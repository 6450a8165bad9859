"""YOLOv2 detector composed from the pieces of this library -- NOT in the reference (SURVEY.md §8 rows a-x1,
a-x2; the north star's model): Darknet-19 up to the 26x26x512 activation, 2x2 max pool, the 13x13 backbone
layers and two 3x3 head convolutions, the stride-2 passthrough (reorg 26x26x512 -> 13x13x2048, concatenated
with the 13x13x1024 path), a 3x3 convolution 3072 -> 1024, the 1x1 output convolution to B*(5+C) channels,
anchor-box decode and per-image NMS.  Every layer is the reference's conv-BN-leaky type (darknet.py:39-46),
so the three stacks are `engine.Network` contexts; the glue ops are csrc/ext.hip.
`YOLOv2Detector` is the inference graph, `YOLOv2Trainer` the train step: anchor-box loss (csrc/ext.hip
yolov2_loss_kernel, specification oracle/ext_ref.py), backward through the head, the passthrough concat, the
13x13 stack, the 2x2 pool and the stem, Adam on the three flat parameter buffers, optional multi-scale.
"""
import numpy as np
import torch

from .. import engine as E

# k-means anchors of YOLOv2 on VOC, in 13x13 cell units (width, height)
ANCHORS_VOC = ((1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071))


def yolov2_specs(num_class=20, num_anchors=5):
    core = [tuple(s) for s in E.CORE_SPEC]
    a = core[:13]
    a[12] = (a[12][0], a[12][1], a[12][2], 0)            # keep the 26x26x512 activation un-pooled
    b = core[13:18] + [(3, 1024, 1024, 0), (3, 1024, 1024, 0)]
    c = [(3, 4 * 512 + 1024, 1024, 0), (1, 1024, num_anchors * (5 + num_class), 0)]
    return a, b, c


class YOLOv2Detector:
    def __init__(self, batch, image_size=416, num_class=20, anchors=ANCHORS_VOC, dtype="f16", seed=0,
                 device="cuda:0"):
        assert image_size % 32 == 0
        self.batch, self.size, self.S = batch, image_size, image_size // 32
        self.num_class, self.anchors = num_class, np.asarray(anchors, np.float32)
        self.B = len(self.anchors)
        sa, sb, sc = yolov2_specs(num_class, self.B)
        S = self.S
        self.stem = E.Network(sa, batch, image_size, image_size, dtype=dtype, training=False, device=device)
        self.deep = E.Network(sb, batch, S, S, dtype=dtype, training=False, device=device)
        self.head = E.Network(sc, batch, S, S, dtype=dtype, training=False, device=device)
        for i, net in enumerate((self.stem, self.deep, self.head)):
            net.init_params(seed + i)

    def forward(self, images):
        """images [N,size,size,3] fp32 on the device -> raw grid [N,S,S,B,5+C]"""
        fine = self.stem.forward(images, False, False)                      # [N,2S,2S,512]
        coarse = self.deep.forward(E.max_pool_2x2(fine), False, False)      # [N,S,S,1024]
        cat = E.passthrough_concat(fine, coarse)                            # [N,S,S,3072]
        out = self.head.forward(cat, False, False)                          # [N,S,S,B*(5+C)]
        return out.view(self.batch, self.S, self.S, self.B, 5 + self.num_class)

    def detect(self, images, score_thresh=0.3, iou_thresh=0.45, max_out=100, class_aware=True):
        """-> boxes [N,K,4] (cx,cy,w,h relative), best score [N,K], class id [N,K], keep [N,max_out], count [N]"""
        grid = self.forward(images)
        boxes, scores = E.decode_anchors(grid.contiguous(), self.anchors)
        best, cls = E.class_argmax(scores)
        keep, count = E.nms(boxes, best, cls, iou_thresh, score_thresh, max_out, class_aware)
        return boxes, best, cls, keep, count


class YOLOv2Trainer:
    """Train step of the YOLOv2 detector (NOT in the reference: its trainer is the YOLOv1 grid model).
    Three conv-BN-leaky stacks, each with its own flat parameter / gradient buffer and Adam state.  With `sizes`
    the input size may change from step to step (multi-scale {320..608}): one context + workspace per size, all
    bound to the same three parameter sets (engine.Network(share_with=...))."""

    def __init__(self, batch, image_size=416, num_class=20, anchors=ANCHORS_VOC, dtype="f16", seed=0,
                 device="cuda:0", scales=None, width_div=1):
        from ..trainer import GradReducer
        self._GradReducer = GradReducer
        self.batch, self.num_class, self.dtype, self.device, self.seed = batch, num_class, dtype, device, seed
        self.anchors = np.asarray(anchors, np.float32)
        self.B = len(self.anchors)
        self.scales = scales
        sa, sb, sc = yolov2_specs(num_class, self.B)
        if width_div > 1:      # narrow variant for tests: every inner width divided, the output width kept
            div = lambda c: max(32, c // width_div // 32 * 32)
            sa = [(k, ci if i == 0 else div(ci), div(co), p) for i, (k, ci, co, p) in enumerate(sa)]
            sb = [(k, div(ci), div(co), p) for (k, ci, co, p) in sb]
            sc = [(3, 4 * sa[-1][2] + sb[-1][2], div(1024), 0), (1, div(1024), sc[1][2], 0)]
        self.specs = (sa, sb, sc)
        self.cf = sa[-1][2]
        self.ctx = {}
        self.opts = None
        self.default_size = image_size
        self._nets(image_size)

    def _nets(self, size):
        if size not in self.ctx:
            assert size % 32 == 0
            S = size // 32
            first = next(iter(self.ctx.values())) if self.ctx else None
            sa, sb, sc = self.specs
            mk = lambda spec, h, share: E.Network(spec, self.batch, h, h, dtype=self.dtype, training=True,
                                                  device=self.device, share_with=share)
            nets = (mk(sa, size, first[0] if first else None), mk(sb, S, first[1] if first else None),
                    mk(sc, S, first[2] if first else None))
            if first is None:
                for i, net in enumerate(nets):
                    net.init_params(self.seed + i)
                self.opts = [E.AdamOptimizer(net) for net in nets]
                # ONE loss scale, overflow flag and step counter for the composed graph: an inf confined to
                # one stack must skip the step of all three, and a skipped step must not advance any Adam bias correction
                for opt in self.opts[1:]:
                    opt.scaler = self.opts[0].scaler
            self.ctx[size] = nets
            self.reducers = getattr(self, "reducers", {})
            self.reducers[size] = [self._GradReducer(net) for net in nets]
        return self.ctx[size]

    @property
    def nets(self):
        return self.ctx[self.default_size]

    def forward(self, images, training=True, update_moving=False):
        size = int(images.shape[1])
        stem, deep, head = self._nets(size)
        S = size // 32
        fine = stem.forward(images, training, training, update_moving=update_moving)          # [N,2S,2S,Cf]
        pooled = E.max_pool_2x2(fine)
        coarse = deep.forward(pooled, training, training, update_moving=update_moving)         # [N,S,S,Cc]
        cat = E.passthrough_concat(fine, coarse)
        out = head.forward(cat, training, training, update_moving=update_moving)
        return out.view(self.batch, S, S, self.B, 5 + self.num_class), fine

    def step(self, images, labels):
        """one train step; returns loss[5] = coord, object, noobject, class, total"""
        from ..trainer import _dist
        size = int(images.shape[1])
        stem, deep, head = self._nets(size)
        for net in (stem, deep, head):
            if len(self.ctx) > 1:
                net.params_changed()              # shared parameters moved under another size's contexts
        grid, fine = self.forward(images, True, update_moving=True)
        loss, dnet = E.yolov2_loss(grid.contiguous(), labels, self.anchors, size, True, self.scales)
        dist = _dist()
        red = self.reducers[size]
        dcat = head.backward_input(dnet.view(self.batch, size // 32, size // 32, -1))
        if dist is not None:
            red[2].reduce_all_async()
        dfine, dcoarse = E.passthrough_concat_backward(dcat, self.cf)
        dpooled = deep.backward_input(dcoarse)
        if dist is not None:
            red[1].reduce_all_async()
        E.accumulate(dfine, E.max_pool_2x2_backward(fine, dpooled))
        world = red[0].backward_and_reduce(dfine)
        if dist is not None:
            red[1].join(); red[2].join()
        scaler = self.opts[0].scaler
        for opt, net in zip(self.opts, (stem, deep, head)):
            opt.net = net
            if scaler is not None:
                scaler.attach(net)
        if scaler is None:
            for opt in self.opts:
                opt.step(grad_mult=1.0 / world)
            return loss
        # every stack's sentinels into the one flag, then the three guarded updates: all of them or none
        for i, net in enumerate((stem, deep, head)):
            scaler.scan(net, more=(i > 0))
        for i, opt in enumerate(self.opts):
            opt.step(grad_mult=1.0 / world, joint="first" if i == 0 else "next")
        return loss

    def networks(self, size=None):
        """the three stack contexts (stem, 13x13 stack, head) of an input size"""
        return self._nets(size or self.default_size)

    def flops_per_step(self, size=None, mfma_launches_only=False):
        """algorithmic FLOPs of one train step (3 x forward conv FLOPs, SURVEY 8(d) convention);
        mfma_launches_only: without the 3-channel first layer, which has its own (non-GEMM) kernels"""
        size = size or self.default_size
        tot = 0.0
        for spec, h0 in zip(self.specs, (size, size // 32, size // 32)):
            h = h0
            for (k, ci, co, pool) in spec:
                if not (mfma_launches_only and ci == 3):
                    tot += 2.0 * self.batch * h * h * k * k * ci * co
                if pool:
                    h = (h + 1) // 2
        return 3.0 * tot
