"""YOLOv2 detector composed from the pieces of this library -- NOT in the reference (SURVEY.md §8 rows a-x1,
a-x2; the north star's model): Darknet-19 up to the 26x26x512 activation, 2x2 max pool, the 13x13 backbone
layers and two 3x3 head convolutions, the stride-2 passthrough (reorg 26x26x512 -> 13x13x2048, concatenated
with the 13x13x1024 path), a 3x3 convolution 3072 -> 1024, the 1x1 output convolution to B*(5+C) channels,
anchor-box decode and per-image NMS.  Every layer is the reference's conv-BN-leaky type (darknet.py:39-46),
so the three stacks are `engine.Network` contexts; the glue ops are csrc/ext.hip.  Inference only.
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
        best, cls = scores.max(dim=2)
        keep, count = E.nms(boxes, best.contiguous(), cls.to(torch.int32).contiguous(), iou_thresh, score_thresh, max_out,
                            class_aware)
        return boxes, best, cls, keep, count
