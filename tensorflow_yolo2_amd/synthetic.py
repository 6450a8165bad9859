"""Synthetic inputs of the benchmark / smoke workloads:
images uniform in [-1, 1) (the range x/255*2-1 produces), labels with 1-3 random
boxes per image encoded exactly like src/img_dataset/pascal_voc.py:125-165."""
import numpy as np

from .img_dataset.pascal_voc import encode_boxes


def images(batch, size, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, (batch, size, size, 3)).astype(np.float32)


def det_labels(batch, size, S, seed, num_class=20):
    rng = np.random.default_rng(seed)
    labels = np.zeros((batch, S, S, 5 + num_class), np.float32)
    for i in range(batch):
        objs = []
        for _ in range(int(rng.integers(1, 4))):
            x1, y1 = rng.uniform(1, size * 0.7, 2)
            bw, bh = rng.uniform(size * 0.05, size * 0.3, 2)
            objs.append((x1, y1, x1 + bw, y1 + bh, int(rng.integers(0, num_class))))
        labels[i] = encode_boxes(objs, size, size, size, S, num_class)
    return labels


def cls_labels(batch, seed, classes=1000):
    return np.random.default_rng(seed).integers(0, classes, batch).astype(np.int32)
