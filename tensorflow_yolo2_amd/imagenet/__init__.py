"""Counterparts of src/imagenet/imagenet_{train,test,predict}_darknet.py (the callers of the classifier path,
SURVEY.md section 2 row 14).  The reference's ILSVRC loader (img_dataset/ilsvrc2017_cls_multithread.py: ten prefetch
processes, cv2 augmentation) is out of scope: batches come from synthetic data or from a plain image list."""
import numpy as np


def read_image_list(path):
    """lines `image-path label` -> [(path, int label)] (paths relative to the list file's directory)"""
    import os
    base = os.path.dirname(os.path.abspath(path))
    out = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        name, label = line.rsplit(None, 1)
        out.append((name if os.path.isabs(name) else os.path.join(base, name), int(label)))
    return out


def load_batch(items, size):
    """the loader's image_read without augmentation (ilsvrc2017_cls_multithread.py:320-415: cv2.imread -> resize ->
    x / 255 * 2 - 1 on BGR pixels) -> (images [n, size, size, 3] fp32, labels [n] int32)"""
    from PIL import Image
    from ..img_dataset import pascal_voc
    ims = []
    for path, _ in items:
        rgb = np.array(Image.open(path).convert("RGB"), dtype=np.uint8)
        ims.append(pascal_voc.image_read(rgb[:, :, ::-1], size))
    return np.ascontiguousarray(np.stack(ims), dtype=np.float32), np.asarray([l for _, l in items], np.int32)
