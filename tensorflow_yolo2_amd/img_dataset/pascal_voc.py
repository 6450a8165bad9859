"""Host-side VOC label encoder and image preprocessing -- the data formats on the
input side of the hot path (src/img_dataset/pascal_voc.py:60-67,125-165).  Pure
numpy host code (the reference's is numpy/cv2 too); file IO over a real VOCdevkit
is out of scope (no dataset here), but everything that touches the arithmetic is
kept: the [S,S,25] grid, first-object-wins, no `difficult` filtering."""
import xml.etree.ElementTree as ET

import numpy as np

CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat',
           'bottle', 'bus', 'car', 'cat', 'chair',
           'cow', 'diningtable', 'dog', 'horse',
           'motorbike', 'person', 'pottedplant',
           'sheep', 'sofa', 'train', 'tvmonitor')       # pascal_voc.py:23-27


def encode_boxes(objs, im_h, im_w, image_size, cell_size, num_class=20):
    """load_pascal_annotation's arithmetic (pascal_voc.py:133-165).
    objs: (xmin, ymin, xmax, ymax, class_index) in 1-based pixels of the original image."""
    h_ratio = 1.0 * image_size / im_h
    w_ratio = 1.0 * image_size / im_w
    label = np.zeros((cell_size, cell_size, 5 + num_class))
    for (xmin, ymin, xmax, ymax, cls_ind) in objs:
        x1 = max(min((float(xmin) - 1) * w_ratio, image_size - 1), 0)
        y1 = max(min((float(ymin) - 1) * h_ratio, image_size - 1), 0)
        x2 = max(min((float(xmax) - 1) * w_ratio, image_size - 1), 0)
        y2 = max(min((float(ymax) - 1) * h_ratio, image_size - 1), 0)
        boxes = [(x2 + x1) / 2.0, (y2 + y1) / 2.0, x2 - x1, y2 - y1]
        x_ind = int(boxes[0] * cell_size / image_size)
        y_ind = int(boxes[1] * cell_size / image_size)
        if label[y_ind, x_ind, 0] == 1:
            continue
        label[y_ind, x_ind, 0] = 1
        label[y_ind, x_ind, 1:5] = boxes
        label[y_ind, x_ind, 5 + cls_ind] = 1
    return label


def load_pascal_annotation(xml_path_or_text, image_size, cell_size, im_shape=None):
    """(label [S,S,25], number of objects).  The reference opens the JPEG only for its
    shape (pascal_voc.py:131-134); the XML's <size> carries the same numbers."""
    text = xml_path_or_text
    if "<annotation" not in text:
        with open(xml_path_or_text) as f:
            text = f.read()
    root = ET.fromstring(text)
    if im_shape is None:
        size = root.find('size')
        im_shape = (int(size.find('height').text), int(size.find('width').text))
    objs = []
    for obj in root.findall('object'):
        bb = obj.find('bndbox')
        cls_ind = CLASSES.index(obj.find('name').text.lower().strip())
        objs.append((float(bb.find('xmin').text), float(bb.find('ymin').text),
                     float(bb.find('xmax').text), float(bb.find('ymax').text), cls_ind))
    return encode_boxes(objs, im_shape[0], im_shape[1], image_size, cell_size), len(objs)


def resize_bilinear_u8(img, out_h, out_w):
    """cv2.resize(img, (w, h)) INTER_LINEAR on uint8 (pascal_voc.py:62): half-pixel
    centres, no antialias, 11-bit fixed-point coefficients (OpenCV's documented scheme)."""
    img = np.asarray(img)
    in_h, in_w = img.shape[:2]

    def coeffs(n_in, n_out):
        f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = np.where(i0 < 0, 0.0, f - i0)
        w1 = np.rint(frac * 2048).astype(np.int64)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), 2048 - w1, w1

    y0, y1, wy0, wy1 = coeffs(in_h, out_h)
    x0, x1, wx0, wx1 = coeffs(in_w, out_w)
    a = img.astype(np.int64)
    r0, r1 = a[y0], a[y1]
    top = r0[:, x0] * wx0[None, :, None] + r0[:, x1] * wx1[None, :, None]
    bot = r1[:, x0] * wx0[None, :, None] + r1[:, x1] * wx1[None, :, None]
    return ((top * wy0[:, None, None] + bot * wy1[:, None, None] + (1 << 21)) >> 22).astype(np.uint8)


def image_read(image_bgr_u8, image_size, flipped=False):
    """pascal_voc.image_read (:60-67) on an already decoded BGR uint8 array."""
    image = resize_bilinear_u8(image_bgr_u8, image_size, image_size).astype(np.float32)
    image = (image / 255.0) * 2.0 - 1.0
    return image[:, ::-1, :] if flipped else image


def imread_bgr(path):
    """cv2.imread(imname) (pascal_voc.py:61): uint8 [H,W,3] in BGR order.  Decoder: PIL (no OpenCV in this image);
    libjpeg builds may differ by one level on some pixels, which is why the C1 fixture pins the DECODED input."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def flip_label(label, image_size):
    """the flipped copy of a label grid (pascal_voc.prepare, :74-84): columns mirrored, x -> image_size - 1 - x"""
    out = label[:, ::-1, :].copy()
    resp = out[:, :, 0] == 1
    out[:, :, 1] = np.where(resp, image_size - 1 - out[:, :, 1], out[:, :, 1])
    return out


class pascal_voc(object):
    """The batcher of src/img_dataset/pascal_voc.py:13-86 (`imdb.get()` feeds one sess.run per step,
    pascal_train_darknet.py:96-102): VOC2007 devkit layout (ImageSets/Main/<image_set>.txt, JPEGImages/<i>.jpg,
    Annotations/<i>.xml), images without objects dropped (:116-118), optional flip duplication (:72-85), one
    shuffle at start and one at every wrap of the cursor (:55-57,86).

    get() returns the reference's (images [B,size,size,3] float32 in [-1,1], labels [B,S,S,25]); get_u8() returns
    the same batch with the images still uint8 BGR (resized, before / 255 * 2 - 1) for y2_forward_u8, which
    applies that conversion on the device.  Decoded + resized images are kept in host memory after their first
    use (cache_images; 520 KB per 416x416 image) -- the reference re-decodes every time.

    Data parallelism (round 6; not in the reference): `rank` / `world` shard every epoch by stride.
    All ranks hold the SAME shuffled list (same seed, same number of shuffles: the generator streams stay in lockstep);
    rank r reads positions r, r + world, r + 2 world, ... of it, len // world positions per epoch on every rank (the
    < world entries at the tail of an epoch's order are skipped that epoch -- another order the next one), then all
    reshuffle together.  world = 1 is the reference's cursor, entry for entry."""

    def __init__(self, image_set, batch_size=None, rebuild=False, devkit_path=None, image_size=None, cell_size=None,
                 flipped=None, seed=0, cache_images=True, rank=0, world=1):
        import os
        from .. import config as cfg
        self.name = 'voc_2007'
        self.devkit_path = devkit_path or os.path.join('data', 'VOCdevkit')
        self.data_path = os.path.join(self.devkit_path, 'VOC2007')
        self.batch_size = cfg.BATCH_SIZE if batch_size is None else batch_size
        self.image_size = cfg.IMAGE_SIZE if image_size is None else image_size
        self.cell_size = (self.image_size // 32) if cell_size is None else cell_size
        self.classes = CLASSES
        self.num_class = len(CLASSES)
        self.class_to_ind = dict(zip(self.classes, range(self.num_class)))
        self.flipped = bool(getattr(cfg, "FLIPPED", False)) if flipped is None else bool(flipped)
        self.image_set = image_set
        self.cursor = 0
        assert world >= 1 and 0 <= rank < world, (rank, world)
        self.rank, self.world = int(rank), int(world)
        self.rng = np.random.default_rng(seed)
        self.cache_images = cache_images
        self._cache = {}
        assert os.path.exists(self.data_path), 'Path does not exist: {}'.format(self.data_path)
        self.gt_labels = self.prepare()
        self.per_rank = len(self.gt_labels) // self.world      # positions of one epoch on every rank
        assert self.per_rank >= 1, "fewer images (%d) than ranks (%d)" % (len(self.gt_labels), self.world)

    # ---- pascal_voc.py:69-124
    def load_labels(self):
        import os
        txtname = os.path.join(self.data_path, 'ImageSets', 'Main', self.image_set + '.txt')
        assert os.path.exists(txtname), 'Path does not exist: {}'.format(txtname)
        with open(txtname) as f:
            self.image_index = [x.strip() for x in f.readlines() if x.strip()]
        gt_labels = []
        for index in self.image_index:
            imname = os.path.join(self.data_path, 'JPEGImages', index + '.jpg')
            xml = os.path.join(self.data_path, 'Annotations', index + '.xml')
            # the reference reads the JPEG for its shape (:131-134); the header is enough
            from PIL import Image
            with Image.open(imname) as im:
                w, h = im.size
            label, num = load_pascal_annotation(xml, self.image_size, self.cell_size, im_shape=(h, w))
            if num == 0:
                continue
            gt_labels.append({'imname': imname, 'label': label, 'flipped': False})
        return gt_labels

    def prepare(self):
        gt_labels = self.load_labels()
        if self.flipped:
            gt_labels = gt_labels + [{'imname': g['imname'], 'label': flip_label(g['label'], self.image_size),
                                      'flipped': True} for g in gt_labels]
        self.rng.shuffle(gt_labels)
        return gt_labels

    # ---- pascal_voc.py:60-67
    def image_read_u8(self, imname, flipped=False):
        img = self._cache.get(imname)
        if img is None:
            img = resize_bilinear_u8(imread_bgr(imname), self.image_size, self.image_size)
            if self.cache_images:
                self._cache[imname] = img
        return img[:, ::-1, :] if flipped else img

    def image_read(self, imname, flipped=False):
        image = self.image_read_u8(imname, flipped).astype(np.float32)
        return (image / 255.0) * 2.0 - 1.0

    # ---- pascal_voc.py:42-58
    def _next(self):
        g = self.gt_labels[self.cursor * self.world + self.rank]
        self.cursor += 1
        if self.cursor >= self.per_rank:
            self.rng.shuffle(self.gt_labels)
            self.cursor = 0
        return g

    def get(self):
        images = np.zeros((self.batch_size, self.image_size, self.image_size, 3), np.float32)
        labels = np.zeros((self.batch_size, self.cell_size, self.cell_size, 25), np.float32)
        for count in range(self.batch_size):
            g = self._next()
            images[count] = self.image_read(g['imname'], g['flipped'])
            labels[count] = g['label']
        return images, labels

    def get_u8(self, images_out=None, labels_out=None):
        """the batch get() would return, images as uint8 BGR; writes into caller buffers (pinned memory) if given"""
        images = np.empty((self.batch_size, self.image_size, self.image_size, 3), np.uint8) if images_out is None \
            else images_out
        labels = np.empty((self.batch_size, self.cell_size, self.cell_size, 25), np.float32) if labels_out is None \
            else labels_out
        for count in range(self.batch_size):
            g = self._next()
            images[count] = self.image_read_u8(g['imname'], g['flipped'])
            labels[count] = g['label']
        return images, labels
