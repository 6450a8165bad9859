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
