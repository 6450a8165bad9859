"""numpy restatement of the host-side data formats either side of the hot path
(src/img_dataset/pascal_voc.py:60-67,125-165; src/pascal/pascal_detect_darknet.py:34-38).
TEST INFRASTRUCTURE ONLY.
"""
import xml.etree.ElementTree as ET
import numpy as np

VOC_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat',
               'bottle', 'bus', 'car', 'cat', 'chair',
               'cow', 'diningtable', 'dog', 'horse',
               'motorbike', 'person', 'pottedplant',
               'sheep', 'sofa', 'train', 'tvmonitor')      # pascal_voc.py:23-27


def encode_boxes(objs, im_h, im_w, image_size, cell_size, num_class=20):
    """pascal_voc.py:133-165.  objs: iterable of (xmin, ymin, xmax, ymax, cls_ind)
    in VOC 1-based pixel coordinates of the ORIGINAL image.
    Returns label [S,S,5+num_class] float64 (np.zeros default dtype, :137)."""
    h_ratio = 1.0 * image_size / im_h                                  # :133
    w_ratio = 1.0 * image_size / im_w                                  # :134
    label = np.zeros((cell_size, cell_size, 5 + num_class))            # :137
    for (xmin, ymin, xmax, ymax, cls_ind) in objs:
        x1 = max(min((float(xmin) - 1) * w_ratio, image_size - 1), 0)  # :146-147
        y1 = max(min((float(ymin) - 1) * h_ratio, image_size - 1), 0)
        x2 = max(min((float(xmax) - 1) * w_ratio, image_size - 1), 0)
        y2 = max(min((float(ymax) - 1) * h_ratio, image_size - 1), 0)
        boxes = [(x2 + x1) / 2.0, (y2 + y1) / 2.0, x2 - x1, y2 - y1]   # :156
        x_ind = int(boxes[0] * cell_size / image_size)                 # :157
        y_ind = int(boxes[1] * cell_size / image_size)                 # :158
        if label[y_ind, x_ind, 0] == 1:                                # :159 first object wins
            continue
        label[y_ind, x_ind, 0] = 1
        label[y_ind, x_ind, 1:5] = boxes
        label[y_ind, x_ind, 5 + cls_ind] = 1
    return label


def parse_voc_xml(xml_text):
    """pascal_voc.py:140-155: (width, height, [(xmin,ymin,xmax,ymax,cls_ind)]).
    `difficult` objects are NOT filtered (the reference does not)."""
    root = ET.fromstring(xml_text)
    size = root.find('size')
    w, h = int(size.find('width').text), int(size.find('height').text)
    objs = []
    for obj in root.findall('object'):
        bb = obj.find('bndbox')
        cls_ind = VOC_CLASSES.index(obj.find('name').text.lower().strip())
        objs.append((float(bb.find('xmin').text), float(bb.find('ymin').text),
                     float(bb.find('xmax').text), float(bb.find('ymax').text), cls_ind))
    return w, h, objs


def normalise(image_u8_or_f32):
    """(image / 255.0) * 2.0 - 1.0 in float32 (pascal_voc.py:63-64)."""
    image = np.asarray(image_u8_or_f32).astype(np.float32)
    return (image / 255.0) * 2.0 - 1.0


def resize_bilinear_u8(img, out_h, out_w):
    """cv2.resize(img, (w,h)) default INTER_LINEAR on uint8: half-pixel centres,
    no antialias, fixed-point 11-bit coefficients.  cv2 is absent, so bit-equality
    with OpenCV is UNPINNED; this restates its documented algorithm."""
    img = np.asarray(img)
    in_h, in_w = img.shape[:2]

    def coeffs(n_in, n_out):
        scale = n_in / n_out
        f = (np.arange(n_out) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = f - i0
        lo_clip = i0 < 0
        i0c = np.clip(i0, 0, n_in - 1)
        i1c = np.clip(i0 + 1, 0, n_in - 1)
        frac = np.where(lo_clip, 0.0, frac)
        w1 = np.rint(frac * 2048).astype(np.int64)          # INTER_RESIZE_COEF_SCALE = 1<<11
        w0 = 2048 - w1
        return i0c, i1c, w0, w1

    y0, y1, wy0, wy1 = coeffs(in_h, out_h)
    x0, x1, wx0, wx1 = coeffs(in_w, out_w)
    a = img.astype(np.int64)
    rows0, rows1 = a[y0], a[y1]
    top = rows0[:, x0] * wx0[None, :, None] + rows0[:, x1] * wx1[None, :, None]
    bot = rows1[:, x0] * wx0[None, :, None] + rows1[:, x1] * wx1[None, :, None]
    acc = top * wy0[:, None, None] + bot * wy1[:, None, None]
    return ((acc + (1 << 21)) >> 22).astype(np.uint8)
