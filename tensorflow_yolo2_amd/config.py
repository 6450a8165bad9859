"""Counterpart of src/config.py: the hyper-parameters that feed the hot path."""
import numpy as np

TRAIN_SNAPSHOT_PREFIX = 'train'
BATCH_SIZE = 48          # config.py:32
IMAGE_SIZE = 224         # config.py:34

# YOLO1 VOC settings (config.py:37-45)
S = 7
B = 2
LAMBDA_COORD = 5
LAMBDA_NOOBJ = 0.5


def yolo_grid_offset(S=S, B=B):
    """config.py:40-42 (py2 `range(S) * S * B`): [S(y), S(x), B] array whose value is the column index."""
    off = np.array(list(range(S)) * S * B)
    off = np.reshape(off, (B, S, S))
    return np.transpose(off, (1, 2, 0))


YOLO_GRID_OFFSET = yolo_grid_offset(S, B)


def get_ckpts_dir(network_name, imdb_name, root="ckpts"):
    """config.py:80-90: <root>/<network_name>/<imdb_name>, created on demand"""
    import os
    path = os.path.abspath(os.path.join(root, network_name, imdb_name))
    os.makedirs(path, exist_ok=True)
    return path
