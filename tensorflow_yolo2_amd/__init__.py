"""MI355X-native Darknet-19 / YOLO grid detector (hot path of wenxichen/tensorflow_yolo2).

Importing this package does not touch the GPU; the HIP library is loaded (and
must exist) on first use -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
