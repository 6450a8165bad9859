"""CPU oracle for the Darknet-19 / YOLO grid-detector hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tensorflow_yolo2_amd/`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

It is a numpy restatement of the arithmetic that the reference
(wenxichen/tensorflow_yolo2, TF1/py2, cannot run here) delegates to TensorFlow
ops.  Every function cites the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * 3x3 SAME conv + stride-2 subsample: PINNED by the reference's own
    known-answer tests (src/slim_dir/nets/resnet_v1_test.py:58-152).
  * label encode (pascal_voc.py:125-165): PINNED by hand-derived answers for
    the reference fixture tests/testImg2Anno.xml.
  * YOLO_GRID_OFFSET (config.py:40-42): PINNED (closed form).
  * batch-norm, leaky, max-pool, get_iou, get_loss, decode, Adam/Momentum:
    PARITY UNPINNED by the reference (it holds no tests or golden vectors for
    them and TF1 cannot be imported here); they are cross-checked against an
    independent PyTorch-CPU implementation + autograd in tests/.
"""
from . import nn_ref, loss_ref, optim_ref, data_ref  # noqa: F401
