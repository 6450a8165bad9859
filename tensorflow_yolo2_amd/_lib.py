"""ctypes binding of libyolo2_hip.so (C ABI: include/yolo2_hip.h).

There is NO fallback: if the HIP library is missing or a symbol is absent the
import fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C tensorflow_yolo2_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolo2_hip.so")
# development tools (scripts/bench_*.py) opt into the `make dev` library, which adds the kernel variant tables
if os.environ.get("Y2_DEV_LIB") == "1":
    LIB_PATH = os.path.join(_HERE, "libyolo2_hip_dev.so")
# same-box A/B of two builds (scripts/ab_layers.sh): an explicit library file; it must export the whole header too
if os.environ.get("Y2_LIB_PATH"):
    LIB_PATH = os.environ["Y2_LIB_PATH"]

Y2_F32, Y2_F16, Y2_BF16, Y2_F16X2, Y2_F16X2F = 0, 1, 2, 3, 4
# modes whose gradients ride on the f16 loss scale (dY is stored in f16 planes)
LOSS_SCALED = (Y2_F16, Y2_F16X2, Y2_F16X2F)
Y2_TAIL_NONE, Y2_TAIL_AVGPOOL = 0, 1
DTYPES = {"f32": Y2_F32, "fp32": Y2_F32, "float32": Y2_F32,
          "f16": Y2_F16, "fp16": Y2_F16, "float16": Y2_F16,
          "bf16": Y2_BF16, "bfloat16": Y2_BF16,
          # split-operand mode: (hi, lo) f16 pairs, three MFMAs per product, fp32-width storage (include/yolo2_hip.h)
          "f16x2": Y2_F16X2,
          # round 6: f16x2 forward, backward contractions on the hi planes only (one f16 MFMA per product)
          "f16x2f": Y2_F16X2F}

_vp, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64
_pi = C.POINTER(C.c_int)
_psz = C.POINTER(C.c_size_t)

# name -> (restype, argtypes); must list every symbol include/yolo2_hip.h declares
SIGNATURES = {
    "y2_last_error": (C.c_char_p, []),
    "y2_version": (_i, []),
    "y2_darknet19_spec": (_i, [_i, _i, _pi, _i]),
    "y2_ctx_create": (_i, [C.POINTER(_vp), _pi, _i, _i, _i, _i, _i, _i, _i, _i]),
    "y2_ctx_destroy": (None, [_vp]),
    "y2_num_layers": (_i, [_vp]),
    "y2_layer_info": (_i, [_vp, _i, _pi]),
    "y2_param_count": (_sz, [_vp]),
    "y2_state_count": (_sz, [_vp]),
    "y2_param_offsets": (_i, [_vp, _i, _psz]),
    "y2_output_shape": (_i, [_vp, _pi]),
    "y2_workspace_bytes": (_sz, [_vp, _i]),
    "y2_bind": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "y2_set_options": (_i, [_vp, _f, _i]),
    "y2_set_layer_options": (_i, [_vp, _vp, _i, _f, _f, _i]),
    "y2_init_params": (_i, [_vp, _u64, _vp]),
    "y2_params_changed": (_i, [_vp]),
    "y2_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "y2_forward_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "y2_forward_join": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "y2_update_moving_stats": (_i, [_vp, _vp]),
    "y2_backward": (_i, [_vp, _vp, _i, _i, _vp]),
    "y2_backward_marks": (_i, [_vp, _vp, _i, _pi, _vp]),
    "y2_wait_mark": (_i, [_vp, _i, _vp]),
    "y2_backward_input": (_i, [_vp, _vp, _vp, _vp]),
    "y2_debug_read": (_i, [_vp, _i, _i, _vp, _vp]),
    "y2_profile_enable": (_i, [_vp, _i]),
    "y2_profile_collect": (_i, [_vp, C.POINTER(C.c_double), _pi, _i]),
    "y2_profile_busy": (_i, [_vp, _i, _vp, _vp]),
    "y2_profile_layers": (_i, [_vp, C.POINTER(C.c_double)]),
    "y2_yolo_loss_workspace_bytes": (_sz, [_i, _i]),
    "y2_yolo_loss": (_i, [_vp, _vp, _i, _i, _f, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "y2_get_iou": (_i, [_vp, _vp, _vp, _i, _vp]),
    "y2_decode_detections": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "y2_softmax_cross_entropy": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "y2_accuracy": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "y2_maxpool2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_maxpool2x2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_reorg": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "y2_passthrough_concat": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "y2_passthrough_concat_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "y2_accumulate": (_i, [_vp, _vp, _sz, _vp]),
    "y2_add_relu": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "y2_add_relu_backward": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "y2_range_check": (_i, [_vp, _sz, _f, _vp, _vp]),
    "y2_fc_adam_apply_guarded": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _f, _f, _f, _vp]),
    "y2_bordered_bytes": (_sz, [_i, _i, _i, _i, _i, _psz]),
    "y2_link": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "y2_join_backward": (_i, [_i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "y2_pack_group_table": (_i, [_vp, _i, _vp, _sz, _pi, _pi]),
    "y2_pack_group_run": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "y2_join_backward_s2": (_i, [_i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "y2_subsample_bordered": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_scale": (_i, [_vp, _sz, _f, _vp]),
    "y2_class_argmax": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "y2_decode_anchors": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_nms": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp]),
    "y2_yolov2_loss_workspace_bytes": (_sz, [_i]),
    "y2_yolov2_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "y2_batch_norm_forward": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp]),
    "y2_batch_norm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp]),
    "y2_subsample": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "y2_maxpool3x3s2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_maxpool3x3s2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_fully_connected": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "y2_fully_connected_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_conv7x7s2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_conv7x7s2_t": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "y2_conv7x7s2_backward_filter": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "y2_conv7x7s2_backward_filter_t": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "y2_bias_relu": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "y2_bias_relu_backward": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "y2_dropout": (_i, [_vp, _vp, _sz, _f, _u64, _vp]),
    "y2_dropout_dev": (_i, [_vp, _vp, _sz, _f, _vp, _vp]),
    "y2_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _f, _f, _f, _f, _f, _vp]),
    "y2_momentum_step": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _vp]),
    "y2_grad_check": (_i, [_vp, _vp, _vp]),
    "y2_grad_check_more": (_i, [_vp, _vp, _vp]),
    "y2_grad_check_full": (_i, [_vp, _sz, _vp, _vp]),
    "y2_adam_step_guarded": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _vp]),
    "y2_momentum_step_guarded": (_i, [_vp, _vp, _vp, _sz, _vp, _f, _f, _f, _vp]),
    "y2_adam_step_packed": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _vp]),
    "y2_momentum_step_packed": (_i, [_vp, _vp, _vp, _f, _f, _f, _vp]),
    "y2_backward_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _vp]),
    "y2_backward_momentum": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _vp]),
    "y2_conv2d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "y2_conv2d": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "y2_conv2d_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "y2_crc32c": (C.c_uint32, [_vp, _sz, C.c_uint32]),
}


class Y2Error(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library and bind every symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Y2Error("HIP extension not built: %s is missing (run __graft_entry__.build())" % LIB_PATH)
    # torch first: the process must hold ONE HIP runtime.  torch brings its own libamdhip64; loaded after
    # /opt/rocm's copy (which this library would pull in) the two disagree and every HIP call of this
    # library fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("Y2_LIB_PATH") and not hasattr(lib, name):
            continue                     # an older build under A/B: its missing entry points simply cannot be called
        fn = getattr(lib, name)          # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise Y2Error("libyolo2_hip: error %d: %s" % (rc, load().y2_last_error().decode()))


def darknet19_spec(kind, output_filter=30):
    buf = (C.c_int * (4 * 32))()
    n = load().y2_darknet19_spec(kind, output_filter, buf, 32)
    if n < 0:
        check(n)
    return [tuple(buf[i * 4 + j] for j in range(4)) for i in range(n)]
