"""Dev tool (GPU): run ONE conv shape / variant a few times (for rocprofv3 --pmc)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import _lib
lib = _lib.load()
lib.y2dev_bench_conv.restype = C.c_int
lib.y2dev_bench_conv.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_float)]
hw, ci, co, k = [int(v) for v in sys.argv[1].split(",")]
for v in [int(x) for x in sys.argv[2].split(",")]:
    ms = C.c_float()
    rc = lib.y2dev_bench_conv(64, hw, hw, ci, co, k, v, 3, C.byref(ms))
    print(v, rc, ms.value)
