"""Dev tool (GPU): run ONE wgrad shape / variant (for rocprofv3 --pmc)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import _lib
lib = _lib.load()
lib.y2dev_bench_wgrad.restype = C.c_int
lib.y2dev_bench_wgrad.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_float)]
hw, ci, co, k = [int(v) for v in sys.argv[1].split(",")]
for c in sys.argv[2].split(","):
    v, sk = [int(x) for x in c.split(":")]
    ms = C.c_float()
    rc = lib.y2dev_bench_wgrad(64, hw, hw, ci, co, k, v, sk, 2, C.byref(ms))
    print(v, sk, rc, ms.value)
