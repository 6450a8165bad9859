"""Dev tool (GPU, dev library): per-phase s_memtime totals of the 128-cout register-filter kernel, workgroup 0."""
import ctypes as C, os, sys
os.environ["Y2_DEV_LIB"] = "1"
os.environ.setdefault("Y2DEV_BENCH_STATS", "1")
os.environ.setdefault("Y2DEV_BENCH_ROT", "3")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import _lib
lib = _lib.load()
lib.y2dev_bench_conv.restype = C.c_int
lib.y2dev_bench_conv.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_float)]
ms = C.c_float()
rc = lib.y2dev_bench_conv(64, 104, 104, 64, 128, 3, 100, 10, C.byref(ms))
print("rc", rc, "us", ms.value * 1e3)
buf = (C.c_ulonglong * 64)()
lib.y2dev_rf_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
print("stamps rc", lib.y2dev_rf_stamps(buf))
names = ["wait groups", "barrier A", "record+K loop", "barrier B", "stage+patch+stats", "barrier C", "sweep", "-"]
for w in range(8):
    row = [buf[w * 8 + k] for k in range(8)]
    tot = sum(row) or 1
    print("wave", w, " ".join("%s=%d (%.0f%%)" % (names[k], row[k], 100.0 * row[k] / tot) for k in range(7)), "total", tot)
