"""Dev tool (GPU): A/B the implicit-GEMM conv variants on the Darknet-19 layer shapes (f16)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import _lib
lib = _lib.load()
lib.y2dev_bench_conv.restype = C.c_int
lib.y2dev_bench_conv.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_float)]
N = int(os.environ.get("BATCH", "64"))
shapes = [("L2 64->128 104^2", 104, 64, 128, 3), ("L5 128->256 52^2", 52, 128, 256, 3), ("L8 256->512 26^2", 26, 256, 512, 3),
          ("L13 512->1024 13^2", 13, 512, 1024, 3), ("head 1024->1024 13^2", 13, 1024, 1024, 3), ("L14 1x1 1024->512", 13, 1024, 512, 1)]
if os.environ.get("SHAPES"):   # "hw,ci,co,k;hw,ci,co,k"
    shapes = [("%s^2 %s->%s k%s" % tuple(q.split(",")[i] for i in (0, 1, 2, 3)),) + tuple(int(v) for v in q.split(","))
              for q in os.environ["SHAPES"].split(";")]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
rounds = 3
for name, hw, ci, co, k in shapes:
    fl = 2.0 * N * hw * hw * k * k * ci * co
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            ms = C.c_float()
            rc = lib.y2dev_bench_conv(N, hw, hw, ci, co, k, v, 10, C.byref(ms))
            res[v].append(ms.value if rc == 0 else float("nan"))
    print(name.ljust(24), " ".join("v%d:%6.1fus %4.0fTF" % (v, min(res[v]) * 1e3, fl / (min(res[v]) * 1e-3) / 1e12) for v in variants))
