"""Dev tool (GPU): A/B the weight-gradient kernels on the Darknet-19 layer shapes (f16).
variant 0 = one tap per block (wgrad.hip), 1 = nine taps per block (wgrad9.hip); sk = split-K (0 = auto)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import _lib
lib = _lib.load()
lib.y2dev_bench_wgrad.restype = C.c_int
lib.y2dev_bench_wgrad.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_float)]
N = int(os.environ.get("BATCH", "64"))
shapes = [("L1 32->64 208^2", 208, 32, 64, 3), ("L2 64->128 104^2", 104, 64, 128, 3), ("L5 128->256 52^2", 52, 128, 256, 3),
          ("L8 256->512 26^2", 26, 256, 512, 3), ("L13 512->1024 13^2", 13, 512, 1024, 3),
          ("head 1024->1024 13^2", 13, 1024, 1024, 3)]
if os.environ.get("SHAPES"):   # "hw,ci,co,k;hw,ci,co,k"
    shapes = [("%s^2 %s->%s k%s" % tuple(q.split(",")[i] for i in (0, 1, 2, 3)),) + tuple(int(v) for v in q.split(","))
              for q in os.environ["SHAPES"].split(";")]
cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0:0", "1:0"])]
for name, hw, ci, co, k in shapes:
    fl = 2.0 * N * hw * hw * k * k * ci * co
    out = []
    for (v, sk) in cases:
        best = 1e9
        for r in range(2):
            ms = C.c_float()
            rc = lib.y2dev_bench_wgrad(N, hw, hw, ci, co, k, v, sk, 5, C.byref(ms))
            best = min(best, ms.value if rc == 0 else float("nan"))
        out.append("v%d/sk%d:%7.1fus %4.0fTF" % (v, sk, best * 1e3, fl / (best * 1e-3) / 1e12))
    print(name.ljust(22), " ".join(out))
