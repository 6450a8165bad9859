"""time the fused dW + guarded Adam update of the ResNet swap's fc1 (100352 x 4096, batch 32) (scripts only)"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import engine as E, _lib
lib = _lib.load()
M, K, N = 32, 100352, 4096
x = torch.randn(M, K, device="cuda"); dz = torch.randn(M, N, device="cuda") * 1e-3
w = torch.randn(K, N, device="cuda") * 0.01; m = torch.zeros_like(w); v = torch.zeros_like(w)
ctrl = torch.zeros(8, dtype=torch.int32, device="cuda")
ctrl[1] = 1
ctrl.view(torch.float32)[4] = 5e-4
def f():
    E.check(lib.y2_fc_adam_apply_guarded(E._ptr(x), E._ptr(dz), E._ptr(w), E._ptr(m), E._ptr(v), M, K, N, _lib.DTYPES["f16"],
                                         E._ptr(ctrl), 0.9, 0.999, 1e-8, 1.0, E._stream()))
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print("fc1 dW + Adam: %.1f us  = %.2f TB/s over 6 x 1.64 GB" % (us, 6 * K * N * 4 / us / 1e6))
