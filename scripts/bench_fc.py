"""time the fc1 products of the ResNet swap's head (100352 x 4096, batch 32): forward, dx (scripts only)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo2_amd import engine as E
M, K, N = 32, 100352, 4096
x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.01; dy = torch.randn(M, N, device="cuda")
b = torch.zeros(N, device="cuda")
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("fwd us %.1f" % t(lambda: E.fully_connected(x, w, b, True, "f16")))
print("dx  us %.1f" % t(lambda: E.fully_connected_backward(x, w, dy, "f16", want_dw=False)))
