"""Library-GEMM ceiling for the 1x1 layers' shapes (M = bordered pixels, K = Cin, N = Cout), f16 -> f16:
what a tuned plain GEMM reaches on this box, to place the hand-written 1x1 kernels against.  Measurement only."""
import torch, time
shapes = [("L6 52^2 256->128", 64 * 53 * 53, 256, 128), ("L9 26^2 512->256", 64 * 27 * 27, 512, 256),
          ("L14 13^2 1024->512", 64 * 14 * 14, 1024, 512), ("L21 13^2 1024->30(32)", 64 * 14 * 14, 1024, 32),
          ("dg6 52^2 128->256", 64 * 53 * 53, 128, 256), ("dg14 13^2 512->1024", 64 * 14 * 14, 512, 1024)]
dev = "cuda:0"
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    b = torch.randn(K, N, device=dev, dtype=torch.float16)
    for _ in range(5): c = a @ b
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): c = a @ b
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print(f"{name:26s} M={M} K={K} N={N}: {us:7.1f} us  {2.0*M*K*N/us/1e6:7.1f} TF  {(M*K+M*N)*2/us/1e3:6.0f} GB/s")
# wgrad form: dW[K x N] = X^T[K x M] dY[M x N]
for name, M, K, N in shapes[:3]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    d = torch.randn(M, N, device=dev, dtype=torch.float16)
    for _ in range(5): c = a.t() @ d
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): c = a.t() @ d
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print(f"wgrad {name:20s}: {us:7.1f} us  {2.0*M*K*N/us/1e6:7.1f} TF")
