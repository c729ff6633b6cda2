"""What the vendor library sustains on this box for a PLAIN bf16 GEMM of the bench's shapes (torch.matmul -> hipBLASLt / rocBLAS), with random
and with zero operands: the practical ceiling the f32x3 GEMM's EXECUTED rate (6 x algorithmic, profiles/r05_x3_lab.log) can be read against.
Not part of the product path (the library calls no vendor GEMM); evidence for DESIGN.md 4.1 only.

    python tools/vendor_bf16_gemm_probe.py
"""
import torch

dev = torch.device("cuda:0")
M = 147712                                             # 256 images x 577 tokens
shapes = [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]


def rate(a, w, iters=20):
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * a.shape[0] * w.shape[0] * a.shape[1] / ms / 1e9, ms


for name, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
    r, ms = rate(a, w)
    z, zms = rate(torch.zeros_like(a), torch.zeros_like(w))
    print(f"{name:4s} M={M} N={N:5d} K={K:5d}: random operands {r:7.1f} TFLOP/s ({ms:.3f} ms)   zero operands {z:7.1f} TFLOP/s ({zms:.3f} ms)   "
          f"= {r / 2500:.2f} / {z / 2500:.2f} of the dense bf16 peak")
