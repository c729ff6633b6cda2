"""Micro-benchmarks of the individual HIP kernels (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib

lib = _lib.load()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    s = torch.cuda.current_stream().cuda_stream
    for B in (1, 16, 44, 64):
        M = B * 577
        for (N, K, act, name) in [(2304, 768, 0, "qkv"), (768, 768, 0, "out"), (3072, 768, 1, "fc1"), (768, 3072, 0, "fc2")]:
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda") * K ** -0.5
            b = torch.randn(N, device="cuda")
            C = torch.empty(M, N, device="cuda")
            ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32(A.data_ptr(), W.data_ptr(), C.data_ptr(), b.data_ptr(), None, M, N, K, act, s)))
            print(f"gemm B={B:3d} {name:4s} M={M:6d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
        qkv = torch.randn(M, 2304, device="cuda")
        out = torch.empty(M, 768, device="cuda")
        ms = timeit(lambda: _lib.check(lib.tstar_attention_f32(qkv.data_ptr(), out.data_ptr(), B, 577, 12, 0, None, s)))
        fl = 4.0 * B * 12 * 577 * 577 * 64
        print(f"attn B={B:3d} {ms:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
        x = torch.randn(M, 768, device="cuda"); y = torch.empty_like(x); w = torch.randn(768, device="cuda")
        ms = timeit(lambda: _lib.check(lib.tstar_layernorm_f32(x.data_ptr(), y.data_ptr(), w.data_ptr(), w.data_ptr(), M, 768, s)))
        print(f"ln   B={B:3d} {ms:8.3f} ms  {2*M*768*4/ms/1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
