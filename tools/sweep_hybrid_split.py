"""Sweep the hybrid launch's split (rows covered by 128x128 tiles) per GEMM shape: data for pick_cfg's model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


Bs = [int(b) for b in sys.argv[1:]] or [4, 8, 16, 24, 32, 44, 52, 64, 96, 128]
for B in Bs:
    M = B * 577
    for (N, K, nm) in [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc1"), (768, 3072, "fc2")]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        nt = N // 128
        mt = (M + 127) // 128

        def tf(cfg):
            ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32_cfg(A.data_ptr(), W.data_ptr(), C.data_ptr(), b.data_ptr(), None, M, N, K, 0, cfg, s)))
            return 2.0 * M * N * K / ms / 1e9

        base = {n: tf(c) for n, c in (("auto", -1), ("128", 0), ("64n", 1), ("64", 2))}
        # candidate splits: whole waves of 512 big tiles (k = 1..), in row tiles
        cands = sorted({min(mt, (w * 512) // nt) for w in range(1, (mt * nt) // 512 + 1)} | {mt // 2})
        hy = {c: tf(16 + c) for c in cands if 0 < c < mt}
        best_c = max(hy, key=hy.get) if hy else None
        print(f"B={B:3d} {nm:4s} M={M:6d} N={N:5d} K={K:5d} tiles={mt * nt:6d} " + " ".join(f"{k}={v:6.1f}" for k, v in base.items())
              + "  hyb: " + " ".join(f"{c}({c * nt / 512:.2f}w)={v:.1f}" for c, v in hy.items()) + f"  best_hyb={best_c}", flush=True)
