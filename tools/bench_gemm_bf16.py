"""bf16-weight GEMM tiles on the bench's shapes: exact three-term split vs two-term, narrow tiles vs the 128x256 tile.

    python tools/bench_gemm_bf16.py [B ...]     # TFLOP/s algorithmic (2MNK / time); executed = x3 (three-term), x2 (two-term), x6 (f32x3)
"""
import os
import sys
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


Bs = [int(b) for b in sys.argv[1:]] or [16, 64, 225, 256]
variants = [("3-term auto", 3, -1), ("2-term narrow", 2, 5), ("2-term auto", 2, -1), ("2-term all-wide", 2, 4)]
for B in Bs:
    M = B * 577
    for (N, K, nm, res) in [(2304, 768, "qkv", False), (768, 768, "out", True), (3072, 768, "fc1", False), (768, 3072, "fc2", True)]:
        A = torch.randn(M, K, device="cuda")
        Wb = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        C = torch.zeros(M, N, device="cuda")
        row = {}
        # round 4: the f32x3 mode (f32 weights as three exact bf16 planes, six products) and the native f32 tile on the same shape
        Wf = torch.randn(N, K, device="cuda") * K ** -0.5
        Wp = torch.empty(N * K * 6, dtype=torch.uint8, device="cuda")
        _lib.check(lib.tstar_pack_f32x3(Wf.data_ptr(), Wp.data_ptr(), N, K, s))
        for _ in range(2):
            ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32x3_pre(A.data_ptr(), Wp.data_ptr(), C.data_ptr(), b.data_ptr(),
                                                                    C.data_ptr() if res else None, M, N, K, 0, -1, s)))
            row["f32x3"] = max(row.get("f32x3", 0.0), 2.0 * M * N * K / ms / 1e9)
            ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32_cfg(A.data_ptr(), Wf.data_ptr(), C.data_ptr(), b.data_ptr(),
                                                                  C.data_ptr() if res else None, M, N, K, 0, -1, s)))
            row["native f32"] = max(row.get("native f32", 0.0), 2.0 * M * N * K / ms / 1e9)
        for _ in range(2):
            for name, terms, cfg in variants:
                ms = timeit(lambda: _lib.check(lib.tstar_gemm_bf16w_pre(A.data_ptr(), Wb.data_ptr(), C.data_ptr(), b.data_ptr(),
                                                                        C.data_ptr() if res else None, M, N, K, 0, terms, cfg, s)))
                row[name] = max(row.get(name, 0.0), 2.0 * M * N * K / ms / 1e9)
        print(f"B={B:3d} {nm:4s} M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(f"{k}={v:6.1f}" for k, v in row.items())
              + f"   executed (2-term auto x2) = {2 * row['2-term auto']:.0f} TFLOP/s", flush=True)
