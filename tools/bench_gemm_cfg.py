"""Time every GEMM tile configuration (and the launcher's automatic choice) on the bench's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


names = {-1: "auto", 0: "128x128", 1: "64x128", 2: "64x64", 3: "hybrid", 4: "hyb-max"}
Bs = [int(b) for b in sys.argv[1:]] or [1, 4, 16, 32, 44, 52, 64, 96, 128]
for B in Bs:
    M = B * 577
    for (N, K, nm) in [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc1"), (768, 3072, "fc2")]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        nt, mt = N // 128, (M + 127) // 128
        big = ((mt * nt) // 512) * 512 // nt          # row tiles covered by whole waves of 128x128 tiles
        cfgs = [-1, 0, 1, 2, 3] + ([16 + big] if 0 < big < mt else [])
        # three interleaved rounds, median per configuration (clocks drift by a few % between back-to-back runs)
        samples = {c: [] for c in cfgs}
        for _ in range(3):
            for c in cfgs:
                ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32_cfg(A.data_ptr(), W.data_ptr(), C.data_ptr(), b.data_ptr(), None, M, N, K, 0, c, s)), iters=6)
                samples[c].append(2.0 * M * N * K / ms / 1e9)
        row = {c: sorted(v)[1] for c, v in samples.items()}
        best = max(v for c, v in row.items() if c != -1)
        print(f"B={B:3d} {nm:4s} M={M:6d} N={N:5d} K={K:5d} waves={mt * nt / 512:6.2f}  "
              + "  ".join(f"{names[min(c, 4)]}={v:6.1f}" for c, v in row.items()) + f"   auto/best={row[-1] / best:.3f}", flush=True)
