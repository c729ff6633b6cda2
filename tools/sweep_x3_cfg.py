"""The f32x3 GEMM tile (gemm_tile_x3, round 5) on the shapes of a forward at batch B: every pure tile shape, the hybrid launches and
the wide (128x256) launch against the launcher's automatic choice.  Data behind launch_mode<4>'s policy (csrc/gemm_f32.hip).
    python tools/sweep_x3_cfg.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tstar_amd import _lib  # noqa: E402

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


names = {-1: "auto", 0: "128x128", 1: "64x128", 2: "64x64", 3: "hybrid", 4: "wide-all", 5: "no-wide"}
Bs = [int(b) for b in sys.argv[1:]] or [1, 2, 4, 8, 10, 16, 24, 32, 45, 64, 96, 128, 180, 256]
worst = 1.0
for B in Bs:
    M = B * 577
    for (N, K, nm) in [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc1"), (768, 3072, "fc2")]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        Wp = torch.empty(N * K * 6, dtype=torch.uint8, device="cuda")
        _lib.check(lib.tstar_pack_f32x3(W.data_ptr(), Wp.data_ptr(), N, K, s))
        cfgs = [-1, 0, 1, 2, 3, 4, 5]
        samples = {c: [] for c in cfgs}
        for _ in range(3):
            for c in cfgs:
                if c == 4 and M < 128:
                    continue
                ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32x3_pre(A.data_ptr(), Wp.data_ptr(), C.data_ptr(), b.data_ptr(), None, M, N, K, 0, c, s)), iters=6)
                samples[c].append(2.0 * M * N * K / ms / 1e9)
        row = {c: sorted(v)[1] for c, v in samples.items() if v}
        best = max(v for c, v in row.items() if c != -1)
        worst = min(worst, row[-1] / best)
        print(f"B={B:3d} {nm:4s} M={M:6d} N={N:5d} K={K:5d} b128={((M + 127) // 128) * (N // 128):5d}  "
              + "  ".join(f"{names[c]}={v:6.1f}" for c, v in row.items()) + f"   auto/best={row[-1] / best:.3f}", flush=True)
print(f"worst auto/best = {worst:.3f}")
