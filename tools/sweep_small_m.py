"""Sub-wave GEMM launches (M = B x 577 for small B, the shapes of the reference-default 4x4 grid forwards): every pure tile
shape and every hybrid split (n row tiles of 128 rows, the rest 64 x 128) against the launcher's automatic choice.
Data behind pick_cfg's sub-wave branch (csrc/gemm_f32.hip).   python tools/sweep_small_m.py [B ...]"""
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


Bs = [int(b) for b in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 8, 10, 12, 16]
worst = 1.0
for B in Bs:
    for (rows, N, K, nm) in [(577, 2304, 768, "qkv"), (577, 768, 768, "out"), (577, 3072, 768, "fc1"), (577, 768, 3072, "fc2"),
                             (576, 768, 3072, "patch"), (576, 512, 768, "cls"), (576, 768, 768, "box")]:
        M = B * rows
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        nt, mt = N // 128, (M + 127) // 128

        def tf(cfg):
            v = []
            for _ in range(3):
                ms = timeit(lambda: _lib.check(lib.tstar_gemm_f32_cfg(A.data_ptr(), W.data_ptr(), C.data_ptr(), b.data_ptr(), None, M, N, K, 0, cfg, s)))
                v.append(2.0 * M * N * K / ms / 1e9)
            return sorted(v)[1]

        res = {"auto": tf(-1), "128": tf(0), "64n": tf(1), "64": tf(2)}
        hy = {n: tf(16 + n) for n in sorted({max(1, mt // 4), max(1, mt // 3), max(1, mt // 2), max(1, (2 * mt) // 3), max(1, (3 * mt) // 4)}) if 0 < n < mt}
        best = max(max(v for k, v in res.items() if k != "auto"), max(hy.values(), default=0.0))
        worst = min(worst, res["auto"] / best)
        print(f"B={B:3d} {nm:5s} M={M:6d} N={N:5d} K={K:5d} b128={mt * nt:5d} " + " ".join(f"{k}={v:6.1f}" for k, v in res.items())
              + "  hyb: " + " ".join(f"{n}/{mt}={v:.1f}" for n, v in hy.items()) + f"  auto/best={res['auto'] / best:.3f}", flush=True)
print(f"worst auto/best = {worst:.3f}")
