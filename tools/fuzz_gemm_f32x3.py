import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, random
from tstar_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
rnd = random.Random(7)
bad = 0
for it in range(160):
    M = rnd.choice([1, 2, 31, 33, 63, 64, 65, 127, 128, 129, 577, 600, 1154, rnd.randint(1, 4000), rnd.randint(4000, 70000)])
    N = 128 * rnd.choice([1, 2, 3, 4, 6, 8, 18, 24])
    K = 32 * rnd.choice([1, 2, 3, 8, 24, 25, 96])
    act = rnd.choice([0, 0, 1, 2]); res = rnd.random() < 0.4 and act == 0
    g = torch.Generator().manual_seed(it)
    A = torch.randn(M, K, generator=g) * (10 ** rnd.uniform(-3, 3)); W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g); R = torch.randn(M, N, generator=g) if res else None
    ref = A.double() @ W.double().t() + b.double()
    mag = A.abs().double() @ W.abs().double().t() + b.abs().double()
    if res: ref += R.double(); mag += R.abs().double()
    dA, dW, db = A.cuda(), W.cuda(), b.cuda(); dR = R.cuda() if res else None
    outs = {}
    for cfg in (5, -1, 0, 1, 2, 3, 4):
        dC = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.tstar_gemm_f32x3(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), dR.data_ptr() if res else None, M, N, K, 0, cfg, st))
        outs[cfg] = dC
        if not torch.equal(dC, outs[5]): bad += 1; print("MISMATCH across cfg", M, N, K, cfg)
    out = outs[5].cpu().double()
    e = ((out - ref).abs() / mag).max().item()
    if not torch.isfinite(out).all() or e > 2 ** -19: bad += 1; print("ERR", M, N, K, e)
    if act:
        dC = torch.empty((M, N), device="cuda")
        _lib.check(lib.tstar_gemm_f32x3(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), None, M, N, K, act, -1, st))
        r32 = ref.float(); want = r32 * torch.sigmoid(1.702 * r32) if act == 1 else torch.nn.functional.gelu(r32)
        if (dC.cpu() - want).abs().max().item() > 3e-5 * max(1.0, want.abs().max().item()): bad += 1; print("ACT", M, N, K, act)
print("fuzz done, bad =", bad)
