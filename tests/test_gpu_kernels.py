"""Kernel-level parity: each HIP kernel (through the C ABI) vs a plain torch fp32 CPU
statement of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from tstar_amd import _lib
    return _lib.load()


def _check(rc):
    from tstar_amd import _lib
    _lib.check(rc)


@pytest.mark.parametrize("M,N,K", [(577, 768, 768), (128, 128, 32), (1, 128, 64), (1154, 2304, 768),
                                   (1000, 768, 3072), (64, 512, 512)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_bias_act(lib, M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N + K + act)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    ref = F.linear(A, W, b)
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    elif act == 2:
        ref = F.gelu(ref)
    dA, dW, db = A.cuda(), W.cuda(), b.cuda()
    dC = torch.full((M, N), float("nan"), device="cuda")
    _check(lib.tstar_gemm_f32(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), None, M, N, K, act,
                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    err = (dC.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(577, 768, 3072), (130, 256, 64), (1154, 512, 768), (25388, 768, 768)])
def test_gemm_tile_configs(lib, cfg, M, N, K):
    """All three block-tile shapes compute the same product (the launcher picks per shape)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    ref = F.linear(A, W, b)
    dA, dW, db = A.cuda(), W.cuda(), b.cuda()
    dC = torch.full((M, N), float("nan"), device="cuda")
    _check(lib.tstar_gemm_f32_cfg(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), None, M, N, K, 0, cfg,
                                  torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (dC.cpu() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K,act,res", [(73856, 768, 768, 0, True), (30004, 2304, 768, 1, False), (2308, 3072, 768, 0, False),
                                           (1155, 768, 3072, 2, False)])
def test_gemm_result_independent_of_tiling(lib, M, N, K, act, res):
    """Full-size property: every tile shape, the hybrid launch at any split and the launcher's own choice give
    BIT-IDENTICAL outputs (each element is one MFMA accumulation chain over K in the same order), including the
    guarded last row panel, the fused activation and the in-place residual."""
    g = torch.Generator().manual_seed(M + N)
    dA = torch.randn(M, K, generator=g).cuda()
    dW = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    db = torch.randn(N, generator=g).cuda()
    dR = torch.randn(M, N, generator=g).cuda() if res else None
    mt = (M + 127) // 128
    outs = {}
    for cfg in (0, 1, 2, 3, -1, 16 + 1, 16 + mt // 3, 16 + mt):
        dC = dR.clone() if res else torch.full((M, N), float("nan"), device="cuda")
        _check(lib.tstar_gemm_f32_cfg(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), dC.data_ptr() if res else None,
                                      M, N, K, act, cfg, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs[cfg] = dC
    assert torch.isfinite(outs[0]).all()
    for cfg, o in outs.items():
        assert torch.equal(o, outs[0]), cfg
    with pytest.raises(Exception):
        _check(lib.tstar_gemm_f32_cfg(dA.data_ptr(), dW.data_ptr(), outs[0].data_ptr(), db.data_ptr(), None, M, N, K, act, 7,
                                      torch.cuda.current_stream().cuda_stream))


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K,act", [(577, 768, 3072, 0), (130, 256, 64, 1), (1154, 512, 768, 2), (25388, 768, 768, 0)])
def test_gemm_bf16_weights_exact_split(lib, cfg, M, N, K, act):
    """bf16-weight GEMM: f32 activations split exactly into 3 bf16 terms -> the f32-accumulated product
    with the bf16-ROUNDED weights (compared against a float64 product of the same operands)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    A[::7] *= 1e-3                                             # wide dynamic range across rows
    A[:, ::5] *= 37.0
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    Wr = W.to(torch.bfloat16).to(torch.float64)
    ref = A.to(torch.float64) @ Wr.t() + b.to(torch.float64)
    mag = A.abs().to(torch.float64) @ Wr.abs().t() + b.abs().to(torch.float64)
    dA, dW, db = A.cuda(), W.cuda(), b.cuda()
    dC = torch.full((M, N), float("nan"), device="cuda")
    _check(lib.tstar_gemm_bf16w(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), None, M, N, K, 0, cfg,
                                torch.cuda.current_stream().cuda_stream))
    out = dC.cpu().to(torch.float64)
    assert torch.isfinite(out).all()
    err = ((out - ref).abs() / mag).max().item()
    assert err < 1e-6                                              # f32-roundoff class (grows slowly with K); bf16 activations would be ~4e-3
    # the native f32 MFMA kernel on the same bf16-valued weights sits in the same error class
    dWr = Wr.to(torch.float32).cuda()
    dC32 = torch.empty((M, N), device="cuda")
    _check(lib.tstar_gemm_f32(dA.data_ptr(), dWr.data_ptr(), dC32.data_ptr(), db.data_ptr(), None, M, N, K, 0,
                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    err32 = ((dC32.cpu().to(torch.float64) - ref).abs() / mag).max().item()
    assert err < 3 * err32 + 1e-7, (err, err32)
    if act:
        dC2 = torch.empty((M, N), device="cuda")
        _check(lib.tstar_gemm_bf16w(dA.data_ptr(), dW.data_ptr(), dC2.data_ptr(), db.data_ptr(), None, M, N, K, act, cfg,
                                    torch.cuda.current_stream().cuda_stream))
        r32 = ref.to(torch.float32)
        want = r32 * torch.sigmoid(1.702 * r32) if act == 1 else F.gelu(r32)
        assert (dC2.cpu() - want).abs().max().item() < 3e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,N,K,act,res", [(577, 768, 3072, 0, False), (130, 256, 64, 1, False), (1154, 512, 768, 2, False),
                                           (25388, 768, 768, 0, True), (29427, 2304, 768, 0, False), (30004, 768, 3072, 0, True),
                                           (23657, 3072, 768, 1, False)])
def test_gemm_bf16_weights_two_term(lib, M, N, K, act, res):
    """bf16-weight GEMM with two-term activations (weights_mode "bf16", BASELINE config 5): a = a_hi + a_lo, both
    round-to-nearest bf16 (|a - a_hi - a_lo| <= 2^-17 |a|), C += a_lo*w + a_hi*w with exact products and f32 accumulation.
    Hard bound |C - ref| <= 2^-17 * sum|a w| (+ f32 accumulation); with random signs the relative rms is a few 1e-6.
    Every tile choice -- tile_cfg 0..3 (128x128 / 64x128 / 64x64 / hybrid), 4 = the 128x256 tile on every full 128-row panel
    (64x128 tiles on the ragged rest), 5 = wide tile forbidden, -1 = the launcher's choice (whole waves of wide tiles from 512
    of them on: the last three shapes) -- must agree with the float64 product of the same operands, and with each other bit
    for bit (same K order, same products).  The float64 reference is formed once per shape.
    Round 6: 6 = the wide tile with its weight fragments streamed global -> VGPR from the fragment-packed plane (gemm_tile_w2v, what the
    launcher picks on its own for N = 768 from one wave of wide tiles on: shapes four and six, with their residual epilogues) -- the
    same bits again, on every shape whose N is a multiple of 256 (K = 64: two K tiles; ragged M; quick-GELU / GELU epilogues)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    A[::7] *= 1e-3                                             # wide dynamic range across rows
    A[:, ::5] *= 37.0
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if res else None
    Wr = W.to(torch.bfloat16).to(torch.float64)
    ref = A.to(torch.float64) @ Wr.t() + b.to(torch.float64)
    mag = A.abs().to(torch.float64) @ Wr.abs().t() + b.abs().to(torch.float64)
    if res:
        ref = ref + R.to(torch.float64)
        mag = mag + R.abs().to(torch.float64)
    dA, dW, db = A.cuda(), W.cuda(), b.cuda()
    dR = R.cuda() if res else None
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for cfg in (5, -1, 0, 1, 2, 3, 4) + ((6,) if N % 256 == 0 and M >= 128 else ()):
        dC = torch.full((M, N), float("nan"), device="cuda")
        _check(lib.tstar_gemm_bf16w2(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), dR.data_ptr() if res else None, M, N, K, 0,
                                     cfg, st))
        outs[cfg] = dC
        assert torch.equal(dC, outs[5]), cfg                   # every tile shape computes the same bits as the narrow tiles
    out = outs[5].cpu().to(torch.float64)
    assert torch.isfinite(out).all()
    err = ((out - ref).abs() / mag).max().item()
    assert err < 2.0 ** -17 + 1e-6, err
    rms = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rms < 8e-6, rms
    if act:
        r32 = ref.to(torch.float32)
        want = r32 * torch.sigmoid(1.702 * r32) if act == 1 else F.gelu(r32)
        first = None
        for cfg in (-1, 4) + ((6,) if N % 256 == 0 and M >= 128 else ()):
            dC2 = torch.empty((M, N), device="cuda")
            _check(lib.tstar_gemm_bf16w2(dA.data_ptr(), dW.data_ptr(), dC2.data_ptr(), db.data_ptr(), None, M, N, K, act, cfg, st))
            assert (dC2.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
            first = dC2 if first is None else first
            assert torch.equal(dC2, first), cfg


@pytest.mark.parametrize("M,N,K,act,res", [(577, 768, 3072, 0, False), (130, 256, 64, 1, False), (1154, 512, 768, 2, False),
                                           (25388, 768, 768, 0, True), (29427, 2304, 768, 0, False), (30004, 768, 3072, 0, True),
                                           (23657, 3072, 768, 1, False), (1, 128, 32, 0, False),
                                           (577, 128, 96, 0, False), (70, 256, 160, 0, True)])   # odd K-tile counts: the peeled tile of the 64x64 ring
def test_gemm_f32x3(lib, M, N, K, act, res):
    """f32x3 GEMM (weights_mode 4, round 4): both f32 operands as THREE exact round-to-nearest bf16 terms (all 24 significand
    bits), the six partial products with ka + kw <= 2 on the bf16 matrix pipe, f32 accumulation.  The gate the round-3 review
    set for a bf16-pipe mode to count as f32: its error against float64 must be NO LARGER than the native f32 MFMA tile's on
    the same inputs -- asserted here on every shape (rms and worst element over sum|a w|), both printed.  Inputs span a wide
    dynamic range (rows scaled by 1e-3, columns by 37).  Every tile choice (tile_cfg 0..3, 4 = 128x256 on every full panel,
    5 = wide tile forbidden, -1 = the launcher's) must give the same bits: same K order, same products."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    A[::7] *= 1e-3
    A[:, ::5] *= 37.0
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if res else None
    ref = A.to(torch.float64) @ W.to(torch.float64).t() + b.to(torch.float64)
    mag = A.abs().to(torch.float64) @ W.abs().to(torch.float64).t() + b.abs().to(torch.float64)
    if res:
        ref = ref + R.to(torch.float64)
        mag = mag + R.abs().to(torch.float64)
    dA, dW, db = A.cuda(), W.cuda(), b.cuda()
    dR = R.cuda() if res else None
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for cfg in (5, -1, 0, 1, 2, 3, 4):
        dC = torch.full((M, N), float("nan"), device="cuda")
        _check(lib.tstar_gemm_f32x3(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), db.data_ptr(), dR.data_ptr() if res else None, M, N, K, 0,
                                    cfg, st))
        outs[cfg] = dC
        assert torch.equal(dC, outs[5]), cfg
    out = outs[5].cpu().to(torch.float64)
    assert torch.isfinite(out).all()
    dC32 = torch.empty((M, N), device="cuda")
    _check(lib.tstar_gemm_f32(dA.data_ptr(), dW.data_ptr(), dC32.data_ptr(), db.data_ptr(), dR.data_ptr() if res else None, M, N, K, 0, st))
    torch.cuda.synchronize()
    o32 = dC32.cpu().to(torch.float64)
    rms = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    rms32 = ((o32 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    worst = ((out - ref).abs() / mag).max().item()
    worst32 = ((o32 - ref).abs() / mag).max().item()
    print(f"f32x3 M={M} N={N} K={K}: rms err / rms C {rms:.3e} (native f32 MFMA tile {rms32:.3e}); max err / sum|aw| {worst:.3e} ({worst32:.3e})")
    assert rms <= rms32 * 1.02 + 1e-9, (rms, rms32)            # the gate; 2 % = sampling noise of an rms over M * N elements (K = 32: both ~1e-8)
    assert worst < 2.0 ** -19 and worst < 2 * worst32 + 2.0 ** -23, (worst, worst32)   # worst element: the same class as the native tile's (a max is a noisy statistic)
    if act:
        r32 = ref.to(torch.float32)
        want = r32 * torch.sigmoid(1.702 * r32) if act == 1 else F.gelu(r32)
        for cfg in (-1, 4):
            dC2 = torch.empty((M, N), device="cuda")
            _check(lib.tstar_gemm_f32x3(dA.data_ptr(), dW.data_ptr(), dC2.data_ptr(), db.data_ptr(), None, M, N, K, act, cfg, st))
            assert (dC2.cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_f32x3_operand_split_is_exact(lib):
    """a = a0 + a1 + a2 with a0 = bf16_rn(a), a1 = bf16_rn(a - a0), a2 = a - a0 - a1: the third remainder is itself a bfloat16
    for every finite float32 whose terms stay normal -- checked through the kernel: with W = identity rows (1.0 is exact in
    one plane) the six-product GEMM must return A bit for bit (only a0 w0 + a1 w0 + a2 w0 are non-zero), for random
    mantissas over 60 binades, signed zeros of a term (exact bf16 inputs) and the ties of round-to-nearest-even."""
    K = N = 128
    g = torch.Generator().manual_seed(5)
    bits = torch.randint(0, 2 ** 23, (4096, K), generator=g, dtype=torch.int32)
    expo = torch.randint(97, 157, (4096, K), generator=g, dtype=torch.int32)           # 2^-30 .. 2^29
    sign = torch.randint(0, 2, (4096, K), generator=g, dtype=torch.int32)
    A = ((sign << 31) | (expo << 23) | bits).view(torch.float32).clone()
    A[0] = torch.randn(K, generator=g).to(torch.bfloat16).to(torch.float32)           # a1 = a2 = 0
    A[1] = (torch.randint(0, 2 ** 7, (K,), generator=g, dtype=torch.int32) << 16 | 0x3F808000).view(torch.float32)       # ties of the first rounding: low half exactly 0x8000
    A[2] = 0.0
    W = torch.eye(N, K)
    dA, dW = A.cuda(), W.cuda()
    dC = torch.full((A.shape[0], N), float("nan"), device="cuda")
    _check(lib.tstar_gemm_f32x3(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), None, None, A.shape[0], N, K, 0, -1,
                                torch.cuda.current_stream().cuda_stream))
    assert torch.equal(dC.cpu().view(torch.int32), A.view(torch.int32))          # no -0 among the inputs: bitwise
    # and with the roles swapped: A = identity block, W random -> C = W^T rows bit for bit (the weight planes are exact too)
    Wr = A[:N].clone()
    I = torch.eye(128, K)
    dI, dWr = I.cuda(), Wr.cuda()
    dC = torch.full((128, N), float("nan"), device="cuda")
    _check(lib.tstar_gemm_f32x3(dI.data_ptr(), dWr.data_ptr(), dC.data_ptr(), None, None, 128, N, K, 0, -1,
                                torch.cuda.current_stream().cuda_stream))
    assert torch.equal(dC.cpu().view(torch.int32), Wr.t().contiguous().view(torch.int32))


def test_gemm_asymmetric_identity(lib):
    """A = I against an asymmetric W: catches row/col swaps in the C/D layout."""
    N = K = 128
    A = torch.eye(K)
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 1013) / 7.0
    dC = torch.empty((K, N), device="cuda")
    dA, dW = A.cuda(), W.cuda()          # keep the device tensors alive across the call
    _check(lib.tstar_gemm_f32(dA.data_ptr(), dW.data_ptr(), dC.data_ptr(), None, None, K, N, K, 0,
                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(dC.cpu(), W.t().contiguous())


def test_gemm_residual_inplace(lib):
    M, N, K = 300, 768, 3072
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    X = torch.randn(M, N, generator=g)
    ref = X + F.linear(A, W, b)
    dX, dA, dW, db = X.cuda(), A.cuda(), W.cuda(), b.cuda()
    _check(lib.tstar_gemm_f32(dA.data_ptr(), dW.data_ptr(), dX.data_ptr(), db.data_ptr(),
                              dX.data_ptr(), M, N, K, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (dX.cpu() - ref).abs().max().item() < 5e-5


def test_gemm_rejects_bad_shapes(lib):
    from tstar_amd import _lib
    d = torch.zeros(256 * 256, device="cuda")
    rc = lib.tstar_gemm_f32(d.data_ptr(), d.data_ptr(), d.data_ptr(), None, None, 16, 100, 32, 0, None)
    assert rc == 1 and b"multiple of 128" in lib.tstar_last_error()
    with pytest.raises(_lib.TStarHipError):
        _lib.check(rc)


@pytest.mark.parametrize("rows,D", [(577, 768), (5, 768), (64, 512), (1, 512)])
def test_layernorm(lib, rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    w = torch.randn(D, generator=g)
    b = torch.randn(D, generator=g)
    ref = F.layer_norm(x, (D,), w, b, 1e-5)
    dy = torch.empty((rows, D), device="cuda")
    dx, dw, db = x.cuda(), w.cuda(), b.cuda()
    _check(lib.tstar_layernorm_f32(dx.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                   rows, D, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (dy.cpu() - ref).abs().max().item() < 1e-5


def _attn_ref(qkv, B, T, heads, mask=None):
    D = heads * 64
    q, k, v = qkv.view(B, T, 3 * D).split(D, dim=-1)
    q = q.view(B, T, heads, 64).transpose(1, 2)
    k = k.view(B, T, heads, 64).transpose(1, 2)
    v = v.view(B, T, heads, 64).transpose(1, 2)
    att = torch.matmul(q, k.transpose(2, 3)) * 0.125
    if mask is not None:
        att = att + mask
    att = torch.softmax(att, dim=-1)
    return torch.matmul(att, v).transpose(1, 2).reshape(B * T, D)


@pytest.mark.parametrize("B,T,heads", [(1, 577, 12), (2, 577, 12), (3, 16, 8), (1, 33, 2), (1, 128, 1), (1, 1, 1), (2, 97, 3), (1, 600, 2),
                                       (1, 65, 2), (3, 193, 4), (2, 321, 1), (5, 577, 3)])
def test_attention_full(lib, B, T, heads):
    # T = 128 n + 65 (65, 193, 321, 577) takes the 16-query tail workgroup for the last 65 queries of every (image, head):
    # short workgroups after all full ones, four waves x 16 queries + the straggler query T-1 shared as VALU work
    g = torch.Generator().manual_seed(B * 1000 + T + heads)
    D = heads * 64
    qkv = torch.randn(B * T, 3 * D, generator=g)
    ref = _attn_ref(qkv, B, T, heads)
    out = torch.full((B * T, D), float("nan"), device="cuda")
    dqkv = qkv.cuda()
    _check(lib.tstar_attention_f32(dqkv.data_ptr(), out.data_ptr(), B, T, heads, 0, None,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("last_key_spike", [False, True])
def test_attention_spiked_scores(lib, last_key_spike):
    """Force large online-softmax rescales: one key dominates late in the sequence; with ``last_key_spike`` it is
    the straggler key 576 that the kernel folds in after its block loop."""
    B, T, heads = 1, 577, 2
    g = torch.Generator().manual_seed(11)
    D = heads * 64
    qkv = torch.randn(B * T, 3 * D, generator=g)
    qkv[500, D:2 * D] *= 40.0        # huge key late -> max jumps at key block 15
    qkv[3, D:2 * D] *= 25.0          # and an early big one
    if last_key_spike:
        qkv[576, D:2 * D] *= 60.0
    # the straggler QUERY 576 with one dominant key in each of two different waves' key shares (keys 8 w .. 8 w + 7 of a tile):
    # its four partial online-softmax states differ by hundreds of exponent units when they are merged
    qkv[576, :D] *= 6.0
    qkv[101, D:2 * D] *= 30.0
    qkv[250, D:2 * D] *= 18.0
    ref = _attn_ref(qkv, B, T, heads)
    out = torch.empty((B * T, D), device="cuda")
    dqkv = qkv.cuda()
    _check(lib.tstar_attention_f32(dqkv.data_ptr(), out.data_ptr(), B, T, heads, 0, None,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert (out.cpu() - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("B,T,heads", [(1, 577, 12), (2, 577, 3), (1, 33, 2), (1, 128, 1), (1, 1, 1), (2, 97, 3), (1, 600, 2), (1, 64, 1)])
def test_attention_split(lib, B, T, heads):
    """Full attention with f32-split operands on the bf16 matrix pipe (the bf16-pipe weights modes): each operand
    carries 16 significand bits, so the result tracks the float64 reference to ~1e-5 of the value range (the exact-f32
    kernel: ~1e-6), far inside what the detector-score contract (1e-3) needs."""
    g = torch.Generator().manual_seed(B * 1000 + T + heads)
    D = heads * 64
    qkv = torch.randn(B * T, 3 * D, generator=g)
    qkv[:, :D] *= 3.0                                    # scores up to ~ +-25: a peaky softmax
    ref = _attn_ref(qkv.double(), B, T, heads)
    out = torch.full((B * T, D), float("nan"), device="cuda")
    out32 = torch.empty((B * T, D), device="cuda")
    dqkv = qkv.cuda()
    _check(lib.tstar_attention_split(dqkv.data_ptr(), out.data_ptr(), B, T, heads, torch.cuda.current_stream().cuda_stream))
    _check(lib.tstar_attention_f32(dqkv.data_ptr(), out32.data_ptr(), B, T, heads, 0, None,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    err = (out.cpu().double() - ref).abs().max().item()
    err32 = (out32.cpu().double() - ref).abs().max().item()
    assert err < 2e-4, (err, err32)
    assert err < 60 * err32 + 1e-5, (err, err32)


def test_attention_split_spiked(lib):
    """Large online-softmax rescales (a dominant key late, and the straggler key 576 dominating) in the split kernel."""
    B, T, heads = 1, 577, 2
    D = heads * 64
    for spike_last in (False, True):
        g = torch.Generator().manual_seed(11)
        qkv = torch.randn(B * T, 3 * D, generator=g)
        qkv[500, D:2 * D] *= 40.0
        qkv[3, D:2 * D] *= 25.0
        if spike_last:
            qkv[576, D:2 * D] *= 60.0
        ref = _attn_ref(qkv.double(), B, T, heads)
        out = torch.empty((B * T, D), device="cuda")
        dqkv = qkv.cuda()
        _check(lib.tstar_attention_split(dqkv.data_ptr(), out.data_ptr(), B, T, heads, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        # a dominant logit of magnitude ~300 carries an absolute error of ~300 * 2^-17 = 2e-3 in the exponent
        assert (out.cpu().double() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("B,T,heads", [(1, 577, 12), (2, 577, 3), (1, 33, 2), (1, 128, 1), (1, 1, 1), (2, 97, 3), (1, 600, 2), (1, 64, 1),
                                        (1, 65, 1), (3, 193, 2)])
def test_attention_x3(lib, B, T, heads):
    """The f32x3 mode's vision attention (round 5): Q (pre-scaled), K, V and the probabilities each as THREE exact bf16 terms,
    six products per MFMA step on the bf16 matrix pipe, f32 accumulation, software-pipelined over key tiles.  The gate for a
    bf16-pipe kernel to count as f32: its error against float64 must be no larger than the exact-f32 MFMA kernel's on the
    same inputs (peaky softmax: scores up to ~ +-25) -- max and rms, both printed."""
    g = torch.Generator().manual_seed(B * 1000 + T + heads)
    D = heads * 64
    qkv = torch.randn(B * T, 3 * D, generator=g)
    qkv[:, :D] *= 3.0
    ref = _attn_ref(qkv.double(), B, T, heads)
    out = torch.full((B * T, D), float("nan"), device="cuda")
    out32 = torch.empty((B * T, D), device="cuda")
    dqkv = qkv.cuda()
    st = torch.cuda.current_stream().cuda_stream
    _check(lib.tstar_attention_x3(dqkv.data_ptr(), out.data_ptr(), B, T, heads, st))
    _check(lib.tstar_attention_f32(dqkv.data_ptr(), out32.data_ptr(), B, T, heads, 0, None, st))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    e, e32 = (out.cpu().double() - ref), (out32.cpu().double() - ref)
    err, err32 = e.abs().max().item(), e32.abs().max().item()
    rms, rms32 = e.pow(2).mean().sqrt().item(), e32.pow(2).mean().sqrt().item()
    print(f"attention_x3 B={B} T={T} heads={heads}: max err {err:.3e} (f32 kernel {err32:.3e}), rms {rms:.3e} ({rms32:.3e})")
    assert err < 2e-5
    assert rms <= rms32 * 1.05 + 1e-9, (rms, rms32)
    assert err <= 1.5 * err32 + 1e-7, (err, err32)                  # a max over B * T * D elements is a noisy statistic


def test_attention_x3_spiked(lib):
    """Large online-softmax rescales in the pipelined kernel (a dominant key late, an early big one, the straggler key 576
    dominating): the same bound as the exact-f32 kernel's test."""
    B, T, heads = 1, 577, 2
    D = heads * 64
    for spike_last in (False, True):
        g = torch.Generator().manual_seed(11)
        qkv = torch.randn(B * T, 3 * D, generator=g)
        qkv[500, D:2 * D] *= 40.0
        qkv[3, D:2 * D] *= 25.0
        qkv[101, D:2 * D] *= 30.0
        if spike_last:
            qkv[576, D:2 * D] *= 60.0
        ref = _attn_ref(qkv, B, T, heads)
        out = torch.empty((B * T, D), device="cuda")
        dqkv = qkv.cuda()
        _check(lib.tstar_attention_x3(dqkv.data_ptr(), out.data_ptr(), B, T, heads, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert (out.cpu() - ref).abs().max().item() < 5e-5


def test_attention_causal_padded(lib):
    B, T, heads = 4, 16, 8
    g = torch.Generator().manual_seed(5)
    D = heads * 64
    qkv = torch.randn(B * T, 3 * D, generator=g)
    lens = [3, 4, 16, 2]
    km = torch.zeros(B, T, dtype=torch.uint8)
    for i, n in enumerate(lens):
        km[i, :n] = 1
    neg = torch.finfo(torch.float32).min
    causal = torch.full((T, T), neg).triu(1)
    pad = torch.zeros(B, 1, 1, T).masked_fill(km.view(B, 1, 1, T) == 0, neg)
    mask = (causal.view(1, 1, T, T) + pad).clamp_min(neg)
    ref = _attn_ref(qkv, B, T, heads, mask)
    out = torch.empty((B * T, D), device="cuda")
    dqkv, dkm = qkv.cuda(), km.cuda()
    _check(lib.tstar_attention_f32(dqkv.data_ptr(), out.data_ptr(), B, T, heads, 1, dkm.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().view(B, T, D)
    r = ref.view(B, T, D)
    # rows whose own position is padding attend to nothing valid beyond the causal prefix in HF
    # (softmax over all-min rows is uniform); only the non-padded query rows feed the pooled output.
    for i, n in enumerate(lens):
        assert (o[i, :n] - r[i, :n]).abs().max().item() < 2e-5
