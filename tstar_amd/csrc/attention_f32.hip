// Flash-style fp32 multi-head self-attention on the CDNA4 matrix cores, head_dim 64.
//
// Replaces OwlViTAttention.forward (HF modeling_owlvit.py:428-459; eager math
// :377-402: softmax(Q K^T / 8 + mask) V) for both towers: vision T=577 / 12
// heads (no mask) and text T=16 / 8 heads (causal + key-padding mask,
// modeling_owlvit.py:631-663).  Input is the fused QKV projection
// [B*T, 3*D] (q | k | v), output [B*T, D] ready for out_proj.
//
// Mapping (wave64, v_mfma_f32_32x32x2_f32, exact f32):
//  * block = 4 waves = 128 query rows of one (image, head); wave w owns 32 rows.
//  * scores are computed TRANSPOSED, S^T = K Q^T (A = K tile from LDS, B = Q
//    fragment held in 32 VGPRs for the whole kernel, pre-scaled by
//    log2(e)/8).  In the 32x32 C/D layout a lane then owns ONE query
//    (column lane&31) and 16 keys ((r&3)+8(r>>2)+4(lane>>5)): the row max / row
//    sum are 15 in-lane ops + one cross-half exchange, and the probabilities
//    are already in B-operand layout for O^T = V^T P^T -- no LDS round trip,
//    no permutes.  O^T columns are queries too, so the online-softmax rescale is
//    lane-local.
//  * T = 32 n + 1 (577 = 576 patches + CLS): the straggler key is folded in after
//    the block loop with VALU ops (tail_key) instead of a 19th, 97 % empty block.
//  * K/V tiles (32 keys) are staged global -> VGPR -> LDS, double-buffered, one
//    barrier per tile; K rows are 64 floats with an XOR swizzle of the float4
//    column (conflict-free ds_read_b128, 32 KB of LDS per workgroup),
//    V read as ds_read_b32 rows (two 32-lane halves never conflict).
//  * measured and NOT adopted (round 2, tools/bench_attention.py, B = 256): 3- / 6-wave
//    workgroups that tile the 576 patch queries exactly (92 / 67 TFLOP/s vs 118 for 4 waves:
//    more K/V staging per MFMA, no SIMD imbalance to win back -- a workgroup's waves land on
//    the SIMDs cyclically from a varying start); the straggler query 576 in its own VALU-only
//    workgroup (-15 %: it re-reads all K/V of its head, 20 % more L2 traffic); K/V tiles by direct
//    global -> LDS DMA (global_load_lds_dwordx4 through inline asm, no staging VGPRs / ds_write:
//    114.4 vs 114.6 in a same-session A/B -- no gain); the same plus scalar-base addressing, a loop
//    unrolled over the two LDS buffers and a subtract-free softmax (accumulator started at -max):
//    107 (register spills).  Ablations of the adopted kernel at B = 256: no barrier +-0, no K/V
//    reloads +7 %, no exp / sub / add in the softmax +3.6 %: no single limiter is left; the rest is
//    the MFMA issue pattern (one dependent 32-MFMA chain for S^T, two chains for PV, LDS fragment
//    reads between them) against 4 independent chains in the GEMM tile.  That last hypothesis was then tested
//    too: a software-pipelined variant (S^T of tile t issued as s, o0, s, o1 with PV of tile t-1 -- three
//    independent chains, fragments of step r+1 requested before the MFMAs of step r, order pinned with
//    sched_barrier, LDS-DMA staging; 140 VGPRs = 3 waves / SIMD) measured 113.9 vs 117.9 in a same-session
//    A/B: dependent-MFMA stalls are not the limiter either, the lost wave of occupancy costs more.  Two more same-session
//    A/Bs against 119.8: the V tile staged TRANSPOSED so that one ds_read_b128 feeds four PV MFMAs (8 b128 instead of 16
//    ds_read2_b32 per tile, transposing ds_write_b32 staging): 109.3; an XCD-aware block order that keeps the five query
//    tiles of a head on one L2: 119.8 -- the K/V re-reads already hit; 2-wave workgroups with 64 queries per wave (two Q
//    fragments share every K / V fragment read: half the LDS fragment traffic per MFMA, 2 + 4 independent MFMA chains, 235
//    VGPRs = 2 waves / SIMD): 114.7 vs 119.2 -- the LDS fragment reads are not the limiter either.
//    Round 3.  (a) The wave whose tile holds the single straggler query (T - 1 = 576) scoring it with VALU ops from the staged
//    K / V tiles instead of a 19th MFMA wave tile (5.3 % fewer MFMAs): same-session A/B at B = 256 111.1 / 112.9 / 113.8
//    (round-2 kernel) vs 110.9 / 112.5 / 112.6, B = 64 +2 % -- not adopted on its own.  (b) ADOPTED: the last query block of a
//    head (65 queries at T = 577) as FOUR 16-query wave tiles on v_mfma_f32_16x16x4_f32 plus the straggler shared by the four
//    waves as VALU work (attention_tail16 below), dispatched after all full blocks: B = 256 119.5 / 119.8 / 119.9 -> 123.3 /
//    124.0 / 123.8 TFLOP/s (+3.4 %), B = 64 105.6 -> 109.5, B = 16 90 -> 95 (profiles/r03_attention_tail16_ab.md).  The review's
//    gate of 125 is not met: the gain is the 5 % of MFMA work the idle / single-query wave tiles of that block used to issue plus
//    part of their slot occupancy; the 16x16x4 shape has the same rate and the same accumulator footprint per query as 32x32x2,
//    so it buys nothing by itself.  What remains between 124 and the 136-140 of a register-only MFMA loop is per-wave
//    serialisation (LDS fragment reads and the softmax between dependent MFMA chains) under 4 waves per SIMD.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include <math.h>

namespace tstar {

constexpr int HD = 64, KB = 32;
// K tile rows are 64 floats (one 256-B LDS bank row), un-padded; float4 column j of key row k is stored at column
// j ^ (k & 15).  ds_read_b128 is serviced in 16-lane groups over 64 banks (MI355X guide, LDS): the 16 lanes of a group
// must hit 16 distinct 16-B slots, and the S^T fragment read (16 keys with distinct k & 15, one logical column) does.
__device__ __forceinline__ int kswz(int key, int j) { return (j ^ (key & 15)) * 4; }

// ---------------------------------------------------------------------------------------------
// The last query block of a head when T = 128 n + 65 (577 = 4 x 128 + 64 + 1): with 32-query wave tiles that workgroup
// keeps its CU slot (LDS, four wave slots) for all 18 key tiles while two of its waves work, the third carries ONE query
// and the fourth none -- 20 wave slots for 18.03 tiles of queries.  Here the same 65 queries are spread over all four waves
// as 16-query tiles on v_mfma_f32_16x16x4_f32 (same rate per flop, half the MFMAs per wave and key tile, so the workgroup is
// done in roughly half the time), and the straggler query T-1 is scored by wave 3 with VALU ops from the K / V tiles the
// workgroup stages anyway.  These short workgroups are dispatched AFTER all full ones (block index order), so they fill the tail.
// Layouts (16x16x4: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15], C/D: n = lane & 15, m = 4 (lane >> 4) + r):
//   S^T tile kt (keys 16 kt .. +15): A = K[key = lane & 15][d = 16 j + 4 g + e] (one swizzled ds_read_b128 per j), B = Q fragment;
//   a lane ends up with the scores of query lane & 15 against keys 16 kt + 4 g + r -- which is exactly the B operand
//   P[key][q] of O^T += V^T P^T at step (kt, r), as in the 32-query form; softmax statistics need two cross-lane steps (g).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// V rows are rotated by 16 ((key >> 2) & 3) floats in the short workgroups: the PV fragment read of a 32-lane service group
// spans two key rows 4 apart (g = 0 / 1), which would otherwise sit on the same 16 banks
__device__ __forceinline__ int vrot(int key) { return 16 * ((key >> 2) & 3); }
__device__ __forceinline__ void attention_tail16(const float* __restrict__ qkv, float* __restrict__ out, int T, int heads, int b, int head,
                                              float (&Ks)[2][32][64], float (&Vs)[2][32][64]) {
    const int D = heads * HD, D3 = 3 * D;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const size_t rowbase = (size_t)b * T;
    const int q = (T - 65) + wave * 16 + l15;                          // 64 full queries; all valid
    const float sc = 0.125f * 1.44269504088896340736f;
    f32x4 qf[4];
    {
        const float* qp = qkv + (rowbase + q) * D3 + head * HD + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) { qf[j] = *reinterpret_cast<const f32x4*>(qp + 16 * j); qf[j] *= sc; }
    }
    // The straggler query T-1 is shared by the four waves: wave w scores it against keys 8 w .. 8 w + 7 of every tile with VALU
    // ops (lane = (key k8, 8-dim group dg)) and keeps its own online-softmax state over THOSE keys; the four partial states are
    // merged once at the end.  Per tile and wave: 8 FMAs + three 3-step reductions + 8 x (readlane, ds_read, FMA).
    const int k8 = lane >> 3, dg = lane & 7;
    f32x4 qs[2];
    {
        const float* qp = qkv + (rowbase + T - 1) * D3 + head * HD + 8 * dg;
        qs[0] = *reinterpret_cast<const f32x4*>(qp); qs[1] = *reinterpret_cast<const f32x4*>(qp + 4);
        qs[0] *= sc; qs[1] *= sc;
    }
    float mv_run = -INFINITY, lv_run = 0.f, o_v = 0.f;

    const int f4 = t & 15, sr = t >> 4;
    const float* kbase = qkv + D + head * HD + f4 * 4;
    const float* vbase = qkv + 2 * D + head * HD + f4 * 4;
    const int nkb = T / KB;                                            // T = 32 n + 1: the last key is folded in after the loop
    f32x4 rk[2], rv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t ro = (rowbase + sr + 16 * i) * D3;
        rk[i] = *reinterpret_cast<const f32x4*>(kbase + ro);
        rv[i] = *reinterpret_cast<const f32x4*>(vbase + ro);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<f32x4*>(&Ks[0][sr + 16 * i][kswz(sr + 16 * i, f4)]) = rk[i];
        *reinterpret_cast<f32x4*>(&Vs[0][sr + 16 * i][(f4 * 4 + vrot(sr + 16 * i)) & 63]) = rv[i];
    }
    __syncthreads();

    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    int cur = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const bool more = kb + 1 < nkb;
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const size_t ro = (rowbase + (kb + 1) * KB + sr + 16 * i) * D3;
                rk[i] = *reinterpret_cast<const f32x4*>(kbase + ro);
                rv[i] = *reinterpret_cast<const f32x4*>(vbase + ro);
            }
        }
        f32x4_t s[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const float* kp = &Ks[cur][16 * kt + l15][0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(kp + (((4 * j + g) ^ l15) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[e], qf[j][e], s[kt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // the wave's share of the straggler query: keys 8 wave + k8 of this tile (issued here so that it overlaps the MFMA chain)
        float sx;
        {
            const int key = 8 * wave + k8;
            const float* kp = &Ks[cur][key][0];
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp + kswz(key, 2 * dg));
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(kp + kswz(key, 2 * dg + 1));
            sx = k0[0] * qs[0][0];
            sx = fmaf(k0[1], qs[0][1], sx); sx = fmaf(k0[2], qs[0][2], sx); sx = fmaf(k0[3], qs[0][3], sx);
            sx = fmaf(k1[0], qs[1][0], sx); sx = fmaf(k1[1], qs[1][1], sx); sx = fmaf(k1[2], qs[1][2], sx); sx = fmaf(k1[3], qs[1][3], sx);
            sx += __shfl_xor(sx, 1); sx += __shfl_xor(sx, 2); sx += __shfl_xor(sx, 4);      // all 8 lanes of key k8 hold its score
        }
        float mb = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mb = fmaxf(mb, __shfl_xor(mb, 16));
        mb = fmaxf(mb, __shfl_xor(mb, 32));
        const float m_new = fmaxf(m_run, mb);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // exp2(-inf) = 0 on the first tile
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m_new); ps += s[kt][r]; }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                const float* vrow = &Vs[cur][key][0];
                const int c0 = l15 + vrot(key);                        // vrot is a multiple of 16: (c0 + 16 dt) & 63 keeps the lane's column
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[(c0 + 16 * dt) & 63], s[kt][r], o[dt], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        {
            // online softmax of the straggler over this wave's 8 keys, then O[d = lane] += sum_k p_k V[k][lane]
            float mx = sx;
            mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(mv_run, mx);
            const float al = __builtin_amdgcn_exp2f(mv_run - mn);
            const float p = __builtin_amdgcn_exp2f(sx - mn);
            float pv = p;
            pv += __shfl_xor(pv, 8); pv += __shfl_xor(pv, 16); pv += __shfl_xor(pv, 32);
            lv_run = lv_run * al + pv;
            mv_run = mn;
            o_v *= al;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int key = 8 * wave + k;
                const float pk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 8 * k));     // lane 8 k holds key k's p
                o_v = fmaf(pk, Vs[cur][key][(lane + vrot(key)) & 63], o_v);
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<f32x4*>(&Ks[cur ^ 1][sr + 16 * i][kswz(sr + 16 * i, f4)]) = rk[i];
                *reinterpret_cast<f32x4*>(&Vs[cur ^ 1][sr + 16 * i][(f4 * 4 + vrot(sr + 16 * i)) & 63]) = rv[i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // the straggler KEY T-1 for the 16 queries of this wave: the lane holds a quarter of its query's 64 dims
    const size_t ro = (rowbase + (T - 1)) * D3 + head * HD;
    {
        float sx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(qkv + ro + D + 16 * j + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) sx += kv[e] * qf[j][e];
        }
        sx += __shfl_xor(sx, 16);
        sx += __shfl_xor(sx, 32);
        const float m_new = fmaxf(m_run, sx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float p = __builtin_amdgcn_exp2f(sx - m_new);
        l_run = l_run * alpha + p;
        const float inv = 1.0f / l_run;
        float* op = out + (rowbase + q) * D + head * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(qkv + ro + 2 * D + 16 * dt + 4 * g);
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (o[dt][e] * alpha + p * vv[e]) * inv;
            *reinterpret_cast<f32x4*>(op + 16 * dt) = a;
        }
    }
    // merge the four waves' partial states of the straggler query (through the K tile buffer, free now), add key T-1, store
    float* mrg = &Ks[0][0][0];                                         // [4][66]: o[64], m, l per wave
    __syncthreads();
    mrg[wave * 66 + lane] = o_v;
    if (lane == 0) { mrg[wave * 66 + 64] = mv_run; mrg[wave * 66 + 65] = lv_run; }
    __syncthreads();
    if (wave == 0) {
        float sx = 0.f;                                                // score of key T-1: lane (k8 unused) -> 8-dim group dg
        {
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(qkv + ro + D + 8 * dg);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(qkv + ro + D + 8 * dg + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sx = fmaf(k0[e], qs[0][e], sx);
#pragma unroll
            for (int e = 0; e < 4; ++e) sx = fmaf(k1[e], qs[1][e], sx);
            sx += __shfl_xor(sx, 1); sx += __shfl_xor(sx, 2); sx += __shfl_xor(sx, 4);
        }
        float m = sx;
#pragma unroll
        for (int w = 0; w < 4; ++w) m = fmaxf(m, mrg[w * 66 + 64]);
        float l = __builtin_amdgcn_exp2f(sx - m), ov = l * qkv[ro + 2 * D + lane];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __builtin_amdgcn_exp2f(mrg[w * 66 + 64] - m);
            l = fmaf(mrg[w * 66 + 65], f, l);
            ov = fmaf(mrg[w * 66 + lane], f, ov);
        }
        out[(rowbase + T - 1) * D + head * HD + lane] = ov / l;
    }
}

template <int MODE, bool T16 = false>
__global__ __launch_bounds__(256, 4) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int T, int heads, int qtiles,
                                                            const uint8_t* __restrict__ key_mask) {
    __shared__ __attribute__((aligned(16))) float Ks[2][KB][HD];
    __shared__ __attribute__((aligned(16))) float Vs[2][KB][HD];

    const int D = heads * HD, D3 = 3 * D;
    int bid = blockIdx.x, qt;
    if (T16) {
        // all full 128-query blocks first, then one short block per (image, head): the short ones fill the tail
        const int n_full = (gridDim.x / qtiles) * (qtiles - 1);
        if (bid >= n_full) {
            bid -= n_full;
            attention_tail16(qkv, out, T, heads, bid / heads, bid % heads, Ks, Vs);
            return;
        }
        qt = bid % (qtiles - 1); bid /= (qtiles - 1);
    } else {
        qt = bid % qtiles; bid /= qtiles;
    }
    const int head = bid % heads;
    const int b = bid / heads;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int kx4 = 4 * (h ^ (l31 & 15));
    const size_t rowbase = (size_t)b * T;

    // ---- Q fragment (B operand): lane holds Q[q][8c + 4h .. +3], c = 0..7, scaled by log2e/8
    const int q = qt * 128 + wave * 32 + l31;
    const int qc = q < T ? q : T - 1;
    const bool wave_active = (qt * 128 + wave * 32) < T;
    f32x4 qf[8];
    {
        const float* qp = qkv + (rowbase + qc) * D3 + head * HD + 4 * h;
        const float sc = 0.125f * 1.44269504088896340736f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[c] = *reinterpret_cast<const f32x4*>(qp + 8 * c);
            qf[c] *= sc;
        }
    }

    // ---- staging assignment for K/V tiles: thread -> rows (t>>4), (t>>4)+16; float4 column t&15
    const int f4 = t & 15, sr = t >> 4;
    const float* kbase = qkv + D + head * HD + f4 * 4;
    const float* vbase = qkv + 2 * D + head * HD + f4 * 4;
    auto krow = [&](int key) { return (rowbase + (key < T ? key : T - 1)) * D3; };

    // T = 32 n + 1 (the ViT's 576 patches + CLS): the single straggler key is folded in after the loop with a
    // few VALU ops instead of costing a whole 32-key MFMA block (1/19 of the kernel at T = 577)
    const bool tail_key = MODE == 0 && (T % KB) == 1;
    const int nkb = tail_key ? T / KB : (T + KB - 1) / KB;
    f32x4 rk[2], rv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        size_t ro = krow(sr + 16 * i);
        rk[i] = *reinterpret_cast<const f32x4*>(kbase + ro);
        rv[i] = *reinterpret_cast<const f32x4*>(vbase + ro);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<f32x4*>(&Ks[0][sr + 16 * i][kswz(sr + 16 * i, f4)]) = rk[i];
        *reinterpret_cast<f32x4*>(&Vs[0][sr + 16 * i][f4 * 4]) = rv[i];
    }
    __syncthreads();

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    int cur = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const bool more = kb + 1 < nkb;
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                size_t ro = krow((kb + 1) * KB + sr + 16 * i);
                rk[i] = *reinterpret_cast<const f32x4*>(kbase + ro);
                rv[i] = *reinterpret_cast<const f32x4*>(vbase + ro);
            }
        }
        if (wave_active) {
            // S^T[key][q] = sum_d K[key][d] * Q[q][d]
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const float* kp = &Ks[cur][l31][0];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // kswz(l31, 2c + h) = (8c) ^ (4 (h ^ (l31 & 15))): one XOR with a per-lane constant
                f32x4 ka = *reinterpret_cast<const f32x4*>(kp + ((8 * c) ^ kx4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[c][e], s, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            // masks (only the last key block of the full-attention mode can hold invalid keys) + block max
            float mb = -INFINITY;
            if (MODE == 1 || (!tail_key && kb == nkb - 1)) {
                const int key0 = kb * KB + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2);
                    bool ok = key < T;
                    if (MODE == 1) ok = ok && key <= q && key_mask[(size_t)b * T + (key < T ? key : 0)] != 0;
                    s[r] = ok ? s[r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mb = fmaxf(mb, s[r]);
            mb = fmaxf(mb, __shfl_xor(mb, 32));
            const float m_new = fmaxf(m_run, mb);
            // m_new is finite as soon as one key of this or an earlier block is valid
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);     // exp2(-inf) = 0 on the first block
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
                ps += s[r];
            }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {                  // running maxima settle after a few blocks: skip the rescale
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
            // O^T[d][q] += sum_key V[key][d] * P[q][key]
            const float* vp = &Vs[cur][4 * h][l31];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = (r & 3) + 8 * (r >> 2);
                const float v0 = vp[kr * HD];
                const float v1 = vp[kr * HD + 32];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<f32x4*>(&Ks[cur ^ 1][sr + 16 * i][kswz(sr + 16 * i, f4)]) = rk[i];
                *reinterpret_cast<f32x4*>(&Vs[cur ^ 1][sr + 16 * i][f4 * 4]) = rv[i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    if (tail_key && wave_active) {
        const size_t ro = (rowbase + (T - 1)) * D3 + head * HD;
        // score of every query of this wave against key T-1: the lane holds half of the query's 64 dims
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(qkv + ro + D + 8 * c + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) sx += kv[e] * qf[c][e];
        }
        sx += __shfl_xor(sx, 32);
        const float m_new = fmaxf(m_run, sx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // 0 when no key came before (T = 1)
        const float p = __builtin_amdgcn_exp2f(sx - m_new);
        l_run = l_run * alpha + p;
        m_run = m_new;
        // O^T[d][q] = O^T[d][q] * alpha + p * V[T-1][d], d = (r&3) + 8(r>>2) + 4h (+32 for the second tile)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qkv + ro + 2 * D + 8 * g4 + 4 * h);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qkv + ro + 2 * D + 32 + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[g4 * 4 + e] = o0[g4 * 4 + e] * alpha + p * v0[e];
                o1[g4 * 4 + e] = o1[g4 * 4 + e] * alpha + p * v1[e];
            }
        }
    }

    if (q < T) {
        const float inv = 1.0f / l_run;
        float* op = out + (rowbase + q) * D + head * HD + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = o0[g4 * 4 + e] * inv; c[e] = o1[g4 * 4 + e] * inv; }
            *reinterpret_cast<f32x4*>(op + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(op + 32 + 8 * g4) = c;
        }
    }
}

int attention_f32(const float* qkv, float* out, int B, int T, int heads, int mode,
                  const uint8_t* key_mask, hipStream_t s) {
    TSTAR_REQUIRE(B > 0 && T > 0 && heads > 0, "attention_f32: empty problem");
    TSTAR_REQUIRE(mode == 0 || (mode == 1 && key_mask != nullptr), "attention_f32: mode 1 needs key_mask");
    const int qtiles = cdiv(T, 128);
    const int grid = B * heads * qtiles;
    const bool prof = prof_enabled();
    if (prof) prof_start(PROF_ATTN, s, 4.0 * B * heads * (double)T * T * HD);
    // TSTAR_ATTN_T16=0 runs the round-2 form (the last 65 queries of a head as 32-query wave tiles) for same-session A/Bs
    static const bool t16 = [] { const char* e = getenv("TSTAR_ATTN_T16"); return e ? atoi(e) != 0 : true; }();
    if (mode == 0 && t16 && T % 128 == 65)
        hipLaunchKernelGGL((attention_f32_kernel<0, true>), dim3(grid), dim3(256), 0, s, qkv, out, T, heads, qtiles, key_mask);
    else if (mode == 0)
        hipLaunchKernelGGL(attention_f32_kernel<0>, dim3(grid), dim3(256), 0, s, qkv, out, T, heads, qtiles, key_mask);
    else
        hipLaunchKernelGGL(attention_f32_kernel<1>, dim3(grid), dim3(256), 0, s, qkv, out, T, heads, qtiles, key_mask);
    if (prof) prof_stop(PROF_ATTN, s);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

}  // namespace tstar
