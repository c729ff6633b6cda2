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
//    Round 3, the one bounded experiment the review asked for (commit "attention: VALU straggler query"): the wave whose tile
//    holds the single straggler query (T - 1 = 576) scores it with VALU ops from the K / V tiles the block stages anyway
//    (32 FMAs + two 5-step wave reductions + 32 readlane/FMA per key tile; every lane of that wave already holds the query in
//    fragment layout) instead of a 19th MFMA wave tile: 5.3 % fewer MFMAs.  Same-session A/B, three alternations:
//    B = 256: 111.1 / 112.9 / 113.8 (round-2 kernel) vs 110.9 / 112.5 / 112.6; B = 64: 95.2 / 96.0 / 96.0 vs 98.0 / 97.8 / 97.9.
//    No gain at the bench's batch size (gate was 125): removing MFMA work does not shorten anything, because the cost of
//    T = 4 x 128 + 65 is SLOT occupancy -- the fifth workgroup of every head holds its CU slot (LDS, 4 wave slots) for all
//    18 key tiles with two of its four waves working: 20 wave slots for 18.03 tiles of queries = the 0.90 already
//    accounted for.  The 16x16x4 MFMA shape itself has the same rate (2048 flops / 32 cycles) and the same accumulator
//    footprint per query as 32x32x2, so it buys registers only with fewer queries per wave, i.e. more K / V staging per
//    MFMA (the 3- / 6-wave result above); what would recover the 10 % is a fifth workgroup that finishes in half the
//    time (16 queries per wave there): a second loop body in this kernel for +0.9 % of a step -- not built.  Topic closed.
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

template <int MODE>
__global__ __launch_bounds__(256, 4) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int T, int heads, int qtiles,
                                                            const uint8_t* __restrict__ key_mask) {
    __shared__ __attribute__((aligned(16))) float Ks[2][KB][HD];
    __shared__ __attribute__((aligned(16))) float Vs[2][KB][HD];

    const int D = heads * HD, D3 = 3 * D;
    int bid = blockIdx.x;
    const int qt = bid % qtiles; bid /= qtiles;
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
    if (mode == 0)
        hipLaunchKernelGGL(attention_f32_kernel<0>, dim3(grid), dim3(256), 0, s, qkv, out, T, heads, qtiles, key_mask);
    else
        hipLaunchKernelGGL(attention_f32_kernel<1>, dim3(grid), dim3(256), 0, s, qkv, out, T, heads, qtiles, key_mask);
    if (prof) prof_stop(PROF_ATTN, s);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

}  // namespace tstar
