// Flash-style multi-head self-attention on the bf16 matrix pipe with f32-split operands, head_dim 64,
// full (unmasked) attention only -- the vision tower's attention in the opt-in bf16-pipe modes
// (TSTAR_WEIGHTS_BF16 / TSTAR_WEIGHTS_BF16_EXACT; the f32x3 mode keeps the exact-f32 attention).  Same math as attention_f32.hip (HF modeling_owlvit.py
// :377-402, softmax(Q K^T / 8) V), same block/wave mapping and the same transposed-score trick; what
// changes is the arithmetic of the two contractions:
//
//  * every f32 operand x is carried as two round-to-nearest bfloat16 terms x_hi + x_lo (16 significand
//    bits, |x - x_hi - x_lo| <= 2^-18 |x|) and a product a*b runs as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
//    v_mfma_f32_32x32x16_bf16 (exact products, f32 accumulation, the 2^-18 lo*lo term dropped) -- the
//    two-terms-per-operand scheme of the f32-split GEMM tile of rounds 1-3 (retired in round 4).  Per 32-key block a wave issues 24 MFMAs of
//    32 cycles instead of 64 of 64 cycles.
//  * S^T = K Q^T: A = K tile (LDS, [key][d] bf16 planes, d contiguous), B = Q fragment (registers, split
//    once per block, pre-scaled by log2(e)/8).
//  * O^T = V^T P^T: B = P (the lane's 16 probabilities, split in registers; its keys (r&3)+8(r>>2)+4h are
//    exactly two 4-key runs per K=16 step), A = V^T from LDS.  V is TRANSPOSED while it is staged
//    (global [key][d] f32 -> LDS [d][key] bf16 planes; a thread packs two adjacent keys into one dword), so
//    a lane fetches its 2 x 4 keys at a fixed d with two ds_read_b64.
//  * LDS (16 KB per buffer, double buffered): K planes 2 x 32 x 128 B, 16-B chunks XOR-swizzled by
//    (key >> 1) & 7; V^T planes 2 x 64 x 64 B, 8-B units XOR-swizzled by (d >> 2) & 7: fragment reads are
//    conflict-free, staging writes at most 2-way (free for ds_write_b32).
//  * T = 32 n + 1: the straggler key is folded in with VALU ops after the loop, as in attention_f32.hip.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include <math.h>

namespace tstar {

namespace {

constexpr int HD = 64, KB = 32;
constexpr int K_PLANE = KB * 128, V_PLANE = HD * 64, BUF = 2 * K_PLANE + 2 * V_PLANE;   // bytes: 16 KB per buffer

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> {packed bf16 hi pair, packed lo pair}, round to nearest: x = hi + lo + O(2^-18 x)
__device__ __forceinline__ u32x2 split2(float x0, float x1) {
    f32x2 x; x[0] = x0; x[1] = x1;
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 r;
    r[0] = x0 - __uint_as_float(hi << 16);
    r[1] = x1 - __uint_as_float(hi & 0xFFFF0000u);
    u32x2 o;
    o[0] = hi;
    o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    return o;
}

__device__ __forceinline__ int k_off(int key, int chunk) { return key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int d, int unit) { return d * 64 + ((unit ^ ((d >> 2) & 7)) << 3); }

__global__ __launch_bounds__(256, 3) void attention_split_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 int T, int heads, int qtiles) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int D = heads * HD, D3 = 3 * D;
    int bid = blockIdx.x;
    const int qt = bid % qtiles; bid /= qtiles;
    const int head = bid % heads;
    const int b = bid / heads;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const size_t rowbase = (size_t)b * T;

    // ---- Q fragment (B operand of S^T = K Q^T): lane holds Q[q][16 s + 8 h .. +7], s = 0..3, split hi / lo
    const int q = qt * 128 + wave * 32 + l31;
    const int qc = q < T ? q : T - 1;
    const bool wave_active = (qt * 128 + wave * 32) < T;
    bf16x8 qh[4], ql[4];
    {
        const float* qp = qkv + (rowbase + qc) * D3 + head * HD + 8 * h;
        const float sc = 0.125f * 1.44269504088896340736f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s) * sc;
            const f32x4 c = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4) * sc;
            u32x4 hi, lo;
            const u32x2 p0 = split2(a[0], a[1]), p1 = split2(a[2], a[3]), p2 = split2(c[0], c[1]), p3 = split2(c[2], c[3]);
            hi[0] = p0[0]; hi[1] = p1[0]; hi[2] = p2[0]; hi[3] = p3[0];
            lo[0] = p0[1]; lo[1] = p1[1]; lo[2] = p2[1]; lo[3] = p3[1];
            qh[s] = __builtin_bit_cast(bf16x8, hi);
            ql[s] = __builtin_bit_cast(bf16x8, lo);
        }
    }

    // ---- staging assignment
    // K: thread -> keys (t >> 4), (t >> 4) + 16; float4 column t & 15 (d = 4 (t & 15) .. +3)
    // V: thread -> key pair 2 (t & 15), 2 (t & 15) + 1; d = 4 (t >> 4) .. +3 (transposed into LDS)
    const int kc4 = t & 15, kr = t >> 4;
    const int vkp = t & 15, vdq = t >> 4;
    const float* kbase = qkv + D + head * HD + kc4 * 4;
    const float* vbase = qkv + 2 * D + head * HD + vdq * 4;
    auto krow = [&](int key) { return (rowbase + (key < T ? key : T - 1)) * D3; };

    const bool tail_key = (T % KB) == 1;
    const int nkb = tail_key ? T / KB : (T + KB - 1) / KB;

    f32x4 rk[2], rv[2];
    auto gload = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rk[i] = *reinterpret_cast<const f32x4*>(kbase + krow(kb * KB + kr + 16 * i));
            rv[i] = *reinterpret_cast<const f32x4*>(vbase + krow(kb * KB + 2 * vkp + i));
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x2 p0 = split2(rk[i][0], rk[i][1]), p1 = split2(rk[i][2], rk[i][3]);
            u32x2 hi, lo;
            hi[0] = p0[0]; hi[1] = p1[0];
            lo[0] = p0[1]; lo[1] = p1[1];
            const int off = k_off(kr + 16 * i, kc4 >> 1) + (kc4 & 1) * 8;
            *reinterpret_cast<u32x2*>(base + off) = hi;
            *reinterpret_cast<u32x2*>(base + K_PLANE + off) = lo;
        }
        char* vb = base + 2 * K_PLANE;
#pragma unroll
        for (int e = 0; e < 4; ++e) {            // one dword = (key 2 kp, key 2 kp + 1) at d = 4 dq + e
            const u32x2 p = split2(rv[0][e], rv[1][e]);
            const int off = v_off(4 * vdq + e, vkp >> 1) + (vkp & 1) * 4;
            *reinterpret_cast<unsigned*>(vb + off) = p[0];
            *reinterpret_cast<unsigned*>(vb + V_PLANE + off) = p[1];
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    if (nkb > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const bool more = kb + 1 < nkb;
        if (more) gload(kb + 1);
        if (wave_active) {
            const char* base = smem + cur * BUF;
            // S^T[key][q]: two accumulators break the 12-deep dependent chain; summed below
            f32x16 sa, sb;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(base + k_off(l31, 2 * s + h));
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(base + K_PLANE + k_off(l31, 2 * s + h));
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], sb, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], sb, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], sa, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = sa[r] + sb[r];
            if (!tail_key && kb == nkb - 1) {           // a partial last block: mask the keys past T
                const int key0 = kb * KB + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (key0 + (r & 3) + 8 * (r >> 2)) < T ? s[r] : -INFINITY;
            }
            float mb = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mb = fmaxf(mb, s[r]);
            mb = fmaxf(mb, __shfl_xor(mb, 32));
            const float m_new = fmaxf(m_run, mb);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // every block holds at least one valid key
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
                ps += s[r];
            }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
            // O^T[d][q] += sum_key V[key][d] P[q][key]; K=16 step ks covers keys 16 ks .. 16 ks + 15, of which this lane's
            // P values are r = 8 ks .. 8 ks + 7 <-> keys 16 ks + 4 h + {0..3} and 16 ks + 8 + 4 h + {0..3}
            const char* vb = base + 2 * K_PLANE;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 phi, plo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x2 p = split2(s[8 * ks + 2 * j], s[8 * ks + 2 * j + 1]);
                    phi[j] = p[0];
                    plo[j] = p[1];
                }
                const bf16x8 ph = __builtin_bit_cast(bf16x8, phi), pl = __builtin_bit_cast(bf16x8, plo);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = l31 + 32 * dt;
                    u32x4 vh, vl;
                    const u32x2 a0 = *reinterpret_cast<const u32x2*>(vb + v_off(d, 4 * ks + h));
                    const u32x2 a1 = *reinterpret_cast<const u32x2*>(vb + v_off(d, 4 * ks + 2 + h));
                    const u32x2 c0 = *reinterpret_cast<const u32x2*>(vb + V_PLANE + v_off(d, 4 * ks + h));
                    const u32x2 c1 = *reinterpret_cast<const u32x2*>(vb + V_PLANE + v_off(d, 4 * ks + 2 + h));
                    vh[0] = a0[0]; vh[1] = a0[1]; vh[2] = a1[0]; vh[3] = a1[1];
                    vl[0] = c0[0]; vl[1] = c0[1]; vl[2] = c1[0]; vl[3] = c1[1];
                    const bf16x8 vhi = __builtin_bit_cast(bf16x8, vh), vlo = __builtin_bit_cast(bf16x8, vl);
                    f32x16& o = dt == 0 ? o0 : o1;
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, ph, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vhi, pl, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vhi, ph, o, 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if (tail_key && wave_active) {
        // straggler key T-1 in f32 VALU arithmetic; q is rebuilt from its two terms (16 significand bits, as in the MFMAs)
        const size_t ro = (rowbase + (T - 1)) * D3 + head * HD;
        float sx = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(qkv + ro + D + 16 * s + 8 * h);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(qkv + ro + D + 16 * s + 8 * h + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sx += k0[e] * ((float)qh[s][e] + (float)ql[s][e]);
                sx += k1[e] * ((float)qh[s][4 + e] + (float)ql[s][4 + e]);
            }
        }
        sx += __shfl_xor(sx, 32);
        const float m_new = fmaxf(m_run, sx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float p = __builtin_amdgcn_exp2f(sx - m_new);
        l_run = l_run * alpha + p;
        m_run = m_new;
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

}  // namespace

int attention_split(const float* qkv, float* out, int B, int T, int heads, hipStream_t s) {
    TSTAR_REQUIRE(B > 0 && T > 0 && heads > 0, "attention_split: empty problem");
    const int qtiles = cdiv(T, 128);
    const int grid = B * heads * qtiles;
    const bool prof = prof_enabled();
    if (prof) prof_start(PROF_ATTN, s, 4.0 * B * heads * (double)T * T * HD);
    hipLaunchKernelGGL(attention_split_kernel, dim3(grid), dim3(256), 0, s, qkv, out, T, heads, qtiles);
    if (prof) prof_stop(PROF_ATTN, s);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

}  // namespace tstar
