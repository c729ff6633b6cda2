// Vision-tower attention of the f32x3 mode (round 5): softmax(Q K^T / 8) V on the bf16 matrix pipe with EXACT operands.
//
// Replaces OwlViTAttention.forward (HF modeling_owlvit.py:377-402, 428-459) for the vision tower on behalf of
// /root/reference/TStar/interface_heuristic.py:237-239, like attention_f32.hip, whose block / wave mapping and
// transposed-score trick it keeps (block = 4 waves = 128 queries of one (image, head); S^T = K Q^T so that a lane owns one
// query and its softmax statistics are lane-local; O^T = V^T P^T with P already in B-operand layout).  What changes:
//
//  * arithmetic: every f32 operand (Q pre-scaled by log2(e)/8, K, V, and the probabilities P) is carried as THREE
//    round-to-nearest bfloat16 terms x = x0 + x1 + x2 (exact: 8 + 8 + 8 significand bits with signed remainders) and a
//    contraction step runs the six partial products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 (each product exact, f32
//    accumulation) -- the scheme of gemm_tile_x3.  The three products left out are <= 2^-24 of a term each.  Per 32-key
//    tile a wave issues 48 MFMAs of 32 cycles (the f32 kernel: 64 of 64 cycles).
//  * schedule: software-pipelined per wave across key tiles.  Phase A of iteration t interleaves the 24 MFMAs of S(t+1) with
//    the VALU work of softmax(t) and the three-term split of P(t); phase B interleaves the 24 MFMAs of O += V(t)^T P(t)^T
//    with the split + LDS stores of tile t + 2 (loaded a tile earlier) and the global loads of tile t + 3.  Operand fragments
//    are re-read from LDS into the registers of a plane as soon as that plane's last product of the step has issued (products
//    ordered x0-first for that reason), the order is pinned with sched_barrier, ONE barrier per key tile.
//  * LDS: K planes [key][d] (128-B rows, 16-B chunks XOR-swizzled by (key >> 1) & 7) double-buffered, V^T planes [d][key]
//    (80-B rows, keys permuted so that a lane's eight keys of a step are one 16-B chunk; V is transposed while staged) triple-buffered because tile t + 2 is
//    written while tile t is read: 69 KB per workgroup, 2 workgroups per CU (256 registers per wave).
//  * T = 32 n + 1 (577): the straggler key is folded in with f32 VALU ops after the loop (q rebuilt exactly from its terms).
//  * measured (B = 256, T = 577, 12 heads; tools/lab/attn_lab.hip, profiles/r05_attention_x3_lab.log, r05_attention_x3_counters.md):
//    154 TFLOP/s algorithmic (0.92 PFLOP/s executed) against 120-125 for attention_f32_kernel, 147 vs 103 inside the bench; error
//    against float64 below the f32 kernel's (tests/test_gpu_kernels.py::test_attention_x3).  Matrix pipe busy 0.51: the kernel is
//    bound by VALU issue, 7.5 VALU instructions per MFMA (the exact splits of P and of the staged K / V cost 5.5 per element).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <utility>

namespace tstar {
namespace ax3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int HD = 64, KB = 32;
constexpr int VPITCH = 80;                              // bytes per V^T row: 64 of keys + 16 of padding (see v_off)
constexpr int KP = KB * 128, VP = HD * VPITCH;          // bytes of one K plane / one V^T plane of a tile
constexpr int KBUF = 3 * KP, VBUF = 3 * VP, NKB = 2, NVB = 3;
constexpr int LDS_BYTES = NKB * KBUF + NVB * VBUF;      // 69 KB

#define AX3_FENCE() __builtin_amdgcn_sched_barrier(0)

// (x0, x1) -> three packed bf16 pairs: round-to-nearest terms of the running remainder (the third is exact)
__device__ __forceinline__ void split2_rn3(float x0, float x1, unsigned (&o)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f32x2 x; x[0] = x0; x[1] = x1;
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        o[k] = hb;
        if (k < 2) {
            x0 = x0 - __uint_as_float(hb << 16);
            x1 = x1 - __uint_as_float(hb & 0xFFFF0000u);
        }
    }
}
// value of lane ^ 32 combined with this lane's, without the LDS pipe (gfx950 v_permlane32_swap)
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ int k_off(int key, int chunk) { return key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4); }
// V^T rows hold their 32 keys in the order [0-3, 8-11 | 4-7, 12-15 | 16-19, 24-27 | 20-23, 28-31]: the eight keys a lane needs for K = 16
// step ks (4 h + {0..3} and 8 + 4 h + {0..3} of keys 16 ks ..) are ONE 16-byte chunk 2 ks + h.  Rows are 80 bytes apart: a 16-lane
// service group of ds_read_b128 reads 16 rows d, 5 d + chunk (mod 16) is then a distinct 16-byte slot of the 256-B bank row, and the
// staging stores of a wave (four rows 4 apart, 64 bytes each) start 80 dwords = 16 banks apart and cover all 64 banks once (with
// 64-byte rows they were 4-way conflicts: 38 % of the LDS-busy cycles of the first form, profiles/r05_attention_x3_counters.md)
__device__ __forceinline__ int v_off(int d, int chunk) { return d * VPITCH + (chunk << 4); }
// byte position of key k (0..31) inside its row's un-swizzled order
__device__ __forceinline__ int v_keypos(int k) { const int g = k >> 2; const int gp = (g & 4) | ((g & 1) << 1) | ((g >> 1) & 1); return (gp * 4 + (k & 3)) * 2; }

__device__ __forceinline__ f32x16 mfma(const bf16x8 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// PLANES = false: K / V come from the f32 qkv rows and are split while they are staged (any caller: tstar_attention_x3).
// PLANES = true: the qkv GEMM's epilogue has already written them as plane tiles (kv_plane_tile() below: one 27-KB LDS image per
// GLOBAL 32-row tile of the token matrix and head) and the kernel copies them with the LDS-DMA engine (global_load_lds_dwordx4:
// no staging registers, no split, no LDS stores in the loop -- the split happens ONCE per K / V element instead of once per
// query block); key tiles are then the global tiles that overlap the image's rows, masked at both ends.
template <bool PLANES>
__global__ __launch_bounds__(256, 2) void attention_x3_kernel(const float* __restrict__ qkv, const char* __restrict__ planes,
                                                              float* __restrict__ out, int T, int heads, int qtiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kbuf = smem;
    char* const vbuf = smem + NKB * KBUF;

    const int D = heads * HD, D3 = 3 * D;
    int bid = blockIdx.x;
    const int qt = bid % qtiles; bid /= qtiles;
    const int head = bid % heads;
    const int b = bid / heads;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const size_t rowbase = (size_t)b * T;

    const int q = qt * 128 + wave * 32 + l31;
    const int qc = q < T ? q : T - 1;
    const bool wave_active = (qt * 128 + wave * 32) < T;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][16 s + 8 h .. +7], s = 0..3, three planes
    u32x4 qp[3][4];
    {
        const float* src = qkv + (rowbase + qc) * D3 + head * HD + 8 * h;
        const float sc = 0.125f * 1.44269504088896340736f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + 16 * s) * sc;
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + 16 * s + 4) * sc;
            unsigned o0[3], o1[3], o2[3], o3[3];
            split2_rn3(a[0], a[1], o0); split2_rn3(a[2], a[3], o1); split2_rn3(c[0], c[1], o2); split2_rn3(c[2], c[3], o3);
#pragma unroll
            for (int k = 0; k < 3; ++k) { qp[k][s][0] = o0[k]; qp[k][s][1] = o1[k]; qp[k][s][2] = o2[k]; qp[k][s][3] = o3[k]; }
        }
    }

    // ---- staging assignment (as attention_split.hip)
    // K: thread -> keys (t >> 4), (t >> 4) + 16; float4 column t & 15.  V: thread -> keys 2 (t & 15), 2 (t & 15) + 1; d = 4 (t >> 4) .. +3
    const int kc4 = t & 15, kr = t >> 4;
    const int vkp = t & 15, vdq = t >> 4;
    // buffer loads over THIS image's rows: a 32-bit byte offset per staged row + a uniform tile offset; rows past T read as zeros
    const __amdgpu_buffer_rsrc_t kv_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv + rowbase * D3), 0, (int)((size_t)T * D3 * 4), 0x00020000);
    int kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        kvo[i] = ((kr + 16 * i) * D3 + D + head * HD + kc4 * 4) * 4;
        vvo[i] = ((2 * vkp + i) * D3 + 2 * D + head * HD + vdq * 4) * 4;
    }
    const int kwr0 = k_off(kr, kc4 >> 1) + (kc4 & 1) * 8, kwr1 = k_off(kr + 16, kc4 >> 1) + (kc4 & 1) * 8;
    int vwr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int pos = v_keypos(2 * vkp);                                   // keys 2 vkp, 2 vkp + 1 stay adjacent (runs of 4 move as a whole)
        vwr[e] = v_off(4 * vdq + e, pos >> 4) + (pos & 15);
    }

    const int R0 = b * T;                                                    // first global row of the image (PLANES: tiles are global)
    const int kt0 = R0 >> 5;
    const bool tail_key = !PLANES && (T % KB) == 1;
    const int nkb = PLANES ? ((R0 + T - 1) >> 5) - kt0 + 1 : (tail_key ? T / KB : (T + KB - 1) / KB);

    f32x4 rk[2], rv[2];
    auto gload = [&](int kb) __attribute__((always_inline)) {
        const int so = (kb < nkb ? kb : nkb) * (KB * 4) * D3;               // past the last tile: zeros (never staged); clamped so that the look-ahead
                                                                            // tiles nkb + 1, nkb + 2 cannot overflow the int offset near the 2-GiB row limit
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, kvo[i], so, 0));
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kv_rsrc, vvo[i], so, 0));
        }
    };
    // staging piece p of 8: 0..3 the four K pairs (the 8-byte stores follow the second pair of a float4), 4..7 the four V^T dwords
    unsigned ksp[2][3];
    auto stage_piece = [&](auto P, char* kdst, char* vdst) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        if constexpr (p < 4) {
            constexpr int i = p >> 1, hf = p & 1;
            split2_rn3(rk[i][2 * hf], rk[i][2 * hf + 1], ksp[hf]);
            if constexpr (hf == 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u32x2 v; v[0] = ksp[0][k]; v[1] = ksp[1][k];
                    *reinterpret_cast<u32x2*>(kdst + k * KP + (i == 0 ? kwr0 : kwr1)) = v;
                }
            }
        } else {
            constexpr int e = p - 4;
            unsigned o[3];
            split2_rn3(rv[0][e], rv[1][e], o);
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<unsigned*>(vdst + k * VP + vwr[e]) = o[k];
        }
    };
    auto stage_all = [&](int kb) __attribute__((always_inline)) {
        char* kd = kbuf + (kb & 1) * KBUF;
        char* vd = vbuf + (kb % NVB) * VBUF;
        [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
            (stage_piece(std::integral_constant<int, P>{}, kd, vd), ...);
        }(std::make_integer_sequence<int, 8>{});
    };

    // PLANES: tile kb of this image = global tile kt0 + kb; its one-KB chunks (12 K, 15 V^T) are copied by the four waves in turn,
    // lane-linear (the image in global memory IS the swizzled LDS image).  M0 carries the LDS destination (saved / restored inside
    // the statement: it is compiler-reserved); completion is counted by the issuing wave (s_waitcnt vmcnt) before the tile barrier.
    auto dma = [&](int kb) __attribute__((always_inline)) {
        if constexpr (PLANES) {
            if (kb < nkb) {
                const char* src = planes + ((size_t)(kt0 + kb) * heads + head) * (size_t)(KBUF + VBUF) + lane * 16;
                const unsigned kdst = (unsigned)(size_t)(kbuf + (kb & 1) * KBUF), vdst = (unsigned)(size_t)(vbuf + (kb % NVB) * VBUF);
                constexpr int NCH = (KBUF + VBUF) / 1024;                      // one-KB chunks of a tile image (K planes, then V^T planes)
                static_assert((KBUF + VBUF) % 1024 == 0 && KBUF % 1024 == 0, "tile images are copied in 1-KB wave loads");
#pragma unroll
                for (int c = 0; c < (NCH + 3) / 4; ++c) {
                    const int chunk = wave + 4 * c;                          // wave-uniform
                    if (chunk >= NCH) break;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(chunk < KBUF / 1024 ? kdst + chunk * 1024 : vdst + (chunk - KBUF / 1024) * 1024);
                    const char* g = src + chunk * 1024;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
                }
            }
        }
    };

    // fragment reads: per-lane offsets formed once (the XOR swizzles), planes / d halves as immediates
    int kro[4], vro[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) kro[s] = k_off(l31, 2 * s + h);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) vro[ks] = v_off(l31, 2 * ks + h);
    auto rdk = [&](const char* kb_, int plane, int s) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(kb_ + plane * KP + kro[s]);
    };
    auto rdv = [&](const char* vb_, int plane, int dt, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(vb_ + plane * VP + dt * (32 * VPITCH) + vro[ks]);
    };

    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    f32x16 o0, o1, s0_, s1_;                                                // scores: the current tile's and the next tile's, roles alternate
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s0_[r] = 0.f; s1_[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f, alpha = 1.f, m_new = 0.f, ps = 0.f, mb = 0.f;
    u32x4 pp[3][2];                                                          // P planes: [term][ks]
    bf16x8 vf[3][2];                                                         // V^T fragments of the current ks: [term][dt]

    // ---- prologue: tiles 0 and 1 into LDS, tile 2 into the staging registers, S(0)
    if constexpr (PLANES) {
        dma(0);
        dma(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (nkb > 0) {
        gload(0);
        stage_all(0);
        gload(1);
        if (nkb > 1) stage_all(1);
        gload(2);
    }
    __syncthreads();
    if (wave_active && nkb > 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 k0 = rdk(kbuf, 0, s), k1 = rdk(kbuf, 1, s), k2 = rdk(kbuf, 2, s);
            s0_ = mfma(k0, qp[2][s], s0_); s0_ = mfma(k0, qp[1][s], s0_);
            s0_ = mfma(k0, qp[0][s], s0_); s0_ = mfma(k1, qp[1][s], s0_);
            s0_ = mfma(k1, qp[0][s], s0_); s0_ = mfma(k2, qp[0][s], s0_);
        }
    }
    __syncthreads();          // K buffer 0 is rewritten (tile 2) in phase B of iteration 0: every wave's reads of it are done

    // softmax pieces of the CURRENT tile's scores sc_ (interleaved with the MFMAs of the next tile's scores)
    auto soft_piece = [&](auto P, auto MASK, f32x16& sc_, int kb) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        if constexpr (p == 0) {
            if constexpr (PLANES) {
                if (decltype(MASK)::value) {                                 // the first / last global tile also holds rows of the neighbouring images
                    const int row0 = (kt0 + kb) * KB + 4 * h - R0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc_[r] = (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) < (unsigned)T ? sc_[r] : -INFINITY;
                }
            } else if (decltype(MASK)::value && !tail_key && kb == nkb - 1) {                               // a partial last tile: mask the keys past T
                const int key0 = kb * KB + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc_[r] = (key0 + (r & 3) + 8 * (r >> 2)) < T ? sc_[r] : -INFINITY;
            }
            mb = sc_[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mb = fmaxf(mb, sc_[r]);
        } else if constexpr (p == 1) {
            mb = xhalf_max(mb);
            m_new = fmaxf(m_run, mb);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);                   // exp2(-inf) = 0 on the first tile; every tile holds a valid key
            m_run = m_new;
            ps = 0.f;
        } else if constexpr (p < 6) {
            constexpr int c = p - 2;
#pragma unroll
            for (int r = 4 * c; r < 4 * c + 4; ++r) { sc_[r] = __builtin_amdgcn_exp2f(sc_[r] - m_new); ps += sc_[r]; }
        } else if constexpr (p == 6) {
            ps = xhalf_sum(ps);
            l_run = l_run * alpha + ps;
        } else {
            constexpr int j = p - 7;                                         // pair j: r = 2 j, 2 j + 1 -> K = 16 step j >> 2, dword j & 3
            unsigned o[3];
            split2_rn3(sc_[2 * j], sc_[2 * j + 1], o);
#pragma unroll
            for (int k = 0; k < 3; ++k) pp[k][j >> 2][j & 3] = o[k];
        }
    };

    // phase A of iteration kb: softmax + P split of tile kb, and (DOQK) S(kb + 1) into sa / sb from K buffer (kb + 1) & 1
    auto phase_a = [&](auto DOQK, auto MASK, f32x16& sc_, f32x16& sn, int kb) __attribute__((always_inline)) {
        constexpr bool doqk = decltype(DOQK)::value;
        const char* kb_ = kbuf + ((kb + 1) & 1) * KBUF;
        bf16x8 k0, k1, k2;
        if constexpr (doqk) {
            k0 = rdk(kb_, 0, 0); k1 = rdk(kb_, 1, 0); k2 = rdk(kb_, 2, 0);
        }
        soft_piece(std::integral_constant<int, 0>{}, MASK, sc_, kb);
        AX3_FENCE();
        [&]<int... S>(std::integer_sequence<int, S...>) __attribute__((always_inline)) {
            ([&] {
                if constexpr (doqk) { sn = mfma(k0, qp[2][S], S == 0 ? zero16 : sn); sn = mfma(k0, qp[1][S], sn); }
                soft_piece(std::integral_constant<int, 1 + 3 * S>{}, MASK, sc_, kb);
                AX3_FENCE();
                if constexpr (doqk) {
                    sn = mfma(k0, qp[0][S], sn);
                    if constexpr (S < 3) k0 = rdk(kb_, 0, S + 1);
                    sn = mfma(k1, qp[1][S], sn);
                }
                soft_piece(std::integral_constant<int, 2 + 3 * S>{}, MASK, sc_, kb);
                AX3_FENCE();
                if constexpr (doqk) {
                    sn = mfma(k1, qp[0][S], sn);
                    if constexpr (S < 3) k1 = rdk(kb_, 1, S + 1);
                    sn = mfma(k2, qp[0][S], sn);
                    if constexpr (S < 3) k2 = rdk(kb_, 2, S + 1);
                }
                soft_piece(std::integral_constant<int, 3 + 3 * S>{}, MASK, sc_, kb);
                AX3_FENCE();
            }(), ...);
        }(std::make_integer_sequence<int, 4>{});
        {   // the first V^T fragments of phase B: their LDS latency hides behind the last two split pieces
            const char* vb_ = vbuf + (kb % NVB) * VBUF;
#pragma unroll
            for (int k = 0; k < 3; ++k) { vf[k][0] = rdv(vb_, k, 0, 0); vf[k][1] = rdv(vb_, k, 1, 0); }
        }
        soft_piece(std::integral_constant<int, 13>{}, MASK, sc_, kb);
        soft_piece(std::integral_constant<int, 14>{}, MASK, sc_, kb);
        AX3_FENCE();
    };

    // phase B of iteration kb: O += V(kb)^T P(kb)^T, and (DOST) the split + LDS stores of tile kb + 2, then the loads of tile kb + 3
    auto phase_b = [&](auto DOST, int kb) __attribute__((always_inline)) {
        constexpr bool dost = decltype(DOST)::value;
        const char* vb_ = vbuf + (kb % NVB) * VBUF;
        char* kd = kbuf + (kb & 1) * KBUF;                                   // tile kb + 2 -> K buffer (kb + 2) & 1 = kb & 1
        char* vd = vbuf + ((kb + 2) % NVB) * VBUF;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        AX3_FENCE();
        auto extra = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;                            // MFMA group 0..11 of this phase
            if constexpr (dost && !PLANES) {
                if constexpr (g < 8) stage_piece(std::integral_constant<int, g>{}, kd, vd);
                if constexpr (g == 8) gload(kb + 3);
            }
            AX3_FENCE();
        };
        [&]<int... KS>(std::integer_sequence<int, KS...>) __attribute__((always_inline)) {
            ([&] {
                // x0-first order: a plane's fragment registers are refilled (next K = 16 step) right after its last product
                o0 = mfma(vf[0][0], pp[2][KS], o0); o1 = mfma(vf[0][1], pp[2][KS], o1);
                extra(std::integral_constant<int, 6 * KS + 0>{});
                o0 = mfma(vf[0][0], pp[1][KS], o0); o1 = mfma(vf[0][1], pp[1][KS], o1);
                extra(std::integral_constant<int, 6 * KS + 1>{});
                o0 = mfma(vf[0][0], pp[0][KS], o0); o1 = mfma(vf[0][1], pp[0][KS], o1);
                if constexpr (KS == 0) { vf[0][0] = rdv(vb_, 0, 0, 1); vf[0][1] = rdv(vb_, 0, 1, 1); }
                extra(std::integral_constant<int, 6 * KS + 2>{});
                o0 = mfma(vf[1][0], pp[1][KS], o0); o1 = mfma(vf[1][1], pp[1][KS], o1);
                extra(std::integral_constant<int, 6 * KS + 3>{});
                o0 = mfma(vf[1][0], pp[0][KS], o0); o1 = mfma(vf[1][1], pp[0][KS], o1);
                if constexpr (KS == 0) { vf[1][0] = rdv(vb_, 1, 0, 1); vf[1][1] = rdv(vb_, 1, 1, 1); }
                extra(std::integral_constant<int, 6 * KS + 4>{});
                o0 = mfma(vf[2][0], pp[0][KS], o0); o1 = mfma(vf[2][1], pp[0][KS], o1);
                if constexpr (KS == 0) { vf[2][0] = rdv(vb_, 2, 0, 1); vf[2][1] = rdv(vb_, 2, 1, 1); }
                extra(std::integral_constant<int, 6 * KS + 5>{});
            }(), ...);
        }(std::make_integer_sequence<int, 2>{});
    };

    auto tile_barrier = [&]() __attribute__((always_inline)) {
        if constexpr (PLANES) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's DMA pieces (issued a tile ago) + its LDS reads
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        AX3_FENCE();
    };
    // one iteration of the generic (peeled) form: roles of the two score registers by the parity of kb, S(kb + 1) only if it exists
    auto iteration = [&](int kb) __attribute__((always_inline)) {
        dma(kb + 2);
        if (kb & 1) { if (kb + 1 < nkb) phase_a(std::true_type{}, std::true_type{}, s1_, s0_, kb); else phase_a(std::false_type{}, std::true_type{}, s1_, s0_, kb); }
        else { if (kb + 1 < nkb) phase_a(std::true_type{}, std::true_type{}, s0_, s1_, kb); else phase_a(std::false_type{}, std::true_type{}, s0_, s1_, kb); }
        if (kb + 2 < nkb) phase_b(std::true_type{}, kb); else phase_b(std::false_type{}, kb);
        tile_barrier();
    };
    if (wave_active) {
        __builtin_amdgcn_s_setprio(1);
        int kb = 0;
        if constexpr (PLANES) {
            if (nkb > 0) { iteration(0); kb = 1; }                           // the first global tile is masked: peeled
            // full, unmasked iterations in pairs (kb odd): neither is the last tile
            for (; kb + 2 < nkb; kb += 2) {
                dma(kb + 2);
                phase_a(std::true_type{}, std::false_type{}, s1_, s0_, kb);
                phase_b(std::true_type{}, kb);
                tile_barrier();
                dma(kb + 3);
                phase_a(std::true_type{}, std::false_type{}, s0_, s1_, kb + 1);
                phase_b(std::true_type{}, kb + 1);
                tile_barrier();
            }
        } else {
            // full iterations in pairs: S(kb + 1) and the staging of tile kb + 2 both exist (kb + 2 < nkb for both halves)
            for (; kb + 3 < nkb; kb += 2) {
                phase_a(std::true_type{}, std::false_type{}, s0_, s1_, kb);
                phase_b(std::true_type{}, kb);
                tile_barrier();
                phase_a(std::true_type{}, std::false_type{}, s1_, s0_, kb + 1);
                phase_b(std::true_type{}, kb + 1);
                tile_barrier();
            }
        }
        for (; kb < nkb; ++kb) iteration(kb);                                // the last one to three tiles
        __builtin_amdgcn_s_setprio(0);
    } else {
        // a wave without queries (the last query block of a head) only stages
        for (int kb = 0; kb < nkb; ++kb) {
            if constexpr (PLANES) dma(kb + 2);
            else if (kb + 2 < nkb) { stage_all(kb + 2); gload(kb + 3); }
            tile_barrier();
        }
    }

    if (tail_key && wave_active) {
        // straggler key T - 1 in f32 VALU arithmetic; q is rebuilt exactly from its three terms
        const size_t ro = (rowbase + (T - 1)) * D3 + head * HD;
        float sx = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(qkv + ro + D + 16 * s + 8 * h);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(qkv + ro + D + 16 * s + 8 * h + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // element e of a plane dword pair: even -> low half, odd -> high half
                auto term = [&](int k) { const unsigned dw = qp[k][s][e >> 1]; return __uint_as_float((e & 1) ? (dw & 0xFFFF0000u) : (dw << 16)); };
                const float qv = (term(0) + term(1)) + term(2);
                sx += (e < 4 ? k0[e] : k1[e - 4]) * qv;
            }
        }
        sx = xhalf_sum(sx);
        const float mn = fmaxf(m_run, sx);
        const float al = __builtin_amdgcn_exp2f(m_run - mn);
        const float p = __builtin_amdgcn_exp2f(sx - mn);
        l_run = l_run * al + p;
        m_run = mn;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qkv + ro + 2 * D + 8 * g4 + 4 * h);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qkv + ro + 2 * D + 32 + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[g4 * 4 + e] = o0[g4 * 4 + e] * al + p * v0[e];
                o1[g4 * 4 + e] = o1[g4 * 4 + e] * al + p * v1[e];
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

}  // namespace ax3

// One plane tile: K planes [3][32 keys][64 d] then V^T planes [3][64 d][32 keys], bfloat16, in the swizzled LDS layout (k_off / v_off /
// v_keypos above) -- 27 KB per (global 32-row tile, head); tile (kt, head) starts at ((kt * heads) + head) * (KBUF + VBUF).
inline size_t kv_planes_bytes(int rows, int heads) { return (size_t)((rows + 31) / 32) * heads * (size_t)(ax3::KBUF + ax3::VBUF); }

#ifdef TSTAR_ATTN_X3_LAB
namespace ax3 {
// f32 qkv rows -> plane tiles (what the qkv GEMM's epilogue writes in the f32x3 mode; this kernel is the stand-alone form for the
// C-ABI entry / tests / the lab).  One thread per (row pair, head, 4 d): rows past `rows` are written as zeros.
__global__ __launch_bounds__(256) void kv_planes_kernel(const float* __restrict__ qkv, char* __restrict__ planes, int rows, int heads) {
    const int D = heads * HD, D3 = 3 * D;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int dq = (int)(gid % 16), head = (int)((gid / 16) % heads);
    const size_t rp = gid / (16 * (size_t)heads);                            // row pair
    const int row = (int)(rp * 2);
    if (row >= ((rows + 31) / 32) * 32) return;
    const int kt = row >> 5, key = row & 31;
    char* tile = planes + ((size_t)kt * heads + head) * (size_t)(KBUF + VBUF);
    f32x4 kx[2], vx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (row + i < rows) {
            kx[i] = *reinterpret_cast<const f32x4*>(qkv + (size_t)(row + i) * D3 + D + head * HD + dq * 4);
            vx[i] = *reinterpret_cast<const f32x4*>(qkv + (size_t)(row + i) * D3 + 2 * D + head * HD + dq * 4);
        } else {
            kx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            vx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                             // K: [key][d], d contiguous
        unsigned o0[3], o1[3];
        split2_rn3(kx[i][0], kx[i][1], o0);
        split2_rn3(kx[i][2], kx[i][3], o1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u32x2 v; v[0] = o0[k]; v[1] = o1[k];
            *reinterpret_cast<u32x2*>(tile + k * KP + k_off(key + i, dq >> 1) + (dq & 1) * 8) = v;
        }
    }
    const int pos = v_keypos(key);                                            // V^T: [d][key], the pair (key, key + 1) stays adjacent
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned o[3];
        split2_rn3(vx[0][e], vx[1][e], o);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<unsigned*>(tile + KBUF + k * VP + v_off(4 * dq + e, pos >> 4) + (pos & 15)) = o[k];
    }
}
}  // namespace ax3
#endif

// launches the kernel on `s`; returns 0, or a hipError_t value.  `planes` = nullptr: K / V are split from the f32 rows in the kernel
// (the library's form).  The plane-tile + LDS-DMA form is compiled for the lab only (TSTAR_ATTN_X3_LAB): measured at the bench shape
// it is 7 % faster (159.7 vs 148.8 TFLOP/s, profiles/r05_attention_x3_planes_dma_lab.log) -- about 1 % of a step, less than what
// writing the tiles from the qkv GEMM's epilogue would cost -- so the library does not use it.
inline int attention_x3_launch(const float* qkv, float* out, int B, int T, int heads, hipStream_t s, const char* planes = nullptr) {
    const int qtiles = (T + 127) / 128;
#ifdef TSTAR_ATTN_X3_LAB
    // the lab harness is one device, one host thread: a process-wide flag is enough there.  The LIBRARY raises the dynamic-LDS
    // limit through ensure_dyn_lds (keyed on kernel AND device, mutex-protected: the attribute applies to the current device only)
    // before it calls this launcher -- csrc/attention_x3.hip.
    static bool attr[2] = {false, false};
    if (planes) {
        if (!attr[1]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ax3::attention_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ax3::LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr[1] = true;
        }
        hipLaunchKernelGGL(ax3::attention_x3_kernel<true>, dim3(B * heads * qtiles), dim3(256), ax3::LDS_BYTES, s, qkv, planes, out, T, heads, qtiles);
        return (int)hipGetLastError();
    }
    if (!attr[0]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ax3::attention_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ax3::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr[0] = true;
    }
#endif
    if (planes) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ax3::attention_x3_kernel<false>, dim3(B * heads * qtiles), dim3(256), ax3::LDS_BYTES, s, qkv, planes, out, T, heads, qtiles);
    return (int)hipGetLastError();
}

#ifdef TSTAR_ATTN_X3_LAB
// f32 qkv [rows, 3 * heads * 64] -> plane tiles (kv_planes_bytes(rows, heads) bytes)
inline int kv_planes_launch(const float* qkv, char* planes, int rows, int heads, hipStream_t s) {
    const size_t n = (size_t)((rows + 31) / 32) * 16 * 16 * heads;           // (row pairs of whole tiles) x heads x 16 float4 columns
    hipLaunchKernelGGL(ax3::kv_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, qkv, planes, rows, heads);
    return (int)hipGetLastError();
}
#endif

}  // namespace tstar
