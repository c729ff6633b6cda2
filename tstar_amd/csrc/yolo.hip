// YOLO-World-v2 image path on gfx950 WITHOUT the matrix cores (BASELINE configs[3]: "CSPDarknet conv path, no
// MFMA"): the second detector backend behind the reference's heuristic plug-in surface.
//
// Replaces YoloWorldInterface.inference_detector (/root/reference/TStar/interface_heuristic.py:136-168: mmdet test
// pipeline on images[0], model.test_step, score > 0.12, top-50) and, for the searcher's fast path, the
// detection -> grid-cell loop of TStarSearcher.imageGridScoreFunction (interface_searcher.py:129-150).  The model
// source is NOT part of the reference tree (dangling symlink; mmdet/mmyolo absent): the architecture is restated from
// the published YOLO-World-v2 design and its parity is UNPINNED -- see oracle/yolo_ref.py; tests compare this file
// against that independent CPU statement on seeded weights.
//
// The network is data: tstar_amd/yolo_world.py flattens it into a float32 blob (BatchNorm folded) and a table of ops
// over NHWC activation buffers; this file interprets the table.
//   conv_valu_kernel     implicit-GEMM convolution on the f32 VALU (fmaf, no MFMA): 128 pixels x 64 channels per
//                        workgroup, 8 x 4 outputs per lane, K = (ky, kx, ci) staged through LDS in slices of 16 with
//                        ds_read_b128 fragment reads; fused bias / SiLU / residual / attention-gate epilogue; reads and
//                        writes at channel offsets so torch.cat never copies.  Bound: f32 VALU issue (157.3 TFLOP/s
//                        spec; tools/lab/pkfma_rate.hip measures 140 for a pure v_pk_fma_f32 stream at 8 waves / SIMD and
//                        108 at the 2 waves / SIMD a 128-accumulator tile leaves) shared with the LDS pipe.
//   conv_sw_kernel       the same convolution with the weights as SCALAR operands (s_load_dwordx16 rows of the transposed
//                        matrix feeding v_pk_fma_f32 from SGPRs): a quarter of the LDS traffic per FMA; used for the
//                        layers with >= 400 workgroups of 512 pixels x 64 channels (see the kernel's comment).
//   conv_halo_kernel     its form for 3x3 / stride-1 layers: an 8 x 40 output patch per workgroup, the 10 x 42 halo patch of a
//                        16-channel slice staged once and read by all nine taps.
//   pool5 / upcopy / attn / letterbox / head_decode  HBM-bound elementwise & small reductions (wave shuffles).
//   sort_nms_kernel      per image: bitonic sort of the (score, anchor, class) candidates, class-aware greedy NMS run by
//                        ONE wavefront (kept boxes in LDS, lanes test them in parallel, __ballot decides), top-k.
#include "../../include/tstar_hip.h"
#include "common.h"
#include "heads.h"
#include "prof.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace tstar {

enum { OP_CONV = 0, OP_POOL5 = 1, OP_UPCOPY = 2, OP_ATTN = 3 };
enum { YACT_NONE = 0, YACT_SILU = 1 };
enum { MODE_PLAIN = 0, MODE_RESIDUAL = 1, MODE_ATTN_MUL = 2 };
constexpr int YOLO_IMG = 640;
constexpr int YOLO_REG_MAX = 16;
constexpr int YOLO_TEXT = 512;
constexpr int YOLO_MAX_Q = TSTAR_OWL_MAX_QUERIES;
constexpr int YOLO_SETS = TSTAR_OWL_MAX_SETS;
constexpr int YOLO_NMS_PRE = 30000, YOLO_MAX_PER_IMG = 300;
constexpr float YOLO_IOU_THR = 0.7f, YOLO_SCORE_THR = 0.001f;

struct ConvArgs {
    const float* src; int src_ld, src_off, cin, H, W;       // input NHWC [B,H,W,src_ld], channels [src_off, src_off+cin)
    float* dst; int dst_ld, dst_off, cout, Ho, Wo;
    const float* w;                                          // [cout][ks*ks*cin]
    const float* wt;                                         // the same matrix transposed, [ks*ks*cin][cout] (scalar-weight kernel)
    const float* bias;                                       // [cout] or null
    unsigned zoff;                                           // element offset (from src) of the zero quad kept behind the source buffer
    const float* aux; int aux_ld, aux_off;                   // residual tensor (same spatial size as dst) or attention [P, heads]
    int ks, stride, act, mode, heads;
    int M;                                                   // B * Ho * Wo
};

// K is walked in slices of 16 input channels, the ks*ks taps of a slice back to back (the nine shifted reads of one
// 16-channel activation slice then hit the CU's L1: +2 % over tap-major order).  Every conv kernel uses this order, so
// their outputs are bit-identical.  Returns the (tap, first input channel) of slice k0 / 16; the weight row of its first
// k is tap * cin + ci.
template <int KS>
__device__ __forceinline__ void slice_pos(int k0, int& tap, int& ci) {
    const int sl = k0 / 16;
    tap = sl % (KS * KS);
    ci = (sl / (KS * KS)) * 16;
}

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

// ------------------------------------------------------------------ implicit-GEMM conv on the VALU
// Workgroup tile 16 TM pixels x BN channels (TM = 8 or 4: 128 or 64 pixels; BN = 64 or 128), K slices of 16; a lane owns
// TM pixels x TN channels (TN = 4 or 8).  The 64-pixel tile serves layers with fewer than 512 128-pixel tiles (small maps,
// small batches): twice the workgroups, so CUs hold two that overlap each other's barriers and load latency (L model
// forward: B = 1 15.7 -> 10.1 ms, B = 4 19.3 -> 16.2 ms).  Per k the lane reads 2 + TN/4 LDS quads for 8 TN FMAs: with TN = 8 the LDS pipe (shared by the CU's four
// SIMDs) carries 4 reads per 64 FMAs instead of 3 per 32, which is what keeps the VALU fed on the >= 128-channel layers.
// LDS rows k >= 8 are rotated by 8 floats: the transposing ds_write_b32 of lanes with k-quad 0 / 2 (and 1 / 3) would
// otherwise land on the same banks (row stride = 16 mod 32 banks); the b128 fragment reads stay 16-byte aligned.
constexpr int CBK = 16;

template <int KS, int TN, int TM>
__global__ __launch_bounds__(256) void conv_valu_kernel(ConvArgs a, int mt, int nt) {
    constexpr int BN = 16 * TN, CLD_W = BN + 4, NWQ = BN / 64;       // NWQ weight float4 per thread per slice
    constexpr int CBM = 16 * TM, CLD_A = CBM + 4, NAQ = CBM / 64;    // TM = 8 (4): 128 (64) pixels per workgroup, NAQ activation float4 per thread
    constexpr int DEPTH = TN == 8 ? 1 : 3;                           // register prefetch depth in slices (the 8 x 8 tile has no registers to spare)
    __shared__ __attribute__((aligned(16))) float As[CBK][CLD_A];
    __shared__ __attribute__((aligned(16))) float Ws[CBK][CLD_W];
    const int tid = threadIdx.x;
    // 1-D grid, XCD-aware: an XCD gets a contiguous run of tiles, the channel tiles of one pixel tile adjacent (shared L2)
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * CBM, n0 = (tile % nt) * BN;
    // staging roles: A tile 128 x 16 = 512 float4 -> 2 per thread (rows tid/4 and tid/4 + 64, k quad tid%4);
    //                W tile  BN x 16 -> NWQ per thread (rows tid/4 (+64))
    const int kq = (tid & 3) * 4;
    const int rot = (kq >> 3) * 8;                                   // column rotation of LDS rows k >= 8
    const int ar0 = tid >> 2;
    int ab[NAQ], ay[NAQ], ax[NAQ];
    bool aval[NAQ];
#pragma unroll
    for (int r = 0; r < NAQ; ++r) {
        const int m = m0 + ar0 + r * 64;
        aval[r] = m < a.M;
        const int mm = aval[r] ? m : 0;
        const int hw = a.Ho * a.Wo;
        ab[r] = mm / hw;
        const int rem = mm - ab[r] * hw;
        ay[r] = (rem / a.Wo) * a.stride - (KS >> 1);
        ax[r] = (rem % a.Wo) * a.stride - (KS >> 1);
    }
    const int K = KS * KS * a.cin;
    const float* wrow[NWQ];
#pragma unroll
    for (int r = 0; r < NWQ; ++r) {
        const int wn = n0 + ar0 + r * 64;
        wrow[r] = a.w + (size_t)(wn < a.cout ? wn : 0) * K + kq;   // rows >= cout read row 0: their outputs are never stored
    }

    // compute roles: 8 rows x TN cols per lane
    const int tx = tid & 15, ty = tid >> 4;                          // cols tx*4 (+64).., rows ty*8..
    float acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;

    // global -> register staging, DEPTH = three slices deep (one for the 8 x 8 tile): a small layer runs one workgroup (or none) per CU, so nothing else hides
    // the L2 latency of a slice's loads; issued three slices ahead they have ~1.5 us to land.  Loads are unconditional
    // (a padded tap / out-of-range row reads a valid dummy address and is zeroed when the slice is stored; past the last
    // slice, slice 0 is re-read and dropped), which keeps them back to back and lets the in-order vmcnt wait be exact.
    struct Stage { f32x4 ra[NAQ], rw[NWQ]; unsigned ok; };
    auto load_tile = [&](Stage& st, int k0) {
        // k0 is a multiple of 16 and cin % 16 == 0, so the 16-wide slice stays inside one (ky, kx) tap
        if (k0 >= K) k0 = 0;
        int tap, ci0;
        slice_pos<KS>(k0, tap, ci0);
        const int ci = ci0 + kq, krow = tap * a.cin + ci0;
        const int ky = KS == 1 ? 0 : tap / KS, kx = KS == 1 ? 0 : tap - ky * KS;
        st.ok = 0;
#pragma unroll
        for (int r = 0; r < NAQ; ++r) {
            const int iy = ay[r] + ky, ix = ax[r] + kx;
            const bool ok = aval[r] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const size_t off = ok ? ((size_t)(ab[r] * a.H + iy) * a.W + ix) * a.src_ld + a.src_off + ci : (size_t)0;
            st.ra[r] = *reinterpret_cast<const f32x4*>(a.src + off);
            st.ok |= ok ? 1u << r : 0u;
        }
#pragma unroll
        for (int r = 0; r < NWQ; ++r) st.rw[r] = *reinterpret_cast<const f32x4*>(wrow[r] + krow);   // rows >= cout read row 0: their outputs are never stored
    };
    auto store_tile = [&](const Stage& st) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NAQ; ++r) {
            const f32x4 v = ((st.ok >> r) & 1) ? st.ra[r] : z;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[kq + e][(ar0 + r * 64 + rot) & (CBM - 1)] = v[e];
        }
#pragma unroll
        for (int r = 0; r < NWQ; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[kq + e][(ar0 + r * 64 + rot) & (BN - 1)] = st.rw[r][e];
    };
    auto step = [&](Stage& st, int k0) {
        __syncthreads();
        store_tile(st);
        __syncthreads();
        load_tile(st, k0 + DEPTH * CBK);
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            const int rk = (k >> 3) * 8;                             // compile-time after unrolling
            f32x4 av[TM / 4];
#pragma unroll
            for (int h2 = 0; h2 < TM / 4; ++h2) av[h2] = *reinterpret_cast<const f32x4*>(&As[k][(ty * TM + 4 * h2 + rk) & (CBM - 1)]);
            f32x4 w4[TN / 4];
#pragma unroll
            for (int q = 0; q < TN / 4; ++q) w4[q] = *reinterpret_cast<const f32x4*>(&Ws[k][(tx * 4 + q * 64 + rk) & (BN - 1)]);
#pragma unroll
            for (int q = 0; q < TN / 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int h2 = 0; h2 < TM / 4; ++h2) acc[4 * h2 + i][q * 4 + j] = fmaf(av[h2][i], w4[q][j], acc[4 * h2 + i][q * 4 + j]);
                    }
                }
        }
    };
    if constexpr (DEPTH == 3) {
        Stage s0, s1, s2;
        load_tile(s0, 0);
        load_tile(s1, CBK);
        load_tile(s2, 2 * CBK);
        for (int k0 = 0;;) {                                         // the back edge always follows step(s2, ..)
            step(s0, k0); k0 += CBK; if (k0 >= K) break;
            step(s1, k0); k0 += CBK; if (k0 >= K) break;
            step(s2, k0); k0 += CBK; if (k0 >= K) break;
        }
    } else {
        Stage s0;
        load_tile(s0, 0);
        for (int k0 = 0; k0 < K; k0 += CBK) step(s0, k0);
    }
    // epilogue: bias, activation, residual (after the activation: DarknetBottleneck adds the identity last) or the
    // max-sigmoid attention gate (after project_conv's BatchNorm, no activation)
    const int ch_per_head = a.mode == MODE_ATTN_MUL ? a.cout / a.heads : 1;
#pragma unroll
    for (int q = 0; q < TN / 4; ++q) {
        const int nb = n0 + tx * 4 + q * 64;
        if (nb >= a.cout) continue;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + nb);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + ty * TM + i;
            if (m >= a.M) continue;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[i][q * 4 + j] + bias[j];
                if (a.act == YACT_SILU) t = silu(t);
                v[j] = t;
            }
            if (a.mode == MODE_RESIDUAL) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(a.aux + (size_t)m * a.aux_ld + a.aux_off + nb);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += r[j];
            } else if (a.mode == MODE_ATTN_MUL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= a.aux[(size_t)m * a.aux_ld + a.aux_off + (nb + j) / ch_per_head];
            }
            *reinterpret_cast<f32x4*>(a.dst + (size_t)m * a.dst_ld + a.dst_off + nb) = v;
        }
    }
}

// ------------------------------------------------------------------ scalar-weight conv on the VALU
// The tile kernel above needs 16 LDS floats (8 pixels + 8 weights) per 64 FMAs and a lane; with the CU's four SIMDs on
// one LDS pipe that is as many LDS cycles as packed-FMA cycles, and neither pipe gets past half rate.  Here the weights
// never touch LDS or VGPRs: every lane of a wave works on the SAME 16 output channels, so the weight row of a k is
// wave-uniform -- one s_load_dwordx16 from the transposed matrix [K][cout] into SGPRs -- and feeds v_pk_fma_f32 as its
// scalar operand (channel pairs are the packed halves, the pixel value is broadcast with op_sel).  A lane owns P = 8 (or
// 4) pixels x 16 channels (128 / 64 accumulators: 2 / 4 waves per SIMD); the four waves of a workgroup share one 64 P-pixel
// x 16-k activation tile in LDS and take one 16-channel group each.  Per k and wave (P = 8): 2 ds_read_b128 + 1 s_load for 64
// v_pk_fma_f32, a quarter of the LDS traffic per FMA of the tile kernel.  Scalar loads are issued in batches of two k with the
// (out-of-order, hence lgkmcnt(0)) wait placed before the next batch; padded taps read a zero quad kept behind every
// activation buffer instead of branching.  The accumulation order over k is the same
// sequential fmaf chain (slice_pos), so all conv kernels give bit-identical outputs (tested).  Measured on the way (L model,
// B = 32 forward, ms): tile kernel only 80.5; + this kernel 73.3 (8 pixels per lane) / 70.0 (per-layer 4 or 8); DMA staging
// + slice-major K 68.3; 32-bit saddr addressing 67.0.  No gain: a second register stage of global prefetch, touching the
// next batch's weight rows to pre-load the scalar cache, 64-byte aligned rows alone.
constexpr int SWK = 16, SWN = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// acc.xy += a.x * w.xy  /  acc.xy += a.y * w.xy   (w in an SGPR pair)
__device__ __forceinline__ void pkfma_lo(f32x2& acc, f32x2 av, f32x2 w) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(av), "s"(w)); }
__device__ __forceinline__ void pkfma_hi(f32x2& acc, f32x2 av, f32x2 w) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(av), "s"(w)); }

template <int P>
__device__ __forceinline__ void conv_sw_epilogue(const ConvArgs& a, f32x2 (&acc)[P][8], int m0, int cg0, int lane) {
    // as in the tile kernel; a lane writes 16 consecutive channels of each of its P pixels
    const int ch_per_head = a.mode == MODE_ATTN_MUL ? a.cout / a.heads : 1;
    float bias[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) bias[c] = a.bias ? a.bias[cg0 + c] : 0.f;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int m = m0 + lane + 64 * j;
        if (m >= a.M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[j][2 * q + (e >> 1)][e & 1] + bias[4 * q + e];
                if (a.act == YACT_SILU) t = silu(t);
                v[e] = t;
            }
            const int nb = cg0 + 4 * q;
            if (a.mode == MODE_RESIDUAL) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(a.aux + (size_t)m * a.aux_ld + a.aux_off + nb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r[e];
            } else if (a.mode == MODE_ATTN_MUL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= a.aux[(size_t)m * a.aux_ld + a.aux_off + (nb + e) / ch_per_head];
            }
            *reinterpret_cast<f32x4*>(a.dst + (size_t)m * a.dst_ld + a.dst_off + nb) = v;
        }
    }
}

// The activation tile is staged by direct global -> LDS DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write
// phase) into a double-buffered tile: ONE barrier per slice, and the loads of slice s+1 land in the other
// buffer while slice s is computed.  A wave instruction fills 1 KB of LDS contiguously (lane L at byte 16 L), so tile rows
// are un-padded 64-byte pixel rows and the bank-conflict swizzle is applied on the GLOBAL side: slot q of pixel p holds
// k-quad q ^ ((p >> 2) & 3); the fragment read of k-quad kq then takes slot kq ^ ((lane >> 2) & 3), and the 16 lanes of a
// ds_read_b128 group (16 consecutive pixels) hit 16 distinct 16-byte slots.
template <int KS, int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P == 8 ? 2 : 4, P == 8 ? 2 : 4))) void conv_sw_kernel(ConvArgs a, int mt, int nt) {
    constexpr int SWM = 64 * P;
    __shared__ __attribute__((aligned(16))) float As[2][SWM][SWK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * SWM;
    const int cg0 = (tile % nt) * SWN + wave * 16;
    const bool active = cg0 < a.cout;
    // staging roles: DMA instruction i of wave w fills pixels (4 i + w) 16 .. +15; lane L -> pixel (L >> 2), slot L & 3
    unsigned poff[P], pval[P];
    {
        const int hw = a.Ho * a.Wo;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int pl = (4 * i + wave) * 16 + (lane >> 2);              // pixel within the tile
            const int kq = (lane & 3) ^ ((pl >> 2) & 3);                   // the k-quad this slot holds
            const int m = m0 + pl;
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            const int y0 = oy * a.stride - (KS >> 1), x0 = ox * a.stride - (KS >> 1);
            poff[i] = 4u * (unsigned)(((b * a.H + y0) * a.W + x0) * a.src_ld + a.src_off + 4 * kq);   // BYTE offset from a.src; wraps for y0 / x0 = -1, only used with a valid tap
            unsigned v = 0;
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                const int iy = y0 + t / KS, ix = x0 + t % KS;
                v |= (unsigned)(ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) << t;
            }
            pval[i] = v;
        }
    }
    const int K = KS * KS * a.cin;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto stage = [&](int k0, int buf) {
        int tap, ci;
        slice_pos<KS>(k0, tap, ci);
        const int ky = KS == 1 ? 0 : tap / KS, kx = KS == 1 ? 0 : tap - ky * KS;
        const unsigned toff = 4u * (unsigned)((ky * a.W + kx) * a.src_ld + ci), tbit = 1u << tap, zoffb = 4u * a.zoff;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            // scalar base + 32-bit byte offset (zoff < 2^30 elements, launcher): bit test, add, select per load
            const unsigned off = (pval[i] & tbit) ? poff[i] + toff : zoffb;
            const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)&As[buf][(4 * i + wave) * 16][0]);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(a.src), "s"(la) : "memory");
        }
    };
    f32x2 acc[P][8];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = f32x2{0.f, 0.f};
    typedef __attribute__((address_space(4))) const f32x16 cw16;
    const float* wcol = a.wt + (active ? cg0 : 0);
    const int xs = ((lane >> 2) & 3) * 4;                                  // read-side swizzle, in floats

    stage(0, 0);
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += SWK) {
        int wtap, wci;
        slice_pos<KS>(k0, wtap, wci);
        const float* wrow0 = wcol + (size_t)(wtap * a.cin + wci) * a.cout;   // weight row of the slice's first k
        f32x16 wna = *(const cw16*)(unsigned long long)(wrow0);
        f32x16 wnb = *(const cw16*)(unsigned long long)(wrow0 + a.cout);
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): this wave's DMA of slice k0 has landed
        __syncthreads();                                                    // everyone's has, and nobody still reads the other buffer
        if (k0 + SWK < K) stage(k0 + SWK, cur ^ 1);
        if (active) {
            const float* ab = &As[cur][lane][0];
            f32x4 av[P], an[P];
#pragma unroll
            for (int j = 0; j < P; ++j) av[j] = *reinterpret_cast<const f32x4*>(ab + j * 64 * SWK + (0 ^ xs));
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    asm volatile("" :: "s"(wna[0]), "s"(wnb[0]));
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x16 wa = wna, wb = wnb;
                    const int kn = kq * 4 + kp * 2 + 2;
                    if (kn < SWK) {
                        wna = *(const cw16*)(unsigned long long)(wrow0 + (size_t)kn * a.cout);
                        wnb = *(const cw16*)(unsigned long long)(wrow0 + (size_t)(kn + 1) * a.cout);
                    }
                    if (kp == 0 && kq < 3) {
#pragma unroll
                        for (int j = 0; j < P; ++j) an[j] = *reinterpret_cast<const f32x4*>(ab + j * 64 * SWK + ((4 * (kq + 1)) ^ xs));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x2 w2 = {wa[2 * c], wa[2 * c + 1]};
#pragma unroll
                        for (int j = 0; j < P; ++j) pkfma_lo(acc[j][c], kp ? f32x2{av[j][2], av[j][3]} : f32x2{av[j][0], av[j][1]}, w2);
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x2 w2 = {wb[2 * c], wb[2 * c + 1]};
#pragma unroll
                        for (int j = 0; j < P; ++j) pkfma_hi(acc[j][c], kp ? f32x2{av[j][2], av[j][3]} : f32x2{av[j][0], av[j][1]}, w2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kq < 3) {
#pragma unroll
                    for (int j = 0; j < P; ++j) av[j] = an[j];
                }
            }
        }
        cur ^= 1;
    }
    if (!active) return;
    conv_sw_epilogue<P>(a, acc, m0, cg0, lane);
}

// ------------------------------------------------------------------ scalar-weight 3x3 conv with a halo tile
// For 3x3 / stride-1 layers: a workgroup owns a TH x TW patch of output pixels (320 of them: 8 x 40 on the 160-, 80- and
// 40-wide maps of a 640 x 640 input, 16 x 20 on the 20-wide maps) and stages the patch's halo of a 16-channel slice ONCE
// (9 DMA instructions per wave, out-of-image pixels from the zero quad); the nine taps then read shifted windows of it --
// every shift is an immediate ds_read offset -- instead of re-loading a shifted tile per tap as conv_sw_kernel does: 6.5x
// fewer DMA loads and two barriers per 144 k instead of one per 16.  Halo pixels are 80-byte rows (16 floats + one pad
// quad, filled by a fifth dummy lane so that a DMA instruction still writes 1 KB contiguously): consecutive pixels at an
// 80-byte stride make the ds_read_b128 of a 16-lane group conflict-free.  A lane owns 5 pixels x NCH channels.
//
// Patches are cut from the batch's images STACKED row-wise (row R = b * H + y; NHWC makes pixel (R, x) element R * W + x), so
// a map whose height is not a multiple of TH (20 rows under 16-row patches) still tiles without ragged patches: a patch may
// then span the boundary between two images, and the LDS halo gets ONE extra all-zero row at that boundary (XROW) -- it is the
// bottom padding of the upper image and the top padding of the lower one at once; the output rows below it just sit one
// halo row lower (a per-lane constant folded into the lane's base offset), so the tap loop is unchanged.
//
// NCH = 16 (3 waves per SIMD) is the form for layers with enough workgroups; NCH = 8 halves a wave's channel group (a
// workgroup covers 32 channels: twice the workgroups, 40 accumulators: 4 waves per SIMD) for the small layers and small
// batches, where 16-channel waves leave most SIMDs with one wave or none.  Same (ci slice, tap, k) fmaf order as every other
// conv kernel: bit-identical outputs.
constexpr int HLD = SWK + 4;
template <int TH, int TW, bool XROW> struct HaloGeom {
    static constexpr int HW2 = TW + 2, NROW = TH + 2 + (XROW ? 1 : 0), NP = NROW * HW2, P = TH * TW / 64;
    // LDS row pitch: HW2 pixels of 5 quads (4 data + 1 pad) plus 6 pad quads = 96 B.  A lane group of a ds_read_b128 covers 16
    // CONSECUTIVE patch pixels, which wrap from one patch row into the next (TW = 40 or 20 pixels per row); 16 pixels at 80 B
    // cover the 64 banks exactly once only if pixel 0 of the next row sits where pixel TW of this row would, i.e. the pitch is
    // TW * 80 B modulo the 256-B bank period: (TW + 2) * 80 + 96 (round 3 measured SQ_LDS_BANK_CONFLICT / SQ_BUSY_CYCLES = 0.51
    // on this kernel with the unpadded rows: every group that straddled two rows collided on two banks)
    static constexpr int RQ = HW2 * 5 + 6, RPF = RQ * 4;
    static constexpr int NDMA = (NROW * RQ + 63) / 64, NIT = (NDMA + 3) / 4, LDSF = NDMA * 256 + HLD;
    static_assert(TH * TW % 64 == 0, "a halo patch is a whole number of pixels per lane");
};
template <int N> struct WRow;
template <> struct WRow<16> { typedef f32x16 T; };
template <> struct WRow<8> { typedef float T __attribute__((ext_vector_type(8))); };

template <int TH, int TW, int NCH, bool XROW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NCH == 16 ? 3 : 4, NCH == 16 ? 3 : 4))) void conv_halo_kernel(ConvArgs a, int mt, int nt) {
    using G = HaloGeom<TH, TW, XROW>;
    constexpr int HW2 = G::HW2, HP = G::P, NIT = G::NIT, NDMA = G::NDMA, NC2 = NCH / 2;
    constexpr int RQ = G::RQ, RPF = G::RPF;
    __shared__ __attribute__((aligned(16))) float Hs[G::LDSF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int cg0 = (tile % nt) * (4 * NCH) + wave * NCH;
    const bool active = cg0 < a.cout;
    const int tpr = a.W / TW;
    const int rows_total = a.M / a.W;                                 // B * H stacked rows
    const int pt = tile / nt, row0 = (pt / tpr) * TH, x0 = (pt % tpr) * TW;
    // image boundary strictly inside the patch (XROW only; the launcher guarantees H >= TH, so there is at most one):
    // local row rb is the first row of the next image
    int rb = 1 << 20;
    if (XROW) {
        const int rem = row0 % a.H;
        if (rem != 0 && a.H - rem < TH) rb = a.H - rem;
    }
    const bool nb = XROW && rb < TH;
    const int last = TH + 1 + (nb ? 1 : 0);                           // LDS row of the bottom halo
    // staging roles: DMA instruction i = wave + 4 it fills LDS quads 64 i .. 64 i + 63; quad s = halo row s / RQ, and inside the row
    // pixel (s % RQ) / 5, quad (s % RQ) % 5 (quad 4 of a pixel and the six quads behind the last pixel are padding)
    unsigned hoff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int sidx = (wave + 4 * it) * 64 + lane;
        const int hy = sidx / RQ, rem = sidx - hy * RQ;
        const int hx = rem / 5, q = rem - hx * 5;
        const int sr = row0 + hy - 1 - ((nb && hy > rb + 1) ? 1 : 0);  // stacked source row of LDS row hy
        const int ix = x0 + hx - 1;
        bool ok = hy < G::NROW && hx < HW2 && q < 4 && hy <= last && ix >= 0 && ix < a.W && sr >= 0 && sr < rows_total;
        if (nb && hy == rb + 1) ok = false;                            // the zero row between two images
        if (hy == 0 && row0 % a.H == 0) ok = false;                    // top halo above an image's first row
        if (hy == last && (row0 + TH) % a.H == 0) ok = false;          // bottom halo below an image's last row
        hoff[it] = ok ? 4u * (unsigned)((sr * a.W + ix) * a.src_ld + a.src_off + 4 * q) : 0xffffffffu;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned zoffb = 4u * a.zoff;
    auto stage = [&](int ci0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (wave + 4 * it < NDMA) {                                    // wave-uniform
                const unsigned off = hoff[it] == 0xffffffffu ? zoffb : hoff[it] + 4u * (unsigned)ci0;
                const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)&Hs[0]) + (unsigned)(wave + 4 * it) * 1024u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(a.src), "s"(la) : "memory");
            }
        }
    };
    // compute roles: lane -> pixels p = lane + 64 j of the patch; hb[j] = its top-left tap in the halo patch (floats)
    int hb[HP];
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int p = lane + 64 * j, r = p / TW, c = p - r * TW;
        hb[j] = (r + ((nb && r >= rb) ? 1 : 0)) * RPF + c * HLD;
    }
    f32x2 acc[HP][NC2];
#pragma unroll
    for (int j = 0; j < HP; ++j)
#pragma unroll
        for (int c = 0; c < NC2; ++c) acc[j][c] = f32x2{0.f, 0.f};
    typedef typename WRow<NCH>::T wrow_t;
    typedef __attribute__((address_space(4))) const wrow_t cwrow;
    const float* wcol = a.wt + (active ? cg0 : 0);
    const float* hs = &Hs[0];

    for (int ci0 = 0; ci0 < a.cin; ci0 += SWK) {
        __syncthreads();                                                    // everybody is done with the previous slice's patch
        stage(ci0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
        __syncthreads();
        if (!active) continue;
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int toff = ky * RPF + kx * HLD;                            // floats
            const float* wrow0 = wcol + (size_t)(tap * a.cin + ci0) * a.cout;
            wrow_t wna = *(cwrow*)(unsigned long long)(wrow0);
            wrow_t wnb = *(cwrow*)(unsigned long long)(wrow0 + a.cout);
            f32x4 av[HP], an[HP];
#pragma unroll
            for (int j = 0; j < HP; ++j) av[j] = *reinterpret_cast<const f32x4*>(hs + hb[j] + toff);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    asm volatile("" :: "s"(wna[0]), "s"(wnb[0]));
                    __builtin_amdgcn_sched_barrier(0);
                    const wrow_t wa = wna, wb = wnb;
                    const int kn = kq * 4 + kp * 2 + 2;
                    if (kn < SWK) {
                        wna = *(cwrow*)(unsigned long long)(wrow0 + (size_t)kn * a.cout);
                        wnb = *(cwrow*)(unsigned long long)(wrow0 + (size_t)(kn + 1) * a.cout);
                    }
                    if (kp == 0 && kq < 3) {
#pragma unroll
                        for (int j = 0; j < HP; ++j) an[j] = *reinterpret_cast<const f32x4*>(hs + hb[j] + toff + 4 * (kq + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NC2; ++c) {
                        const f32x2 w2 = {wa[2 * c], wa[2 * c + 1]};
#pragma unroll
                        for (int j = 0; j < HP; ++j) pkfma_lo(acc[j][c], kp ? f32x2{av[j][2], av[j][3]} : f32x2{av[j][0], av[j][1]}, w2);
                    }
#pragma unroll
                    for (int c = 0; c < NC2; ++c) {
                        const f32x2 w2 = {wb[2 * c], wb[2 * c + 1]};
#pragma unroll
                        for (int j = 0; j < HP; ++j) pkfma_hi(acc[j][c], kp ? f32x2{av[j][2], av[j][3]} : f32x2{av[j][0], av[j][1]}, w2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kq < 3) {
#pragma unroll
                    for (int j = 0; j < HP; ++j) av[j] = an[j];
                }
            }
        }
    }
    if (!active) return;
    // epilogue (as conv_sw_epilogue, with the patch's pixel mapping)
    const int ch_per_head = a.mode == MODE_ATTN_MUL ? a.cout / a.heads : 1;
    float bias[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) bias[c] = a.bias ? a.bias[cg0 + c] : 0.f;
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int p = lane + 64 * j, r = p / TW, c0 = p - r * TW;
        if (row0 + r >= rows_total) continue;                              // ragged last patch of the stacked rows
        const size_t m = (size_t)(row0 + r) * a.W + x0 + c0;
#pragma unroll
        for (int q = 0; q < NCH / 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[j][2 * q + (e >> 1)][e & 1] + bias[4 * q + e];
                if (a.act == YACT_SILU) t = silu(t);
                v[e] = t;
            }
            const int nb4 = cg0 + 4 * q;
            if (a.mode == MODE_RESIDUAL) {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(a.aux + m * a.aux_ld + a.aux_off + nb4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[e];
            } else if (a.mode == MODE_ATTN_MUL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= a.aux[m * a.aux_ld + a.aux_off + (nb4 + e) / ch_per_head];
            }
            *reinterpret_cast<f32x4*>(a.dst + m * a.dst_ld + a.dst_off + nb4) = v;
        }
    }
}

// W [cout][K] -> Wt [K][cout], once per model at tstar_yolo_create
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)cout * K) return;
    const int n = (int)(i % cout), k = (int)(i / cout);
    wt[i] = w[(size_t)n * K + k];
}

// direct form for small K = ks*ks*cin (the 3-channel stem: K = 27): the whole weight matrix sits in LDS as [K][cout];
// a lane computes 4 neighbouring pixels x 16 output channels, so every weight quad read from LDS feeds 16 FMAs and the
// 64-channel output row of a pixel is written as 16-byte stores.  HBM-bound by its output (64 channels x 4 B per pixel).
constexpr int DPIX = 4, DCH = 16;
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [K][cout]
    const int K = a.ks * a.ks * a.cin;
    for (int i = threadIdx.x; i < K * a.cout; i += blockDim.x) {
        const int n = i % a.cout, k = i / a.cout;
        wl[i] = a.w[(size_t)n * K + k];
    }
    __syncthreads();
    const int ncg = a.cout / DCH;
    const int pq = threadIdx.x / ncg, cg = threadIdx.x % ncg;
    const int ppb = (blockDim.x / ncg) * DPIX;                       // pixels per block
    const int mbase = blockIdx.x * ppb + pq * DPIX;
    if (pq >= blockDim.x / ncg) return;
    float acc[DPIX][DCH];
#pragma unroll
    for (int i = 0; i < DPIX; ++i)
#pragma unroll
        for (int j = 0; j < DCH; ++j) acc[i][j] = 0.f;
    int pb[DPIX], py[DPIX], px[DPIX];
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < DPIX; ++i) {
        const int m = mbase + i < a.M ? mbase + i : a.M - 1;
        pb[i] = m / hw;
        const int rem = m - pb[i] * hw;
        py[i] = (rem / a.Wo) * a.stride - (a.ks >> 1);
        px[i] = (rem % a.Wo) * a.stride - (a.ks >> 1);
    }
    for (int ky = 0; ky < a.ks; ++ky)
        for (int kx = 0; kx < a.ks; ++kx)
            for (int c = 0; c < a.cin; ++c) {
                const float* wr = wl + (size_t)((ky * a.ks + kx) * a.cin + c) * a.cout + cg * DCH;
                f32x4 w4[DCH / 4];
#pragma unroll
                for (int q = 0; q < DCH / 4; ++q) w4[q] = *reinterpret_cast<const f32x4*>(wr + 4 * q);
#pragma unroll
                for (int i = 0; i < DPIX; ++i) {
                    const int iy = py[i] + ky, ix = px[i] + kx;
                    float v = 0.f;
                    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = a.src[((size_t)(pb[i] * a.H + iy) * a.W + ix) * a.src_ld + a.src_off + c];
#pragma unroll
                    for (int q = 0; q < DCH / 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = fmaf(v, w4[q][e], acc[i][4 * q + e]);
                }
            }
#pragma unroll
    for (int i = 0; i < DPIX; ++i) {
        const int m = mbase + i;
        if (m >= a.M) continue;
        float* o = a.dst + (size_t)m * a.dst_ld + a.dst_off + cg * DCH;
#pragma unroll
        for (int q = 0; q < DCH / 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[i][4 * q + e] + (a.bias ? a.bias[cg * DCH + 4 * q + e] : 0.f);
                if (a.act == YACT_SILU) t = silu(t);
                v[e] = t;
            }
            *reinterpret_cast<f32x4*>(o + 4 * q) = v;
        }
    }
}

// algorithmic HBM bytes of one conv launch: the input channels it reads (every input pixel once), the output channels it
// writes, the weights and bias, the residual / attention-gate tensor when fused
static double conv_algorithmic_bytes(const ConvArgs& a) {
    const double B = (double)a.M / ((double)a.Ho * a.Wo);
    double b = 4.0 * B * a.H * a.W * a.cin + 4.0 * a.M * a.cout + 4.0 * a.cout * a.ks * a.ks * a.cin + 4.0 * a.cout;
    if (a.mode == MODE_RESIDUAL) b += 4.0 * a.M * a.cout;
    else if (a.mode == MODE_ATTN_MUL) b += 4.0 * a.M * a.heads;
    return b;
}

static int launch_conv(const ConvArgs& a, hipStream_t s) {
    TSTAR_REQUIRE(a.cout % 4 == 0 && a.dst_ld % 4 == 0 && a.dst_off % 4 == 0, "yolo conv: output channels must be 16-byte aligned");
    const bool tiled = a.cin % CBK == 0 && a.src_ld % 4 == 0 && a.src_off % 4 == 0 && (a.ks == 1 || a.ks == 3);
    // scalar-weight form, chosen per layer from its size in 512-pixel x 64-channel blocks (per-layer timings of the L model
    // at B = 32, tools/rocpd_conv_align.py -> profiles/r02_yolo_conv_kernels_by_layer.md): >= 800 blocks: 4 pixels per lane
    // (256-pixel workgroups, 4 waves / SIMD); 400..799: 8 pixels per lane (one round of the chip's 512 two-per-CU slots);
    // below that the 128-pixel tiles of the tile kernel spread the layer over more CUs and win by up to 2x
    static const int sw_env = [] { const char* e = getenv("TSTAR_YOLO_SW"); return e ? atoi(e) : -1; }();
    const long long sw_blocks = (long long)cdiv(a.M, 512) * cdiv(a.cout, SWN);
    const bool sw_ok = tiled && a.wt && a.cout % 16 == 0 && a.zoff != 0 && a.zoff < (1u << 30);   // byte offsets fit 32 bits
    static const int sw_min = [] { const char* e = getenv("TSTAR_YOLO_SW_MIN"); return e ? atoi(e) : 400; }();
    static const int sw_p_env = [] { const char* e = getenv("TSTAR_YOLO_SW_P"); return e ? atoi(e) : 0; }();
    // halo form of the scalar-weight kernel for 3x3 / stride-1 layers.  Patch 8 x 40 on maps that tile by it (the 160-, 80-
    // and 40-wide ones), 16 x 20 over the row-stacked batch on 20-wide maps (H >= 16; one zero row at image boundaries).
    // From 320 workgroups of 16-channel waves up that form beats both other kernels on every such layer (B = 32: 10-15 %
    // per layer; B = 8: 221 vs 266 us at 320 workgroups, 445 vs 272 at 160 -- profiles/r02_yolo_conv_halo_by_layer_b*.md);
    // below HALO8_MAX (600) such workgroups the 8-channel-per-wave form takes over (twice the workgroups, 4 waves per SIMD: the
    // small maps and small batches, where 16-channel waves leave SIMDs empty), down to HALO8_MIN (600) of ITS workgroups: a
    // halo workgroup walks all of K, so a launch that does not fill the chip once still lasts one workgroup's ~430 us (256
    // input channels) and the tile kernel's many small workgroups win (B = 8, 40x40: 278 vs 426 us at 320 workgroups; B = 16,
    // 640: 440 vs 543) -- thresholds from profiles/r03_yolo_halo_forms_by_layer_*.md.  TSTAR_YOLO_HALO = 0 never / 2 always (when eligible);
    // TSTAR_YOLO_HALO_NCH = 8 / 16 forces the channel form.
    static const int halo_env = [] { const char* e = getenv("TSTAR_YOLO_HALO"); return e ? atoi(e) : -1; }();
    static const int nch_env = [] { const char* e = getenv("TSTAR_YOLO_HALO_NCH"); return e ? atoi(e) : 0; }();
    static const int halo8_max = [] { const char* e = getenv("TSTAR_YOLO_HALO8_MAX"); return e ? atoi(e) : 600; }();
    static const int halo8_min = [] { const char* e = getenv("TSTAR_YOLO_HALO8_MIN"); return e ? atoi(e) : 600; }();
    const bool halo_3x3 = sw_ok && a.ks == 3 && a.stride == 1 && a.Ho == a.H && a.Wo == a.W;
    const bool halo_a = halo_3x3 && a.H % 8 == 0 && a.W % 40 == 0;                         // 8 x 40 patches, never across images
    const bool halo_b = halo_3x3 && !halo_a && a.W % 20 == 0 && a.H >= 16;                  // 16 x 20 patches over stacked rows
    const int rows_total = a.M / (a.W > 0 ? a.W : 1);
    const long long halo_mt = halo_a ? (long long)(rows_total / 8) * (a.W / 40) : halo_b ? (long long)cdiv(rows_total, 16) * (a.W / 20) : 0;
    const long long wgs16 = halo_mt * cdiv(a.cout, 64), wgs8 = halo_mt * cdiv(a.cout, 32);
    int halo_nch = 0;                                                                       // 0 = not the halo form
    if ((halo_a || halo_b) && halo_env != 0) {
        if (nch_env == 8 || nch_env == 16) halo_nch = (halo_env > 1 || (nch_env == 16 ? wgs16 >= 320 : wgs8 >= halo8_min)) ? nch_env : 0;
        else if (wgs16 >= halo8_max) halo_nch = 16;
        else if (wgs8 >= halo8_min || halo_env > 1) halo_nch = 8;
    }
    if (halo_nch) {
        const int mt = (int)halo_mt, nt = cdiv(a.cout, 4 * halo_nch);
        const bool prof = prof_enabled();
        if (prof) prof_start(PROF_CONV, s, 2.0 * a.M * a.cout * a.ks * a.ks * a.cin, conv_algorithmic_bytes(a));
        const dim3 grid(mt * nt);
        if (halo_a) {
            if (halo_nch == 16) hipLaunchKernelGGL((conv_halo_kernel<8, 40, 16, false>), grid, dim3(256), 0, s, a, mt, nt);
            else hipLaunchKernelGGL((conv_halo_kernel<8, 40, 8, false>), grid, dim3(256), 0, s, a, mt, nt);
        } else {
            if (halo_nch == 16) hipLaunchKernelGGL((conv_halo_kernel<16, 20, 16, true>), grid, dim3(256), 0, s, a, mt, nt);
            else hipLaunchKernelGGL((conv_halo_kernel<16, 20, 8, true>), grid, dim3(256), 0, s, a, mt, nt);
        }
        if (prof) prof_stop(PROF_CONV, s);
    } else if (sw_ok && (sw_env < 0 ? sw_blocks >= sw_min : sw_env > 0)) {
        const int sw_p = sw_p_env ? sw_p_env : (sw_blocks >= 800 ? 4 : 8);
        const int mt = cdiv(a.M, 64 * sw_p), nt = cdiv(a.cout, SWN);
        const dim3 grid(mt * nt);
        const bool prof = prof_enabled();
        if (prof) prof_start(PROF_CONV, s, 2.0 * a.M * a.cout * a.ks * a.ks * a.cin, conv_algorithmic_bytes(a));
        if (sw_p == 8) {
            if (a.ks == 1) hipLaunchKernelGGL((conv_sw_kernel<1, 8>), grid, dim3(256), 0, s, a, mt, nt);
            else hipLaunchKernelGGL((conv_sw_kernel<3, 8>), grid, dim3(256), 0, s, a, mt, nt);
        } else {
            if (a.ks == 1) hipLaunchKernelGGL((conv_sw_kernel<1, 4>), grid, dim3(256), 0, s, a, mt, nt);
            else hipLaunchKernelGGL((conv_sw_kernel<3, 4>), grid, dim3(256), 0, s, a, mt, nt);
        }
        if (prof) prof_stop(PROF_CONV, s);
    } else if (tiled) {
        // 128-channel tiles (8 x 8 outputs per lane) when the layer has the channels and enough pixels to fill the chip;
        // 64-pixel tiles (4 x 4 per lane) when 128-pixel tiles would leave CUs without a workgroup to overlap with
        static const int tn_env = [] { const char* e = getenv("TSTAR_YOLO_TN"); return e ? atoi(e) : 0; }();
        static const int tm_env = [] { const char* e = getenv("TSTAR_YOLO_TM"); return e ? atoi(e) : 0; }();
        static const int tm_min = [] { const char* e = getenv("TSTAR_YOLO_TM_MIN"); return e ? atoi(e) : 512; }();
        const bool wide = tn_env ? tn_env == 8 : (a.cout % 128 == 0 && (long long)cdiv(a.M, 128) * (a.cout / 128) >= 512);
        const bool small = !wide && (tm_env ? tm_env == 4 : (long long)cdiv(a.M, 128) * cdiv(a.cout, 64) < tm_min);
        const int mt = cdiv(a.M, small ? 64 : 128), nt = cdiv(a.cout, wide ? 128 : 64);
        const dim3 grid(mt * nt);
        const bool prof = prof_enabled();
        if (prof) prof_start(PROF_CONV, s, 2.0 * a.M * a.cout * a.ks * a.ks * a.cin, conv_algorithmic_bytes(a));
        if (a.ks == 1) {
            if (wide) hipLaunchKernelGGL((conv_valu_kernel<1, 8, 8>), grid, dim3(256), 0, s, a, mt, nt);
            else if (small) hipLaunchKernelGGL((conv_valu_kernel<1, 4, 4>), grid, dim3(256), 0, s, a, mt, nt);
            else hipLaunchKernelGGL((conv_valu_kernel<1, 4, 8>), grid, dim3(256), 0, s, a, mt, nt);
        } else {
            if (wide) hipLaunchKernelGGL((conv_valu_kernel<3, 8, 8>), grid, dim3(256), 0, s, a, mt, nt);
            else if (small) hipLaunchKernelGGL((conv_valu_kernel<3, 4, 4>), grid, dim3(256), 0, s, a, mt, nt);
            else hipLaunchKernelGGL((conv_valu_kernel<3, 4, 8>), grid, dim3(256), 0, s, a, mt, nt);
        }
        if (prof) prof_stop(PROF_CONV, s);
    } else {
        TSTAR_REQUIRE(a.mode == MODE_PLAIN, "yolo conv: the direct form has no fused residual / gate");
        const int K = a.ks * a.ks * a.cin;
        TSTAR_REQUIRE(a.cout % DCH == 0 && a.cout / DCH <= 256 && (size_t)K * a.cout * 4 <= 64 * 1024,
                      "yolo conv: the direct (small-K) form needs cout % 16 == 0 and its weights in 64 KB of LDS");
        const int ncg = a.cout / DCH, nthreads = (256 / ncg) * ncg, ppb = (nthreads / ncg) * DPIX;
        hipLaunchKernelGGL(conv_direct_kernel, dim3(cdiv(a.M, ppb)), dim3(nthreads), (size_t)K * a.cout * 4, s, a);
    }
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// ------------------------------------------------------------------ elementwise ops
// SPPF: dst[.., doff + c] = max over the 5x5 window (stride 1, pad 2, -inf padding) of src[.., soff + c]
__global__ __launch_bounds__(256) void pool5_kernel(const float* __restrict__ buf, int ld, int soff, int doff, int C, int H, int W, float* __restrict__ out, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int cq = C / 4;
    const int c = (int)(gid % cq) * 4;
    const size_t p = gid / cq;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const size_t img = p / ((size_t)W * H);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(buf + ((img * H + yy) * W + xx) * ld + soff + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
        }
    }
    *reinterpret_cast<f32x4*>(out + p * ld + doff + c) = m;
}

// dst[b, y, x, doff + c] = src[b, y / f, x / f, soff + c]   (f = 1: channel-offset copy; f = 2: nearest upsample)
__global__ __launch_bounds__(256) void upcopy_kernel(const float* __restrict__ src, int sld, int soff, int C, int Hs, int Ws, int f,
                                                     float* __restrict__ dst, int dld, int doff, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int cq = C / 4;
    const int c = (int)(gid % cq) * 4;
    const size_t p = gid / cq;
    const int Wd = Ws * f, Hd = Hs * f;
    const int x = (int)(p % Wd), y = (int)((p / Wd) % Hd);
    const size_t img = p / ((size_t)Wd * Hd);
    *reinterpret_cast<f32x4*>(dst + p * dld + doff + c) =
        *reinterpret_cast<const f32x4*>(src + ((img * Hs + y / f) * Ws + x / f) * sld + soff + c);
}

// MaxSigmoidAttnBlock gate: attn[p, m] = sigmoid(max_n <embed[p, m, :], guide[n, m, :]> / sqrt(hc) + bias[m]);
// guide = guide_fc(text) of the image's query set, [Q][embed] per set, staged in LDS
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ emb, int ld, int off, int embed, int heads, int HW,
                                                   const float* __restrict__ guide_all, const int* __restrict__ setQ,
                                                   const int* __restrict__ image_set, const float* __restrict__ bias,
                                                   float* __restrict__ attn, int P) {
    extern __shared__ __attribute__((aligned(16))) float g[];   // [Q][embed]
    const int img = blockIdx.y;
    const int set = image_set ? image_set[img] : 0;
    const int Q = setQ[set];
    const float* gs = guide_all + (size_t)set * YOLO_MAX_Q * embed;
    for (int i = threadIdx.x; i < Q * embed; i += blockDim.x) g[i] = gs[i];
    __syncthreads();
    const int hc = embed / heads;
    const float inv = 1.0f / sqrtf((float)hc);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < HW * heads; t += gridDim.x * blockDim.x) {
        const int pix = t / heads, m = t - pix * heads;
        const size_t p = (size_t)img * HW + pix;
        const float* e = emb + p * ld + off + m * hc;
        float best = -INFINITY;
        if (hc == 32 && ((ld | off) & 3) == 0) {
            // the pixel's 32 head channels once, as eight 16-byte loads (every published scale has 32 channels per head); the
            // guide vectors come from LDS as float4 too.  Same sequential fmaf order as the scalar loop.
            f32x4 ev[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ev[q] = *reinterpret_cast<const f32x4*>(e + 4 * q);
            for (int n = 0; n < Q; ++n) {
                const f32x4* gv = reinterpret_cast<const f32x4*>(g + n * embed + m * 32);
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 w = gv[q];
                    d = fmaf(ev[q][0], w[0], d); d = fmaf(ev[q][1], w[1], d); d = fmaf(ev[q][2], w[2], d); d = fmaf(ev[q][3], w[3], d);
                }
                best = fmaxf(best, d);
            }
        } else {
            for (int n = 0; n < Q; ++n) {
                const float* gv = g + n * embed + m * hc;
                float d = 0.f;
                for (int c = 0; c < hc; ++c) d = fmaf(e[c], gv[c], d);
                best = fmaxf(best, d);
            }
        }
        const float v = best * inv + bias[m];
        attn[p * heads + m] = 1.0f / (1.0f + __expf(-v));
    }
}

// guide[set][n][e] = guide_fc(text_n) = W[e,:] . t_n + b[e]
__global__ void guide_fc_kernel(const float* __restrict__ text, int Q, const float* __restrict__ W, const float* __restrict__ b,
                                int embed, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q * embed) return;
    const int n = i / embed, e = i - n * embed;
    const float* t = text + (size_t)n * YOLO_TEXT;
    const float* w = W + (size_t)e * YOLO_TEXT;
    float d = 0.f;
    for (int c = 0; c < YOLO_TEXT; ++c) d = fmaf(t[c], w[c], d);
    out[i] = d + b[e];
}

// F.normalize(text, dim=-1): t / max(||t||, 1e-12); one wave per query
__global__ void text_normalize_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int q = blockIdx.x, lane = threadIdx.x;
    float v[8], s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = in[(size_t)q * YOLO_TEXT + i * 64 + lane]; s += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float den = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[(size_t)q * YOLO_TEXT + i * 64 + lane] = v[i] / den;
}

// ------------------------------------------------------------------ test pipeline
// INTER_AREA-style down-scale (the build's definition: exact-footprint box average in integers): per axis, output o
// covers source units [o*n_in, (o+1)*n_in) of 1/n_out pixel.  u8 [B,H,W,3] -> u8 [B,oh,ow,3]
__global__ __launch_bounds__(256) void area_resize_kernel(const uint8_t* __restrict__ in, int H, int W, uint8_t* __restrict__ out, int oh, int ow, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % ow), oy = (int)((gid / ow) % oh);
    const size_t img = gid / ((size_t)ow * oh);
    const long long xlo = (long long)ox * W, xhi = xlo + W, ylo = (long long)oy * H, yhi = ylo + H;
    const int px0 = (int)(xlo / ow), px1 = (int)((xhi - 1) / ow), py0 = (int)(ylo / oh), py1 = (int)((yhi - 1) / oh);
    long long acc[3] = {0, 0, 0};
    for (int py = py0; py <= py1; ++py) {
        const long long a = (long long)py * oh, b = a + oh;
        const long long wy = (yhi < b ? yhi : b) - (ylo > a ? ylo : a);
        long long row[3] = {0, 0, 0};
        for (int px = px0; px <= px1; ++px) {
            const long long c = (long long)px * ow, d = c + ow;
            const long long wx = (xhi < d ? xhi : d) - (xlo > c ? xlo : c);
            const uint8_t* q = in + ((img * H + py) * W + px) * 3;
            row[0] += wx * q[0]; row[1] += wx * q[1]; row[2] += wx * q[2];
        }
        acc[0] += wy * row[0]; acc[1] += wy * row[1]; acc[2] += wy * row[2];
    }
    const long long den = (long long)W * H;
    uint8_t* o = out + gid * 3;
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((2 * acc[c] + den) / (2 * den));
}

// u8 [B,nh,nw,3] RGB -> f32 NHWC [B,640,640,3]: pad 114, channels REVERSED (the reference feeds RGB arrays to a BGR
// pipeline whose preprocessor then swaps them, interface_heuristic.py:137), / 255
__global__ __launch_bounds__(256) void letterbox_pack_kernel(const uint8_t* __restrict__ in, int nh, int nw, int top, int left,
                                                             float* __restrict__ out, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = (int)(gid % YOLO_IMG), y = (int)((gid / YOLO_IMG) % YOLO_IMG);
    const size_t img = gid / ((size_t)YOLO_IMG * YOLO_IMG);
    int r = 114, g = 114, b = 114;
    const int sy = y - top, sx = x - left;
    if (sy >= 0 && sy < nh && sx >= 0 && sx < nw) {
        const uint8_t* q = in + ((img * nh + sy) * nw + sx) * 3;
        r = q[0]; g = q[1]; b = q[2];
    }
    float* o = out + gid * 3;
    o[0] = (float)b / 255.0f; o[1] = (float)g / 255.0f; o[2] = (float)r / 255.0f;
}

// ------------------------------------------------------------------ head
struct DecodeArgs {
    const float* E; const float* R;                           // [B*HW, 512], [B*HW, 64]
    int HW, Wl, stride, anchor0, n_anchor;                    // level geometry; anchor index base inside the image
    float logit_scale, bias;
    const float* textn;                                       // [sets][32][512] F.normalize'd text
    const int* setQ; const int* image_set;
    float pad_left, pad_top, sf_w, sf_h;
    float cand_thr;
    float* boxes;                                             // [B, n_anchor, 4] un-letterboxed, unclamped
    unsigned long long* cand; int cand_cap; int* cand_count;  // per image
    float* dense_scores; int dense_q;                         // optional [B, n_anchor, dense_q]
};

// 16 anchors of ONE image per workgroup (4 per wave, one after the other): DFL expectation + decode + un-letterbox;
// scores against every query of the image's set; (score, anchor, class) pairs above the candidate threshold are
// collected in LDS and appended to the image's list with ONE global atomic per workgroup (every anchor appending on its
// own serialised ~8 x 10^5 atomics on a few counters: 9 ms of a 100 ms forward)
constexpr int HD_APB = 16;                                          // anchors per block
__global__ __launch_bounds__(256) void head_decode_kernel(DecodeArgs a, int rows) {
    __shared__ unsigned long long s_cand[HD_APB * YOLO_MAX_Q];
    __shared__ int s_n, s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int row0 = blockIdx.x * HD_APB;
    const int img = row0 / a.HW;                                     // HW % 16 == 0: a block never straddles two images
    const int set = a.image_set ? a.image_set[img] : 0;
    const int Q = a.setQ[set];
    for (int ai = 0; ai < HD_APB / 4; ++ai) {
        const int row = row0 + wave * (HD_APB / 4) + ai;
        if (row >= rows) break;
        const int p = row - img * a.HW;
        const int anchor = a.anchor0 + p;
        // DFL: lane = side * 16 + bin
        const float r = a.R[(size_t)row * 64 + lane];
        float mx = r;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(r - mx);
        float se = e, sw = e * (float)(lane & 15);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { se += __shfl_xor(se, o); sw += __shfl_xor(sw, o); }
        const float dist = (sw / se) * (float)a.stride;
        const float d0 = __shfl(dist, 0), d1 = __shfl(dist, 16), d2 = __shfl(dist, 32), d3 = __shfl(dist, 48);
        const float px = ((float)(p % a.Wl) + 0.5f) * (float)a.stride, py = ((float)(p / a.Wl) + 0.5f) * (float)a.stride;
        if (lane == 0) {
            f32x4 bx;
            bx[0] = (px - d0 - a.pad_left) / a.sf_w; bx[1] = (py - d1 - a.pad_top) / a.sf_h;
            bx[2] = (px + d2 - a.pad_left) / a.sf_w; bx[3] = (py + d3 - a.pad_top) / a.sf_h;
            *reinterpret_cast<f32x4*>(a.boxes + ((size_t)img * a.n_anchor + anchor) * 4) = bx;
        }
        // classification: <E[row], textn[k]> * exp(logit_scale) + bias -> sigmoid
        const float* er = a.E + (size_t)row * YOLO_TEXT;
        float ev[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ev[i] = er[i * 64 + lane];
        for (int k = 0; k < Q; ++k) {
            const float* t = a.textn + ((size_t)set * YOLO_MAX_Q + k) * YOLO_TEXT;
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf(ev[i], t[i * 64 + lane], d);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (lane == 0) {
                const float logit = d * a.logit_scale + a.bias;
                const float sc = 1.0f / (1.0f + expf(-logit));
                if (a.dense_scores) a.dense_scores[((size_t)img * a.n_anchor + anchor) * a.dense_q + k] = sc;
                if (sc > a.cand_thr) {
                    const unsigned id = (unsigned)anchor * YOLO_MAX_Q + (unsigned)k;
                    s_cand[atomicAdd(&s_n, 1)] = ((unsigned long long)__float_as_uint(sc) << 32) | (0xFFFFFFFFu - id);
                }
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (threadIdx.x == 0 && n > 0) s_base = atomicAdd(&a.cand_count[img], n);
    __syncthreads();
    if (n > 0) {
        const int base = s_base;
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (base + i < a.cand_cap) a.cand[(size_t)img * a.cand_cap + base + i] = s_cand[i];
    }
}

// One workgroup per image.  mmyolo's order of business -- every (anchor, class) pair with score > 0.001 is a candidate,
// sorted by descending score, the first nms_pre = 30000 kept, class-aware NMS with mmcv's coordinate-offset trick (the offset
// is the largest coordinate among ALL kept candidates + 1), max_per_img = 300, clamp -- followed by the reference wrapper
// (score > 0.12, top max_dets).  Only candidates above the wrapper threshold can ever be output, and a candidate can only
// be suppressed by a higher-scoring survivor, so the full sort is not needed:
//   1. if there are more than nms_pre candidates, an 8-pass radix select finds the nms_pre-th largest key (keys are unique);
//   2. the largest coordinate is reduced over the candidates at or above that key;
//   3. the candidates above the wrapper threshold (and the cut) are compacted into LDS (up to 16384 keys = 128 KB) and
//      bitonic-sorted there -- a global-memory sort of all 8400 x Q keys took 3.6 ms per batch; beyond 16384 the kernel
//      falls back to sorting the whole list in global memory;
//   4. greedy NMS by ONE wavefront (survivors in LDS, lanes test them in parallel, __ballot decides), cut, clamp, top-k.
constexpr int NMS_LDS_KEYS = 16384;
__global__ __launch_bounds__(1024) void sort_nms_kernel(unsigned long long* __restrict__ cand_all, int cand_cap, const int* __restrict__ cand_count,
                                                        const float* __restrict__ boxes_all, int n_anchor, float img_w, float img_h,
                                                        float wrapper_thr, int max_dets, float* __restrict__ det_scores,
                                                        int* __restrict__ det_labels, float* __restrict__ det_boxes, int* __restrict__ n_det) {
    extern __shared__ unsigned long long s_keys[];             // NMS_LDS_KEYS
    __shared__ float k_box[YOLO_MAX_PER_IMG][4];               // offset boxes of the survivors
    __shared__ float k_area[YOLO_MAX_PER_IMG];
    __shared__ unsigned k_id[YOLO_MAX_PER_IMG];
    __shared__ float k_score[YOLO_MAX_PER_IMG];
    __shared__ float s_red[1024 / 64];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned long long s_pref[2];
    __shared__ int s_kept, s_m;
    const int img = blockIdx.x, t = threadIdx.x;
    unsigned long long* cand = cand_all + (size_t)img * cand_cap;
    const float* boxes = boxes_all + (size_t)img * n_anchor * 4;
    int n = cand_count[img];
    n = n < cand_cap ? n : cand_cap;
    // ---- 1. the nms_pre cut: key of rank nms_pre - 1 in descending order (0 = no cut)
    unsigned long long kth = 0ull;
    if (n > YOLO_NMS_PRE) {
        unsigned long long prefix = 0ull, mask = 0ull;
        int kk = YOLO_NMS_PRE - 1;                             // rank from the top
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int i = t; i < 256; i += 1024) s_hist[i] = 0u;
            __syncthreads();
            for (int i = t; i < n; i += 1024) {
                const unsigned long long c = cand[i];
                if ((c & mask) == prefix) atomicAdd(&s_hist[(c >> shift) & 0xFF], 1u);
            }
            __syncthreads();
            if (t == 0) {
                int acc = 0, d = 255;
                for (; d >= 0; --d) { if (acc + (int)s_hist[d] > kk) break; acc += s_hist[d]; }
                s_pref[0] = prefix | ((unsigned long long)d << shift);
                s_pref[1] = (unsigned long long)(kk - acc);
            }
            __syncthreads();
            prefix = s_pref[0];
            kk = (int)s_pref[1];
            mask |= 0xFFULL << shift;
            __syncthreads();
        }
        kth = prefix;
    }
    // ---- 2. mmcv batched_nms: boxes + label * (max coordinate + 1), in float32; 3. compaction into LDS
    if (t == 0) { s_kept = 0; s_m = 0; }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = t; i < n; i += 1024) {
        const unsigned long long c = cand[i];
        if (c < kth) continue;
        const unsigned id = 0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFu);
        const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)(id / YOLO_MAX_Q) * 4);
        mx = fmaxf(fmaxf(fmaxf(mx, b[0]), fmaxf(b[1], b[2])), b[3]);
        if (__uint_as_float((unsigned)(c >> 32)) > wrapper_thr) {
            const int slot = atomicAdd(&s_m, 1);
            if (slot < NMS_LDS_KEYS) s_keys[slot] = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) s_red[t >> 6] = mx;
    __syncthreads();
    float maxc = s_red[0];
    for (int i = 1; i < 1024 / 64; ++i) maxc = fmaxf(maxc, s_red[i]);
    const float off_unit = maxc + 1.0f;
    int m = s_m;
    unsigned long long* keys = s_keys;
    if (m > NMS_LDS_KEYS) {                                    // rare: sort the whole list in global memory instead
        keys = cand;
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int i = n + t; i < n2; i += 1024) cand[i] = 0ull;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < n2; i += 1024) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long x = cand[i], y = cand[l];
                        if (((i & k) == 0) ? x < y : x > y) { cand[i] = y; cand[l] = x; }
                    }
                }
                __syncthreads();
            }
        m = n < YOLO_NMS_PRE ? n : YOLO_NMS_PRE;
    } else {
        int n2 = 1;
        while (n2 < m) n2 <<= 1;
        for (int i = m + t; i < n2; i += 1024) s_keys[i] = 0ull;          // padding sorts last (score bits 0)
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1)                                   // bitonic sort in LDS, descending
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < n2; i += 1024) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long x = s_keys[i], y = s_keys[l];
                        if (((i & k) == 0) ? x < y : x > y) { s_keys[i] = y; s_keys[l] = x; }
                    }
                }
                __syncthreads();
            }
    }
    if (t < 64) {                                              // ---- 4. ONE wave: lock-step, no barriers
        // candidates are taken 64 at a time: every lane fetches one candidate's key and box (64 independent loads in flight),
        // then the wave walks them in order with v_readlane broadcasts -- a dependent global load per candidate cost ~1.5 us
        int kept = 0;
        bool done = false;
        for (int base = 0; base < m && kept < YOLO_MAX_PER_IMG && !done; base += 64) {
            const int mine = base + t;
            const unsigned long long cm = mine < m ? keys[mine] : 0ull;
            const unsigned idm = 0xFFFFFFFFu - (unsigned)(cm & 0xFFFFFFFFu);
            f32x4 bm = {0.f, 0.f, 0.f, 0.f};
            if (mine < m) bm = *reinterpret_cast<const f32x4*>(boxes + (size_t)(idm / YOLO_MAX_Q) * 4);
            const int scm = (int)(unsigned)(cm >> 32);
            const int cnt = (m - base) < 64 ? (m - base) : 64;
            for (int j = 0; j < cnt && kept < YOLO_MAX_PER_IMG; ++j) {
                const float sc = __int_as_float(__builtin_amdgcn_readlane(scm, j));
                // the wrapper drops survivors at or below its threshold, and they cannot suppress anything before them
                if (!(sc > wrapper_thr)) { done = true; break; }
                const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)idm, j);
                const int label = (int)(id % YOLO_MAX_Q);
                const float o = (float)label * off_unit;
                const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bm[0]), j)) + o;
                const float y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bm[1]), j)) + o;
                const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bm[2]), j)) + o;
                const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bm[3]), j)) + o;
                const float area = (x1 - x0) * (y1 - y0);
                bool sup = false;
                for (int q = t; q < kept; q += 64) {
                    const float iw = fmaxf(fminf(x1, k_box[q][2]) - fmaxf(x0, k_box[q][0]), 0.f);
                    const float ih = fmaxf(fminf(y1, k_box[q][3]) - fmaxf(y0, k_box[q][1]), 0.f);
                    const float inter = iw * ih;
                    sup |= inter / (area + k_area[q] - inter) > YOLO_IOU_THR;
                }
                if (__ballot(sup) == 0ull) {
                    if (t == 0) {
                        k_box[kept][0] = x0; k_box[kept][1] = y0; k_box[kept][2] = x1; k_box[kept][3] = y1;
                        k_area[kept] = area; k_id[kept] = id; k_score[kept] = sc;
                    }
                    // lane 0's survivor must be visible to every lane's LDS reads of the next candidate: a wavefront-scope
                    // release fence + wave barrier pins the order for the compiler too (no load may be hoisted above it)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    ++kept;
                }
            }
        }
        if (t == 0) s_kept = kept;
    }
    __syncthreads();
    // survivors are in descending score order and all above the wrapper threshold: the output is a prefix
    const int kept = s_kept;
    const int nd = kept < max_dets ? kept : max_dets;
    if (t < max_dets) {
        float* b = det_boxes + ((size_t)img * max_dets + t) * 4;
        if (t < nd) {
            const unsigned id = k_id[t];
            const f32x4 rb = *reinterpret_cast<const f32x4*>(boxes + (size_t)(id / YOLO_MAX_Q) * 4);
            b[0] = fminf(fmaxf(rb[0], 0.f), img_w); b[1] = fminf(fmaxf(rb[1], 0.f), img_h);
            b[2] = fminf(fmaxf(rb[2], 0.f), img_w); b[3] = fminf(fmaxf(rb[3], 0.f), img_h);
            det_scores[(size_t)img * max_dets + t] = k_score[t];
            det_labels[(size_t)img * max_dets + t] = (int)(id % YOLO_MAX_Q);
        } else {
            b[0] = b[1] = b[2] = b[3] = 0.f;
            det_scores[(size_t)img * max_dets + t] = 0.f;
            det_labels[(size_t)img * max_dets + t] = -1;
        }
    }
    if (t == 0) n_det[img] = nd;
}

// imageGridScoreFunction's loop (interface_searcher.py:129-150) over the <= max_dets detections of each image
__global__ __launch_bounds__(64) void det_cells_kernel(const float* __restrict__ det_scores, const int* __restrict__ det_labels,
                                                       const float* __restrict__ det_boxes, const int* __restrict__ n_det, int max_dets,
                                                       const double* __restrict__ qweight_all, const int* __restrict__ image_set,
                                                       int img_w, int img_h, int grows, int gcols, double* __restrict__ cell_conf,
                                                       uint32_t* __restrict__ cell_mask) {
    extern __shared__ unsigned long long sm[];
    const int ncell = grows * gcols, b = blockIdx.x;
    unsigned long long* cbits = sm;
    uint32_t* cmask = reinterpret_cast<uint32_t*>(sm + ncell);
    const double* qweight = qweight_all + (image_set ? image_set[b] : 0) * YOLO_MAX_Q;
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) { cbits[i] = 0ull; cmask[i] = 0u; }
    __syncthreads();
    const double cw = (double)img_w / (double)gcols, ch = (double)img_h / (double)grows;
    for (int d = threadIdx.x; d < n_det[b]; d += blockDim.x) {
        const size_t r = (size_t)b * max_dets + d;
        const int lab = det_labels[r];
        const double conf = (double)det_scores[r] * qweight[lab];
        const float* bb = det_boxes + r * 4;
        const float cx = (bb[0] + bb[2]) * 0.5f, cy = (bb[1] + bb[3]) * 0.5f;
        int gx = (int)floor((double)cx / cw), gy = (int)floor((double)cy / ch);
        gx = gx < gcols - 1 ? gx : gcols - 1; gy = gy < grows - 1 ? gy : grows - 1;
        gx = gx < 0 ? 0 : gx; gy = gy < 0 ? 0 : gy;
        atomicMax(&cbits[gy * gcols + gx], (unsigned long long)__double_as_longlong(conf));
        atomicOr(&cmask[gy * gcols + gx], 1u << lab);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) {
        cell_conf[(size_t)b * ncell + i] = __longlong_as_double((long long)cbits[i]);
        cell_mask[(size_t)b * ncell + i] = cmask[i];
    }
}

}  // namespace tstar

using namespace tstar;

struct YoloOp { int w[24]; };
struct YoloGuide { int embed, heads, w_off, b_off, bias_off; float* d_guide; };   // d_guide [sets][32][embed]
struct YoloLevel { int e_buf, r_buf, size, stride, param_off; float logit_scale, bias; };

struct tstar_yolo {
    float* d_blob = nullptr; size_t n_blob = 0;
    float* d_blob_t = nullptr;                                // conv matrices transposed to [K][cout], each at a 64-byte aligned offset
    std::vector<size_t> wt_off;                               // ... wt_off[op] (floats), one entry per op
    std::vector<float> h_small;                               // host copy of the per-level scalars
    std::vector<YoloOp> ops;
    std::vector<int> buf_h, buf_w, buf_c;
    std::vector<float*> bufs;
    std::vector<YoloGuide> guides;
    std::vector<YoloLevel> levels;
    int input_buf = 0, max_batch = 0, n_anchor = 0;
    int Q[YOLO_SETS] = {0};
    float *d_text = nullptr, *d_textn = nullptr;              // [sets][32][512] raw / normalised
    double* d_qweight = nullptr;
    int *d_setQ = nullptr, *d_image_set = nullptr, *d_iota = nullptr, *d_cand_count = nullptr;
    int image_set_cap = 0;
    uint8_t* d_tmp_u8 = nullptr; size_t tmp_u8_bytes = 0;
    float* d_boxes = nullptr;                                 // [max_batch, n_anchor, 4]
    unsigned long long* d_cand = nullptr; int cand_cap = 0;   // [max_batch, cand_cap]
};

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

static size_t pow2_at_least(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

extern "C" {

int tstar_yolo_destroy(tstar_yolo* h) {
    if (!h) return TSTAR_OK;
    void* ptrs[] = {h->d_blob, h->d_blob_t, h->d_text, h->d_textn, h->d_qweight, h->d_setQ, h->d_image_set, h->d_iota, h->d_cand_count,
                    h->d_tmp_u8, h->d_boxes, h->d_cand};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (float* p : h->bufs) if (p) (void)hipFree(p);
    for (auto& g : h->guides) if (g.d_guide) (void)hipFree(g.d_guide);
    delete h;
    return TSTAR_OK;
}

int tstar_yolo_create(tstar_yolo** out, const float* h_blob, size_t n_blob, const int32_t* h_ops, int n_ops, int op_words,
                      const int32_t* h_bufs, int n_bufs, const int32_t* h_guides, int n_guides, const int32_t* h_levels,
                      int n_levels, int input_buf, int max_batch) {
    TSTAR_REQUIRE(out && h_blob && h_ops && h_bufs && h_levels, "tstar_yolo_create: null argument");
    TSTAR_REQUIRE(op_words == 24 && n_ops >= 1 && n_bufs >= 1 && n_levels >= 1 && n_levels <= 8, "tstar_yolo_create: bad program shape");
    TSTAR_REQUIRE(max_batch >= 1 && max_batch <= 1024, "tstar_yolo_create: max_batch must be in 1..1024");
    TSTAR_REQUIRE(input_buf >= 0 && input_buf < n_bufs, "tstar_yolo_create: bad input buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("tstar_yolo_create: no HIP device visible (this library has no CPU path)");
        return TSTAR_ERR_HIP;
    }
    tstar_yolo* h = new tstar_yolo();
    h->max_batch = max_batch; h->input_buf = input_buf; h->n_blob = n_blob;
    auto fail = [&](int rc) { tstar_yolo_destroy(h); return rc; };
    for (int i = 0; i < n_bufs; ++i) {
        const int H = h_bufs[i * 3], W = h_bufs[i * 3 + 1], Cc = h_bufs[i * 3 + 2];
        if (H < 1 || W < 1 || Cc < 1) { set_error("tstar_yolo_create: bad buffer shape"); return fail(TSTAR_ERR_ARG); }
        h->buf_h.push_back(H); h->buf_w.push_back(W); h->buf_c.push_back(Cc);
    }
    if (h->buf_h[input_buf] != YOLO_IMG || h->buf_w[input_buf] != YOLO_IMG || h->buf_c[input_buf] != 3) {
        set_error("tstar_yolo_create: the input buffer must be 640 x 640 x 3"); return fail(TSTAR_ERR_ARG);
    }
    auto buf_ok = [&](int b) { return b >= 0 && b < n_bufs; };
    auto off_ok = [&](long long off, long long n) { return off >= 0 && (size_t)(off + n) <= n_blob; };
    for (int i = 0; i < n_ops; ++i) {
        YoloOp op; memcpy(op.w, h_ops + (size_t)i * op_words, sizeof(op.w));
        const int* w = op.w;
        bool ok = true;
        if (w[0] == OP_CONV) {
            ok = buf_ok(w[1]) && buf_ok(w[4]) && w[3] >= 1 && w[6] >= 1 && (w[7] == 1 || w[7] == 3) && (w[8] == 1 || w[8] == 2) &&
                 w[2] >= 0 && w[2] + w[3] <= h->buf_c[w[1]] && w[5] >= 0 && w[5] + w[6] <= h->buf_c[w[4]] && h->buf_c[w[4]] % 4 == 0 &&
                 off_ok(w[10], (long long)w[6] * w[7] * w[7] * w[3]) && (w[11] < 0 || off_ok(w[11], w[6])) &&
                 (w[12] == MODE_PLAIN || (buf_ok(w[13]) && w[14] >= 0));
            if (ok) {
                const int Ho = (h->buf_h[w[1]] + 2 * (w[7] / 2) - w[7]) / w[8] + 1, Wo = (h->buf_w[w[1]] + 2 * (w[7] / 2) - w[7]) / w[8] + 1;
                ok = Ho == h->buf_h[w[4]] && Wo == h->buf_w[w[4]];
            }
        } else if (w[0] == OP_POOL5) {
            ok = buf_ok(w[1]) && w[4] == w[1] && w[3] % 4 == 0 && w[2] % 4 == 0 && w[5] % 4 == 0 && h->buf_c[w[1]] % 4 == 0 &&
                 w[2] + w[3] <= h->buf_c[w[1]] && w[5] + w[3] <= h->buf_c[w[1]];
        } else if (w[0] == OP_UPCOPY) {
            ok = buf_ok(w[1]) && buf_ok(w[4]) && (w[6] == 1 || w[6] == 2) && w[3] % 4 == 0 && w[2] % 4 == 0 && w[5] % 4 == 0 &&
                 h->buf_c[w[1]] % 4 == 0 && h->buf_c[w[4]] % 4 == 0 && w[2] + w[3] <= h->buf_c[w[1]] &&
                 w[5] + w[3] <= h->buf_c[w[4]] && h->buf_h[w[1]] * w[6] == h->buf_h[w[4]] && h->buf_w[w[1]] * w[6] == h->buf_w[w[4]];
        } else if (w[0] == OP_ATTN) {
            ok = buf_ok(w[1]) && buf_ok(w[4]) && w[7] >= 0 && w[7] < n_guides && w[6] >= 1 && w[3] % w[6] == 0 && h->buf_c[w[4]] == w[6] &&
                 w[2] + w[3] <= h->buf_c[w[1]];
        } else ok = false;
        if (!ok) { set_error("tstar_yolo_create: malformed op " + std::to_string(i)); return fail(TSTAR_ERR_ARG); }
        h->ops.push_back(op);
    }
    hipError_t e = hipMalloc(&h->d_blob, n_blob * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_blob, h_blob, n_blob * sizeof(float), hipMemcpyHostToDevice);
    // transposed copies for the scalar-weight kernel: a weight row of 16 channels is one 64-byte scalar-cache line
    size_t n_t = 0;
    for (size_t i = 0; i < h->ops.size(); ++i) {
        const int* w = h->ops[i].w;
        h->wt_off.push_back(n_t);
        if (w[0] == OP_CONV) n_t += ((size_t)w[6] * w[7] * w[7] * w[3] + 15) / 16 * 16;
    }
    if (e == hipSuccess) e = hipMalloc(&h->d_blob_t, (n_t + 16) * sizeof(float));
    for (size_t i = 0; i < h->ops.size() && e == hipSuccess; ++i) {
        const int* w = h->ops[i].w;
        if (w[0] != OP_CONV) continue;
        const int cout = w[6], K = w[7] * w[7] * w[3];
        hipLaunchKernelGGL(transpose_w_kernel, dim3(cdiv(cout * K, 256)), dim3(256), 0, 0, h->d_blob + w[10], h->d_blob_t + h->wt_off[i], cout, K);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    for (int i = 0; i < n_guides && e == hipSuccess; ++i) {
        YoloGuide g{h_guides[i * 5], h_guides[i * 5 + 1], h_guides[i * 5 + 2], h_guides[i * 5 + 3], h_guides[i * 5 + 4], nullptr};
        if (g.embed < 1 || g.heads < 1 || g.embed % g.heads || !off_ok(g.w_off, (long long)g.embed * YOLO_TEXT) || !off_ok(g.b_off, g.embed) ||
            !off_ok(g.bias_off, g.heads) || (size_t)YOLO_MAX_Q * g.embed * sizeof(float) > 96 * 1024) {
            set_error("tstar_yolo_create: malformed attention layer"); return fail(TSTAR_ERR_ARG);
        }
        e = hipMalloc(&g.d_guide, (size_t)YOLO_SETS * YOLO_MAX_Q * g.embed * sizeof(float));
        h->guides.push_back(g);
    }
    for (int i = 0; i < n_levels; ++i) {
        YoloLevel l{h_levels[i * 8], h_levels[i * 8 + 1], h_levels[i * 8 + 2], h_levels[i * 8 + 3], h_levels[i * 8 + 4], 0.f, 0.f};
        if (!buf_ok(l.e_buf) || !buf_ok(l.r_buf) || (l.size * l.size) % 16 != 0 || h->buf_c[l.e_buf] != YOLO_TEXT || h->buf_c[l.r_buf] != 4 * YOLO_REG_MAX ||
            h->buf_h[l.e_buf] != l.size || h->buf_h[l.r_buf] != l.size || !off_ok(l.param_off, 2)) {
            set_error("tstar_yolo_create: malformed head level"); return fail(TSTAR_ERR_ARG);
        }
        l.logit_scale = h_blob[l.param_off]; l.bias = h_blob[l.param_off + 1];
        h->levels.push_back(l);
        h->n_anchor += l.size * l.size;
    }
    for (int i = 0; i < n_bufs && e == hipSuccess; ++i) {
        float* p = nullptr;
        const size_t n = (size_t)max_batch * h->buf_h[i] * h->buf_w[i] * h->buf_c[i];
        e = hipMalloc(&p, (n + 4) * sizeof(float));                  // + the zero quad padded conv taps read (conv_sw_kernel)
        if (e == hipSuccess) e = hipMemset(p + n, 0, 4 * sizeof(float));
        h->bufs.push_back(p);
    }
    const size_t nsq = (size_t)YOLO_SETS * YOLO_MAX_Q;
    if (e == hipSuccess) e = hipMalloc(&h->d_text, nsq * YOLO_TEXT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&h->d_textn, nsq * YOLO_TEXT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&h->d_qweight, nsq * sizeof(double));
    if (e == hipSuccess) e = hipMemset(h->d_qweight, 0, nsq * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_setQ, YOLO_SETS * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->d_setQ, 0, YOLO_SETS * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_iota, max_batch * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_cand_count, max_batch * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_boxes, (size_t)max_batch * h->n_anchor * 4 * sizeof(float));
    if (e == hipSuccess) {
        std::vector<int> io(max_batch);
        for (int i = 0; i < max_batch; ++i) io[i] = i;
        e = hipMemcpy(h->d_iota, io.data(), max_batch * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        set_error(std::string("tstar_yolo_create: allocation failed: ") + hipGetErrorString(e));
        return fail(TSTAR_ERR_HIP);
    }
    *out = h;
    return TSTAR_OK;
}

#define YCHECK_SET(set, fn) TSTAR_REQUIRE((set) >= 0 && (set) < YOLO_SETS, fn ": query_set must be in 0..63")

int tstar_yolo_set_text_feats(tstar_yolo* h, int query_set, const float* h_text, const double* h_class_weight, int Q, void* stream) {
    TSTAR_REQUIRE(h && h_text && h_class_weight, "tstar_yolo_set_text_feats: null argument");
    YCHECK_SET(query_set, "tstar_yolo_set_text_feats");
    TSTAR_REQUIRE(Q >= 1 && Q <= YOLO_MAX_Q, "tstar_yolo_set_text_feats: Q must be in 1..32");
    hipStream_t s = (hipStream_t)stream;
    const size_t qo = (size_t)query_set * YOLO_MAX_Q;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_text + qo * YOLO_TEXT, h_text, (size_t)Q * YOLO_TEXT * sizeof(float), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_qweight + qo, h_class_weight, Q * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(text_normalize_kernel, dim3(Q), dim3(64), 0, s, h->d_text + qo * YOLO_TEXT, h->d_textn + qo * YOLO_TEXT);
    for (auto& g : h->guides)
        hipLaunchKernelGGL(guide_fc_kernel, dim3(cdiv(Q * g.embed, 256)), dim3(256), 0, s, h->d_text + qo * YOLO_TEXT, Q,
                           h->d_blob + g.w_off, h->d_blob + g.b_off, g.embed, g.d_guide + qo * g.embed);
    TSTAR_HIP_CHECK(hipGetLastError());
    h->Q[query_set] = Q;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_setQ, h->Q, sizeof(h->Q), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}

int tstar_yolo_set_class_weights(tstar_yolo* h, int query_set, const double* h_class_weight, int Q, void* stream) {
    TSTAR_REQUIRE(h && h_class_weight, "tstar_yolo_set_class_weights: null argument");
    YCHECK_SET(query_set, "tstar_yolo_set_class_weights");
    TSTAR_REQUIRE(Q == h->Q[query_set] && Q >= 1, "tstar_yolo_set_class_weights: Q does not match the installed text features");
    hipStream_t s = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_qweight + (size_t)query_set * YOLO_MAX_Q, h_class_weight, Q * sizeof(double), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}

static int run_program(tstar_yolo* h, int B, const int* d_image_set, hipStream_t s) {
    for (size_t oi = 0; oi < h->ops.size(); ++oi) {
        const int* w = h->ops[oi].w;
        if (w[0] == OP_CONV) {
            ConvArgs a{};
            a.src = h->bufs[w[1]]; a.src_ld = h->buf_c[w[1]]; a.src_off = w[2]; a.cin = w[3]; a.H = h->buf_h[w[1]]; a.W = h->buf_w[w[1]];
            a.dst = h->bufs[w[4]]; a.dst_ld = h->buf_c[w[4]]; a.dst_off = w[5]; a.cout = w[6]; a.Ho = h->buf_h[w[4]]; a.Wo = h->buf_w[w[4]];
            a.ks = w[7]; a.stride = w[8]; a.act = w[9]; a.w = h->d_blob + w[10]; a.wt = h->d_blob_t + h->wt_off[oi]; a.bias = w[11] >= 0 ? h->d_blob + w[11] : nullptr;
            a.mode = w[12];
            if (a.mode != MODE_PLAIN) { a.aux = h->bufs[w[13]]; a.aux_ld = h->buf_c[w[13]]; a.aux_off = w[14]; a.heads = h->buf_c[w[13]]; }
            a.M = B * a.Ho * a.Wo;
            const size_t zo = (size_t)h->max_batch * a.H * a.W * a.src_ld;
            a.zoff = zo < (1ull << 30) ? (unsigned)zo : 0;
            RC(launch_conv(a, s));
        } else if (w[0] == OP_POOL5) {
            const int H = h->buf_h[w[1]], W = h->buf_w[w[1]];
            const size_t total = (size_t)B * H * W * (w[3] / 4);
            hipLaunchKernelGGL(pool5_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->bufs[w[1]], h->buf_c[w[1]], w[2], w[5],
                               w[3], H, W, h->bufs[w[1]], total);
        } else if (w[0] == OP_UPCOPY) {
            const int Hs = h->buf_h[w[1]], Ws = h->buf_w[w[1]], f = w[6];
            const size_t total = (size_t)B * Hs * f * Ws * f * (w[3] / 4);
            hipLaunchKernelGGL(upcopy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->bufs[w[1]], h->buf_c[w[1]], w[2], w[3],
                               Hs, Ws, f, h->bufs[w[4]], h->buf_c[w[4]], w[5], total);
        } else {
            const YoloGuide& g = h->guides[w[7]];
            const int HW = h->buf_h[w[1]] * h->buf_w[w[1]];
            const int bx = cdiv(HW * g.heads, 256);               // one (pixel, head) per thread (the cap of 64 blocks per image left the 80x80 maps at a quarter of the threads they need)
            const size_t lds = (size_t)YOLO_MAX_Q * g.embed * sizeof(float);
            RC(ensure_dyn_lds(reinterpret_cast<const void*>(attn_kernel), 96 * 1024));
            hipLaunchKernelGGL(attn_kernel, dim3(bx, B), dim3(256), lds, s, h->bufs[w[1]], h->buf_c[w[1]], w[2], g.embed, g.heads, HW,
                               g.d_guide, h->d_setQ, d_image_set, h->d_blob + g.bias_off, h->bufs[w[4]], B * HW);
        }
        TSTAR_HIP_CHECK(hipGetLastError());
    }
    return TSTAR_OK;
}

int tstar_yolo_detect(tstar_yolo* h, const uint8_t* d_images, int B, int H, int W, int grid_rows, int grid_cols,
                      const int32_t* h_image_query_set, float score_threshold, int max_dets, float* d_det_scores,
                      int32_t* d_det_labels, float* d_det_boxes, int32_t* d_n_det, double* d_cell_conf, uint32_t* d_cell_mask,
                      float* d_dense_scores, float* d_dense_boxes, void* stream) {
    TSTAR_REQUIRE(h && d_images && d_det_scores && d_det_labels && d_det_boxes && d_n_det, "tstar_yolo_detect: null argument");
    TSTAR_REQUIRE(B >= 1 && H >= 2 && W >= 2, "tstar_yolo_detect: empty batch or image");
    TSTAR_REQUIRE(max_dets >= 1 && max_dets <= YOLO_MAX_PER_IMG, "tstar_yolo_detect: max_dets must be in 1..300");
    TSTAR_REQUIRE(!d_cell_conf == !d_cell_mask, "tstar_yolo_detect: cell_conf and cell_mask go together");
    TSTAR_REQUIRE(!d_cell_conf || (grid_rows >= 1 && grid_cols >= 1 && grid_rows * grid_cols <= 4096), "tstar_yolo_detect: grid must have 1..4096 cells");
    hipStream_t s = (hipStream_t)stream;
    int q_uniform = -1, q_max = 0;
    for (int b = 0; b < B; ++b) {
        const int set = h_image_query_set ? h_image_query_set[b] : 0;
        YCHECK_SET(set, "tstar_yolo_detect");
        if (h->Q[set] == 0) { set_error("tstar_yolo_detect: no text features installed in the requested query set (call tstar_yolo_set_text_feats first)"); return TSTAR_ERR_STATE; }
        q_uniform = (b == 0 || q_uniform == h->Q[set]) ? h->Q[set] : 0;
        q_max = q_max > h->Q[set] ? q_max : h->Q[set];
    }
    TSTAR_REQUIRE(!d_dense_scores || q_uniform > 0, "tstar_yolo_detect: dense scores need the same query count for every image");
    if (h_image_query_set) {
        if (B > h->image_set_cap) {
            TSTAR_HIP_CHECK(hipStreamSynchronize(s));
            if (h->d_image_set) TSTAR_HIP_CHECK(hipFree(h->d_image_set));
            h->d_image_set = nullptr; h->image_set_cap = 0;
            TSTAR_HIP_CHECK(hipMalloc(&h->d_image_set, (size_t)B * sizeof(int)));
            h->image_set_cap = B;
        }
        TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_image_set, h_image_query_set, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
    }
    // candidate lists: every (anchor, class) pair can qualify
    const int need_cap = (int)pow2_at_least((size_t)h->n_anchor * q_max);
    if (need_cap > h->cand_cap) {
        TSTAR_HIP_CHECK(hipStreamSynchronize(s));
        if (h->d_cand) TSTAR_HIP_CHECK(hipFree(h->d_cand));
        h->d_cand = nullptr; h->cand_cap = 0;
        TSTAR_HIP_CHECK(hipMalloc(&h->d_cand, (size_t)h->max_batch * need_cap * sizeof(unsigned long long)));
        h->cand_cap = need_cap;
    }
    // mmyolo test pipeline geometry: YOLOv5KeepRatioResize(640) then LetterResize(640, allow_scale_up=False, pad 114)
    const double ratio = fmin(640.0 / (H > W ? H : W), 640.0 / (H < W ? H : W));
    const int rw = ratio != 1.0 ? (int)(W * ratio) : W, rh = ratio != 1.0 ? (int)(H * ratio) : H;
    TSTAR_REQUIRE(rw >= 1 && rh >= 1 && rw <= YOLO_IMG && rh <= YOLO_IMG, "tstar_yolo_detect: image shape outside the letterbox geometry");
    const double sfw = (double)rw / W, sfh = (double)rh / H;
    const int ph = YOLO_IMG - rh, pw = YOLO_IMG - rw;
    // LetterResize: top = int(round(padding_h // 2 - 0.1)) = padding_h // 2 (the -0.1 only breaks the .5 tie of a float half)
    const int top = ph / 2, left = pw / 2;
    // candidates exactly as mmyolo's predict_by_feat forms them (multi_label, score > score_thr = 0.001): the class-aware
    // NMS offsets depend on the largest coordinate among ALL of them, so the wrapper's 0.12 is not applied early
    const float cand_thr = YOLO_SCORE_THR;
    // chunks of max_batch + one remainder.  Cutting a batch into near-equal chunks instead (156 -> 78 + 78 under a capacity of
    // 96) was measured and is WORSE than 76 + 76 + 4 (bench conv average 98.3 vs 100.5 TFLOP/s): the chunk size is chosen so
    // that the halo layers fill whole rounds of the chip's 768 workgroup slots (20 B workgroups on a 40x40 / 256-channel
    // layer), and two images too many start a third, almost empty round on every such layer.
    for (int b0 = 0; b0 < B; b0 += h->max_batch) {
        const int Bc = (B - b0) < h->max_batch ? (B - b0) : h->max_batch;
        const uint8_t* imgs = d_images + (size_t)b0 * H * W * 3;
        const uint8_t* packed_src = imgs;
        if (rw != W || rh != H) {
            const size_t need = (size_t)Bc * rh * rw * 3;
            if (need > h->tmp_u8_bytes) {
                TSTAR_HIP_CHECK(hipStreamSynchronize(s));
                if (h->d_tmp_u8) TSTAR_HIP_CHECK(hipFree(h->d_tmp_u8));
                h->d_tmp_u8 = nullptr; h->tmp_u8_bytes = 0;
                TSTAR_HIP_CHECK(hipMalloc(&h->d_tmp_u8, need));
                h->tmp_u8_bytes = need;
            }
            if (ratio < 1.0) {
                const size_t total = (size_t)Bc * rh * rw;
                hipLaunchKernelGGL(area_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, imgs, H, W, h->d_tmp_u8, rh, rw, total);
                TSTAR_HIP_CHECK(hipGetLastError());
            } else {
                RC(bilinear_gather_u8(imgs, H, W, h->d_iota, Bc, rw, rh, h->d_tmp_u8, 0, s));
            }
            packed_src = h->d_tmp_u8;
        }
        {
            const size_t total = (size_t)Bc * YOLO_IMG * YOLO_IMG;
            hipLaunchKernelGGL(letterbox_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, packed_src, rh, rw, top, left,
                               h->bufs[h->input_buf], total);
            TSTAR_HIP_CHECK(hipGetLastError());
        }
        const int* d_sets = h_image_query_set ? h->d_image_set + b0 : nullptr;
        RC(run_program(h, Bc, d_sets, s));
        TSTAR_HIP_CHECK(hipMemsetAsync(h->d_cand_count, 0, Bc * sizeof(int), s));
        int anchor0 = 0;
        for (const YoloLevel& l : h->levels) {
            DecodeArgs a{};
            a.E = h->bufs[l.e_buf]; a.R = h->bufs[l.r_buf]; a.HW = l.size * l.size; a.Wl = l.size; a.stride = l.stride;
            a.anchor0 = anchor0; a.n_anchor = h->n_anchor; a.logit_scale = l.logit_scale; a.bias = l.bias;
            a.textn = h->d_textn; a.setQ = h->d_setQ; a.image_set = d_sets;
            a.pad_left = (float)left; a.pad_top = (float)top; a.sf_w = (float)sfw; a.sf_h = (float)sfh; a.cand_thr = cand_thr;
            a.boxes = h->d_boxes; a.cand = h->d_cand; a.cand_cap = h->cand_cap; a.cand_count = h->d_cand_count;
            a.dense_scores = d_dense_scores ? d_dense_scores + (size_t)b0 * h->n_anchor * q_uniform : nullptr; a.dense_q = q_uniform;
            const int rows = Bc * a.HW;
            hipLaunchKernelGGL(head_decode_kernel, dim3(cdiv(rows, HD_APB)), dim3(256), 0, s, a, rows);
            TSTAR_HIP_CHECK(hipGetLastError());
            anchor0 += a.HW;
        }
        RC(ensure_dyn_lds(reinterpret_cast<const void*>(sort_nms_kernel), NMS_LDS_KEYS * 8));
        hipLaunchKernelGGL(sort_nms_kernel, dim3(Bc), dim3(1024), (size_t)NMS_LDS_KEYS * 8, s, h->d_cand, h->cand_cap, h->d_cand_count, h->d_boxes, h->n_anchor,
                           (float)W, (float)H, score_threshold, max_dets, d_det_scores + (size_t)b0 * max_dets, d_det_labels + (size_t)b0 * max_dets,
                           d_det_boxes + (size_t)b0 * max_dets * 4, d_n_det + b0);
        TSTAR_HIP_CHECK(hipGetLastError());
        if (d_dense_boxes)
            TSTAR_HIP_CHECK(hipMemcpyAsync(d_dense_boxes + (size_t)b0 * h->n_anchor * 4, h->d_boxes, (size_t)Bc * h->n_anchor * 4 * sizeof(float),
                                           hipMemcpyDeviceToDevice, s));
        if (d_cell_conf) {
            const int ncell = grid_rows * grid_cols;
            hipLaunchKernelGGL(det_cells_kernel, dim3(Bc), dim3(64), (size_t)ncell * 12, s, d_det_scores + (size_t)b0 * max_dets,
                               d_det_labels + (size_t)b0 * max_dets, d_det_boxes + (size_t)b0 * max_dets * 4, d_n_det + b0, max_dets,
                               h->d_qweight, d_sets, W, H, grid_rows, grid_cols, d_cell_conf + (size_t)b0 * ncell, d_cell_mask + (size_t)b0 * ncell);
            TSTAR_HIP_CHECK(hipGetLastError());
        }
    }
    return TSTAR_OK;
}

int tstar_yolo_num_anchors(tstar_yolo* h) { return h ? h->n_anchor : 0; }

}  // extern "C"
