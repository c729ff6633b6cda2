// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue).
//
// This is the dominant kernel of the OWL-ViT scorer (patch-embed, QKV, out-proj,
// fc1, fc2, head projections: 102.5 of the 114.8 GFLOP per detector image (attention is the other 12.3),
// SURVEY.md 8d).  It replaces the torch.nn.Linear / Conv2d calls HF makes in
// modeling_owlvit.py:336-337, 428-459, 471-475, 993-999, 1020 on behalf of
// /root/reference/TStar/interface_heuristic.py:237-239.
//
// Design (gfx950):
//  * v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain), 64 cycles per
//    SIMD per issue, 157.3 TFLOP/s chip peak.  There is no TF32 on CDNA4.
//  * 128x128x32 block tile, 256 threads = 4 waves (2x2), each wave 64x64 =
//    2x2 MFMA tiles (64 accumulator VGPRs).  2 blocks per CU so one block's
//    MFMA stream covers the other's barrier / LDS-write bubbles.  Two smaller
//    tile shapes (64x128, 64x64) share the code; the launcher (pick_cfg) uses
//    them for grids under one wave of 128x128 tiles and, from one full wave
//    on, launches whole waves of 128x128 tiles followed by a 64x128 tail
//    (the hybrid kernel), so the last wave is quantised at half a tile.
//  * both operands are K-contiguous (activations row-major, nn.Linear weight
//    [out,in]) -> identical staging for A and W: global_load_dwordx4 ->
//    ds_write_b128 into a [128][32+4] padded tile (row stride 144 B makes the
//    ds_read_b128 fragment reads conflict-free: 36*r mod 64 hits 16 distinct
//    4-bank slots for r = 0..15), double-buffered, ONE barrier per K tile.
//  * a lane's b128 read delivers 4 consecutive k values; lanes 0-31 take
//    k = 8q..8q+3, lanes 32-63 k = 8q+4..8q+7, so four MFMAs consume one read
//    (the k-slot <-> k-index assignment of an MFMA is free as long as A and B
//    agree).
//  * 1-D grid with an XCD-aware bijective remap: consecutive tiles (same A row
//    panel) land on the same XCD's L2.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include <type_traits>
#include <utility>

namespace tstar {

constexpr int BK = 32, LDS_LD = BK + 4;

// Linear tile index -> (row tile, column tile).  gm <= 1: row-panel major (all column tiles of a row panel are consecutive).
// gm > 1: SUPER-PANELS of gm row panels, column-major inside one -- the ~64 tiles an XCD has in flight then form a
// gm x (64 / gm) patch that shares gm A panels and 64 / gm W tiles in that XCD's L2, instead of one or two A panels against
// every W tile: with the panel-major order a wide layer (fc1: 24 column tiles) re-streams the whole weight matrix from the
// Infinity Cache for every 2-3 row panels (profiles/r03_pmc_gemm_traffic.json: 2.5x the algorithmic bytes per launch).
__device__ __forceinline__ void tile_mn(int tile, int mt, int nt, int gm, int& mi, int& ni) {
    if (gm <= 1) { mi = tile / nt; ni = tile - mi * nt; return; }
    const int per = gm * nt;
    const int sp = tile / per, r = tile - sp * per;
    const int left = mt - sp * gm;
    const int rows = left < gm ? left : gm;              // the last super-panel may be short
    ni = r / rows;
    mi = sp * gm + (r - ni * rows);
}

__device__ __forceinline__ float epi_act(float v, int act) {
    // quick_gelu(v) = v * sigmoid(1.702 v); exp through the raw v_exp_f32 (2^x, ~1 ulp): the epilogue runs
    // and the reciprocal through v_rcp_f32 (1 ulp): the epilogue runs while the matrix pipe idles, so its VALU
    // cost is on the critical path of short-K launches
    if (act == ACT_QGELU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896340736f * v));
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

// Tile configuration: WM x WN waves (always 4), each wave TM x TN MFMA tiles of 32x32.
//   Cfg128: 128x128 block, 2 blocks/CU  -- best when the grid fills whole waves of 512 slots
//   Cfg64N: 64x128 block,  2 blocks/CU  -- halves the tail for narrow-N / mid-size grids
//   Cfg64:  64x64 block,   4 blocks/CU  -- small M (one grid image, text tower)
template <int TM_, int TN_, int MINW_>
struct Cfg { static constexpr int TM = TM_, TN = TN_, MINW = MINW_, BM = 2 * TM_ * 32, BN = 2 * TN_ * 32; };
using Cfg128 = Cfg<2, 2, 2>;
using Cfg64N = Cfg<1, 2, 2>;
using Cfg64 = Cfg<1, 1, 4>;
using Cfg128W = Cfg<2, 4, 2>;      // 128x256 block (each wave 64x128): the two-term bf16-weight tile, where LDS traffic per MFMA decides

// Epilogue shared by the f32 and the bf16-pipe tiles.  C/D layout of the 32x32 MFMA:
// col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
//
// Structure matters here: with one bounds-checked store per basic block the compiler has to re-wait
// (s_waitcnt vmcnt(0)) for the bias / residual loads in EVERY block, which also waits for the previous
// store's write acknowledgement -- 64 serialised round trips (~300 cycles each, ~20k cycles per tile,
// measured as a flat C-bytes / 4 TB/s surcharge per launch).  So: per 32x32 sub-tile all loads (residual,
// position embedding) are issued first, the 16 results are formed in registers, then the 16 stores go out
// back to back; tiles that lie completely inside M (all but the last row panel) take an unguarded path
// without any per-element branch.
template <class CF, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH, bool GUARD>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& g, f32x16 (&acc)[CF::TM][CF::TN], const int m0, const int n0,
                                                   const int wm, const int wn, const int l31, const int h) {
    constexpr int TM = CF::TM, TN = CF::TN;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        const float bv = HAS_BIAS ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
            size_t off[16];
            float extra[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = rbase + (r & 3) + 8 * (r >> 2);
                if (GUARD) row = row < g.M ? row : g.M - 1;          // loads stay in bounds; the store is predicated
                size_t orow = row;
                float e = 0.f;
                if (PATCH) {
                    // patch-embed rows (b, p) -> token rows (b, 1 + p), + position embedding
                    const int b = row / g.patch_np, p = row - b * g.patch_np;
                    orow = (size_t)b * (g.patch_np + 1) + 1 + p;
                    e = g.pos[(size_t)(1 + p) * g.N + col];
                }
                off[r] = orow * g.ldc + col;
                if (HAS_RES) e += g.res[off[r]];
                extra[r] = e;
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = epi_act(acc[i][j][r] + bv, ACT);
                if (PATCH || HAS_RES) v[r] += extra[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!GUARD || rbase + (r & 3) + 8 * (r >> 2) < g.M) g.C[off[r]] = v[r];
            }
        }
    }
}

// The 128x256 tile carries 128 accumulator registers into the epilogue.  Same scheme as above, but row-major over the
// sub-tiles (the 16 row offsets of a 32-row band are formed once, as 32-bit element offsets, and serve its four column
// tiles) with a scheduling barrier per sub-tile: in the column-major order the compiler kept the 64-bit offsets of all 32
// rows live across the column tiles and spilled ~130 registers to scratch.
template <class CF, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH, bool GUARD>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmArgs& g, f32x16 (&acc)[CF::TM][CF::TN], const int m0, const int n0,
                                                   const int wm, const int wn, const int l31, const int h) {
    constexpr int TM = CF::TM, TN = CF::TN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
        unsigned roff[16], poff[PATCH ? 16 : 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = rbase + (r & 3) + 8 * (r >> 2);
            if (GUARD) row = row < g.M ? row : g.M - 1;
            unsigned orow = (unsigned)row;
            if (PATCH) {
                const int b = row / g.patch_np, p = row - b * g.patch_np;
                orow = (unsigned)(b * (g.patch_np + 1) + 1 + p);
                poff[r] = (unsigned)(1 + p) * (unsigned)g.N;
            }
            roff[r] = orow * (unsigned)g.ldc;                 // element offsets fit 32 bits (checked by the launcher)
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
            const float bv = HAS_BIAS ? g.bias[col] : 0.f;
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = 0.f;
                if (PATCH) x = g.pos[(size_t)poff[r] + col];
                if (HAS_RES) x += g.res[(size_t)roff[r] + col];
                e[r] = x;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = epi_act(acc[i][j][r] + bv, ACT);
                if (PATCH || HAS_RES) v += e[r];
                e[r] = v;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!GUARD || rbase + (r & 3) + 8 * (r >> 2) < g.M) g.C[(size_t)roff[r] + col] = e[r];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <class CF, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[CF::TM][CF::TN], const int m0, const int n0,
                                              const int wm, const int wn, const int l31, const int h) {
    if constexpr (CF::TN >= 4) {
        if (m0 + CF::BM <= g.M) gemm_epilogue_wide<CF, ACT, HAS_BIAS, HAS_RES, PATCH, false>(g, acc, m0, n0, wm, wn, l31, h);
        else gemm_epilogue_wide<CF, ACT, HAS_BIAS, HAS_RES, PATCH, true>(g, acc, m0, n0, wm, wn, l31, h);
        return;
    }
    if (m0 + CF::BM <= g.M) gemm_epilogue_impl<CF, ACT, HAS_BIAS, HAS_RES, PATCH, false>(g, acc, m0, n0, wm, wn, l31, h);
    else gemm_epilogue_impl<CF, ACT, HAS_BIAS, HAS_RES, PATCH, true>(g, acc, m0, n0, wm, wn, l31, h);
}

// ---------------------------------------------------------------------------------------------
// bf16-WEIGHT tile (BASELINE config 5, "bf16 ViT weights"): W is stored as bfloat16 (half the
// bytes), the float32 activations are split EXACTLY into three bfloat16 terms
// a = a0 + a1 + a2 (8 + 8 + 8 significand bits, truncation split) while they are staged into LDS,
// and C += a0*w + a1*w + a2*w runs on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate per
// instruction, 3 instructions per K=16).  Every product is exact in f32, so the result is an
// f32-accumulated dot product of the f32 activations with the bf16 weights -- the same quantity
// the f32 tile computes from bf16-valued weights (measured error 1.2e-7 * sum|a w|, i.e.
// f32-roundoff class) at 2.4-2.6x the speed.
// LDS tiles are [rows][32 bf16] = 64 B per row with the 16-B chunk index XOR-swizzled by
// (row >> 2) & 3, which makes both the ds_read_b128 fragment reads and the staging writes
// conflict-free without padding (3 A-term tiles + 1 W tile, double buffered = 64 KB at 128x128).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int bfw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void split4_bf16x3(const f32x4 v, u32x2 (&out)[3]) {
    unsigned t[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = v[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned b = __float_as_uint(r) & 0xFFFF0000u;      // top 8 significand bits
            t[k][e] = b;
            r = r - __uint_as_float(b);                               // exact
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out[k][0] = (t[k][0] >> 16) | t[k][1];
        out[k][1] = (t[k][2] >> 16) | t[k][3];
    }
}

// Two-term round-to-nearest split for the SPLIT mode: hi = bf16(v), lo = bf16(v - hi); v - hi - lo is
// below 2^-17 |v| and unbiased.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4_bf16x2(const f32x4 v, u32x2 (&out)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 x; x[0] = v[2 * p]; x[1] = v[2 * p + 1];
        const bf16x2 hi = __builtin_convertvector(x, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, hi);
        f32x2 r;
        r[0] = x[0] - __uint_as_float(hb << 16);                      // exact
        r[1] = x[1] - __uint_as_float(hb & 0xFFFF0000u);
        const bf16x2 lo = __builtin_convertvector(r, bf16x2);
        out[0][p] = hb;
        out[1][p] = __builtin_bit_cast(unsigned, lo);
    }
}

// WMODE 1: bf16 weights (Wb), activations split exactly into 3 terms.
// (WMODE 2, the round-1..3 "f32 split" of both operands into two bf16 terms, is retired: the f32x3 tile below carries
//  all 24 bits of both operands.)
// WMODE 3: bf16 weights (exact in ONE term), activations as TWO round-to-nearest bf16 terms a_hi + a_lo
//          (16 significand bits, |a - a_hi - a_lo| <= 2^-17 |a|): C += a_lo*w + a_hi*w, 2 MFMAs per K = 16
//          instead of 3.  Products are exact, accumulation is f32; the dropped third term is 2^-17 relative
//          per product with random sign -- three orders of magnitude inside the 1e-3 score contract (tests
//          state the measured bound).  The default of BASELINE config 5; mode 1 stays selectable.
//          With two thirds of the MFMAs the 128x128 tile would be bound by LDS traffic (A is written as two
//          planes and read back once per N tile: 499 LDS cycles against 512 matrix-pipe cycles per K tile),
//          so whole waves of the launch use a 128x256 tile (Cfg128W, each wave 64x128: 8 fragment reads per
//          16 MFMAs instead of 6 per 8; 666 LDS cycles against 1024), register prefetch one K tile deep
//          (a K tile is 1024 matrix-pipe cycles per wave, two waves per SIMD: longer than an L2 round trip).
template <class CF, int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__device__ __forceinline__ void gemm_tile_bf16w(const GemmArgs& g, const int m0, const int n0, float* smem_f) {
    constexpr int TM = CF::TM, TN = CF::TN, BM = CF::BM, BN = CF::BN;
    constexpr int NAT = WMODE == 1 ? 3 : 2, NWT = 1;                      // operand terms
    constexpr int NPROD = WMODE == 3 ? 2 : 3;                             // MFMA products per algorithmic product
    constexpr int DEPTH = TN >= 4 ? 1 : 2;                                // register prefetch depth in K tiles
    constexpr int A_T = BM * 64, W_T = BN * 64, BUF = NAT * A_T + NWT * W_T;       // bytes
    char* smem = reinterpret_cast<char*>(smem_f);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int NA = BM / 32, NB = BN / 64;
    const int c4 = t & 7, r0 = t >> 3;            // A: rows r0 + 32 i, float4 column c4
    const int wc = t & 3, wr = t >> 2;            // W: rows wr + 64 i, 16-B chunk wc
    const float* ap[NA];
    size_t bo[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int ar = m0 + r0 + 32 * i;
        ar = ar < g.M ? ar : g.M - 1;
        ap[i] = g.A + (size_t)ar * g.lda + c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) bo[i] = (size_t)(n0 + wr + 64 * i) * g.K + wc * 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register prefetch is TWO K tiles deep (sets 0/1): with the bf16 pipe a K tile is only ~770 matrix-pipe
    // cycles per wave, less than one L2 round trip, so a one-deep prefetch (as in the f32 tile, whose K tile
    // is 4096 cycles) would expose the load latency at every tile.
    f32x4 ra[DEPTH][NA];
    u32x4 rb[DEPTH][NWT][NB];
    auto gload = [&](auto SET, int kt) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[st][i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[st][0][i] = *reinterpret_cast<const u32x4*>(g.Wb + bo[i] + kt * BK);
        }
    };
    auto lstore = [&](auto SET, int buf) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 sp[NAT];
            if constexpr (WMODE == 1) split4_bf16x3(ra[st][i], sp);
            else split4_bf16x2(ra[st][i], sp);
            const int off = bfw_off(r0 + 32 * i, c4 >> 1) + (c4 & 1) * 8;
#pragma unroll
            for (int k = 0; k < NAT; ++k) *reinterpret_cast<u32x2*>(base + k * A_T + off) = sp[k];
        }
#pragma unroll
        for (int w = 0; w < NWT; ++w)
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<u32x4*>(base + NAT * A_T + w * W_T + bfw_off(wr + 64 * i, wc)) = rb[st][w][i];
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* base = smem + buf * BUF;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {                       // two K = 16 steps per tile
            bf16x8 fa[NAT][TM], fw[NWT][TN];
#pragma unroll
            for (int k = 0; k < NAT; ++k)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[k][i] = *reinterpret_cast<const bf16x8*>(base + k * A_T + bfw_off(wm * TM * 32 + i * 32 + l31, 2 * s + h));
#pragma unroll
            for (int w = 0; w < NWT; ++w)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fw[w][j] = *reinterpret_cast<const bf16x8*>(base + NAT * A_T + w * W_T + bfw_off(wn * TN * 32 + j * 32 + l31, 2 * s + h));
            // smallest term first: mode 1 a2*w, a1*w, a0*w; mode 3 a_lo*w, a_hi*w
#pragma unroll
            for (int p = NPROD - 1; p >= 0; --p) {
                const int ka = p, kw = 0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ka][i], fw[kw][j], acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, DEPTH - 1>;      // (DEPTH == 1 never reaches the two-deep loop below)
    const int nk = g.K / BK;
    if constexpr (DEPTH == 1) {
        gload(S0{}, 0);
        lstore(S0{}, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) gload(S0{}, kt + 1);
            compute(kt & 1);
            if (more) lstore(S0{}, (kt + 1) & 1);
            __syncthreads();
        }
        gemm_epilogue<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, acc, m0, n0, wm, wn, l31, h);
        return;
    }
    gload(S0{}, 0);
    if (nk > 1) gload(std::integral_constant<int, DEPTH - 1>{}, 1);
    lstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even tile: LDS buffer 0; set 0 is free again (tile kt went to LDS), set 1 holds tile kt+1
        if (kt + 2 < nk) gload(S0{}, kt + 2);
        compute(0);
        if (kt + 1 < nk) lstore(S1{}, 1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd tile: LDS buffer 1; set 1 is free, set 0 holds tile kt+2
        if (kt + 3 < nk) gload(S1{}, kt + 3);
        compute(1);
        if (kt + 2 < nk) lstore(S0{}, 0);
        __syncthreads();
    }
    gemm_epilogue<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, acc, m0, n0, wm, wn, l31, h);
}

// ---------------------------------------------------------------------------------------------
// f32x3 tile (WMODE 4, weights_dtype "f32x3", round 4): an f32 x f32 product on the bf16 matrix pipe with ALL 24
// significand bits of both operands.  a = a0 + a1 + a2 and w = w0 + w1 + w2 exactly, each term a round-to-nearest
// bfloat16 of the running remainder (8 + 8 + 8 bits with signed remainders; v_cvt_pk_bf16_f32), and
//     C += a0 w2 + (a1 w1 + a0 w1) + (a2 w0 + a1 w0 + a0 w0)          per K = 16 step, smallest terms first,
// the six partial products with ka + kw <= 2 on v_mfma_f32_32x32x16_bf16 (each product exact, f32 accumulation).  The
// three products left out (a1 w2, a2 w1, a2 w2) are each <= 2^-24 |a w| and zero-mean: two orders below the rounding of
// the f32 accumulation itself.  Measured against float64 on the same inputs (tools/lab/frag_lab.hip,
// profiles/r04_frag_lab.log; tests/test_gpu_kernels.py::test_gemm_f32x3 asserts it): rms error 3.48e-7 (K = 768) /
// 1.44e-6 (K = 3072) -- the SAME four digits as the nine-product form, and below the native f32 MFMA tile's 4.14e-7 /
// 1.64e-6 (its accumulator is rounded once per k, this one six times per 16 k).
// Why six and not nine: nine 32-cycle products against eight 64-cycle f32 MFMAs is 1.78x on paper, but random operands
// pull the bf16 pipe's clock to 1.6-1.7 GHz where the f32 MFMA keeps 2.3: the nine-product loop measures 133-145 TFLOP/s,
// the native tile 132-138 (profiles/r04_frag_lab_counters.md).  Six products: 177-201.
//
// Main loop (different from the tiles above): the WEIGHT planes never touch LDS.  They are packed once per matrix in
// MFMA-fragment order -- [n-tile of 32 columns][phase q = 3 * (k / 16) + ph][lane][8 bf16], plane 2 - ph, so a wave's
// B operand of one phase is ONE contiguous 1-KB buffer_load_dwordx4 -- and stream global -> VGPR, every fragment a full
// K = 16 step ahead of its use (gemm_tile_x3 below).  LDS carries only the three activation planes (split while staging, the
// swizzled 64-B rows of the tiles above).
// (For the two-term bf16-weight mode a global -> VGPR weight path is SLOWER than its LDS tile -- 4 MFMAs per loaded KB put
// the fragment loads on the 64 B/clk vector-memory path at ~75 % -- so that mode keeps its tile: r04_frag_lab_counters.md.)
template <int NA>
__device__ __forceinline__ void split4_rn(const f32x4 v, u32x2 (&out)[NA]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 x; x[0] = v[2 * p]; x[1] = v[2 * p + 1];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const bf16x2 b = __builtin_convertvector(x, bf16x2);
            const unsigned hb = __builtin_bit_cast(unsigned, b);
            out[k][p] = hb;
            if (k + 1 < NA) {
                x[0] = x[0] - __uint_as_float(hb << 16);                  // exact
                x[1] = x[1] - __uint_as_float(hb & 0xFFFF0000u);
            }
        }
    }
}

// one pair of floats -> three packed bf16 pairs (round-to-nearest terms of the running remainder).  Scalar subtractions: packed
// f32 VALU beside MFMAs costs more than its issue slot (MI355X guide, "price of one filler beside MFMAs")
__device__ __forceinline__ void split2_rn3(float x0, float x1, unsigned (&o)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f32x2 x; x[0] = x0; x[1] = x1;
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        o[k] = hb;
        if (k < 2) {
            x0 = x0 - __uint_as_float(hb << 16);                      // exact
            x1 = x1 - __uint_as_float(hb & 0xFFFF0000u);
        }
    }
}

// Round 5: the loop is software-pipelined per WAVE (lab: tools/lab/x3_lab.hip, profiles/r05_x3_lab.log: 181-200 -> 225-242 TFLOP/s
// algorithmic on the B = 256 shapes, matrix pipe busy 0.60 -> 0.80-0.86; with zero operands, i.e. without the power limit,
// 2.08 PFLOP/s executed = 0.83 of the dense bf16 peak).  The round-4 loop ran, per K tile, [fragment reads -> wait] 48 MFMAs
// [reads -> wait] 48 MFMAs [split of the next tile: ~100 VALU] [LDS stores] [barrier], and the compiler had folded its 3-set
// weight-fragment ring so that most fragment loads were waited for 8-16 MFMAs after their issue: every such section is a hole in
// the wave's MFMA stream that only the other block's wave on the same SIMD can fill.  Now one wave's stream never waits for
// something it has just asked for:
//   * weight fragments: 3 * TN per K = 16 step, each in its OWN 4-register slot, re-loaded for the next step right after its
//     last MFMA of this step (buffer loads: descriptor + 32-bit lane offset + scalar offset, no 64-bit VALU address math) ->
//     42-46 MFMAs (1300+ cycles) between a load and its first use;
//   * MFMA order of a step: per fragment all its products back to back (w0: a2, a1, a0; w1: a1, a0; w2: a0), two column tiles
//     interleaved so that MFMAs on one accumulator are 2 * TM apart.  Per accumulator and step: a2 w0, a1 w0, a0 w0, a1 w1,
//     a0 w1, a0 w2 -- the same six exact products as before in another order (rms error vs float64 3.53e-7 at K = 768,
//     1.45e-6 at K = 3072; round 4's order 3.49e-7 / 1.44e-6; native f32 MFMA tile 4.14e-7 / 1.64e-6);
//   * activation fragments: a2 / a1 single-buffered and re-read from LDS as soon as their last product of the step has issued
//     (24 / 8 MFMAs before their next use at TN = 4), a0 double-buffered;
//   * the split of the NEXT K tile and its LDS stores are cut into pieces that ride between the MFMA groups of the tile's FIRST
//     step; the ONE barrier per K tile sits between the two steps, where every operand of the second step is already in
//     registers (RAW: tile kt + 1 is written in step 0 of tile kt and first read in its step 1; WAR: the buffer written in step 0
//     of tile kt + 1 was last read in step 0 of tile kt, before that tile's barrier -- the wait for those reads precedes it);
//   * loop body without branches: loads past the end are clamped, the stores of a tile past the end go to the idle buffer.
template <class CF, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__device__ __forceinline__ void gemm_tile_x3(const GemmArgs& g, const int m0, const int n0, float* smem_f) {
    constexpr int TM = CF::TM, TN = CF::TN, BM = CF::BM;
    constexpr int A_T = BM * 64, BUF = 3 * A_T;                           // bytes: one plane, one buffer (three planes)
    constexpr int NA = BM / 32;                                           // float4 loads per thread per K tile
    constexpr int PAIR = TN >= 2 ? 2 : 1, GP = TN / PAIR;                 // column tiles interleaved, groups of them
    // weight-fragment prefetch distance in K = 16 steps.  One step is 6 * TM * TN MFMAs: 48 (1500+ cycles) on the batch tiles, but only 6
    // (~190 cycles, a fraction of the L2 latency) on the 64x64 tile of the M = 577 launches, whose 120-480 blocks cannot hide it by
    // occupancy either -- there the fragments of FOUR steps are in flight (a ring of 4 slot sets, the loop unrolled by two K tiles).
    // Same MFMA order per accumulator: same bits.
    constexpr int WD = TM * TN == 1 ? 4 : 1;                              // (2 on the 64x128 / 128x128 tiles measured no different: r05_x3_cfg_sweep_small_m_prefetch.log)
    static_assert(TN % PAIR == 0, "column tiles go in pairs");
    char* smem = reinterpret_cast<char*>(smem_f);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    // activation rows: one base per block + a 32-bit byte offset per staged row (clamped to the last row)
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int ar = m0 + r0 + 32 * i;
        ar = ar < g.M ? ar : g.M - 1;
        aoff[i] = (unsigned)(ar - m0) * (unsigned)g.lda * 4u + c4 * 16;
    }
    const int nsteps = g.K / 16, nk = g.K / BK;
    // fragment stream of this wave's column tile j: (j * nsteps + step) * 3072 + ph * 1024 + lane * 16, ph = 2 - kw
    const char* wrow = static_cast<const char*>(g.Wp) + (size_t)((n0 >> 5) + wn * TN) * nsteps * 3072;
    const __amdgpu_buffer_rsrc_t ra_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)m0 * g.lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wrow), 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    // LDS (64-B rows, 16-B chunk XOR-swizzled by (row >> 2) & 3): fragment read offsets of step 0 / 1, staging write offset
    const int rd0 = (wm * TM * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);
    const int rd1 = rd0 ^ 32;
    const int wr0 = r0 * 64 + (((c4 >> 1) ^ ((r0 >> 2) & 3)) << 4) + (c4 & 1) * 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA];
    u32x4 w[WD][3][TN];                                                   // w[slot][kw][j]: plane kw of column tile j; slot = step % WD
    bf16x8 a0[2][TM], a1[TM], a2[TM];
    unsigned sp[2][3];

    auto gload_a = [&](int kt) __attribute__((always_inline)) {
        kt = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)aoff[i], kt * (BK * 4), 0));
    };
    auto wload = [&](auto SL, auto KW, auto J, int gs) __attribute__((always_inline)) {
        constexpr int sl = decltype(SL)::value, kw = decltype(KW)::value, j = decltype(J)::value;
        gs = gs < nsteps ? gs : nsteps - 1;
        w[sl][kw][j] = __builtin_amdgcn_raw_buffer_load_b128(rw_rsrc, voff + (2 - kw) * 1024, (j * nsteps + gs) * 3072, 0);
    };
    auto rd_a = [&](const char* base, int off, int plane, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + off + plane * A_T + i * 2048);
    };
    // staging piece p of 2 * NA: half p & 1 of float4 p >> 1; the three 8-byte stores follow the second half
    auto stage_piece = [&](auto P, char* wbase) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value, i = p >> 1, hf = p & 1;
        split2_rn3(ra[i][2 * hf], ra[i][2 * hf + 1], sp[hf]);
        if constexpr (hf == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                u32x2 v; v[0] = sp[0][k]; v[1] = sp[1][k];
                *reinterpret_cast<u32x2*>(wbase + wr0 + k * A_T + i * 2048) = v;
            }
        }
    };
    auto mf = [&](const bf16x8& a, const u32x4& b, f32x16& c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    // one K = 16 step.  S = step inside the tile (selects the a0 set); rbase / roff: where the NEXT step's fragments are read;
    // wbase: where the next tile's planes go (only S == 0 stages); gnext: fragment-stream step of the reloads; ktload: the tile
    // whose activations are loaded once the staging registers are free
    auto step = [&](auto SS, auto SLOT, const char* rbase, int roff, char* wbase, int gnext, int ktload) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;
        using SL = decltype(SLOT);
        auto& ws = w[SL::value];                                          // this step's fragments; each is reloaded (step gnext) after its last product
        constexpr int NPIECE = NA * 2, NG = 6 * GP, PPG = (NPIECE + 1 + NG - 1) / NG;
        auto after_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int gi = decltype(G)::value;
            if constexpr (S == 0) {
                [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                    ([&] {
                        constexpr int p = gi * PPG + Q;
                        if constexpr (p < NPIECE) stage_piece(std::integral_constant<int, p>{}, wbase);
                        if constexpr (p == NPIECE) gload_a(ktload);
                    }(), ...);
                }(std::make_integer_sequence<int, PPG>{});
            }
            if constexpr (gi == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[S ^ 1][i] = rd_a(rbase, roff, 0, i);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // w0 fragments: a2, a1, a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = PAIR * JP;
#pragma unroll
                for (int jj = 0; jj < PAIR; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a2[i], ws[0][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < PAIR; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], ws[0][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * JP + 1>{});
#pragma unroll
                for (int jj = 0; jj < PAIR; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], ws[0][j0 + jj], acc[i][j0 + jj]);
                [&]<int... JJ>(std::integer_sequence<int, JJ...>) __attribute__((always_inline)) {
                    (wload(SL{}, std::integral_constant<int, 0>{}, std::integral_constant<int, j0 + JJ>{}, gnext), ...);
                }(std::make_integer_sequence<int, PAIR>{});
                after_group(std::integral_constant<int, 3 * JP + 2>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
        // a2's last product of the step has issued: the next step's a2
#pragma unroll
        for (int i = 0; i < TM; ++i) a2[i] = rd_a(rbase, roff, 2, i);
        __builtin_amdgcn_sched_barrier(0);
        // w1 fragments: a1, a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = PAIR * JP;
#pragma unroll
                for (int jj = 0; jj < PAIR; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], ws[1][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * GP + 2 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < PAIR; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], ws[1][j0 + jj], acc[i][j0 + jj]);
                [&]<int... JJ>(std::integer_sequence<int, JJ...>) __attribute__((always_inline)) {
                    (wload(SL{}, std::integral_constant<int, 1>{}, std::integral_constant<int, j0 + JJ>{}, gnext), ...);
                }(std::make_integer_sequence<int, PAIR>{});
                after_group(std::integral_constant<int, 3 * GP + 2 * JP + 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = rd_a(rbase, roff, 1, i);
        __builtin_amdgcn_sched_barrier(0);
        // w2 fragments: a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = PAIR * JP;
#pragma unroll
                for (int jj = 0; jj < PAIR; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], ws[2][j0 + jj], acc[i][j0 + jj]);
                [&]<int... JJ>(std::integer_sequence<int, JJ...>) __attribute__((always_inline)) {
                    (wload(SL{}, std::integral_constant<int, 2>{}, std::integral_constant<int, j0 + JJ>{}, gnext), ...);
                }(std::make_integer_sequence<int, PAIR>{});
                after_group(std::integral_constant<int, 5 * GP + JP>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
    };

    // ---- prologue: tile 0 into LDS buffer 0, step 0's weight fragments, tile 1 into the staging registers
    gload_a(0);
    [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {      // steps 0 .. WD - 1 into their slots
        ((wload(std::integral_constant<int, Q / TN>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, Q % TN>{}, Q / TN),
          wload(std::integral_constant<int, Q / TN>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, Q % TN>{}, Q / TN),
          wload(std::integral_constant<int, Q / TN>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, Q % TN>{}, Q / TN)), ...);
    }(std::make_integer_sequence<int, WD * TN>{});
    [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
        (stage_piece(std::integral_constant<int, P>{}, smem), ...);
    }(std::make_integer_sequence<int, NA * 2>{});
    gload_a(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { a0[0][i] = rd_a(smem, rd0, 0, i); a1[i] = rd_a(smem, rd0, 1, i); a2[i] = rd_a(smem, rd0, 2, i); }
    // one static priority for the whole loop (over a co-resident block's prologue / epilogue): +1 % in the lab
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
    // one K tile: step 0 (the next fragments are this tile's step 1; stages tile kt + 1 into the other buffer, then loads tile kt + 2),
    // the barrier, step 1 (the next fragments are tile kt + 1, step 0).  SL0 = slot of the tile's first step
    auto tile = [&](auto SL0, int kt) __attribute__((always_inline)) {
        constexpr int s0 = decltype(SL0)::value;
        char* cur = smem + (kt & 1) * BUF;
        char* nxt = smem + ((kt + 1) & 1) * BUF;
        step(std::integral_constant<int, 0>{}, std::integral_constant<int, s0>{}, cur, rd1, nxt, 2 * kt + WD, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        step(std::integral_constant<int, 1>{}, std::integral_constant<int, (s0 + 1) % WD>{}, nxt, rd0, nxt, 2 * kt + 1 + WD, 0);
    };
    if constexpr (WD <= 2) {
        for (int kt = 0; kt < nk; ++kt) tile(std::integral_constant<int, 0>{}, kt);      // a tile's two steps are slots 0 and (WD == 2) 1
    } else {
        static_assert(WD == 4, "the loop below is unrolled by WD / 2 = 2 K tiles");
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            tile(std::integral_constant<int, 0>{}, kt);
            tile(std::integral_constant<int, 2>{}, kt + 1);
        }
        if (kt < nk) tile(std::integral_constant<int, 0>{}, kt);          // odd tile count: kt is even, its steps are slots 0 and 1
    }
    __builtin_amdgcn_s_setprio(0);
    gemm_epilogue<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, acc, m0, n0, wm, wn, l31, h);
}

// Wp[n-tile][q = 3 * ks + ph][lane][e] = plane (2 - ph) of W[32 n-tile + (lane & 31)][16 ks + 8 (lane >> 5) + e]; one thread per
// (row, 8-k chunk): 32 B read, three 16-B stores.
__global__ __launch_bounds__(256) void pack_x3_kernel(const float* __restrict__ W, char* __restrict__ Wp, int N, int K) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int kc = K / 8;
    if (gid >= (size_t)N * kc) return;
    const int n = (int)(gid / kc), c = (int)(gid % kc);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + c * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + c * 8 + 4);
    u32x2 s0[3], s1[3];
    split4_rn<3>(v0, s0);
    split4_rn<3>(v1, s1);
    const int ks = c >> 1, hh = c & 1, nph = (K / 16) * 3;
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        u32x4 o;
        o[0] = s0[2 - ph][0]; o[1] = s0[2 - ph][1]; o[2] = s1[2 - ph][0]; o[3] = s1[2 - ph][1];
        *reinterpret_cast<u32x4*>(Wp + (((size_t)(n >> 5) * nph + ks * 3 + ph) * 64 + hh * 32 + (n & 31)) * 16) = o;
    }
}

int pack_weights_x3(const float* W, void* Wp, int N, int K, hipStream_t s) {
    TSTAR_REQUIRE(N % 32 == 0 && K % 16 == 0, "pack_weights_x3: N must be a multiple of 32, K of 16");
    const size_t n = (size_t)N * (K / 8);
    hipLaunchKernelGGL(pack_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, static_cast<char*>(Wp), N, K);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// ---------------------------------------------------------------------------------------------
// Two-term tile with the weight fragments global -> VGPR (round 6; lab: tools/lab/x3_lab.hip `gemm_w2_v2`, profiles/r05_two_term_v2_lab.log).
// The same arithmetic as gemm_tile_bf16w<.., WMODE 3>: bf16 weights (one exact plane), activations split while staged into
// a_hi = bf16(a), a_lo = bf16(a - a_hi), and per K = 16 step and accumulator  C += a_lo w, then C += a_hi w  -- same products, same order,
// same bits -- on the per-wave software pipeline of gemm_tile_x3: the weight plane packed in MFMA-fragment order
// ([n-tile of 32][k / 16][lane][8 bf16]: a wave's B operand of one step is ONE contiguous 1-KB buffer load), fragments in two slot sets
// re-loaded two steps ahead right after their last product, LDS carries only the two activation planes (32 KB per 128-row block), the split
// of the next K tile in pieces between the MFMA groups of a tile's first step, ONE barrier per K tile between its two steps.
// Measured against the LDS tile on the batch shapes: +12 ... +20 % where N = 768 (out-proj 396 -> 467-480, fc2 496 -> 555-568 TFLOP/s
// algorithmic), +-3 % on the wide layers (4 MFMAs per loaded KB put ~48 B/clk on the 64 B/clk vector-memory path) -- so the launcher
// uses it for N = 768 only (launch_mode).
template <class CF, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__device__ __forceinline__ void gemm_tile_w2v(const GemmArgs& g, const int m0, const int n0, float* smem_f) {
    constexpr int TM = CF::TM, TN = CF::TN, BM = CF::BM;
    static_assert(TN % 2 == 0, "column tiles go in pairs");
    constexpr int A_T = BM * 64, BUF = 2 * A_T;                           // bytes: one plane, one buffer (two planes)
    constexpr int NA = BM / 32;
    char* smem = reinterpret_cast<char*>(smem_f);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int ar = m0 + r0 + 32 * i;
        ar = ar < g.M ? ar : g.M - 1;
        aoff[i] = (unsigned)(ar - m0) * (unsigned)g.lda * 4u + c4 * 16;
    }
    const int nsteps = g.K / 16, nk = g.K / BK;
    const char* wrow = static_cast<const char*>(g.Wq) + (size_t)((n0 >> 5) + wn * TN) * nsteps * 1024;
    const __amdgpu_buffer_rsrc_t ra_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)m0 * g.lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wrow), 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    const int rd0 = (wm * TM * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);
    const int rd1 = rd0 ^ 32;
    const int wr0 = r0 * 64 + (((c4 >> 1) ^ ((r0 >> 2) & 3)) << 4) + (c4 & 1) * 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NA];
    u32x4 w[2][TN];                                                       // w[set][j]: step parity -> set
    bf16x8 a0[2][TM], a1[TM];                                             // a0 = a_hi (double-buffered), a1 = a_lo
    unsigned sp[2][2];

    auto gload_a = [&](int kt) __attribute__((always_inline)) {
        kt = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)aoff[i], kt * (BK * 4), 0));
    };
    auto wload = [&](auto SET, auto J, int gs) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value, j = decltype(J)::value;
        gs = gs < nsteps ? gs : nsteps - 1;
        w[st][j] = __builtin_amdgcn_raw_buffer_load_b128(rw_rsrc, voff, (j * nsteps + gs) * 1024, 0);
    };
    auto rd_a = [&](const char* base, int off, int plane, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + off + plane * A_T + i * 2048);
    };
    // hi = bf16(x), lo = bf16(x - hi): the arithmetic of split4_bf16x2 on one pair
    auto split_pair = [&](float x0, float x1, unsigned (&o)[2]) __attribute__((always_inline)) {
        f32x2 x; x[0] = x0; x[1] = x1;
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        o[0] = hb;
        f32x2 r; r[0] = x0 - __uint_as_float(hb << 16); r[1] = x1 - __uint_as_float(hb & 0xFFFF0000u);
        o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    };
    auto stage_piece = [&](auto P, char* wbase) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value, i = p >> 1, hf = p & 1;
        split_pair(ra[i][2 * hf], ra[i][2 * hf + 1], sp[hf]);
        if constexpr (hf == 1) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                u32x2 v; v[0] = sp[0][k]; v[1] = sp[1][k];
                *reinterpret_cast<u32x2*>(wbase + wr0 + k * A_T + i * 2048) = v;
            }
        }
    };
    auto mf = [&](const bf16x8& a, const u32x4& b, f32x16& c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    auto step = [&](auto SS, const char* rbase, int roff, char* wbase, int gs, int ktload) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;                            // step inside the tile = weight slot set = a0 set
        constexpr int NPIECE = NA * 2, GP = TN / 2, NG = 2 * GP, PPG = (NPIECE + 1 + NG - 1) / NG;
        auto after_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int gi = decltype(G)::value;
            if constexpr (S == 0) {
                [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                    ([&] {
                        constexpr int p = gi * PPG + Q;
                        if constexpr (p < NPIECE) stage_piece(std::integral_constant<int, p>{}, wbase);
                        if constexpr (p == NPIECE) gload_a(ktload);
                    }(), ...);
                }(std::make_integer_sequence<int, PPG>{});
            }
            if constexpr (gi == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[S ^ 1][i] = rd_a(rbase, roff, 0, i);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], w[S][j0 + jj], acc[i][j0 + jj]);          // a_lo w first
                after_group(std::integral_constant<int, 2 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[S][j0 + jj], acc[i][j0 + jj]);       // then a_hi w
                wload(std::integral_constant<int, S>{}, std::integral_constant<int, j0>{}, gs + 2);
                wload(std::integral_constant<int, S>{}, std::integral_constant<int, j0 + 1>{}, gs + 2);
                if constexpr (JP == GP - 1) {                             // a_lo's last product of the step has issued
#pragma unroll
                    for (int i = 0; i < TM; ++i) a1[i] = rd_a(rbase, roff, 1, i);
                }
                after_group(std::integral_constant<int, 2 * JP + 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
    };

    gload_a(0);
    [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
        ((wload(std::integral_constant<int, 0>{}, std::integral_constant<int, J>{}, 0),
          wload(std::integral_constant<int, 1>{}, std::integral_constant<int, J>{}, 1)), ...);
    }(std::make_integer_sequence<int, TN>{});
    [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
        (stage_piece(std::integral_constant<int, P>{}, smem), ...);
    }(std::make_integer_sequence<int, NA * 2>{});
    gload_a(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { a0[0][i] = rd_a(smem, rd0, 0, i); a1[i] = rd_a(smem, rd0, 1, i); }
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * BUF;
        char* nxt = smem + ((kt + 1) & 1) * BUF;
        step(std::integral_constant<int, 0>{}, cur, rd1, nxt, 2 * kt, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        step(std::integral_constant<int, 1>{}, nxt, rd0, nxt, 2 * kt + 1, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    gemm_epilogue<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, acc, m0, n0, wm, wn, l31, h);
}

// Wq[n-tile][ks][lane][e] = bf16(W[32 n-tile + (lane & 31)][16 ks + 8 (lane >> 5) + e]) from the bfloat16 copy Wb [N, K]: one thread per
// (row, 8-k chunk), one 16-byte load, one 16-byte store
__global__ __launch_bounds__(256) void pack_w2_kernel(const __bf16* __restrict__ Wb, char* __restrict__ Wq, int N, int K) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int kc = K / 8;
    if (gid >= (size_t)N * kc) return;
    const int n = (int)(gid / kc), c = (int)(gid % kc);
    const u32x4 v = *reinterpret_cast<const u32x4*>(Wb + (size_t)n * K + c * 8);
    const int ks = c >> 1, hh = c & 1;
    *reinterpret_cast<u32x4*>(Wq + (((size_t)(n >> 5) * (K / 16) + ks) * 64 + hh * 32 + (n & 31)) * 16) = v;
}

int pack_weights_w2(const __bf16* Wb, void* Wq, int N, int K, hipStream_t s) {
    TSTAR_REQUIRE(N % 32 == 0 && K % 16 == 0, "pack_weights_w2: N must be a multiple of 32, K of 16");
    const size_t n = (size_t)N * (K / 8);
    hipLaunchKernelGGL(pack_w2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wb, static_cast<char*>(Wq), N, K);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// One output tile of shape CF at (m0, n0).  `smem` is the block's dynamic LDS.
template <class CF, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int m0, const int n0, float* smem) {
    constexpr int TM = CF::TM, TN = CF::TN, BM = CF::BM, BN = CF::BN;
    float* As = smem;                         // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // staging assignment: thread -> (row r0 + 32 i, float4 column c4); 32 rows per pass
    constexpr int NA = BM / 32, NB = BN / 32;
    const int c4 = t & 7, r0 = t >> 3;
    const float* ap[NA];
    const float* bp[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int ar = m0 + r0 + 32 * i;
        ar = ar < g.M ? ar : g.M - 1;           // clamp: never read past the last row
        ap[i] = g.A + (size_t)ar * g.lda + c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) bp[i] = g.W + (size_t)(n0 + r0 + 32 * i) * g.K + c4 * 4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i]);
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDS_LD + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + 32 * i) * LDS_LD + c4 * 4]) = rb[i];
    __syncthreads();

    const int nk = g.K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + (kt + 1) * BK);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + (kt + 1) * BK);
        }
        const float* Ac = As + cur * BM * LDS_LD + (wm * TM * 32 + l31) * LDS_LD + h * 4;
        const float* Bc = Bs + cur * BN * LDS_LD + (wn * TN * 32 + l31) * LDS_LD + h * 4;
        // raise this wave's issue priority over the co-resident block's staging traffic while it
        // feeds the matrix pipe (+4-7 % measured on gfx950)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) {
            float* Aw = As + (cur ^ 1) * BM * LDS_LD;
            float* Bw = Bs + (cur ^ 1) * BN * LDS_LD;
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&Aw[(r0 + 32 * i) * LDS_LD + c4 * 4]) = ra[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bw[(r0 + 32 * i) * LDS_LD + c4 * 4]) = rb[i];
        }
        __syncthreads();
        cur ^= 1;
    }

    gemm_epilogue<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, acc, m0, n0, wm, wn, l31, h);
}

template <class CF, int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__global__ __launch_bounds__(256, CF::MINW) void gemm_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nt = g.N / CF::BN;
    const int mt = (g.M + CF::BM - 1) / CF::BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    int mi, ni;
    tile_mn(tile, mt, nt, g.group_m, mi, ni);
    if constexpr (WMODE == 4) gemm_tile_x3<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * CF::BM, ni * CF::BN, smem);
    else if constexpr (WMODE != 0) gemm_tile_bf16w<CF, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * CF::BM, ni * CF::BN, smem);
    else gemm_tile<CF, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * CF::BM, ni * CF::BN, smem);
}

// Hybrid launch: rows [0, m_split) in 128x128 tiles (whole waves of the 512 resident slots), the
// remaining rows in 64x128 tiles.  Blocks are dispatched in index order, so the half-size tiles
// arrive last and fill the tail that a pure 128x128 grid leaves on most CUs.
template <int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
__global__ __launch_bounds__(256, 2) void gemm_f32_hybrid_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nt = g.N / 128;
    const int n_big = (g.m_split / 128) * nt;
    if ((int)blockIdx.x < n_big) {
        const int tile = xcd_remap(blockIdx.x, n_big);
        int mi, ni;
        tile_mn(tile, g.m_split / 128, nt, g.group_m, mi, ni);
        if constexpr (WMODE == 4) gemm_tile_x3<Cfg128, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * 128, ni * 128, smem);
        else if constexpr (WMODE != 0) gemm_tile_bf16w<Cfg128, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * 128, ni * 128, smem);
        else gemm_tile<Cfg128, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * 128, ni * 128, smem);
    } else {
        const int n_small = gridDim.x - n_big;
        const int tile = xcd_remap(blockIdx.x - n_big, n_small);
        int mi, ni;
        tile_mn(tile, n_small / nt, nt, g.group_m, mi, ni);
        if constexpr (WMODE == 4) gemm_tile_x3<Cfg64N, ACT, HAS_BIAS, HAS_RES, PATCH>(g, g.m_split + mi * 64, ni * 128, smem);
        else if constexpr (WMODE != 0) gemm_tile_bf16w<Cfg64N, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(g, g.m_split + mi * 64, ni * 128, smem);
        else gemm_tile<Cfg64N, ACT, HAS_BIAS, HAS_RES, PATCH>(g, g.m_split + mi * 64, ni * 128, smem);
    }
}

// Wide hybrid launch of the two-term bf16-weight mode (WMODE 3) and of the f32x3 mode (WMODE 4): rows [0, m_split) in 128x256 tiles (whole waves of the 512
// resident slots), the remaining rows in 64x128 tiles that arrive last and fill the tail.
// VW (two-term mode only): the wide tiles take their weight fragments global -> VGPR from the fragment-packed plane g.Wq (gemm_tile_w2v)
template <int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH, bool VW = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16w2_wide_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntw = g.N / 256;
    const int n_big = (g.m_split / 128) * ntw;
    if ((int)blockIdx.x < n_big) {
        const int tile = xcd_remap(blockIdx.x, n_big);
        int mi, ni;
        tile_mn(tile, g.m_split / 128, ntw, g.group_m, mi, ni);
        if constexpr (WMODE == 4) gemm_tile_x3<Cfg128W, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * 128, ni * 256, smem);
        else if constexpr (VW) gemm_tile_w2v<Cfg128W, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * 128, ni * 256, smem);
        else gemm_tile_bf16w<Cfg128W, 3, ACT, HAS_BIAS, HAS_RES, PATCH>(g, mi * 128, ni * 256, smem);
    } else {
        const int nt = g.N / 128;
        const int n_small = gridDim.x - n_big;
        const int tile = xcd_remap(blockIdx.x - n_big, n_small);
        int mi, ni;
        tile_mn(tile, n_small / nt, nt, g.group_m, mi, ni);
        if constexpr (WMODE == 4) gemm_tile_x3<Cfg64N, ACT, HAS_BIAS, HAS_RES, PATCH>(g, g.m_split + mi * 64, ni * 128, smem);
        else gemm_tile_bf16w<Cfg64N, 3, ACT, HAS_BIAS, HAS_RES, PATCH>(g, g.m_split + mi * 64, ni * 128, smem);
    }
}

// dynamic LDS of one block: double-buffered operand tiles
template <int WMODE>
constexpr int lds_bytes(int bm, int bn) {
    return WMODE == 4 ? 2 * 3 * bm * 64 : WMODE == 3 ? 2 * (2 * bm + bn) * 64 : WMODE == 1 ? 2 * (3 * bm + bn) * 64 : 2 * (bm + bn) * LDS_LD * 4;
}

// algorithmic HBM bytes of one launch: A and W read once, C written once, residual read once, bias / position rows
static double gemm_algorithmic_bytes(const GemmArgs& g) {
    const double wbytes = g.Wp ? 6.0 : g.Wb ? 2.0 : 4.0;                // f32x3: three bf16 planes; bf16 weights: one term
    return 4.0 * g.M * g.K + wbytes * g.N * g.K + 4.0 * g.M * g.N * (g.res ? 2.0 : 1.0) + (g.bias ? 4.0 * g.N : 0.0) +
           (g.pos ? 4.0 * (g.patch_np + 1) * g.N : 0.0);
}

template <class CF, int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    constexpr int lds = lds_bytes<WMODE>(CF::BM, CF::BN);
    auto kern = gemm_f32_kernel<CF, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    const int nwg = cdiv(g.M, CF::BM) * (g.N / CF::BN);
    const bool prof = prof_enabled();
    if (prof) prof_start(PROF_GEMM, stream, 2.0 * g.M * g.N * g.K, gemm_algorithmic_bytes(g));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, stream, g);
    if (prof) prof_stop(PROF_GEMM, stream);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

template <int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
static int launch_hybrid(const GemmArgs& g, hipStream_t stream) {
    constexpr int lds = lds_bytes<WMODE>(128, 128);
    auto kern = gemm_f32_hybrid_kernel<WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    const int nt = g.N / 128;
    const int nwg = (g.m_split / 128) * nt + cdiv(g.M - g.m_split, 64) * nt;
    const bool prof = prof_enabled();
    if (prof) prof_start(PROF_GEMM, stream, 2.0 * g.M * g.N * g.K, gemm_algorithmic_bytes(g));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, stream, g);
    if (prof) prof_stop(PROF_GEMM, stream);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

template <int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH, bool VW = false>
static int launch_wide(const GemmArgs& g, hipStream_t stream) {
    constexpr int lds = lds_bytes<WMODE>(128, 256);       // (VW: the wide tile needs 32 KB, the 64x128 tail tile of the same kernel this much)
    auto kern = gemm_bf16w2_wide_kernel<WMODE, ACT, HAS_BIAS, HAS_RES, PATCH, VW>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    const int nwg = (g.m_split / 128) * (g.N / 256) + cdiv(g.M - g.m_split, 64) * (g.N / 128);
    const bool prof = prof_enabled();
    if (prof) prof_start(PROF_GEMM, stream, 2.0 * g.M * g.N * g.K, gemm_algorithmic_bytes(g));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, stream, g);
    if (prof) prof_stop(PROF_GEMM, stream);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// Tile choice, calibrated on MI355X (tools/sweep_small_m.py for launches under two waves of 128x128 tiles -- the
// B = 1..16 grid forwards of the reference-default 4x4 grid; tools/sweep_hybrid_split.py, tools/bench_gemm_cfg.py for
// the batch shapes).  Blocks are dispatched in index order onto 512 resident slots (256 CUs x 2 blocks; 1024
// quarter-size slots for 64x64).  b128 = number of 128x128 tiles of the problem:
//  * b128 <= 200: 64x64 tiles (up to 800 quarter-size blocks: every CU gets work; measured best up to there);
//  * b128 <= 256: 64x128 tiles;
//  * 256 < b128 < 410: ONE wave of mixed sizes -- the first n row tiles 128 rows high, the rest 64 x 128, with n chosen
//    so that the launch has about 512 blocks (and at least half of the row tiles big): the small tiles finish first and
//    leave their CUs to the big ones instead of idling a third of the chip (qkv at B = 4: 76 -> 103 TFLOP/s);
//  * 410 <= b128 < 512: a pure 128x128 grid (measured best: nearly a full wave);
//  * 512 <= b128 < 1024: hybrid, half of the row tiles big -- a whole wave of big tiles plus a thin second wave leaves
//    the chip idle for up to half a tile time (fc1 at B = 5: 91 -> 109 TFLOP/s);
//  * from two waves on, as many whole 512-block waves of 128x128 tiles as fit, the remaining rows as 64x128 tiles, which
//    arrive last and fill the tail at half the granularity (never worse than a pure grid, up to +15 %).
// Returns 0/1/2 for a pure grid, or 3 for the hybrid launch with *m_split set.
static int pick_cfg(int M, int N, int* m_split) {
    *m_split = 0;
    const int nt = N / 128;
    const int mt = cdiv(M, 128);
    const int b128 = mt * nt;
    if (b128 <= 200) return 2;                               // 64x64
    if (b128 <= 256) return 1;                               // 64x128
    if (b128 < 410 || (b128 >= 512 && b128 < 1024)) {
        int n_big = mt / 2;
        if (b128 < 410) {
            const int fill = 2 * mt - 512 / nt;              // n with nt * (n + 2 (mt - n)) <= 512
            if (fill > n_big) n_big = fill;
        }
        if (n_big > mt - 1) n_big = mt - 1;
        if (n_big < 1) return 1;
        *m_split = n_big * 128;
        return 3;
    }
    if (b128 < 512) return 0;                                // 128x128
    const int big_rows = ((b128 / 512) * 512 / nt) * 128;    // rows covered by whole waves of 128x128 tiles
    if (big_rows >= M) return 0;
    *m_split = big_rows;
    return 3;
}

template <int WMODE, int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
static int launch_mode(const GemmArgs& g, hipStream_t stream) {
    const int forced = g.tile_cfg;                           // -1 = auto
    if constexpr (WMODE == 3 || WMODE == 4) {
        // two-term mode (and the f32x3 mode, whose 128x256 tile measures 6-7 % above its 128x128 one: 197-201 vs 184-188 TFLOP/s): from one wave (512) of 128x256 tiles on, whole waves of them; the remaining full panels are wide
        // too when they make more than half a wave (a 64x128 tile of this mode is LDS-bound and runs at ~0.7 of the wide
        // tile's rate: three rounds of them cost more than one round of wide tiles -- tools/bench_gemm_bf16.py: out-proj
        // at B = 64, 864 wide tiles: 410 all wide vs 384 with a narrow tail), else they and the ragged rows go out as
        // 64x128 tiles that fill the tail (tile_cfg 4 forces every full 128-row panel wide; 5 forces the wide tile OFF)
        if (g.N % 256 == 0 && forced != 5 && (forced == -1 || forced == 4 || forced == 6)) {
            const int ntw = g.N / 256, mt = g.M / 128;       // full 128-row panels only
            const long long bw = (long long)mt * ntw;
            int big = 0;
            if (forced == 4 || forced == 6) big = mt;
            else if (bw >= 512) big = (bw % 512) > 256 ? mt : (int)(((bw / 512) * 512) / ntw);
            // f32x3, under one wave of wide tiles (tools/sweep_x3_cfg.py, profiles/r05_x3_cfg_sweep.log): the wide tile still wins when its
            // blocks nearly fill the 512 slots (>= 400: fc1 at B = 8 195 -> 216, qkv at B = 10 190 -> 212 TFLOP/s) or when every block gets
            // a CU to itself (192..256: out-proj / fc2 at B = 16 +5 / +7 %); in between (a full CU pair next to single ones) it loses
            else if (WMODE == 4 && (bw >= 400 || (bw >= 192 && bw <= 256))) big = mt;
            const bool fits32 = ((long long)(g.M + g.M / (g.patch_np > 0 ? g.patch_np : g.M) + 1) * g.ldc) < (1ll << 32);   // the wide epilogue's offsets
            if (big > 0 && fits32) {
                GemmArgs h = g;
                h.m_split = big * 128;
                if constexpr (WMODE == 3) {
                    // round 6, per-shape dispatch: with a fragment-packed plane at hand the wide tiles of the N = 768 layers (out-proj, fc2,
                    // patch embedding, box head) stream their weights global -> VGPR (+12 ... +20 % there; +-3 % on the wide layers, which
                    // keep the LDS tile).  Same bits either way.  tile_cfg 6 forces this tile on every full panel of any N (tests).
                    if (g.Wq && (forced == 6 || (forced == -1 && g.N == 768))) return launch_wide<3, ACT, HAS_BIAS, HAS_RES, PATCH, true>(h, stream);
                }
                TSTAR_REQUIRE(forced != 6, "gemm_f32: tile_cfg 6 needs the two-term mode with a fragment-packed weight plane (Wq)");
                return launch_wide<WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(h, stream);
            }
        }
    }
    int m_split = 0;
    TSTAR_REQUIRE(forced != 6, "gemm_f32: tile_cfg 6 (two-term wide tile, weights global -> VGPR) needs N % 256 == 0, at least one full 128-row panel and a packed plane");
    int cfg = (forced >= 0 && forced <= 3) || forced >= 16 ? forced : pick_cfg(g.M, g.N, &m_split);
    if (forced == 3) {                                       // forced hybrid: half of the row tiles big
        m_split = (cdiv(g.M, 128) / 2) * 128;
        if (m_split == 0) cfg = 1;
    } else if (forced >= 16) {                               // diagnostic: hybrid with (forced - 16) big row tiles
        m_split = (forced - 16) * 128;
        if (m_split > (g.M / 128) * 128) m_split = (g.M / 128) * 128;
        cfg = m_split == 0 ? 1 : 3;
    }
    if (cfg == 3) {
        GemmArgs h = g;
        h.m_split = m_split;
        return launch_hybrid<WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(h, stream);
    }
    if (cfg == 0) return launch_cfg<Cfg128, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
    if (cfg == 1) return launch_cfg<Cfg64N, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
    return launch_cfg<Cfg64, WMODE, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
}

template <int ACT, bool HAS_BIAS, bool HAS_RES, bool PATCH>
static int launch_one(const GemmArgs& g, hipStream_t stream) {
    if (g.Wp) return launch_mode<4, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
    if (g.Wb && g.a_terms == 2) return launch_mode<3, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
    if (g.Wb) return launch_mode<1, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
    return launch_mode<0, ACT, HAS_BIAS, HAS_RES, PATCH>(g, stream);
}

// rows of a super-panel (tile_mn): TSTAR_GEMM_GM overrides the default for A/B runs (1 = the panel-major order of rounds 1-3)
static int default_group_m() {
    static const int gm = [] { const char* e = getenv("TSTAR_GEMM_GM"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    return gm;
}

int gemm_f32(const GemmArgs& g0, hipStream_t stream) {
    GemmArgs g = g0;
    if (g.group_m <= 0) g.group_m = default_group_m();
    TSTAR_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm_f32: empty problem");
    TSTAR_REQUIRE(g.N % 128 == 0, "gemm_f32: N must be a multiple of 128");
    TSTAR_REQUIRE(g.K % BK == 0, "gemm_f32: K must be a multiple of 32");
    TSTAR_REQUIRE(g.lda % 4 == 0 && g.K % 4 == 0, "gemm_f32: rows must be 16-byte aligned");
    TSTAR_REQUIRE((g.tile_cfg >= -1 && g.tile_cfg <= 6) || g.tile_cfg >= 16, "gemm_f32: tile_cfg must be -1..6 (or 16 + big row tiles)");
    const bool bias = g.bias != nullptr, res = g.res != nullptr, patch = g.pos != nullptr;
    if (patch) {
        TSTAR_REQUIRE(!bias && !res && g.act == ACT_NONE && g.patch_np > 0, "gemm_f32: patch epilogue takes no bias/res/act");
        return launch_one<ACT_NONE, false, false, true>(g, stream);
    }
    if (g.act == ACT_QGELU) {
        TSTAR_REQUIRE(bias && !res, "gemm_f32: quick-gelu epilogue needs bias, no residual");
        return launch_one<ACT_QGELU, true, false, false>(g, stream);
    }
    if (g.act == ACT_GELU) {
        TSTAR_REQUIRE(bias && !res, "gemm_f32: gelu epilogue needs bias, no residual");
        return launch_one<ACT_GELU, true, false, false>(g, stream);
    }
    if (bias && res) return launch_one<ACT_NONE, true, true, false>(g, stream);
    if (bias) return launch_one<ACT_NONE, true, false, false>(g, stream);
    TSTAR_REQUIRE(!res, "gemm_f32: residual without bias is not instantiated");
    return launch_one<ACT_NONE, false, false, false>(g, stream);
}

}  // namespace tstar
