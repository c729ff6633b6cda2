// Lab (round 5): the f32x3 (six-product) GEMM main loop, software-pipelined per WAVE.
//
// Baseline = the library's round-4 loop (gemm_tile_x3): per K tile of 32 every wave runs
//   [ds_read A fragments -> wait] 48 MFMAs [ds_read -> wait] 48 MFMAs [split the next tile: ~100 VALU] [ds_write] [barrier]
// and the compiler folds the 3-set weight-fragment ring so that most fragment loads are waited for 8-16 MFMAs after
// they were issued (ISA of round 4).  Every such section is a hole in that wave's MFMA stream which only the other
// block's wave on the same SIMD can fill: matrix pipe busy 0.60-0.62.
//
// v2 (this file): one wave's stream never waits for something it has just asked for.
//   * weight fragments: 12 per K = 16 step, each in its OWN 4-register slot, re-loaded for the next step right after
//     its last MFMA of this step -> 42-46 MFMAs (1300+ cycles) between the load and its first use;
//   * MFMA order of a step: per fragment, all its products back to back (w0: a2, a1, a0; w1: a1, a0; w2: a0), two column
//     tiles interleaved so that MFMAs on one accumulator are 4 apart;
//   * activation fragments: a2 / a1 single-buffered and re-read from LDS as soon as their last product of the step has
//     issued (24 / 8 MFMAs before their next use), a0 double-buffered;
//   * the split of the NEXT K tile (f32 -> three bf16 planes) and its LDS stores are cut into 8 pieces that ride between
//     the MFMA groups of the tile's FIRST step; the ONE barrier per K tile sits between the two steps, where every operand of
//     the second step is already in registers;
//   * loop body without branches (loads past the end are clamped, the stores of a tile past the end go to the idle buffer).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <utility>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// shader-clock estimate: every block adds its lifetime in core cycles (s_memtime) and in 100 MHz ticks (s_memrealtime)
__device__ unsigned long long g_clk[2];
struct ClockProbe {
    unsigned long long c0, w0;
    __device__ __forceinline__ ClockProbe() { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    __device__ __forceinline__ void done() {
        if (threadIdx.x == 0) { atomicAdd(&g_clk[0], __builtin_readcyclecounter() - c0); atomicAdd(&g_clk[1], wall_clock64() - w0); }
    }
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8; int xcd = bid % nx, idx = bid / nx; int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; return base + idx;
}

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
                x[0] = x[0] - __uint_as_float(hb << 16);
                x[1] = x[1] - __uint_as_float(hb & 0xFFFF0000u);
            }
        }
    }
}
// one pair of floats -> three packed bf16 pairs (round-to-nearest terms of the running remainder); scalar subtractions:
// packed f32 VALU beside MFMAs costs more than its issue slot (MI355X guide)
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

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ------------------------------------------------------------------------------------------------------------------
// baseline: the library's loop (frag_lab's gemm_frag<2, 4, 3, 3, 2, 3, 2>)
template <int TM, int TN, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_x3_base(const float* __restrict__ A, const char* __restrict__ Wp,
                                                          float* __restrict__ C, int M, int N, int K) {
    ClockProbe probe;
    constexpr int NAT = 3, NWT = 3, MAXSUM = 2, D = 3;
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32, NS = BK / 16, PPT = NS * NWT;
    constexpr int A_T = BM * 64, BUF = NAT * A_T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    constexpr int NA = BM / 32;
    const int c4 = t & 7, r0 = t >> 3;
    const float* ap[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; ap[i] = A + (size_t)ar * K + c4 * 4; }
    const int nph = (K / 16) * NWT;
    const char* wb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wb[j] = Wp + (size_t)((n0 >> 5) + wn * TN + j) * nph * 1024 + lane * 16;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NA];
    u32x4 ring[D][TN];
    auto gloadA = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 sp[NAT];
            split4_rn<NAT>(ra[i], sp);
            const int off = lds_off(r0 + 32 * i, c4 >> 1) + (c4 & 1) * 8;
#pragma unroll
            for (int k = 0; k < NAT; ++k) *reinterpret_cast<u32x2*>(base + k * A_T + off) = sp[k];
        }
    };
    const int nk = K / BK;
    gloadA(0);
#pragma unroll
    for (int q = 0; q < D - 1; ++q)
#pragma unroll
        for (int j = 0; j < TN; ++j) ring[q][j] = *reinterpret_cast<const u32x4*>(wb[j] + (size_t)(q < nph ? q : nph - 1) * 1024);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gloadA(kt + 1);
        const char* base = smem + (kt & 1) * BUF;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 fa[NAT][TM];
#pragma unroll
        for (int P = 0; P < PPT; ++P) {
            const int s = P / NWT, ph = P % NWT, kw = NWT - 1 - ph;
            {
                int q = kt * PPT + P + D - 1;
                q = q < nph ? q : nph - 1;
#pragma unroll
                for (int j = 0; j < TN; ++j) ring[(P + D - 1) % D][j] = *reinterpret_cast<const u32x4*>(wb[j] + (size_t)q * 1024);
            }
            if (ph == 0) {
#pragma unroll
                for (int k = 0; k < NAT; ++k)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[k][i] = *reinterpret_cast<const bf16x8*>(base + k * A_T + lds_off(wm * TM * 32 + i * 32 + l31, 2 * s + h));
            }
#pragma unroll
            for (int ka = NAT - 1; ka >= 0; --ka) {
                if (ka + kw > MAXSUM) continue;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ka][i], __builtin_bit_cast(bf16x8, ring[P % D][j]), acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    }
    probe.done();
}

// ------------------------------------------------------------------------------------------------------------------
// v2: software-pipelined per wave (header).  STAGE: 0 = staging pieces interleaved, 1 = staging as one lump after the first
// step's MFMAs (isolates the effect of the interleave).  PRIO: raise the wave's priority for the whole loop (both waves of a
// SIMD equal: no effect expected; kept as a switch).
template <int TM, int TN, int MINW, int STAGE, int PRIO>
__global__ __launch_bounds__(256, MINW) void gemm_x3_v2(const float* __restrict__ A, const char* __restrict__ Wp,
                                                        float* __restrict__ C, int M, int N, int K) {
    static_assert(TN % 2 == 0, "column tiles go in pairs");
    ClockProbe probe;
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32;
    constexpr int A_T = BM * 64, BUF = 3 * A_T;
    constexpr int NA = BM / 32;                 // float4 loads per thread per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    // activation rows: ONE uniform base per K tile (SGPRs) + a 32-bit byte offset per staged row (clamped to the last row)
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; aoff[i] = (unsigned)(ar - m0) * (unsigned)K * 4u + c4 * 16; }
    // STAGE == 3: A is given as three bf16 planes per row, [M][3][K] (what a producer -- LayerNorm, the attention / GELU epilogues --
    // would write): staging is 16-byte loads + ds_write_b128, no split.  Thread -> 16-B chunk t & 3 of (plane, row) = idx / BM, idx % BM,
    // idx = (t >> 2) + 64 i
    constexpr int NPL = 3 * BM / 64, RPP = BM / 64;                 // pieces; row groups of 64 per plane
    // piece i: plane i / RPP, row (t >> 2) + 64 (i % RPP): the swizzle term is the same for rows 64 apart, the plane term is uniform
    unsigned prow[RPP];
#pragma unroll
    for (int i = 0; i < RPP; ++i) { int ar = m0 + (t >> 2) + 64 * i; ar = ar < M ? ar : M - 1; prow[i] = (unsigned)(ar - m0) * (unsigned)K * 6u + (t & 3) * 16; }
    const int pwr0 = (t >> 2) * 64 + (((t & 3) ^ ((t >> 4) & 3)) << 4);
    const char* abase = STAGE == 3 ? reinterpret_cast<const char*>(A) + (size_t)m0 * K * 6 : reinterpret_cast<const char*>(A + (size_t)m0 * K);
    const int nsteps = K / 16, nk = K / BK;
    // fragment stream of this wave's column tile j: (uniform base) + step * 3072 + ph * 1024 + lane * 16, ph = 2 - kw
    const char* wrow = Wp + (size_t)((n0 >> 5) + wn * TN) * nsteps * 3072;
    const unsigned voff = lane * 16;
    // LDS: fragment read offsets of step 0 / 1 (row swizzle term (l31 >> 2) & 3), staging write offset
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
    u32x4 rpl[NPL];
    u32x4 w[3][TN];                 // w[kw][j]: plane kw of column tile j, current step
    bf16x8 a0[2][TM], a1[TM], a2[TM];

    // buffer loads: a resource descriptor per operand (SGPRs), a 32-bit lane offset (VGPR), a uniform 32-bit offset (SGPR, SALU
    // arithmetic) and an immediate -- no 64-bit VALU address arithmetic, no 64-bit pointer registers
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wrow), 0, 0x7fffffff, 0x00020000);
    auto gloadA = [&](int kt) __attribute__((always_inline)) {
        kt = kt < nk ? kt : nk - 1;
        if constexpr (STAGE == 3) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) rpl[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)prow[i % RPP], kt * (BK * 2) + (i / RPP) * K * 2, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)aoff[i], kt * (BK * 4), 0));
    };
    auto wload = [&](auto KW, auto J, int g) __attribute__((always_inline)) {
        constexpr int kw = decltype(KW)::value, j = decltype(J)::value;
        g = g < nsteps ? g : nsteps - 1;
        w[kw][j] = __builtin_amdgcn_raw_buffer_load_b128(rw_rsrc, (int)voff + (2 - kw) * 1024, (j * nsteps + g) * 3072, 0);
    };
    auto rdA = [&](const char* base, int off, int plane, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + off + plane * A_T + i * 2048);
    };
    // staging piece p of NA * 2: half p & 1 of float4 p >> 1; the three 8-byte stores follow the second half
    unsigned sp[2][3];
    auto stage_piece = [&](auto P, char* wbase) __attribute__((always_inline)) {
        if constexpr (STAGE == 3) {
            constexpr int p3 = decltype(P)::value;
            if constexpr (p3 < NPL) *reinterpret_cast<u32x4*>(wbase + pwr0 + (p3 / RPP) * A_T + (p3 % RPP) * 4096) = rpl[p3];
            return;
        }
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
    // wbase: where the next tile's planes are written (only S == 0 stages); gnext: fragment-stream step of the reloads
    auto step = [&](auto SS, const char* rbase, int roff, char* wbase, int gnext, int ktload) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;
        constexpr int NPIECE = STAGE == 3 ? NPL : NA * 2;
        int grp = 0;                        // MFMA groups issued so far in this step (compile-time after unrolling)
        auto after_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            if constexpr (S == 0 && STAGE != 1) {   // 0: split pieces interleaved, 3: plane copies interleaved
                // NPIECE split pieces + the load of the tile after next, PPG of them per MFMA group
                constexpr int NG = 6 * (TN / 2), PPG = (NPIECE + 1 + NG - 1) / NG;
                [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                    ([&] {
                        constexpr int p = g * PPG + Q;
                        if constexpr (p < NPIECE) stage_piece(std::integral_constant<int, p>{}, wbase);
                        if constexpr (p == NPIECE) gloadA(ktload);
                    }(), ...);
                }(std::make_integer_sequence<int, PPG>{});
            }
            if constexpr (g == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[S ^ 1][i] = rdA(rbase, roff, 0, i);
            }
            FENCE();
        };
        (void)grp;
        constexpr int GP = TN / 2;          // column-tile pairs
        // w0 fragments: a2, a1, a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a2[i], w[0][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], w[0][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * JP + 1>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[0][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, 0>{}, std::integral_constant<int, j0>{}, gnext);
                wload(std::integral_constant<int, 0>{}, std::integral_constant<int, j0 + 1>{}, gnext);
                after_group(std::integral_constant<int, 3 * JP + 2>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
        // a2's last product has issued: next step's a2
#pragma unroll
        for (int i = 0; i < TM; ++i) a2[i] = rdA(rbase, roff, 2, i);
        FENCE();
        // w1 fragments: a1, a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], w[1][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * GP + 2 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[1][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, 1>{}, std::integral_constant<int, j0>{}, gnext);
                wload(std::integral_constant<int, 1>{}, std::integral_constant<int, j0 + 1>{}, gnext);
                after_group(std::integral_constant<int, 3 * GP + 2 * JP + 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = rdA(rbase, roff, 1, i);
        FENCE();
        // w2 fragments: a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[2][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, 2>{}, std::integral_constant<int, j0>{}, gnext);
                wload(std::integral_constant<int, 2>{}, std::integral_constant<int, j0 + 1>{}, gnext);
                after_group(std::integral_constant<int, 5 * GP + JP>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
        if constexpr (S == 0 && STAGE == 1) {
            [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
                (stage_piece(std::integral_constant<int, P>{}, wbase), ...);
            }(std::make_integer_sequence<int, NPIECE>{});
            gloadA(ktload);
            FENCE();
        }
    };

    // ---- prologue: tile 0 into LDS buffer 0, step 0's weight fragments, tile 1 into the staging registers
    gloadA(0);
    [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
        ((wload(std::integral_constant<int, 0>{}, std::integral_constant<int, J>{}, 0),
          wload(std::integral_constant<int, 1>{}, std::integral_constant<int, J>{}, 0),
          wload(std::integral_constant<int, 2>{}, std::integral_constant<int, J>{}, 0)), ...);
    }(std::make_integer_sequence<int, TN>{});
    [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
        (stage_piece(std::integral_constant<int, P>{}, smem), ...);
    }(std::make_integer_sequence<int, (STAGE == 3 ? NPL : NA * 2)>{});
    gloadA(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { a0[0][i] = rdA(smem, rd0, 0, i); a1[i] = rdA(smem, rd0, 1, i); a2[i] = rdA(smem, rd0, 2, i); }
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    FENCE();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * BUF;
        char* nxt = smem + ((kt + 1) & 1) * BUF;
        // step 0: next fragments = this tile's step 1; stages tile kt + 1 into nxt; after the staging registers are free, loads tile kt + 2
        step(std::integral_constant<int, 0>{}, cur, rd1, nxt, 2 * kt + 1, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
        // step 1: next fragments = tile kt + 1, step 0
        step(std::integral_constant<int, 1>{}, nxt, rd0, nxt, 2 * kt + 2, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    }
    probe.done();
}

// the same loop with WM waves along M (WM = 4: a 256 x 256 block tile of 8 waves, one block per CU: the four waves of a column pair load the SAME weight
// fragments -- L1 hits instead of L2 requests -- and the activation tile is staged once for twice the MFMAs)
template <int TM, int TN, int WM, int MINW, int STAGE, int PRIO>
__global__ __launch_bounds__(128 * WM, MINW) void gemm_x3_v2w(const float* __restrict__ A, const char* __restrict__ Wp,
                                                        float* __restrict__ C, int M, int N, int K) {
    static_assert(TN % 2 == 0, "column tiles go in pairs");
    ClockProbe probe;
    constexpr int NTHR = 128 * WM, RPP = NTHR / 8, BM = WM * TM * 32, BN = 2 * TN * 32, BK = 32;
    constexpr int A_T = BM * 64, BUF = 3 * A_T;
    constexpr int NA = BM / RPP;                // float4 loads per thread per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    // activation rows: ONE uniform base per K tile (SGPRs) + a 32-bit byte offset per staged row (clamped to the last row)
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + RPP * i; ar = ar < M ? ar : M - 1; aoff[i] = (unsigned)(ar - m0) * (unsigned)K * 4u + c4 * 16; }
    const char* abase = reinterpret_cast<const char*>(A + (size_t)m0 * K);
    const int nsteps = K / 16, nk = K / BK;
    // fragment stream of this wave's column tile j: (uniform base) + step * 3072 + ph * 1024 + lane * 16, ph = 2 - kw
    const char* wrow = Wp + (size_t)((n0 >> 5) + wn * TN) * nsteps * 3072;
    const unsigned voff = lane * 16;
    // LDS: fragment read offsets of step 0 / 1 (row swizzle term (l31 >> 2) & 3), staging write offset
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
    u32x4 w[3][TN];                 // w[kw][j]: plane kw of column tile j, current step
    bf16x8 a0[2][TM], a1[TM], a2[TM];

    // buffer loads: a resource descriptor per operand (SGPRs), a 32-bit lane offset (VGPR), a uniform 32-bit offset (SGPR, SALU
    // arithmetic) and an immediate -- no 64-bit VALU address arithmetic, no 64-bit pointer registers
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wrow), 0, 0x7fffffff, 0x00020000);
    auto gloadA = [&](int kt) __attribute__((always_inline)) {
        kt = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)aoff[i], kt * (BK * 4), 0));
    };
    auto wload = [&](auto KW, auto J, int g) __attribute__((always_inline)) {
        constexpr int kw = decltype(KW)::value, j = decltype(J)::value;
        g = g < nsteps ? g : nsteps - 1;
        w[kw][j] = __builtin_amdgcn_raw_buffer_load_b128(rw_rsrc, (int)voff + (2 - kw) * 1024, (j * nsteps + g) * 3072, 0);
    };
    auto rdA = [&](const char* base, int off, int plane, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + off + plane * A_T + i * 2048);
    };
    // staging piece p of NA * 2: half p & 1 of float4 p >> 1; the three 8-byte stores follow the second half
    unsigned sp[2][3];
    auto stage_piece = [&](auto P, char* wbase) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value, i = p >> 1, hf = p & 1;
        split2_rn3(ra[i][2 * hf], ra[i][2 * hf + 1], sp[hf]);
        if constexpr (hf == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                u32x2 v; v[0] = sp[0][k]; v[1] = sp[1][k];
                *reinterpret_cast<u32x2*>(wbase + wr0 + k * A_T + i * (RPP * 64)) = v;
            }
        }
    };
    auto mf = [&](const bf16x8& a, const u32x4& b, f32x16& c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };

    // one K = 16 step.  S = step inside the tile (selects the a0 set); rbase / roff: where the NEXT step's fragments are read;
    // wbase: where the next tile's planes are written (only S == 0 stages); gnext: fragment-stream step of the reloads
    auto step = [&](auto SS, const char* rbase, int roff, char* wbase, int gnext, int ktload) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;
        constexpr int NPIECE = NA * 2;
        int grp = 0;                        // MFMA groups issued so far in this step (compile-time after unrolling)
        auto after_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            if constexpr (S == 0 && STAGE == 0) {
                // NPIECE split pieces + the load of the tile after next, PPG of them per MFMA group
                constexpr int NG = 6 * (TN / 2), PPG = (NPIECE + 1 + NG - 1) / NG;
                [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                    ([&] {
                        constexpr int p = g * PPG + Q;
                        if constexpr (p < NPIECE) stage_piece(std::integral_constant<int, p>{}, wbase);
                        if constexpr (p == NPIECE) gloadA(ktload);
                    }(), ...);
                }(std::make_integer_sequence<int, PPG>{});
            }
            if constexpr (g == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[S ^ 1][i] = rdA(rbase, roff, 0, i);
            }
            FENCE();
        };
        (void)grp;
        constexpr int GP = TN / 2;          // column-tile pairs
        // w0 fragments: a2, a1, a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a2[i], w[0][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], w[0][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * JP + 1>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[0][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, 0>{}, std::integral_constant<int, j0>{}, gnext);
                wload(std::integral_constant<int, 0>{}, std::integral_constant<int, j0 + 1>{}, gnext);
                after_group(std::integral_constant<int, 3 * JP + 2>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
        // a2's last product has issued: next step's a2
#pragma unroll
        for (int i = 0; i < TM; ++i) a2[i] = rdA(rbase, roff, 2, i);
        FENCE();
        // w1 fragments: a1, a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], w[1][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 3 * GP + 2 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[1][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, 1>{}, std::integral_constant<int, j0>{}, gnext);
                wload(std::integral_constant<int, 1>{}, std::integral_constant<int, j0 + 1>{}, gnext);
                after_group(std::integral_constant<int, 3 * GP + 2 * JP + 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = rdA(rbase, roff, 1, i);
        FENCE();
        // w2 fragments: a0
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[2][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, 2>{}, std::integral_constant<int, j0>{}, gnext);
                wload(std::integral_constant<int, 2>{}, std::integral_constant<int, j0 + 1>{}, gnext);
                after_group(std::integral_constant<int, 5 * GP + JP>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
        if constexpr (S == 0 && STAGE == 1) {
            [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
                (stage_piece(std::integral_constant<int, P>{}, wbase), ...);
            }(std::make_integer_sequence<int, NPIECE>{});
            gloadA(ktload);
            FENCE();
        }
    };

    // ---- prologue: tile 0 into LDS buffer 0, step 0's weight fragments, tile 1 into the staging registers
    gloadA(0);
    [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
        ((wload(std::integral_constant<int, 0>{}, std::integral_constant<int, J>{}, 0),
          wload(std::integral_constant<int, 1>{}, std::integral_constant<int, J>{}, 0),
          wload(std::integral_constant<int, 2>{}, std::integral_constant<int, J>{}, 0)), ...);
    }(std::make_integer_sequence<int, TN>{});
    [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
        (stage_piece(std::integral_constant<int, P>{}, smem), ...);
    }(std::make_integer_sequence<int, NA * 2>{});
    gloadA(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { a0[0][i] = rdA(smem, rd0, 0, i); a1[i] = rdA(smem, rd0, 1, i); a2[i] = rdA(smem, rd0, 2, i); }
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    FENCE();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * BUF;
        char* nxt = smem + ((kt + 1) & 1) * BUF;
        // step 0: next fragments = this tile's step 1; stages tile kt + 1 into nxt; after the staging registers are free, loads tile kt + 2
        step(std::integral_constant<int, 0>{}, cur, rd1, nxt, 2 * kt + 1, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
        // step 1: next fragments = tile kt + 1, step 0
        step(std::integral_constant<int, 1>{}, nxt, rd0, nxt, 2 * kt + 2, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    }
    probe.done();
}

// ------------------------------------------------------------------------------------------------------------------
// two-term mode (bf16 weights = ONE exact plane, activations as two round-to-nearest bf16 terms: BASELINE configs[4]) on the same
// per-wave pipeline: weight fragments global -> VGPR in TWO slot sets (a fragment is re-loaded two steps ahead right after its last
// product: 24+ MFMAs of lead at 16 MFMAs per step), a1 single- / a0 double-buffered, staging pieces in the first step of a tile.
// Weight stream layout: [n-tile][step][lane][8 bf16] (1 KB per fragment).
__device__ __forceinline__ void split2_rn2(float x0, float x1, unsigned (&o)[2]) {
    f32x2 x; x[0] = x0; x[1] = x1;
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    o[0] = hb;
    f32x2 r; r[0] = x0 - __uint_as_float(hb << 16); r[1] = x1 - __uint_as_float(hb & 0xFFFF0000u);
    o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int TM, int TN, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_w2_v2(const float* __restrict__ A, const char* __restrict__ Wp,
                                                        float* __restrict__ C, int M, int N, int K) {
    static_assert(TN % 2 == 0, "column tiles go in pairs");
    ClockProbe probe;
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32;
    constexpr int A_T = BM * 64, BUF = 2 * A_T;
    constexpr int NA = BM / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; aoff[i] = (unsigned)(ar - m0) * (unsigned)K * 4u + c4 * 16; }
    const char* abase = reinterpret_cast<const char*>(A + (size_t)m0 * K);
    const int nsteps = K / 16, nk = K / BK;
    const char* wrow = Wp + (size_t)((n0 >> 5) + wn * TN) * nsteps * 1024;
    const int voff = lane * 16;
    const int rd0 = (wm * TM * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);
    const int rd1 = rd0 ^ 32;
    const int wr0 = r0 * 64 + (((c4 >> 1) ^ ((r0 >> 2) & 3)) << 4) + (c4 & 1) * 8;
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wrow), 0, 0x7fffffff, 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NA];
    u32x4 w[2][TN];                 // w[set][j]: step parity -> set
    bf16x8 a0[2][TM], a1[TM];
    unsigned sp[2][2];

    auto gloadA = [&](int kt) __attribute__((always_inline)) {
        kt = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)aoff[i], kt * (BK * 4), 0));
    };
    auto wload = [&](auto SET, auto J, int g) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value, j = decltype(J)::value;
        g = g < nsteps ? g : nsteps - 1;
        w[st][j] = __builtin_amdgcn_raw_buffer_load_b128(rw_rsrc, voff, (j * nsteps + g) * 1024, 0);
    };
    auto rdA = [&](const char* base, int off, int plane, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + off + plane * A_T + i * 2048);
    };
    auto stage_piece = [&](auto P, char* wbase) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value, i = p >> 1, hf = p & 1;
        split2_rn2(ra[i][2 * hf], ra[i][2 * hf + 1], sp[hf]);
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
    auto step = [&](auto SS, const char* rbase, int roff, char* wbase, int g, int ktload) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;          // step inside the tile = weight slot set = a0 set
        constexpr int NPIECE = NA * 2, GP = TN / 2, NG = 2 * GP, PPG = (NPIECE + 1 + NG - 1) / NG;
        auto after_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int gi = decltype(G)::value;
            if constexpr (S == 0) {
                [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                    ([&] {
                        constexpr int p = gi * PPG + Q;
                        if constexpr (p < NPIECE) stage_piece(std::integral_constant<int, p>{}, wbase);
                        if constexpr (p == NPIECE) gloadA(ktload);
                    }(), ...);
                }(std::make_integer_sequence<int, PPG>{});
            }
            if constexpr (gi == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[S ^ 1][i] = rdA(rbase, roff, 0, i);
            }
            FENCE();
        };
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[i], w[S][j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 2 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[S][j0 + jj], acc[i][j0 + jj]);
                wload(std::integral_constant<int, S>{}, std::integral_constant<int, j0>{}, g + 2);
                wload(std::integral_constant<int, S>{}, std::integral_constant<int, j0 + 1>{}, g + 2);
                if constexpr (JP == GP - 1) {               // a1's last product of the step has issued
#pragma unroll
                    for (int i = 0; i < TM; ++i) a1[i] = rdA(rbase, roff, 1, i);
                }
                after_group(std::integral_constant<int, 2 * JP + 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
    };
    // NOTE a1 of the next step is read after the LAST pair's a0 products were issued, i.e. at the very end of the step: its first use is
    // the next step's first MFMA -- exposed; see the measured result before refining (a1 double-buffered costs TM * 4 registers).

    gloadA(0);
    [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
        ((wload(std::integral_constant<int, 0>{}, std::integral_constant<int, J>{}, 0),
          wload(std::integral_constant<int, 1>{}, std::integral_constant<int, J>{}, 1)), ...);
    }(std::make_integer_sequence<int, TN>{});
    [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
        (stage_piece(std::integral_constant<int, P>{}, smem), ...);
    }(std::make_integer_sequence<int, NA * 2>{});
    gloadA(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { a0[0][i] = rdA(smem, rd0, 0, i); a1[i] = rdA(smem, rd0, 1, i); }
    __builtin_amdgcn_s_setprio(1);
    FENCE();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * BUF;
        char* nxt = smem + ((kt + 1) & 1) * BUF;
        step(std::integral_constant<int, 0>{}, cur, rd1, nxt, 2 * kt, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
        step(std::integral_constant<int, 1>{}, nxt, rd0, nxt, 2 * kt + 1, 0);
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    }
    probe.done();
}

// ------------------------------------------------------------------------------------------------------------------
// two-term mode with BOTH operands through LDS (the library's two-term tile keeps W in LDS: 4 MFMAs per loaded KB are too few for the
// global -> VGPR weight path) on the per-wave pipeline: A fragments double-buffered in registers (read a whole step ahead), W fragments
// re-read right after their pair's last product (8 MFMAs ahead), staging of the next K tile (A split into two planes, W copied) in pieces
// between the MFMA groups of the tile's first step, ONE barrier per K tile between its two steps, buffer loads, no branches in the loop.
// W here is plain bf16 [N][K] row-major (the library's layout for this mode).
template <int TM, int TN, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_w2l_v2(const float* __restrict__ A, const unsigned short* __restrict__ Wb,
                                                         float* __restrict__ C, int M, int N, int K) {
    static_assert(TN % 2 == 0, "column tiles go in pairs");
    ClockProbe probe;
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32;
    constexpr int A_T = BM * 64, W_T = BN * 64, BUF = 2 * A_T + W_T;
    constexpr int NA = BM / 32, NB = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;            // A staging: rows r0 + 32 i, float4 column c4
    const int wc = t & 3, wr = t >> 2;            // W staging: rows wr + 64 i, 16-B chunk wc
    unsigned aoff[NA], woff[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; aoff[i] = (unsigned)(ar - m0) * (unsigned)K * 4u + c4 * 16; }
#pragma unroll
    for (int i = 0; i < NB; ++i) woff[i] = (unsigned)(wr + 64 * i) * (unsigned)K * 2u + wc * 16;
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m0 * K), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wb + (size_t)n0 * K), 0, 0x7fffffff, 0x00020000);
    const int sw = (l31 >> 2) & 3;
    const int rdA0 = (wm * TM * 32 + l31) * 64 + ((h ^ sw) << 4), rdA1 = rdA0 ^ 32;
    const int rdW0 = 2 * A_T + (wn * TN * 32 + l31) * 64 + ((h ^ sw) << 4), rdW1 = rdW0 ^ 32;
    const int wrA = r0 * 64 + (((c4 >> 1) ^ ((r0 >> 2) & 3)) << 4) + (c4 & 1) * 8;
    const int wrW = 2 * A_T + wr * 64 + ((wc ^ ((wr >> 2) & 3)) << 4);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NA];
    u32x4 rb[NB];
    bf16x8 a0[2][TM], a1[2][TM], w[TN];
    unsigned sp[2][2];

    auto gloadAW = [&](int kt) __attribute__((always_inline)) {
        kt = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, (int)aoff[i], kt * (BK * 4), 0));
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw_rsrc, (int)woff[i], kt * (BK * 2), 0);
    };
    auto rd = [&](const char* base, int off, int imm) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + off + imm);
    };
    // staging pieces: 2 * NA split halves (the 8-byte stores follow the second half of a float4), then NB weight stores
    constexpr int NPIECE = 2 * NA + NB;
    auto stage_piece = [&](auto P, char* wbase) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        if constexpr (p < 2 * NA) {
            constexpr int i = p >> 1, hf = p & 1;
            split2_rn2(ra[i][2 * hf], ra[i][2 * hf + 1], sp[hf]);
            if constexpr (hf == 1) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    u32x2 v; v[0] = sp[0][k]; v[1] = sp[1][k];
                    *reinterpret_cast<u32x2*>(wbase + wrA + k * A_T + i * 2048) = v;
                }
            }
        } else {
            constexpr int i = p - 2 * NA;
            *reinterpret_cast<u32x4*>(wbase + wrW + i * 4096) = rb[i];
        }
    };
    auto mf = [&](const bf16x8& a, const bf16x8& b, f32x16& c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };
    // one K = 16 step; S = step inside the tile = set of the A fragments in use; rbase / (roffA, roffW): where the NEXT step's fragments are
    auto step = [&](auto SS, const char* rbase, int roffA, int roffW, char* wbase, int ktload) __attribute__((always_inline)) {
        constexpr int S = decltype(SS)::value;
        constexpr int GP = TN / 2, NG = 2 * GP, PPG = (NPIECE + 1 + NG - 1) / NG;
        auto after_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int gi = decltype(G)::value;
            if constexpr (S == 0) {
                [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                    ([&] {
                        constexpr int p = gi * PPG + Q;
                        if constexpr (p < NPIECE) stage_piece(std::integral_constant<int, p>{}, wbase);
                        if constexpr (p == NPIECE) gloadAW(ktload);
                    }(), ...);
                }(std::make_integer_sequence<int, PPG>{});
            }
            if constexpr (gi == 0) {                 // the next step's A fragments, a whole step ahead
#pragma unroll
                for (int i = 0; i < TM; ++i) { a1[S ^ 1][i] = rd(rbase, roffA, A_T + i * 2048); a0[S ^ 1][i] = rd(rbase, roffA, i * 2048); }
            }
            FENCE();
        };
        [&]<int... JP>(std::integer_sequence<int, JP...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int j0 = 2 * JP;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a1[S][i], w[j0 + jj], acc[i][j0 + jj]);
                after_group(std::integral_constant<int, 2 * JP + 0>{});
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mf(a0[S][i], w[j0 + jj], acc[i][j0 + jj]);
                w[j0] = rd(rbase, roffW, j0 * 2048);                      // this pair's fragments of the next step
                w[j0 + 1] = rd(rbase, roffW, (j0 + 1) * 2048);
                after_group(std::integral_constant<int, 2 * JP + 1>{});
            }(), ...);
        }(std::make_integer_sequence<int, GP>{});
    };

    gloadAW(0);
    [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
        (stage_piece(std::integral_constant<int, P>{}, smem), ...);
    }(std::make_integer_sequence<int, NPIECE>{});
    gloadAW(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { a0[0][i] = rd(smem, rdA0, i * 2048); a1[0][i] = rd(smem, rdA0, A_T + i * 2048); }
#pragma unroll
    for (int j = 0; j < TN; ++j) w[j] = rd(smem, rdW0, j * 2048);
    __builtin_amdgcn_s_setprio(1);
    FENCE();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * BUF;
        char* nxt = smem + ((kt + 1) & 1) * BUF;
        step(std::integral_constant<int, 0>{}, cur, rdA1, rdW1, nxt, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
        step(std::integral_constant<int, 1>{}, nxt, rdA0, rdW0, nxt, 0);
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    }
    probe.done();
}

// ------------------------------------------------------------------------------------------------------------------
static unsigned short f2bf_rn(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

// W [N][K] f32 -> fragment streams [n-tile][q = ks * 3 + ph][lane][8], plane kw = 2 - ph
static void pack_w(const float* W, int N, int K, std::vector<unsigned short>& out) {
    const int nph = (K / 16) * 3;
    out.assign((size_t)(N / 32) * nph * 512, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float r = W[(size_t)n * K + k];
            unsigned short pl[3];
            for (int p = 0; p < 3; ++p) { pl[p] = f2bf_rn(r); r -= bf2f(pl[p]); }
            if (r != 0.f) { printf("W split not exact\n"); exit(1); }
            const int ntile = n / 32, l31 = n % 32, ks = k / 16, hh = (k % 16) / 8, e = k % 8;
            for (int ph = 0; ph < 3; ++ph)
                out[(((size_t)ntile * nph + ks * 3 + ph) * 64 + hh * 32 + l31) * 8 + e] = pl[2 - ph];
        }
}

// f32 [M][K] -> three bf16 planes per row, [M][3][K] (the producer-side split of STAGE == 3)
__global__ void planes_kernel(const float* __restrict__ A, unsigned* __restrict__ P, size_t npairs, int K) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npairs) return;
    const size_t row = g / (K / 2); const int kp = (int)(g % (K / 2));
    unsigned o[3];
    split2_rn3(A[row * K + 2 * kp], A[row * K + 2 * kp + 1], o);
#pragma unroll
    for (int k = 0; k < 3; ++k) P[(row * 3 + k) * (K / 2) + kp] = o[k];
}
static unsigned* g_planes = nullptr;       // planes of the A matrix of the current test (main sets it before the variants run)

template <class KERN>
static float run(KERN k, int BM, int BN, const float* A, const char* Wp, float* C, int M, int N, int K, int iters, int nthr = 256) {
    const int lds = 2 * 3 * BM * 64;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + BM - 1) / BM) * (N / BN);
    if (iters == 0) { hipLaunchKernelGGL(k, dim3(nwg), dim3(nthr), lds, 0, A, Wp, C, M, N, K); CK(hipDeviceSynchronize()); return 0.f; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(nthr), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(nthr), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = 147712, NMAX = 3072, KMAX = 3072;       // M = 256 images x 577 tokens
    const bool zero = argc > 1 && !strcmp(argv[1], "zero");
    const char* only = argc > 2 && !strcmp(argv[1], "one") ? argv[2] : nullptr;       // x3_lab one <variant> [N K]: for counter passes
    std::vector<float> hA((size_t)M * 768), hW((size_t)NMAX * KMAX);
    unsigned s = 12345;
    auto uni = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 16777216.0f; };
    auto gauss = [&]() { float u1 = uni() + 1e-9f, u2 = uni(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    for (auto& v : hA) v = gauss();
    for (auto& v : hW) v = gauss() * 0.03f;
    float *A, *C; char* Wp;
    CK(hipMalloc(&A, (size_t)M * KMAX * 4)); CK(hipMalloc(&C, (size_t)M * NMAX * 4));
    CK(hipMalloc(&Wp, (size_t)3 * NMAX * KMAX * 2));
    for (int rep = 0; rep < 4; ++rep) CK(hipMemcpy((char*)A + (size_t)rep * M * 768 * 4, hA.data(), (size_t)M * 768 * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> pk;

    struct Var { const char* name; int BM, BN; float (*fn)(const float*, const char*, float*, int, int, int, int); };
    static const Var vars[] = {
        {"base 128x256 (library r4 loop)", 128, 256, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_base<2, 4, 2>, 128, 256, a, w, c, m, n, k, it); }},
        {"v2   128x256 staged interleaved", 128, 256, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<2, 4, 2, 0, 0>, 128, 256, a, w, c, m, n, k, it); }},
        {"v2   128x256 A as bf16 planes [M][3][K] (producer-side split)", 128, 256, [](const float*, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<2, 4, 2, 3, 0>, 128, 256, reinterpret_cast<const float*>(g_planes), w, c, m, n, k, it); }},
        {"v2   128x256 staged lump", 128, 256, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<2, 4, 2, 1, 0>, 128, 256, a, w, c, m, n, k, it); }},
        {"v2   128x256 interleaved + setprio", 128, 256, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<2, 4, 2, 0, 1>, 128, 256, a, w, c, m, n, k, it); }},
        {"base 128x128", 128, 128, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_base<2, 2, 2>, 128, 128, a, w, c, m, n, k, it); }},
        {"v2   128x128 interleaved", 128, 128, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<2, 2, 2, 0, 0>, 128, 128, a, w, c, m, n, k, it); }},
        {"v2   128x128 interleaved 3 blocks/CU", 128, 128, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<2, 2, 3, 0, 0>, 128, 128, a, w, c, m, n, k, it); }},
        {"v2w  256x256 8 waves interleaved", 256, 256, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2w<2, 4, 4, 2, 0, 1>, 256, 256, a, w, c, m, n, k, it, 512); }},
        {"base 64x128", 64, 128, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_base<1, 2, 2>, 64, 128, a, w, c, m, n, k, it); }},
        {"v2   64x128 interleaved", 64, 128, [](const float* a, const char* w, float* c, int m, int n, int k, int it) { return run(gemm_x3_v2<1, 2, 2, 0, 0>, 64, 128, a, w, c, m, n, k, it); }},
    };
    const int NV = sizeof(vars) / sizeof(vars[0]);

    // ---- correctness: every variant against float64 (and against the baseline's figures)
    if (!only) for (int k : {64, 768, 3072}) {
        const int m = 300, n = 256;                       // m not a multiple of the tile: the row clamp
        std::vector<float> a((size_t)m * k), w((size_t)n * k);
        for (auto& v : a) v = gauss();
        for (auto& v : w) v = gauss() * 0.03f;
        std::vector<double> ref((size_t)m * n);
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
            double r = 0;
            for (int kk = 0; kk < k; ++kk) r += (double)a[(size_t)i * k + kk] * (double)w[(size_t)j * k + kk];
            ref[(size_t)i * n + j] = r;
        }
        float* a2; CK(hipMalloc(&a2, a.size() * 4)); CK(hipMemcpy(a2, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        pack_w(w.data(), n, k, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> hC((size_t)m * n);
        CK(hipMalloc(&g_planes, (size_t)m * k * 6));
        hipLaunchKernelGGL(planes_kernel, dim3((unsigned)(((size_t)m * k / 2 + 255) / 256)), dim3(256), 0, 0, a2, g_planes, (size_t)m * k / 2, k);
        for (int v = 0; v < NV; ++v) {
            CK(hipMemset(C, 0xFF, hC.size() * 4));
            vars[v].fn(a2, Wp, C, m, n, k, 0);
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, se = 0;
            for (size_t i = 0; i < hC.size(); ++i) { const double e = hC[i] - ref[i]; maxe = fmax(maxe, fabs(e)); se += e * e; }
            printf("  K=%4d %-40s max|err| %.3e  rms err %.3e%s\n", k, vars[v].name, maxe, sqrt(se / hC.size()), (maxe < 1e-4 && maxe == maxe) ? "" : "   <-- WRONG");
        }
        CK(hipFree(a2)); CK(hipFree(g_planes)); g_planes = nullptr;
    }

    // ---- speed on the B = 256 batch shapes (random operands unless `zero`)
    if (zero) CK(hipMemset(A, 0, (size_t)M * KMAX * 4));
    struct Shape { int N, K; } shapes[] = {{3072, 768}, {768, 3072}, {2304, 768}, {768, 768}};
    for (auto sh : shapes) {
        if (only && argc > 4 && (sh.N != atoi(argv[3]) || sh.K != atoi(argv[4]))) continue;
        const double fl = 2.0 * M * sh.N * sh.K;
        pack_w(hW.data(), sh.N, sh.K, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        if (zero) CK(hipMemset(Wp, 0, pk.size() * 2));
        if (g_planes) CK(hipFree(g_planes));
        CK(hipMalloc(&g_planes, (size_t)M * sh.K * 6));
        hipLaunchKernelGGL(planes_kernel, dim3((unsigned)(((size_t)M * sh.K / 2 + 255) / 256)), dim3(256), 0, 0, A, g_planes, (size_t)M * sh.K / 2, sh.K);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < (only ? 1 : 2); ++rep)
            for (int v = 0; v < NV; ++v) {
                if (only && !strstr(vars[v].name, only)) continue;
                if (vars[v].BM < 128 && rep) continue;
                unsigned long long z[2] = {0, 0}, c[2];
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)));
                const float ms = vars[v].fn(A, Wp, C, M, sh.N, sh.K, 5);
                CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
                const double ghz = c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0;
                // matrix-pipe busy fraction implied by time and clock: MFMA cycles needed / (SIMD cycles available)
                const double busy = ghz > 0 ? (6 * fl / 32768.0 * 32.0) / (1024.0 * ghz * 1e9 * ms * 1e-3) : 0.0;
                printf("N=%4d K=%4d %-40s %7.3f ms  %6.1f TFLOP/s algorithmic  %6.0f executed  %.2f GHz  pipe busy %.2f%s\n", sh.N, sh.K, vars[v].name, ms,
                       fl / ms / 1e9, 6 * fl / ms / 1e9, ghz, busy, zero ? "  (ZERO operands)" : "");
                fflush(stdout);
            }
        printf("\n");
    }
    // ---- two-term mode: correctness (vs float64 on bf16-rounded weights) and speed
    {
        auto pack_w1 = [&](const float* W, int N, int K, std::vector<unsigned short>& out) {
            const int ns = K / 16;
            out.assign((size_t)(N / 32) * ns * 512, 0);
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k)
                    out[(((size_t)(n / 32) * ns + k / 16) * 64 + ((k % 16) / 8) * 32 + n % 32) * 8 + k % 8] = f2bf_rn(W[(size_t)n * K + k]);
        };
        auto run_w2 = [&](auto kern, int BM, int BN, const float* a, float* c, int m, int n, int k, int iters) {
            const int lds = 2 * 2 * BM * 64;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const int nwg = ((m + BM - 1) / BM) * (n / BN);
            if (iters == 0) { hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a, Wp, c, m, n, k); CK(hipDeviceSynchronize()); return 0.f; }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a, Wp, c, m, n, k);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a, Wp, c, m, n, k);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms / iters;
        };
        if (!only) for (int k : {64, 768, 3072}) {
            const int m = 300, n = 256;
            std::vector<float> a((size_t)m * k), w((size_t)n * k);
            for (auto& v : a) v = gauss();
            for (auto& v : w) v = bf2f(f2bf_rn(gauss() * 0.03f));
            std::vector<double> ref((size_t)m * n);
            for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
                double r = 0;
                for (int kk = 0; kk < k; ++kk) r += (double)a[(size_t)i * k + kk] * (double)w[(size_t)j * k + kk];
                ref[(size_t)i * n + j] = r;
            }
            float* a2; CK(hipMalloc(&a2, a.size() * 4)); CK(hipMemcpy(a2, a.data(), a.size() * 4, hipMemcpyHostToDevice));
            pack_w1(w.data(), n, k, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
            std::vector<float> hC((size_t)m * n);
            CK(hipMemset(C, 0xFF, hC.size() * 4));
            run_w2(gemm_w2_v2<2, 4, 2>, 128, 256, a2, C, m, n, k, 0);
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, se = 0;
            for (size_t i = 0; i < hC.size(); ++i) { const double e = hC[i] - ref[i]; maxe = fmax(maxe, fabs(e)); se += e * e; }
            printf("  K=%4d two-term v2 128x256                      max|err| %.3e  rms err %.3e%s\n", k, maxe, sqrt(se / hC.size()), (maxe < 1e-3 && maxe == maxe) ? "" : "   <-- WRONG");
            CK(hipFree(a2));
        }
        auto run_w2l = [&](auto kern, int BM, int BN, const float* a, const unsigned short* wb, float* c, int m, int n, int k, int iters) {
            const int lds = 2 * (2 * BM + BN) * 64;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const int nwg = ((m + BM - 1) / BM) * (n / BN);
            if (iters == 0) { hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a, wb, c, m, n, k); CK(hipDeviceSynchronize()); return 0.f; }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a, wb, c, m, n, k);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a, wb, c, m, n, k);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms / iters;
        };
        unsigned short* Wb16; CK(hipMalloc(&Wb16, (size_t)NMAX * KMAX * 2));
        if (!only) for (int k : {64, 768, 3072}) {
            const int m = 300, n = 256;
            std::vector<float> a((size_t)m * k);
            std::vector<unsigned short> wb((size_t)n * k);
            for (auto& v : a) v = gauss();
            for (auto& v : wb) v = f2bf_rn(gauss() * 0.03f);
            std::vector<double> ref((size_t)m * n);
            for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
                double r = 0;
                for (int kk = 0; kk < k; ++kk) r += (double)a[(size_t)i * k + kk] * (double)bf2f(wb[(size_t)j * k + kk]);
                ref[(size_t)i * n + j] = r;
            }
            float* a2; CK(hipMalloc(&a2, a.size() * 4)); CK(hipMemcpy(a2, a.data(), a.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(Wb16, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
            std::vector<float> hC((size_t)m * n);
            CK(hipMemset(C, 0xFF, hC.size() * 4));
            run_w2l(gemm_w2l_v2<2, 4, 2>, 128, 256, a2, Wb16, C, m, n, k, 0);
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, se = 0;
            for (size_t i = 0; i < hC.size(); ++i) { const double e = hC[i] - ref[i]; maxe = fmax(maxe, fabs(e)); se += e * e; }
            printf("  K=%4d two-term v2, W through LDS, 128x256       max|err| %.3e  rms err %.3e%s\n", k, maxe, sqrt(se / hC.size()), (maxe < 1e-3 && maxe == maxe) ? "" : "   <-- WRONG");
            CK(hipFree(a2));
        }
        {
            std::vector<unsigned short> wb((size_t)NMAX * KMAX);
            for (size_t i = 0; i < wb.size(); ++i) wb[i] = f2bf_rn(hW[i]);
            CK(hipMemcpy(Wb16, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
        }
        for (auto sh : shapes) {
            const double fl = 2.0 * M * sh.N * sh.K;
            for (int rep = 0; rep < 2; ++rep) {
                unsigned long long z[2] = {0, 0}, c[2];
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)));
                const float ms = run_w2l(gemm_w2l_v2<2, 4, 2>, 128, 256, A, Wb16, C, M, sh.N, sh.K, 5);
                CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
                const double ghz = c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0;
                printf("N=%4d K=%4d two-term v2, W through LDS, 128x256                    %7.3f ms  %6.1f TFLOP/s algorithmic  %6.0f executed  %.2f GHz\n", sh.N, sh.K, ms,
                       fl / ms / 1e9, 2 * fl / ms / 1e9, ghz);
            }
        }
        for (auto sh : shapes) {
            const double fl = 2.0 * M * sh.N * sh.K;
            pack_w1(hW.data(), sh.N, sh.K, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; ++rep) {
                unsigned long long z[2] = {0, 0}, c[2];
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)));
                const float ms = run_w2(gemm_w2_v2<2, 4, 2>, 128, 256, A, C, M, sh.N, sh.K, 5);
                CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
                const double ghz = c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0;
                printf("N=%4d K=%4d two-term v2 128x256 (W global -> VGPR, 2 slot sets) %7.3f ms  %6.1f TFLOP/s algorithmic  %6.0f executed  %.2f GHz\n", sh.N, sh.K, ms,
                       fl / ms / 1e9, 2 * fl / ms / 1e9, ghz);
            }
        }
    }
    return 0;
}
