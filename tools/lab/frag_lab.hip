// Lab (round 4): bf16-pipe GEMM main loop with the WEIGHT operand out of LDS.
//   * W is pre-packed once, per bf16 plane, in MFMA-fragment order ([n-tile of 32][phase][lane][8 bf16] = 1 KB per
//     wave load) and goes global -> VGPR, through a ring of D fragment sets a few phases ahead of its use;
//   * LDS carries only the activation planes (f32 -> bf16 split while staging, XOR-swizzled 64-B rows);
//   * modes: NAT activation planes x NWT weight planes, a pair (ka, kw) is multiplied when ka + kw <= MAXSUM:
//       3 x 3, MAXSUM 4 = "bf16x9": both f32 operands as three EXACT bf16 terms, all 9 partial products
//       3 x 3, MAXSUM 3 = x8 (drops a2*w2, <= 2^-34 relative), MAXSUM 2 = x6 (drops <= 2^-26 relative per product)
//       2 x 1, MAXSUM 1 = bf16 weights, two-term activations (BASELINE config 5)
//   Splits are round-to-nearest (v_cvt_pk_bf16_f32): a0 = rn(a), a1 = rn(a - a0), a2 = a - a0 - a1 (exact: 8 + 8 + 8
//   significand bits with signed remainders).
// Also: register-only bf16 MFMA rate loops on random data (what the part sustains under its power limit) and the
// error of every mode against float64 next to a native f32-MFMA kernel on the same inputs.
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

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8; int xcd = bid % nx, idx = bid / nx; int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; return base + idx;
}

// round-to-nearest split of 4 floats into NA planes of 4 bf16 (2 dwords each)
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
                x[0] = x[0] - __uint_as_float(hb << 16);               // exact
                x[1] = x[1] - __uint_as_float(hb & 0xFFFF0000u);
            }
        }
    }
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int TM, int TN, int NAT, int NWT, int MAXSUM, int D, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_frag(const float* __restrict__ A, const char* __restrict__ Wp,
                                                       float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32, NS = BK / 16, PPT = NS * NWT;
    static_assert(PPT % D == 0, "ring depth must divide the phases of a K tile");
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
    // W fragment streams: n-tile j of this wave = (n0 + wn * TN * 32) / 32 + j; stream of nph KB
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
    auto wload = [&](auto SET, int q) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        q = q < nph ? q : nph - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) ring[st][j] = *reinterpret_cast<const u32x4*>(wb[j] + (size_t)q * 1024);
    };
    const int nk = K / BK;
    gloadA(0);
    // ring prologue: phases 0 .. D-2
    [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
        (wload(std::integral_constant<int, Q>{}, Q), ...);
    }(std::make_integer_sequence<int, D - 1>{});
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gloadA(kt + 1);
        const char* base = smem + (kt & 1) * BUF;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 fa[NAT][TM];
        [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int s = P / NWT, ph = P % NWT, kw = NWT - 1 - ph;
                // refill the ring D-1 phases ahead
                wload(std::integral_constant<int, (P + D - 1) % D>{}, kt * PPT + P + D - 1);
                if constexpr (ph == 0) {
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
            }(), ...);
        }(std::make_integer_sequence<int, PPT>{});
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
}

template <int TM, int TN, int NAT, int NWT, int MAXSUM, int D, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_frag3(const float* __restrict__ A, const char* __restrict__ Wp,
                                                       float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32, NS = BK / 16, PPT = NS * NWT;
    static_assert(PPT % D == 0, "ring depth must divide the phases of a K tile");
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
    // W fragment streams: n-tile j of this wave = (n0 + wn * TN * 32) / 32 + j; stream of nph KB
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
    auto lstore_part = [&](int buf, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        char* base = smem + buf * BUF;
        {
            u32x2 sp[NAT];
            split4_rn<NAT>(ra[i], sp);
            const int off = lds_off(r0 + 32 * i, c4 >> 1) + (c4 & 1) * 8;
#pragma unroll
            for (int k = 0; k < NAT; ++k) *reinterpret_cast<u32x2*>(base + k * A_T + off) = sp[k];
        }
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
    auto wload = [&](auto SET, int q) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        q = q < nph ? q : nph - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) ring[st][j] = *reinterpret_cast<const u32x4*>(wb[j] + (size_t)q * 1024);
    };
    const int nk = K / BK;
    gloadA(0);
    // ring prologue: phases 0 .. D-2
    [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
        (wload(std::integral_constant<int, Q>{}, Q), ...);
    }(std::make_integer_sequence<int, D - 1>{});
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gloadA(kt + 1);
        const char* base = smem + (kt & 1) * BUF;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 fa[NAT][TM];
        [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int s = P / NWT, ph = P % NWT, kw = NWT - 1 - ph;
                // refill the ring D-1 phases ahead
                wload(std::integral_constant<int, (P + D - 1) % D>{}, kt * PPT + P + D - 1);
                if constexpr (ph == 0) {
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
                // the split + LDS store of the NEXT tile's activations, one f32x4 per phase from phase PPT - NA on: VALU / DS work that
                // issues in the shadow of this phase's MFMAs instead of after the last one
                if constexpr (P >= PPT - NA) { if (more) lstore_part((kt + 1) & 1, std::integral_constant<int, P - (PPT - NA)>{}); }
            }(), ...);
        }(std::make_integer_sequence<int, PPT>{});
        __builtin_amdgcn_s_setprio(0);
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
}

template <int TM, int TN, int NAT, int NWT, int MAXSUM, int D, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_frag2(const float* __restrict__ A, const char* __restrict__ Wp,
                                                       float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32, NS = BK / 16, PPT = NS * NWT;
    static_assert(PPT % D == 0, "ring depth must divide the phases of a K tile");
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
    // W fragment streams: n-tile j of this wave = (n0 + wn * TN * 32) / 32 + j; stream of nph KB
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

    f32x4 ra[2][NA];                                    // A prefetch TWO K tiles deep: loads issued at the top of tile kt are for tile kt + 2
    u32x4 ring[D][TN];
    auto gloadA = [&](auto SET, int kt) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[st][i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
    };
    auto lstore = [&](auto SET, int buf) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 sp[NAT];
            split4_rn<NAT>(ra[st][i], sp);
            const int off = lds_off(r0 + 32 * i, c4 >> 1) + (c4 & 1) * 8;
#pragma unroll
            for (int k = 0; k < NAT; ++k) *reinterpret_cast<u32x2*>(base + k * A_T + off) = sp[k];
        }
    };
    auto wload = [&](auto SET, int q) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        q = q < nph ? q : nph - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) ring[st][j] = *reinterpret_cast<const u32x4*>(wb[j] + (size_t)q * 1024);
    };
    const int nk = K / BK;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    gloadA(S0{}, 0);
    if (nk > 1) gloadA(S1{}, 1);
    [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
        (wload(std::integral_constant<int, Q>{}, Q), ...);
    }(std::make_integer_sequence<int, D - 1>{});
    lstore(S0{}, 0);
    __syncthreads();
    auto ktile = [&](auto SET, int kt) __attribute__((always_inline)) {
        // on entry: LDS buffer kt & 1 holds tile kt; register set SET is free (tile kt went to LDS), the other set holds tile kt + 1
        if (kt + 2 < nk) gloadA(SET, kt + 2);
        const char* base = smem + (kt & 1) * BUF;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 fa[NAT][TM];
        [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int s = P / NWT, ph = P % NWT, kw = NWT - 1 - ph;
                wload(std::integral_constant<int, (P + D - 1) % D>{}, kt * PPT + P + D - 1);
                if constexpr (ph == 0) {
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
            }(), ...);
        }(std::make_integer_sequence<int, PPT>{});
        __builtin_amdgcn_s_setprio(0);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(S0{}, kt);
        if (kt + 1 < nk) lstore(S1{}, 1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        ktile(S1{}, kt + 1);
        if (kt + 2 < nk) lstore(S0{}, 0);
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
}

// Two-term bf16-weight tile, HYBRID W path: of a wave's 4 column tiles NL come through LDS (staged from the fragment-packed
// copy: a fragment is 1 KB contiguous, read back linearly) and 4 - NL straight from global memory.  The all-LDS tile is
// LDS-bound (96 KB per 1024 matrix-pipe cycles), the all-global one vector-memory-bound; this splits the W bytes between the paths.
template <int NL, int D>
__global__ __launch_bounds__(256, 2) void gemm_w2_hybrid(const float* __restrict__ A, const char* __restrict__ Wp,
                                                         float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = 2, TN = 4, NG = TN - NL, BM = 128, BN = 256, BK = 32, NAT = 2;
    constexpr int A_T = BM * 64, W_T = 2 * NL * 2 * 1024;                 // W region per buffer: wn (2) x NL tiles x 2 k-steps x 1 KB
    constexpr int BUF = NAT * A_T + W_T;
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
    const int nph = K / 16;                                               // KB per n-tile stream (one plane)
    // global tiles of this wave: j = NL .. TN-1
    const char* wg[NG > 0 ? NG : 1];
#pragma unroll
    for (int j = 0; j < NG; ++j) wg[j] = Wp + (size_t)((n0 >> 5) + wn * TN + NL + j) * nph * 1024 + lane * 16;
    // LDS staging of the NL tiles of both wn: 2 * NL * 2 fragments of 1 KB per K tile = 4 NL KB = NL * 256 threads x 16 B
    // chunk c (16 B): frag = c / 64 -> (wn_ = frag / (NL * 2), jl = (frag / 2) % NL, s = frag % 2), lane_ = c % 64
    const char* wl[NL > 0 ? NL : 1];
#pragma unroll
    for (int p = 0; p < NL; ++p) {
        const int c = t + 256 * p, frag = c >> 6, ln = c & 63;
        const int wn_ = frag / (NL * 2), jl = (frag >> 1) % NL, s = frag & 1;
        wl[p] = Wp + ((size_t)((n0 >> 5) + wn_ * TN + jl) * nph + s) * 1024 + ln * 16;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NA];
    u32x4 rw[NL > 0 ? NL : 1];
    u32x4 ring[D][NG > 0 ? NG : 1];
    auto gload = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
#pragma unroll
        for (int p = 0; p < NL; ++p) rw[p] = *reinterpret_cast<const u32x4*>(wl[p] + (size_t)kt * 2048);
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
#pragma unroll
        for (int p = 0; p < NL; ++p) *reinterpret_cast<u32x4*>(base + NAT * A_T + (t + 256 * p) * 16) = rw[p];
    };
    auto wload = [&](auto SET, int q) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        q = q < nph ? q : nph - 1;
#pragma unroll
        for (int j = 0; j < NG; ++j) ring[st][j] = *reinterpret_cast<const u32x4*>(wg[j] + (size_t)q * 1024);
    };
    const int nk = K / BK;
    gload(0);
    if constexpr (D > 1) wload(std::integral_constant<int, 0>{}, 0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        const char* base = smem + (kt & 1) * BUF;
        __builtin_amdgcn_s_setprio(1);
        [&]<int... P>(std::integer_sequence<int, P...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int s = P;
                if constexpr (NG > 0) wload(std::integral_constant<int, (P + D - 1) % D>{}, kt * 2 + P + D - 1);
                bf16x8 fa[NAT][TM], fw[NL > 0 ? NL : 1];
#pragma unroll
                for (int k = 0; k < NAT; ++k)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[k][i] = *reinterpret_cast<const bf16x8*>(base + k * A_T + lds_off(wm * TM * 32 + i * 32 + l31, 2 * s + h));
#pragma unroll
                for (int j = 0; j < NL; ++j)
                    fw[j] = *reinterpret_cast<const bf16x8*>(base + NAT * A_T + (((wn * NL + j) * 2 + s) * 64 + lane) * 16);
#pragma unroll
                for (int ka = NAT - 1; ka >= 0; --ka)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int j = 0; j < NL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ka][i], fw[j], acc[i][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < NG; ++j)
                            acc[i][NL + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ka][i], __builtin_bit_cast(bf16x8, ring[P % D][j]), acc[i][NL + j], 0, 0, 0);
                    }
            }(), ...);
        }(std::make_integer_sequence<int, 2>{});
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
}

template <int TM, int TN, int NAT, int NWT, int MAXSUM, int D, int MINW>
float launch3(const float* A, const char* Wp, float* C, int M, int N, int K, int iters) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    static_assert(BM / 32 <= 2 * NWT, "one lstore part per phase");
    const int lds = 2 * NAT * BM * 64;
    auto k = gemm_frag3<TM, TN, NAT, NWT, MAXSUM, D, MINW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + BM - 1) / BM) * (N / BN);
    if (iters == 0) { hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K); CK(hipDeviceSynchronize()); return 0.f; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int TM, int TN, int NAT, int NWT, int MAXSUM, int D, int MINW>
float launch2(const float* A, const char* Wp, float* C, int M, int N, int K, int iters) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    const int lds = 2 * NAT * BM * 64;
    auto k = gemm_frag2<TM, TN, NAT, NWT, MAXSUM, D, MINW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + BM - 1) / BM) * (N / BN);
    if (iters == 0) { hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K); CK(hipDeviceSynchronize()); return 0.f; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int NL, int D>
float launch_h(const float* A, const char* Wp, float* C, int M, int N, int K, int iters) {
    const int lds = 2 * (2 * 128 * 64 + 2 * NL * 2 * 1024);
    auto k = gemm_w2_hybrid<NL, D>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + 127) / 128) * (N / 256);
    if (iters == 0) { hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K); CK(hipDeviceSynchronize()); return 0.f; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

// native f32 MFMA, one wave per 32x32 tile, k in order (error reference only)
__global__ void gemm_native_f32(const float* A, const float* W, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + l31) * K + k + h], W[(size_t)(n0 + l31) * K + k + h], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + n0 + l31] = acc[r];
}

// register-only MFMA rate: NACC independent accumulators, operands rotate over 4 random fragments
template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate(const u32x4* __restrict__ frags, float* out, int iters) {
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = frags[(i * 64 + (threadIdx.x & 63))]; fb[i] = frags[((4 + i) * 64 + (threadIdx.x & 63))]; }
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[(u + a) & 3]), __builtin_bit_cast(bf16x8, fb[(u + 2 * a + 1) & 3]), acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate_f32(const float* __restrict__ frags, float* out, int iters) {
    float fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = frags[(i * 64 + (threadIdx.x & 63))]; fb[i] = frags[((4 + i) * 64 + (threadIdx.x & 63))]; }
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(u + a) & 3], fb[(u + 2 * a + 1) & 3], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;
}

static unsigned short f2bf_rn(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

// pack W [N][K] f32 into fragment streams: [n-tile][q = ks * NWT + ph][lane][8], plane kw = NWT - 1 - ph
static void pack_w(const float* W, int N, int K, int NWT, bool bf16_weights, std::vector<unsigned short>& out) {
    const int nph = (K / 16) * NWT;
    out.assign((size_t)(N / 32) * nph * 512, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float r = W[(size_t)n * K + k];
            unsigned short pl[3] = {0, 0, 0};
            for (int p = 0; p < NWT; ++p) { pl[p] = f2bf_rn(r); r -= bf2f(pl[p]); }
            if (!bf16_weights && NWT == 3 && r != 0.f) { printf("W split not exact\n"); exit(1); }
            const int ntile = n / 32, l31 = n % 32, ks = k / 16, hh = (k % 16) / 8, e = k % 8;
            for (int ph = 0; ph < NWT; ++ph) {
                const int kw = NWT - 1 - ph;
                out[(((size_t)ntile * nph + ks * NWT + ph) * 64 + hh * 32 + l31) * 8 + e] = pl[kw];
            }
        }
}

template <int TM, int TN, int NAT, int NWT, int MAXSUM, int D, int MINW>
float launch(const float* A, const char* Wp, float* C, int M, int N, int K, int iters) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    const int lds = 2 * NAT * BM * 64;
    auto k = gemm_frag<TM, TN, NAT, NWT, MAXSUM, D, MINW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + BM - 1) / BM) * (N / BN);
    if (iters == 0) { hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K); CK(hipDeviceSynchronize()); return 0.f; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const int M = 147712, NMAX = 3072, KMAX = 3072;       // M = 256 images x 577 tokens
    std::vector<float> hA((size_t)M * 768), hW((size_t)NMAX * KMAX);
    unsigned s = 12345;
    auto uni = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 16777216.0f; };
    auto gauss = [&]() { float u1 = uni() + 1e-9f, u2 = uni(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    for (auto& v : hA) v = gauss();
    for (auto& v : hW) v = gauss() * 0.03f;
    float *A, *C; char* Wp; float* Wf;
    CK(hipMalloc(&A, (size_t)M * KMAX * 4)); CK(hipMalloc(&C, (size_t)M * NMAX * 4));
    CK(hipMalloc(&Wp, (size_t)3 * NMAX * KMAX * 2)); CK(hipMalloc(&Wf, (size_t)NMAX * KMAX * 4));
    // A: fill M x 3072 by repeating the 768-wide random block (timing only needs random bits)
    for (int rep = 0; rep < 4; ++rep) CK(hipMemcpy((char*)A + (size_t)rep * M * 768 * 4, hA.data(), (size_t)M * 768 * 4, hipMemcpyHostToDevice));

    // ---- single-variant mode for counter runs: frag_lab one <x9|x9s|x6|x8|w2|w3> <N> <K> <zero 0/1> [iters]
    if (argc > 5 && !strcmp(argv[1], "one")) {
        const int N = atoi(argv[3]), K = atoi(argv[4]), zero = atoi(argv[5]), iters = argc > 6 ? atoi(argv[6]) : 5;
        const bool w1 = argv[2][0] == 'w';
        std::vector<unsigned short> pk;
        pack_w(hW.data(), N, K, w1 ? 1 : 3, w1, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        if (zero) { CK(hipMemset(A, 0, (size_t)M * KMAX * 4)); CK(hipMemset(Wp, 0, pk.size() * 2)); }
        float ms = 0; int np = 9;
        if (!strcmp(argv[2], "x9")) ms = launch<2, 4, 3, 3, 4, 3, 2>(A, Wp, C, M, N, K, iters);
        else if (!strcmp(argv[2], "x9s")) ms = launch<2, 2, 3, 3, 4, 3, 2>(A, Wp, C, M, N, K, iters);
        else if (!strcmp(argv[2], "x8")) { ms = launch<2, 4, 3, 3, 3, 3, 2>(A, Wp, C, M, N, K, iters); np = 8; }
        else if (!strcmp(argv[2], "x6")) { ms = launch<2, 4, 3, 3, 2, 3, 2>(A, Wp, C, M, N, K, iters); np = 6; }
        else if (!strcmp(argv[2], "w2")) { ms = launch<2, 4, 2, 1, 1, 2, 2>(A, Wp, C, M, N, K, iters); np = 2; }
        else if (!strcmp(argv[2], "w3")) { ms = launch<2, 4, 3, 1, 2, 2, 2>(A, Wp, C, M, N, K, iters); np = 3; }
        printf("one %s N=%d K=%d %s: %.3f ms  %.1f TFLOP/s algorithmic  %.0f executed\n", argv[2], N, K, zero ? "ZERO operands" : "random operands", ms,
               2.0 * M * N * K / ms / 1e9, np * 2.0 * M * N * K / ms / 1e9);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "apf")) {
        struct Shape { int N, K; } shapes[] = {{768, 768}, {2304, 768}, {3072, 768}, {768, 3072}};
        std::vector<unsigned short> pk;
        for (auto sh : shapes) {
            const double fl = 2.0 * M * sh.N * sh.K;
            pack_w(hW.data(), sh.N, sh.K, 3, false, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
            auto pr = [&](const char* nm, float ms) { printf("N=%4d K=%4d %-44s %7.3f ms  %6.1f TFLOP/s algorithmic\n", sh.N, sh.K, nm, ms, fl / ms / 1e9); fflush(stdout); };
            for (int rep = 0; rep < 2; ++rep) {
                pr("x6 128x256, A prefetch 1 tile", launch<2, 4, 3, 3, 2, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5));
                pr("x6 128x256, lstore interleaved", launch3<2, 4, 3, 3, 2, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5));
                pr("x6 128x128, A prefetch 1 tile", launch<2, 2, 3, 3, 2, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5));
                pr("x6 128x128, lstore interleaved", launch3<2, 2, 3, 3, 2, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5));
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "hybrid")) {
        struct Shape { int N, K; } shapes[] = {{768, 768}, {2304, 768}, {3072, 768}, {768, 3072}};
        std::vector<unsigned short> pk;
        for (auto sh : shapes) {
            const double fl = 2.0 * M * sh.N * sh.K;
            pack_w(hW.data(), sh.N, sh.K, 1, true, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
            auto pr = [&](const char* nm, float ms) { printf("N=%4d K=%4d %-44s %7.3f ms  %6.1f TFLOP/s algorithmic  %6.0f executed\n", sh.N, sh.K, nm, ms, fl / ms / 1e9, 2 * fl / ms / 1e9); fflush(stdout); };
            pr("two-term, all W global (D2)", launch<2, 4, 2, 1, 1, 2, 2>(A, Wp, C, M, sh.N, sh.K, 5));
            pr("two-term hybrid NL=4 (all W via LDS, packed)", launch_h<4, 1>(A, Wp, C, M, sh.N, sh.K, 5));
            pr("two-term hybrid NL=3 D2", launch_h<3, 2>(A, Wp, C, M, sh.N, sh.K, 5));
            pr("two-term hybrid NL=2 D2", launch_h<2, 2>(A, Wp, C, M, sh.N, sh.K, 5));
            pr("two-term hybrid NL=1 D2", launch_h<1, 2>(A, Wp, C, M, sh.N, sh.K, 5));
        }
        return 0;
    }
    // ---- register-only MFMA rates on random operands
    {
        std::vector<unsigned short> fr(8 * 64 * 8);
        for (auto& v : fr) v = f2bf_rn(gauss());
        std::vector<float> ff(8 * 64);
        for (auto& v : ff) v = gauss();
        u32x4* dfr; float* dff; float* dout;
        CK(hipMalloc(&dfr, fr.size() * 2)); CK(hipMalloc(&dff, ff.size() * 4)); CK(hipMalloc(&dout, 4));
        CK(hipMemcpy(dfr, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dff, ff.data(), ff.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int wps = 1; wps <= 2; ++wps) {
            const int iters = 20000, grid = 256 * wps;
            for (int zero = 0; zero < 2; ++zero) {
                if (zero) CK(hipMemset(dfr, 0, fr.size() * 2));
                hipLaunchKernelGGL(mfma_rate<8>, dim3(grid), dim3(256), 0, 0, dfr, dout, 100);
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mfma_rate<8>, dim3(grid), dim3(256), 0, 0, dfr, dout, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("mfma_f32_32x32x16_bf16 register loop, %d wave(s)/SIMD, %s operands: %.1f ms  %.0f TFLOP/s\n", wps,
                       zero ? "ZERO" : "random", ms, (double)grid * 4 * iters * 32 * 32768.0 / ms / 1e9);
            }
            CK(hipMemcpy(dfr, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(mfma_rate_f32<8>, dim3(grid), dim3(256), 0, 0, dff, dout, 100);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_rate_f32<8>, dim3(grid), dim3(256), 0, 0, dff, dout, iters / 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mfma_f32_32x32x2_f32 register loop, %d wave(s)/SIMD, random operands: %.1f ms  %.1f TFLOP/s\n", wps, ms,
                   (double)grid * 4 * (iters / 2) * 32 * 4096.0 / ms / 1e9);
        }
    }

    // ---- error vs float64 (m x n x k), every mode on the same inputs
    for (int k : {768, 3072}) {
        const int m = 256, n = 256;
        std::vector<float> a((size_t)m * k), w((size_t)n * k);
        for (auto& v : a) v = gauss();
        for (auto& v : w) v = gauss() * 0.03f;
        std::vector<double> ref((size_t)m * n), mag((size_t)m * n), refb((size_t)m * n);
        std::vector<float> wbf((size_t)n * k);
        for (size_t i = 0; i < w.size(); ++i) wbf[i] = bf2f(f2bf_rn(w[i]));
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
            double r = 0, g = 0, rb = 0;
            for (int kk = 0; kk < k; ++kk) {
                const double x = a[(size_t)i * k + kk];
                r += x * (double)w[(size_t)j * k + kk]; g += fabs(x * (double)w[(size_t)j * k + kk]); rb += x * (double)wbf[(size_t)j * k + kk];
            }
            ref[(size_t)i * n + j] = r; mag[(size_t)i * n + j] = g; refb[(size_t)i * n + j] = rb;
        }
        float* a2; CK(hipMalloc(&a2, a.size() * 4)); CK(hipMemcpy(a2, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Wf, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hC((size_t)m * n);
        auto report = [&](const char* nm, const std::vector<double>& rf) {
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, se = 0, sr = 0, maxrel = 0, bias = 0;
            for (size_t i = 0; i < hC.size(); ++i) {
                const double e = hC[i] - rf[i];
                maxe = fmax(maxe, fabs(e)); se += e * e; sr += rf[i] * rf[i]; maxrel = fmax(maxrel, fabs(e) / mag[i]); bias += e * (rf[i] > 0 ? 1 : -1);
            }
            printf("  K=%4d %-28s max|err| %.3e  rms err %.3e  rms err / rms C %.3e  max err/sum|aw| %.3e  mean signed err %.2e\n", k, nm, maxe,
                   sqrt(se / hC.size()), sqrt(se / sr), maxrel, bias / hC.size());
        };
        hipLaunchKernelGGL(gemm_native_f32, dim3(n / 32, m / 32), dim3(64), 0, 0, a2, Wf, C, m, n, k); CK(hipDeviceSynchronize());
        report("native f32 MFMA", ref);
        std::vector<unsigned short> pk;
        pack_w(w.data(), n, k, 3, false, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        launch<2, 4, 3, 3, 4, 3, 2>(a2, Wp, C, m, n, k, 0); report("bf16x9 (2x4 wave tile)", ref);
        launch<2, 2, 3, 3, 4, 3, 2>(a2, Wp, C, m, n, k, 0); report("bf16x9 (2x2 wave tile)", ref);
        launch<2, 4, 3, 3, 3, 3, 2>(a2, Wp, C, m, n, k, 0); report("bf16x8", ref);
        launch<2, 4, 3, 3, 2, 3, 2>(a2, Wp, C, m, n, k, 0); report("bf16x6", ref);
        pack_w(w.data(), n, k, 1, true, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        launch<2, 4, 2, 1, 1, 2, 2>(a2, Wp, C, m, n, k, 0); report("bf16 W, 2-term A (vs bf16 W)", refb);
        launch<2, 4, 3, 1, 2, 1, 2>(a2, Wp, C, m, n, k, 0); report("bf16 W, 3-term A (vs bf16 W)", refb);
        launch_h<2, 2>(a2, Wp, C, m, n, k, 0); report("bf16 W, 2-term, hybrid NL=2", refb);
        launch_h<4, 1>(a2, Wp, C, m, n, k, 0); report("bf16 W, 2-term, hybrid NL=4", refb);
        CK(hipFree(a2));
    }

    // ---- speed on the B = 256 batch shapes
    struct Shape { int N, K; } shapes[] = {{768, 768}, {2304, 768}, {3072, 768}, {768, 3072}};
    std::vector<unsigned short> pk;
    for (auto sh : shapes) {
        const double fl = 2.0 * M * sh.N * sh.K;
        pack_w(hW.data(), sh.N, sh.K, 3, false, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        auto pr = [&](const char* nm, float ms, int nprod) { printf("N=%4d K=%4d %-34s %7.3f ms  %6.1f TFLOP/s algorithmic  %6.0f executed\n", sh.N, sh.K, nm, ms, fl / ms / 1e9, nprod * fl / ms / 1e9); fflush(stdout); };
        pr("x9 128x256 D3 2blk", launch<2, 4, 3, 3, 4, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 9);
        pr("x9 128x128 D3 2blk", launch<2, 2, 3, 3, 4, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 9);
        if (!quick) {
            pr("x9 128x256 D2 2blk", launch<2, 4, 3, 3, 4, 2, 2>(A, Wp, C, M, sh.N, sh.K, 5), 9);
            pr("x9 128x128 D3 3blk", launch<2, 2, 3, 3, 4, 3, 3>(A, Wp, C, M, sh.N, sh.K, 5), 9);
            pr("x9 256x128 D3 2blk (4x2 wave)", launch<4, 2, 3, 3, 4, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 9);
        }
        pr("x8 128x256 D3 2blk", launch<2, 4, 3, 3, 3, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 8);
        pr("x6 128x256 D3 2blk", launch<2, 4, 3, 3, 2, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 6);
        pr("x6 128x128 D3 2blk", launch<2, 2, 3, 3, 2, 3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 6);
        pack_w(hW.data(), sh.N, sh.K, 1, true, pk); CK(hipMemcpy(Wp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        pr("bf16W 2-term 128x256 D2 2blk", launch<2, 4, 2, 1, 1, 2, 2>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        pr("bf16W 2-term 128x256 D1 2blk", launch<2, 4, 2, 1, 1, 1, 2>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        if (!quick) pr("bf16W 2-term 256x128 D2 (4x2)", launch<4, 2, 2, 1, 1, 2, 2>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        pr("bf16W 3-term 128x256 D2 2blk", launch<2, 4, 3, 1, 2, 2, 2>(A, Wp, C, M, sh.N, sh.K, 5), 3);
        pr("bf16W 2-term hybrid NL=4 (all W via LDS)", launch_h<4, 1>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        pr("bf16W 2-term hybrid NL=3 D2", launch_h<3, 2>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        pr("bf16W 2-term hybrid NL=2 D2", launch_h<2, 2>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        pr("bf16W 2-term hybrid NL=1 D2", launch_h<1, 2>(A, Wp, C, M, sh.N, sh.K, 5), 2);
        printf("\n");
    }
    return 0;
}
