// Kernel lab (not part of the library): f32-split GEMM with PRE-SPLIT activations.
// C[M,N] = A W^T with A = a_hi + a_lo and W = w_hi + w_lo held as bfloat16 planes; three products
// (a_lo w_hi, a_hi w_lo, a_hi w_hi) per K = 16 on v_mfma_f32_32x32x16_bf16.  Question: how fast is the tile
// when the operand split is not done inside the GEMM (no VALU split, both operands plain copies)?
//   V = 0: global -> VGPR -> ds_write_b128, register prefetch two K tiles deep
//   V = 1: global -> LDS directly (global_load_lds_dwordx4), LDS ring of NBUF buffers
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8; int xcd = bid % nx, idx = bid / nx; int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; return base + idx;
}
// LDS tile: [rows][32 bf16] = 64 B per row, 16-B chunks XOR-swizzled by (row >> 2) & 3
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

constexpr int BM = 128, BN = 128, BK = 32, TILE_B = 128 * 64;   // one operand-term tile: 8 KB

template <int V, int NBUF>
__global__ __launch_bounds__(256, 2) void gemm_planes(const __bf16* __restrict__ Ap /*[2][M][K]*/, const __bf16* __restrict__ Wp /*[2][N][K]*/,
                                                      float* __restrict__ C, const float* __restrict__ bias, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer b: [a_hi][a_lo][w_hi][w_lo], TILE_B each
    const int nt = N / BN, mt = M / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* base = smem + buf * 4 * TILE_B;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[2][2], fw[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[k][i] = *reinterpret_cast<const bf16x8*>(base + k * TILE_B + lds_off(wm * 64 + i * 32 + l31, 2 * s + h));
                    fw[k][i] = *reinterpret_cast<const bf16x8*>(base + (2 + k) * TILE_B + lds_off(wn * 64 + i * 32 + l31, 2 * s + h));
                }
#pragma unroll
            for (int p = 2; p >= 0; --p) {
                const int ka = p == 2 ? 1 : 0, kw = p == 1 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ka][i], fw[kw][j], acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (V == 0) {
        // thread -> rows wr + 64 i (i = 0, 1), 16-B chunk wc, for each of the 4 operand-term tiles
        const int wc = t & 3, wr = t >> 2;
        size_t ao[2], wo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { ao[i] = (size_t)(m0 + wr + 64 * i) * K + wc * 8; wo[i] = (size_t)(n0 + wr + 64 * i) * K + wc * 8; }
        const size_t aplane = (size_t)M * K, wplane = (size_t)N * K;
        u32x4 r[2][4][2];
        auto gload = [&](auto SET, int kt) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                r[st][0][i] = *reinterpret_cast<const u32x4*>(Ap + ao[i] + kt * BK);
                r[st][1][i] = *reinterpret_cast<const u32x4*>(Ap + aplane + ao[i] + kt * BK);
                r[st][2][i] = *reinterpret_cast<const u32x4*>(Wp + wo[i] + kt * BK);
                r[st][3][i] = *reinterpret_cast<const u32x4*>(Wp + wplane + wo[i] + kt * BK);
            }
        };
        auto lstore = [&](auto SET, int buf) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value;
            char* base = smem + buf * 4 * TILE_B;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(base + k * TILE_B + lds_off(wr + 64 * i, wc)) = r[st][k][i];
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        gload(S0{}, 0);
        if (nk > 1) gload(S1{}, 1);
        lstore(S0{}, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 < nk) gload(S0{}, kt + 2);
            compute(0);
            if (kt + 1 < nk) lstore(S1{}, 1);
            __syncthreads();
            if (kt + 1 >= nk) break;
            if (kt + 3 < nk) gload(S1{}, kt + 3);
            compute(1);
            if (kt + 2 < nk) lstore(S0{}, 0);
            __syncthreads();
        }
    } else {
        // direct-to-LDS: one wave instruction writes 1 KB = 16 rows x 64 B contiguously (lane l -> base + 16 l).
        // 4 operand-term tiles x 8 KB = 32 instructions per K tile per block = 8 per wave: wave w fills tile w
        // (a_hi, a_lo, w_hi, w_lo), 8 instructions of 16 rows.  The swizzle is applied on the SOURCE side: lane l
        // lands in row l/4, physical chunk l%4, so it fetches logical chunk (l%4) ^ ((row >> 2) & 3).
        const int lrow = lane >> 2, pc = lane & 3;
        const __bf16* src;
        {
            const size_t plane = (wave & 1) ? ((wave < 2) ? (size_t)M * K : (size_t)N * K) : 0;
            src = (wave < 2 ? Ap : Wp) + plane + (size_t)((wave < 2 ? m0 : n0)) * K;
        }
        auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
            char* dst = smem + buf * 4 * TILE_B + wave * TILE_B;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int row = g * 16 + lrow;
                const int c = pc ^ ((row >> 2) & 3);
                const __bf16* gp = src + (size_t)row * K + kt * BK + c * 8;
                __builtin_amdgcn_global_load_lds((const void*)gp, (__attribute__((address_space(3))) void*)(dst + g * 1024), 16, 0, 0);
            }
        };
        // ring of NBUF buffers, NBUF - 1 tiles in flight
#pragma unroll
        for (int s = 0; s < NBUF - 1; ++s)
            if (s < nk) issue(s, s);
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt must have landed: at most NBUF - 2 younger groups of 8 loads may still be in flight
            if (kt + NBUF - 2 < nk) {
                if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (NBUF == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                       // everyone's part of tile kt is in LDS; tile kt-1's buffer is free
            if (kt + NBUF - 1 < nk) issue(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
            compute(kt % NBUF);
        }
    }

    // epilogue: form all values, then store back to back (M, N multiples of the tile here)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = m0 + wm * 64 + i * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * N + col] = acc[i][j][r] + bv;
        }
    }
}

static unsigned short f2bf_rn(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int V, int NBUF>
float run(const char* name, const __bf16* Ap, const __bf16* Wp, float* C, const float* bias, int M, int N, int K, int iters) {
    const int lds = (V == 0 ? 2 : NBUF) * 4 * TILE_B;
    auto k = gemm_planes<V, NBUF>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = (M / BM) * (N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, Ap, Wp, C, bias, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, Ap, Wp, C, bias, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-26s M=%6d N=%5d K=%5d lds=%6d  %8.3f ms %7.1f TF(algorithmic)\n", name, M, N, K, lds, ms, 2.0 * M * N * K / ms / 1e9);
    return ms;
}

int main() {
    const int MMAX = 73856 / 128 * 128, NMAX = 3072, KMAX = 3072;
    // correctness on a small problem against float64
    {
        const int M = 256, N = 256, K = 768;
        std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N, 0.f);
        for (size_t i = 0; i < A.size(); ++i) A[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
        for (size_t i = 0; i < W.size(); ++i) W[i] = ((float)((i * 40503u + 7) % 1999) / 1000.f - 1.f) * 0.05f;
        std::vector<unsigned short> Ap(2 * A.size()), Wp(2 * W.size());
        for (size_t i = 0; i < A.size(); ++i) { unsigned short hi = f2bf_rn(A[i]); Ap[i] = hi; Ap[A.size() + i] = f2bf_rn(A[i] - bf2f(hi)); }
        for (size_t i = 0; i < W.size(); ++i) { unsigned short hi = f2bf_rn(W[i]); Wp[i] = hi; Wp[W.size() + i] = f2bf_rn(W[i] - bf2f(hi)); }
        __bf16 *dA, *dW; float *dC, *db;
        CK(hipMalloc(&dA, Ap.size() * 2)); CK(hipMalloc(&dW, Wp.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&db, N * 4));
        CK(hipMemcpy(dA, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(db, 0, N * 4));
        std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0, g = 0; for (int k = 0; k < K; ++k) { double p = (double)A[(size_t)m * K + k] * W[(size_t)n * K + k]; s += p; g += fabs(p); } ref[(size_t)m * N + n] = s; mag[(size_t)m * N + n] = g; }
        std::vector<float> out((size_t)M * N);
        auto check = [&](const char* nm) {
            CK(hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost));
            double maxr = 0; for (size_t i = 0; i < out.size(); ++i) maxr = fmax(maxr, fabs(out[i] - ref[i]) / mag[i]);
            printf("  %-12s max err / sum|a w| = %.3e\n", nm, maxr);
        };
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4)); run<0, 2>("check V0", dA, dW, dC, db, M, N, K, 1); check("V0");
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4)); run<1, 3>("check V1 ring3", dA, dW, dC, db, M, N, K, 1); check("V1/3");
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4)); run<1, 4>("check V1 ring4", dA, dW, dC, db, M, N, K, 1); check("V1/4");
    }
    __bf16 *Ap, *Wp; float *C, *bias;
    CK(hipMalloc(&Ap, (size_t)2 * MMAX * KMAX * 2)); CK(hipMalloc(&Wp, (size_t)2 * NMAX * KMAX * 2));
    CK(hipMalloc(&C, (size_t)MMAX * NMAX * 4)); CK(hipMalloc(&bias, NMAX * 4));
    CK(hipMemset(Ap, 0x3c, (size_t)2 * MMAX * KMAX * 2)); CK(hipMemset(Wp, 0x3c, (size_t)2 * NMAX * KMAX * 2)); CK(hipMemset(bias, 0, NMAX * 4));
    struct Shape { int M, N, K; } shapes[] = {{73856, 2304, 768}, {73856, 768, 768}, {73856, 3072, 768}, {73856, 768, 3072}, {36864, 3072, 768}, {9216, 3072, 768}};
    for (auto s : shapes) {
        const int M = s.M / 128 * 128;
        run<0, 2>("V0 regs, prefetch 2", Ap, Wp, C, bias, M, s.N, s.K, 8);
        run<1, 2>("V1 LDS-DMA ring 2", Ap, Wp, C, bias, M, s.N, s.K, 8);
        run<1, 3>("V1 LDS-DMA ring 3", Ap, Wp, C, bias, M, s.N, s.K, 8);
        run<1, 4>("V1 LDS-DMA ring 4", Ap, Wp, C, bias, M, s.N, s.K, 8);
        printf("\n");
    }
    return 0;
}
