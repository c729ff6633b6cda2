// Lab: fp32-accurate GEMM on the bf16 matrix pipe by exact operand splitting.
//   A (fp32 activations) = a0 + a1 + a2 exactly (three bf16 terms, truncation split)
//   W either bf16 (NW = 1: "bf16 weights", products exact) or fp32 pre-split into NW = 3 bf16 terms
//   C += sum over allowed (ta, tw) of mfma_f32_32x32x16_bf16(A_ta, W_tw)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8; int xcd = bid % nx, idx = bid / nx; int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; return base + idx;
}

// exact 3-way split of 4 floats into 3 x (4 bf16 packed in 2 dwords)
template <int NA>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[NA]) {
    unsigned t[NA][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = v[e];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const unsigned b = __float_as_uint(r) & 0xFFFF0000u;
            t[k][e] = b;
            r = r - __uint_as_float(b);
        }
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        out[k][0] = (t[k][0] >> 16) | t[k][1];
        out[k][1] = (t[k][2] >> 16) | t[k][3];
    }
}

// LDS tile: [rows][32 bf16] = 64 B per row, 16-B chunks XOR-swizzled by (row >> 2) & 3
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int NA, int NW, int MAXSUM>
__global__ __launch_bounds__(256, 2) void gemm_bf16s(const float* __restrict__ A, const __bf16* __restrict__ Wt /*[NW][N][K]*/,
                                                     float* __restrict__ C, const float* __restrict__ bias,
                                                     int M, int N, int K) {
    constexpr int BM = 128, BN = 128, BK = 32, TILE_B = 128 * 64;          // bytes per operand-term tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: buf b: [A terms NA][W terms NW], each TILE_B
    auto a_tile = [&](int buf, int term) { return smem + (size_t)(buf * (NA + NW) + term) * TILE_B; };
    auto w_tile = [&](int buf, int term) { return smem + (size_t)(buf * (NA + NW) + NA + term) * TILE_B; };
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    // A staging: thread -> rows r0 + 32 i, float4 column c4 (k = 4 c4 .. 4 c4 + 3)
    const int c4 = t & 7, r0 = t >> 3;
    const float* ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; ap[i] = A + (size_t)ar * K + c4 * 4; }
    // W staging: per term 512 16-B chunks; thread -> rows wr + 64 i, chunk wc
    const int wc = t & 3, wr = t >> 2;
    const __bf16* wp[NW][2];
#pragma unroll
    for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) wp[k][i] = Wt + ((size_t)k * N + n0 + wr + 64 * i) * K + wc * 8;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4]; u32x4 rw[NW][2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i) rw[k][i] = *reinterpret_cast<const u32x4*>(wp[k][i] + kt * BK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 s[NA];
            split4<NA>(ra[i], s);
            const int row = r0 + 32 * i;
#pragma unroll
            for (int k = 0; k < NA; ++k)
                *reinterpret_cast<u32x2*>(a_tile(buf, k) + lds_off(row, c4 >> 1) + (c4 & 1) * 8) = s[k];
        }
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(w_tile(buf, k) + lds_off(wr + 64 * i, wc)) = rw[k][i];
    };
    gload(0); lstore(0);
    __syncthreads();
    const int nk = K / BK; int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[NA][2], fw[NW][2];
#pragma unroll
            for (int k = 0; k < NA; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[k][i] = *reinterpret_cast<const bf16x8*>(a_tile(cur, k) + lds_off(wm * 64 + i * 32 + l31, 2 * s + h));
#pragma unroll
            for (int k = 0; k < NW; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fw[k][j] = *reinterpret_cast<const bf16x8*>(w_tile(cur, k) + lds_off(wn * 64 + j * 32 + l31, 2 * s + h));
            // smallest terms first
#pragma unroll
            for (int sum = MAXSUM; sum >= 0; --sum)
#pragma unroll
                for (int ta = 0; ta < NA; ++ta) {
                    const int tw = sum - ta;
                    if (tw < 0 || tw >= NW) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta][i], fw[tw][j], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r] + bv;
            }
    }
}

static unsigned short f2bf_trunc(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int NA, int NW, int MAXSUM>
void run(const char* name, const float* A, const __bf16* Wt, float* C, const float* bias, int M, int N, int K, int iters) {
    const int lds = 2 * (NA + NW) * 128 * 64;
    auto k = gemm_bf16s<NA, NW, MAXSUM>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + 127) / 128) * (N / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wt, C, bias, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, A, Wt, C, bias, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-22s M=%6d N=%5d K=%5d lds=%6d  %8.3f ms %7.1f TF(algorithmic)\n", name, M, N, K, lds, ms, 2.0 * M * N * K / ms / 1e9);
}

int main() {
    const int M = 25388, NMAX = 3072, KMAX = 3072;
    std::vector<float> hA((size_t)M * KMAX), hW((size_t)NMAX * KMAX);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd() * 0.05f;
    float *A, *C, *bias; __bf16* Wt;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&C, (size_t)M * NMAX * 4)); CK(hipMalloc(&bias, NMAX * 4));
    CK(hipMalloc(&Wt, (size_t)3 * NMAX * KMAX * 2));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, NMAX * 4));
    // correctness on a small problem (M=256, N=128, K=768) against fp64
    {
        const int m = 256, n = 128, k = 768;
        std::vector<unsigned short> wt((size_t)3 * n * k);
        for (int i = 0; i < n * k; ++i) {
            float r = hW[i];
            for (int t = 0; t < 3; ++t) { unsigned short b = f2bf_trunc(r); wt[(size_t)t * n * k + i] = b; r -= bf2f(b); }
        }
        CK(hipMemcpy(Wt, wt.data(), wt.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> hC((size_t)m * n);
        auto check = [&](const char* nm, bool bf16w) {
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, maxr = 0;
            for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
                double ref = 0, mag = 0;
                for (int kk = 0; kk < k; ++kk) {
                    double w = bf16w ? (double)bf2f(wt[(size_t)j * k + kk]) : (double)hW[(size_t)j * k + kk];
                    ref += (double)hA[(size_t)i * k + kk] * w; mag += fabs((double)hA[(size_t)i * k + kk] * w);
                }
                maxe = fmax(maxe, fabs(hC[(size_t)i * n + j] - ref)); maxr = fmax(maxr, fabs(hC[(size_t)i * n + j] - ref) / mag);
            }
            printf("  %-18s max abs err %.3e   max err / sum|a w| %.3e\n", nm, maxe, maxr);
        };
        // NOTE: A rows use stride K=768 here: build a compact copy
        float* A2; CK(hipMalloc(&A2, (size_t)m * k * 4));
        CK(hipMemcpy2D(A2, k * 4, A, k * 4, k * 4, m, hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL((gemm_bf16s<3, 3, 4>), dim3(2), dim3(256), 2 * 6 * 8192, 0, A2, Wt, C, bias, m, n, k); CK(hipDeviceSynchronize()); check("bf16x9", false);
        hipLaunchKernelGGL((gemm_bf16s<3, 3, 2>), dim3(2), dim3(256), 2 * 6 * 8192, 0, A2, Wt, C, bias, m, n, k); CK(hipDeviceSynchronize()); check("bf16x6", false);
        hipLaunchKernelGGL((gemm_bf16s<3, 1, 2>), dim3(2), dim3(256), 2 * 4 * 8192, 0, A2, Wt, C, bias, m, n, k); CK(hipDeviceSynchronize()); check("bf16w x3 (vs bf16 W)", true);
        hipLaunchKernelGGL((gemm_bf16s<1, 1, 0>), dim3(2), dim3(256), 2 * 2 * 8192, 0, A2, Wt, C, bias, m, n, k); CK(hipDeviceSynchronize()); check("plain bf16 x1", true);
    }
    struct Shape { int N, K; } shapes[] = {{768, 768}, {2304, 768}, {3072, 768}, {768, 3072}};
    for (auto sh : shapes) {
        run<3, 3, 4>("bf16x9", A, Wt, C, bias, M, sh.N, sh.K, 10);
        run<3, 3, 2>("bf16x6", A, Wt, C, bias, M, sh.N, sh.K, 10);
        run<3, 1, 2>("bf16 weights (x3)", A, Wt, C, bias, M, sh.N, sh.K, 10);
        run<1, 1, 0>("plain bf16 (x1)", A, Wt, C, bias, M, sh.N, sh.K, 10);
        printf("\n");
    }
    return 0;
}
