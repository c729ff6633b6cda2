// Kernel lab (not part of the library): variants of the fp32 MFMA GEMM main loop, timed standalone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8; int xcd = bid % nx, idx = bid / nx; int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; return base + idx;
}

// pure MFMA ceiling
__global__ __launch_bounds__(256) void mfma_only(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// WM x WN waves, each wave TM x TN tiles of 32x32; BM = WM*TM*32, BN = WN*TN*32
template <int WM, int WN, int TM, int TN, int BK, int MINW, bool PRIO, int GM = 0, int EPI = 0, int STAG = 0>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_var(const float* __restrict__ A, const float* __restrict__ W,
                                                               float* __restrict__ C, const float* __restrict__ bias,
                                                               int M, int N, int K) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LD = BK + 4, NT = WM * WN * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 2 * BM * LD;
    const int nt = N / BN, mt = (M + BM - 1) / BM;
    if (STAG > 0 && blockIdx.x < 512) {
        // first-wave blocks start in STAG different phases of one tile time (~ nk * 8192 cycles at 2 blocks/CU)
        const int phase = (blockIdx.x / 8) % STAG;            // blockIdx/8 = index within the XCD
        const int sleeps = phase * (K / BK) / STAG;           // one s_sleep(127) ~ 8128 cycles
        for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    int tm, tn;
    if (GM == 0) { tm = tile / nt; tn = tile % nt; }
    else {
        const int per = GM * nt, grp = tile / per, loc = tile - grp * per;
        const int rows = (mt - grp * GM) < GM ? (mt - grp * GM) : GM;      // last group may be short
        tn = loc / rows; tm = grp * GM + loc % rows;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, h = lane >> 5;
    constexpr int F4R = BK / 4;                 // float4 per row
    constexpr int RPP = NT / F4R;               // rows per pass
    constexpr int NA = BM / RPP, NB = BN / RPP;
    const int c4 = t % F4R, r0 = t / F4R;
    const float* ap[NA]; const float* bp[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + RPP * i; ar = ar < M ? ar : M - 1; ap[i] = A + (size_t)ar * K + c4 * 4; }
#pragma unroll
    for (int i = 0; i < NB; ++i) bp[i] = W + (size_t)(n0 + r0 + RPP * i) * K + c4 * 4;
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
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + RPP * i) * LD + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + RPP * i) * LD + c4 * 4]) = rb[i];
    __syncthreads();
    const int nk = K / BK; int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + (kt + 1) * BK);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + (kt + 1) * BK);
        }
        const float* Ac = As + cur * BM * LD + (wm * TM * 32 + l31) * LD + h * 4;
        const float* Bc = Bs + cur * BN * LD + (wn * TN * 32 + l31) * LD + h * 4;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LD + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (more) {
            float* Aw = As + (cur ^ 1) * BM * LD; float* Bw = Bs + (cur ^ 1) * BN * LD;
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&Aw[(r0 + RPP * i) * LD + c4 * 4]) = ra[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bw[(r0 + RPP * i) * LD + c4 * 4]) = rb[i];
        }
        __syncthreads();
        cur ^= 1;
    }
    if (EPI == 1) {      // ablation: no epilogue stores (one conditional store keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 123.456f) C[0] = sum;
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r] + bv;
            }
    }
}


// persistent variant: gridDim = slots; each block walks tiles; next tile's first K-slab is loaded
// before the current tile's epilogue stores.
template <int TM, int TN, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_persist(const float* __restrict__ A, const float* __restrict__ W,
                                                          float* __restrict__ C, const float* __restrict__ bias,
                                                          int M, int N, int K) {
    constexpr int BK = 32, BM = 2 * TM * 32, BN = 2 * TN * 32, LD = BK + 4, NA = BM / 32, NB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 2 * BM * LD;
    const int nt = N / BN, mt = (M + BM - 1) / BM, ntiles = mt * nt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    // XCD-contiguous static schedule: block b (XCD b%8) takes tiles chunk_start + (b/8) + i*(G/8)
    const int G = gridDim.x, xcd = blockIdx.x % 8, lb = blockIdx.x / 8, gpx = G / 8;
    const int q = ntiles / 8, r = ntiles % 8;
    const int cstart = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int clen = (xcd < r) ? q + 1 : q;
    const float* ap[NA]; const float* bp[NB];
    f32x4 ra[NA], rb[NB];
    auto setup = [&](int tile, int& m0, int& n0) {
        m0 = (tile / nt) * BM; n0 = (tile % nt) * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; ap[i] = A + (size_t)ar * K + c4 * 4; }
#pragma unroll
        for (int i = 0; i < NB; ++i) bp[i] = W + (size_t)(n0 + r0 + 32 * i) * K + c4 * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i]);
    };
    int li = lb;
    if (li >= clen) return;
    int m0, n0;
    setup(cstart + li, m0, n0);
    const int nk = K / BK;
    int cur = 0;
    while (true) {
        // stage slab 0 of this tile (registers already hold it)
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[cur * BM * LD + (r0 + 32 * i) * LD + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[cur * BN * LD + (r0 + 32 * i) * LD + c4 * 4]) = rb[i];
        __syncthreads();
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
        const int cm0 = m0, cn0 = n0;
        const int next_li = li + gpx;
        const bool has_next = next_li < clen;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + (kt + 1) * BK);
#pragma unroll
                for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + (kt + 1) * BK);
            } else if (has_next) {
                setup(cstart + next_li, m0, n0);          // prefetch next tile's slab 0 under the last MFMAs
            }
            const float* Ac = As + cur * BM * LD + (wm * TM * 32 + l31) * LD + h * 4;
            const float* Bc = Bs + cur * BN * LD + (wn * TN * 32 + l31) * LD + h * 4;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LD + kk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LD + kk * 8);
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
#pragma unroll
                for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[(cur ^ 1) * BM * LD + (r0 + 32 * i) * LD + c4 * 4]) = ra[i];
#pragma unroll
                for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[(cur ^ 1) * BN * LD + (r0 + 32 * i) * LD + c4 * 4]) = rb[i];
                __syncthreads();
                cur ^= 1;
            }
        }
        // epilogue of the finished tile (next tile's loads are in flight)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = cn0 + wn * TN * 32 + j * 32 + l31;
            const float bv = bias[col];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int row = cm0 + wm * TM * 32 + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                    if (row < M) C[(size_t)row * N + col] = acc[i][j][rr] + bv;
                }
        }
        if (!has_next) break;
        li = next_li;
        cur ^= 1;        // the other buffer is free: every wave passed the barrier before the last compute
        __syncthreads(); // all waves done reading the last slab before it is overwritten two tiles later
    }
}


// persistent + DEFERRED epilogue: the finished tile's accumulators are copied to a second register set
// and stored 4 values per lane per K iteration of the NEXT tile (16 groups of 4 = 64 values, needs
// nk >= 16), so the output writes are spread evenly over time instead of arriving as one 64 KB burst per
// block (bursts that the whole chip issues in phase, stalling the partner blocks' prefetch loads).
__global__ __launch_bounds__(256, 2) void gemm_persist_defer(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ C, const float* __restrict__ bias,
                                                             int M, int N, int K) {
    constexpr int TM = 2, TN = 2, BK = 32, BM = 128, BN = 128, LD = BK + 4, NA = BM / 32, NB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 2 * BM * LD;
    const int nt = N / BN, mt = (M + BM - 1) / BM, ntiles = mt * nt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int c4 = t & 7, r0 = t >> 3;
    const int G = gridDim.x, xcd = blockIdx.x % 8, lb = blockIdx.x / 8, gpx = G / 8;
    const int q = ntiles / 8, r = ntiles % 8;
    const int cstart = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int clen = (xcd < r) ? q + 1 : q;
    const float* ap[NA]; const float* bp[NB];
    f32x4 ra[NA], rb[NB];
    auto setup = [&](int tile, int& m0, int& n0) {
        m0 = (tile / nt) * BM; n0 = (tile % nt) * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) { int ar = m0 + r0 + 32 * i; ar = ar < M ? ar : M - 1; ap[i] = A + (size_t)ar * K + c4 * 4; }
#pragma unroll
        for (int i = 0; i < NB; ++i) bp[i] = W + (size_t)(n0 + r0 + 32 * i) * K + c4 * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i]);
    };
    int li = lb;
    if (li >= clen) return;
    int m0, n0;
    setup(cstart + li, m0, n0);
    const int nk = K / BK;
    int cur = 0;
    f32x16 accp[TM][TN];
    int pm0 = 0, pn0 = 0;
    bool have_prev = false;
    // one group = 4 consecutive accumulator registers of one MFMA tile (4 output rows at one column)
    auto store_group = [&](auto GI) __attribute__((always_inline)) {
        constexpr int g = decltype(GI)::value;
        constexpr int i = g >> 3, j = (g >> 2) & 1, rq = g & 3;
        const int col = pn0 + wn * TN * 32 + j * 32 + l31;
        const float bv = bias[col];
        const int row0 = pm0 + wm * TM * 32 + i * 32 + 8 * rq + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (row0 + e < M) C[(size_t)(row0 + e) * N + col] = accp[i][j][rq * 4 + e] + bv;
    };
    auto store_switch = [&](int g) __attribute__((always_inline)) {
        switch (g) {
#define SG(n) case n: store_group(std::integral_constant<int, n>{}); break;
            SG(0) SG(1) SG(2) SG(3) SG(4) SG(5) SG(6) SG(7) SG(8) SG(9) SG(10) SG(11) SG(12) SG(13) SG(14) SG(15)
#undef SG
            default: break;
        }
    };
    while (true) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[cur * BM * LD + (r0 + 32 * i) * LD + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[cur * BN * LD + (r0 + 32 * i) * LD + c4 * 4]) = rb[i];
        __syncthreads();
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
        const int cm0 = m0, cn0 = n0;
        const int next_li = li + gpx;
        const bool has_next = next_li < clen;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + (kt + 1) * BK);
#pragma unroll
                for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + (kt + 1) * BK);
            } else if (has_next) {
                setup(cstart + next_li, m0, n0);
            }
            if (have_prev && kt < 16) store_switch(kt);
            const float* Ac = As + cur * BM * LD + (wm * TM * 32 + l31) * LD + h * 4;
            const float* Bc = Bs + cur * BN * LD + (wn * TN * 32 + l31) * LD + h * 4;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LD + kk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LD + kk * 8);
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
#pragma unroll
                for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[(cur ^ 1) * BM * LD + (r0 + 32 * i) * LD + c4 * 4]) = ra[i];
#pragma unroll
                for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(&Bs[(cur ^ 1) * BN * LD + (r0 + 32 * i) * LD + c4 * 4]) = rb[i];
                __syncthreads();
                cur ^= 1;
            }
        }
        // hand the finished tile to the deferred-store set
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) accp[i][j] = acc[i][j];
        pm0 = cm0; pn0 = cn0; have_prev = true;
        if (!has_next) break;
        li = next_li;
        cur ^= 1;
        __syncthreads();
    }
    // flush the last tile
#pragma unroll
    for (int g = 0; g < 16; ++g) store_switch(g);
}

float run_defer(const char* name, const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int iters) {
    const int lds = 2 * (128 + 128) * 36 * 4;
    auto k = gemm_persist_defer;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, W, C, bias, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, W, C, bias, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-28s M=%6d N=%5d K=%5d lds=%6d grid=%5d  %8.3f ms %7.1f TF\n", name, M, N, K, lds, grid, ms, 2.0 * M * N * K / ms / 1e9);
    return ms;
}

template <int TM, int TN, int MINW>
float run_persist(const char* name, int bpc, const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int iters) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, LD = 36;
    const int lds = 2 * (BM + BN) * LD * 4;
    auto k = gemm_persist<TM, TN, MINW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = 256 * bpc;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, W, C, bias, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, W, C, bias, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-28s M=%6d N=%5d K=%5d lds=%6d grid=%5d  %8.3f ms %7.1f TF\n", name, M, N, K, lds, grid, ms, 2.0 * M * N * K / ms / 1e9);
    return ms;
}

template <int WM, int WN, int TM, int TN, int BK, int MINW, bool PRIO, int GM = 0, int EPI = 0, int STAG = 0>
float run(const char* name, const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int iters) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LD = BK + 4;
    const int lds = 2 * (BM + BN) * LD * 4;
    auto k = gemm_var<WM, WN, TM, TN, BK, MINW, PRIO, GM, EPI, STAG>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + BM - 1) / BM) * (N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(WM * WN * 64), lds, 0, A, W, C, bias, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(WM * WN * 64), lds, 0, A, W, C, bias, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%-28s M=%6d N=%5d K=%5d lds=%6d nwg=%5d  %8.3f ms %7.1f TF\n", name, M, N, K, lds, nwg, ms, 2.0 * M * N * K / ms / 1e9);
    return ms;
}

int main(int argc, char** argv) {
    const int MMAX = 64 * 577, NMAX = 3072, KMAX = 3072;
    float *A, *W, *C, *bias;
    CK(hipMalloc(&A, (size_t)MMAX * KMAX * 4)); CK(hipMalloc(&W, (size_t)NMAX * KMAX * 4));
    CK(hipMalloc(&C, (size_t)MMAX * NMAX * 4)); CK(hipMalloc(&bias, NMAX * 4));
    std::vector<float> h((size_t)MMAX * KMAX);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)NMAX * KMAX * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, NMAX * 4));
    {   // ceiling
        float* o; CK(hipMalloc(&o, 256 * 8 * 256 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int wpb = 1; wpb <= 2; ++wpb) {
            const int iters = 20000, blocks = 256 * wpb;
            hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(256), 0, 0, o, 100);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(256), 0, 0, o, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mfma_only %d blocks/CU: %.3f ms  %.1f TF\n", wpb, ms, (double)blocks * 4 * iters * 4 * 4096.0 / ms / 1e9);
        }
    }
    if (argc > 1 && !strcmp(argv[1], "overhead")) {
        // where does the per-tile overhead go?  time = a*K + b per shape, with and without epilogue stores
        const int Ms[] = {36928, 73856 / 2}, Ns[] = {768, 3072};
        for (int N : Ns)
            for (int K : {256, 512, 768, 1536, 3072}) {
                const int M = 36928;
                float t0 = run<2, 2, 2, 2, 32, 2, true, 0, 0>("128x128 plain", A, W, C, bias, M, N, K, 10);
                float t1 = run<2, 2, 2, 2, 32, 2, true, 0, 1>("128x128 no-epilogue", A, W, C, bias, M, N, K, 10);
                float t2 = run_persist<2, 2, 2>("128x128 persist", 2, A, W, C, bias, M, N, K, 10);
                float t3 = K >= 512 ? run_defer("128x128 persist+deferred", A, W, C, bias, M, N, K, 10) : 0.f;
                float t4 = run<2, 2, 2, 2, 32, 2, true, 0, 0, 2>("128x128 stagger2", A, W, C, bias, M, N, K, 10);
                float t5 = run<2, 2, 2, 2, 32, 2, true, 0, 0, 4>("128x128 stagger4", A, W, C, bias, M, N, K, 10);
                printf("   N=%d K=%d: plain %.3f  noepi %.3f  persist %.3f  defer %.3f  stag2 %.3f  stag4 %.3f ms\n\n", N, K, t0, t1, t2, t3, t4, t5);
            }
        (void)Ms;
        {   // correctness of the deferred-store kernel
            const int M = 3000, N = 768, K = 768;
            float* C2; CK(hipMalloc(&C2, (size_t)M * N * 4));
            CK(hipMemset(C2, 0xff, (size_t)M * N * 4));
            run<2, 2, 2, 2, 32, 2, true, 0>("ref", A, W, C, bias, M, N, K, 1);
            run_defer("defer", A, W, C2, bias, M, N, K, 1);
            std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
            CK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < h1.size(); ++i) md = fmax(md, fabs(h1[i] - h2[i]));
            printf("   defer vs ref max diff %g\n", md);
        }
        return 0;
    }
    // correctness check of the persistent kernel against the plain one
    {
        const int M = 3000, N = 768, K = 768;
        float* C2; CK(hipMalloc(&C2, (size_t)M * N * 4));
        run<2, 2, 2, 2, 32, 2, true, 0>("ref", A, W, C, bias, M, N, K, 1);
        run_persist<2, 2, 2>("persist", 2, A, W, C2, bias, M, N, K, 1);
        std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
        CK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < h1.size(); ++i) md = fmax(md, fabs(h1[i] - h2[i]));
        printf("persist vs ref max diff %g\n", md);
    }
    struct Shape { int M, N, K; } shapes[] = {{30016, 768, 768}, {30016, 2304, 768}, {30016, 3072, 768}, {30016, 768, 3072}, {25388, 768, 768}, {25388, 2304, 768}, {25388, 768, 3072}, {9232, 768, 768}, {9232, 3072, 768}};
    for (auto s : shapes) {
        run<2, 2, 2, 2, 32, 2, true, 0>("128x128 plain", A, W, C, bias, s.M, s.N, s.K, 10);
        run_persist<2, 2, 2>("128x128 persist 2/CU", 2, A, W, C, bias, s.M, s.N, s.K, 10);
        run<2, 2, 1, 2, 32, 2, true, 0>("64x128 plain", A, W, C, bias, s.M, s.N, s.K, 10);
        run_persist<1, 2, 2>("64x128 persist 2/CU", 2, A, W, C, bias, s.M, s.N, s.K, 10);
        run_persist<1, 1, 4>("64x64 persist 4/CU", 4, A, W, C, bias, s.M, s.N, s.K, 10);
        printf("\n");
    }
    return 0;
}
