// T* searcher state on the device: frame-score aggregation, temporal-window spread,
// sampling-distribution update, percentile mask and weighted-draw support
// (S5-S7, S1, S10 of SURVEY.md 8a), all float64 like the reference's numpy.
//
// Reference: /root/reference/TStar/interface_searcher.py
//   :309-311 score write-back            -> apply_grid_kernel
//   :215-241 update_top_25_with_window   -> apply_grid_kernel (percentile of the g*g
//                                           confidences + the order-dependent spread,
//                                           run by ONE lane in draw order)
//   :260-261 visited (x, y) extraction   -> apply_grid_kernel (ordered compaction)
//   :266-274 spline evaluation, clamp, sigmoid, normalise -> distribution_kernel
//            (FITPACK splev/fpbspl restated; the FITPACK *fit* stays on the host)
//   :345-352 sampler weights             -> sampler_prep_kernel (np.percentile via
//                                           radix select, mask, fallback, normalise)
//   :353-358, :369-372 np.random.choice  -> cdf_kernel / draw_kernel / exclude
//            (numpy legacy algorithm: sequential cumsum, searchsorted-right; the
//            MT19937 draws and the de-dup loop stay on the host)
//
// These kernels touch <= 8*N bytes per array (29-115 KB): they are launch-latency
// bound, not HBM bound; one workgroup each.  Compiled with -ffp-contract=off so the
// float64 arithmetic is the same sequence of IEEE operations numpy/FITPACK perform;
// np.sum's pairwise order is reproduced by a host-built reduction program.
#include "../../include/tstar_hip.h"
#include "common.h"
#include <math.h>
#include <vector>

namespace tstar {

constexpr int ST = 1024;   // threads per searcher workgroup

struct SumProgram {        // numpy pairwise_sum order for length-N arrays
    int n_leaf = 0, n_ops = 0;
    int *d_leaf_off = nullptr, *d_leaf_len = nullptr;   // leaves of <= 128 elements
    signed char* d_ops = nullptr;                       // 0 = push next leaf, 1 = add top two
};

// numpy/_core/src/umath/loops_utils.h.src pairwise sum: n < 8 sequential; n <= 128
// 8-way unrolled; else split at n/2 rounded down to a multiple of 8.
static void build_pairwise(int off, int n, std::vector<int>& lo, std::vector<int>& ll, std::vector<signed char>& ops) {
    if (n <= 128) { lo.push_back(off); ll.push_back(n); ops.push_back(0); return; }
    int n2 = n / 2;
    n2 -= n2 % 8;
    build_pairwise(off, n2, lo, ll, ops);
    build_pairwise(off + n2, n - n2, lo, ll, ops);
    ops.push_back(1);
}
// np.add.reduce hands the inner loop at most one ufunc buffer (np.getbufsize() = 8192 elements, numpy 1.26 and 2.x) at
// a time and accumulates the chunks left to right: a.sum() = (pw(a[0:8192]) + pw(a[8192:16384])) + ...  For N <= 8192 that
// is the plain pairwise sum; beyond it (the 4-hour videos of BASELINE configs[4]) the chunking changes the last bits.
constexpr int NP_BUFSIZE = 8192;
static void build_program(int off, int n, std::vector<int>& lo, std::vector<int>& ll, std::vector<signed char>& ops) {
    for (int c = 0; c < n; c += NP_BUFSIZE) {
        build_pairwise(off + c, n - c < NP_BUFSIZE ? n - c : NP_BUFSIZE, lo, ll, ops);
        if (c > 0) ops.push_back(1);
    }
}

__device__ double leaf_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.;   // numpy: res = 0.; for i: res += a[i]  (for n < 8; -0.0 aside)
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// block-wide: returns numpy's a.sum() to every thread; scratch: LDS double[n_leaf + 64]
__device__ double block_np_sum(const double* a, const SumProgram& sp, double* scratch) {
    __syncthreads();
    for (int l = threadIdx.x; l < sp.n_leaf; l += blockDim.x) scratch[l] = leaf_sum(a + sp.d_leaf_off[l], sp.d_leaf_len[l]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double* stack = scratch + sp.n_leaf;
        int top = 0, leaf = 0;
        for (int i = 0; i < sp.n_ops; ++i) {
            if (sp.d_ops[i] == 0) stack[top++] = scratch[leaf++];
            else { stack[top - 2] = stack[top - 2] + stack[top - 1]; --top; }
        }
        scratch[0] = stack[0];
    }
    __syncthreads();
    const double r = scratch[0];
    __syncthreads();
    return r;
}

// np.percentile(v, 75) lerp of the two neighbouring order statistics
__device__ __forceinline__ double lerp75(double a, double b, double g) {
    const double d = b - a;
    return g >= 0.5 ? b - d * (1 - g) : a + d * g;
}

// ------------------------------------------------------------------ apply grid
__global__ __launch_bounds__(ST) void apply_grid_kernel(double* __restrict__ score, double* __restrict__ unvisited,
                                                        const int* __restrict__ secs, const double* __restrict__ conf,
                                                        int n, int N, int window, int* __restrict__ vis_x,
                                                        double* __restrict__ vis_y, int* __restrict__ n_vis, int use_lds,
                                                        int do_write, int do_compact) {
    __shared__ double s_lohi[2];
    __shared__ int s_cnt[ST / 64];
    extern __shared__ double s_dyn[];            // use_lds: [score N][conf n][secs n (int)]
    const int t = threadIdx.x;
    if (do_write)
        for (int i = t; i < n; i += ST) { unvisited[secs[i]] = 0.0; score[secs[i]] = conf[i]; }
    // order statistics of conf by rank counting (n <= a few hundred)
    const double vi = n * 0.75 + (1 + 0.75 * (1 - 1 - 1)) - 1;
    int lo = (int)floor(vi);
    const double g = vi - lo;
    lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
    const int hi = lo + 1 < n ? lo + 1 : n - 1;
    for (int i = t; i < n; i += ST) {
        const double c = conf[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double o = conf[j]; rank += (o < c) || (o == c && j < i); }
        if (rank == lo) s_lohi[0] = c;
        if (rank == hi) s_lohi[1] = c;
    }
    __syncthreads();
    // the spread is ONE lane walking the samples in draw order (order-dependent, in place); its working set is
    // staged in LDS when it fits, so each step costs an LDS access instead of a global round trip
    double* w_score = score;
    const double* w_conf = conf;
    const int* w_secs = secs;
    if (use_lds) {
        double* l_conf = s_dyn + N;
        int* l_secs = reinterpret_cast<int*>(l_conf + n);
        __threadfence_block();
        __syncthreads();
        for (int i = t; i < N; i += ST) s_dyn[i] = score[i];
        for (int i = t; i < n; i += ST) { l_conf[i] = conf[i]; l_secs[i] = secs[i]; }
        w_score = s_dyn; w_conf = l_conf; w_secs = l_secs;
    }
    __syncthreads();
    if (t == 0) {
        const double thr = lerp75(s_lohi[0], s_lohi[1], g);
        for (int i = 0; i < n; ++i) {            // draw order, in place (order-dependent)
            if (w_conf[i] >= thr) {
                const int f = w_secs[i];
                for (int off = -window; off <= window; ++off) {
                    const int j = f + off;
                    if (j >= 0 && j < N) {
                        const double v = w_score[f] / (double)((off < 0 ? -off : off) + 1);
                        if (v > w_score[j]) w_score[j] = v;          // Python max(score[j], v)
                    }
                }
            }
        }
    }
    __syncthreads();
    if (use_lds) {
        for (int i = t; i < N; i += ST) score[i] = s_dyn[i];
        __threadfence_block();
        __syncthreads();
    }
    if (!do_compact) return;
    // ordered compaction of the visited frames
    const int per = (N + ST - 1) / ST;
    const int b0 = t * per, b1 = (b0 + per < N) ? b0 + per : N;
    int cnt = 0;
    for (int i = b0; i < b1; ++i) cnt += unvisited[i] == 0.0;
    // exclusive prefix of the per-thread counts: shuffle scan inside each wave, 16 wave totals through LDS
    const int lane = t & 63, wv = t >> 6;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) s_cnt[wv] = incl;
    __syncthreads();
    if (t == 0) { int acc = 0; for (int i = 0; i < ST / 64; ++i) { const int c = s_cnt[i]; s_cnt[i] = acc; acc += c; } *n_vis = acc; }
    __syncthreads();
    int o = s_cnt[wv] + incl - cnt;
    for (int i = b0; i < b1; ++i)
        if (unvisited[i] == 0.0) { vis_x[o] = i; vis_y[o] = score[i]; ++o; }
}

// ------------------------------------------------------------------ spline -> P
// FITPACK splev (ext = 0) + fpbspl for degree k <= 5, evaluated at x = 0..N-1.
__device__ double splev_at(const double* __restrict__ t, const double* __restrict__ c, int n, int k, double x) {
    const int k1 = k + 1, nk1 = n - k1;
    // largest l (1-based) in [k1, nk1] with t(l) <= x, else k1: binary search
    int lo_ = k1, hi_ = nk1;
    while (lo_ < hi_) {
        const int mid = (lo_ + hi_ + 1) >> 1;
        if (t[mid - 1] <= x) lo_ = mid; else hi_ = mid - 1;
    }
    const int l = lo_;
    double h[6], hh[5];
    h[0] = 1.0;
    for (int j = 1; j <= k; ++j) {
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; ++i) {
            const int li = l + i, lj = li - j;
            if (t[li - 1] == t[lj - 1]) { h[i] = 0.0; continue; }
            const double f = hh[i - 1] / (t[li - 1] - t[lj - 1]);
            h[i - 1] = h[i - 1] + f * (t[li - 1] - x);
            h[i] = f * (x - t[lj - 1]);
        }
    }
    double sp = 0.0;
    const int ll = l - k1;
    for (int j = 1; j <= k1; ++j) sp = sp + c[ll + j - 1] * h[j - 1];
    return sp;
}

__global__ __launch_bounds__(ST) void distribution_kernel(double* __restrict__ P, const double* __restrict__ t,
                                                          const double* __restrict__ c, int nknots, int k, int N,
                                                          SumProgram sp) {
    extern __shared__ double scratch[];
    const double floor_v = 1.0 / (double)N;
    for (int i = threadIdx.x; i < N; i += ST) {
        const double y = splev_at(t, c, nknots, k, (double)i);
        const double adj = y > floor_v ? y : floor_v;       // np.maximum(1/N, y) (NaN-free inputs)
        P[i] = 1.0 / (1.0 + exp(-adj));
    }
    const double s = block_np_sum(P, sp, scratch);
    for (int i = threadIdx.x; i < N; i += ST) P[i] = P[i] / s;
}

__global__ __launch_bounds__(ST) void fill_kernel(double* __restrict__ a, double v, int N) {
    for (int i = threadIdx.x; i < N; i += ST) a[i] = v;
}

// ------------------------------------------------------------------ k-th order statistic (radix select)
// values are >= 0 doubles: the bit pattern is monotone.  Returns the element of rank k (0-based).
__device__ double block_select(const double* a, int N, int k, unsigned* hist /*LDS [256]*/, unsigned long long* pref /*LDS [2]*/) {
    __shared__ unsigned s_wtot[4];
    unsigned long long prefix = 0, mask = 0;
    int kk = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(a[i]);
            if ((b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        // the digit whose bin holds rank kk: prefix sums of the 256 bins by the first four waves (shuffle scan + wave totals); the
        // one-lane walk over the bins this replaces cost ~12 us per pass, ~100 of the kernel's 131 us
        unsigned v = 0, incl = 0;
        if (threadIdx.x < 256) {
            v = hist[threadIdx.x];
            incl = v;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(incl, d); if (lane >= d) incl += u; }
            if (lane == 63) s_wtot[threadIdx.x >> 6] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            unsigned base = 0;
            for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_wtot[w];
            const unsigned excl = base + incl - v;
            if ((unsigned)kk >= excl && (unsigned)kk < excl + v) {            // exactly one bin
                pref[0] = prefix | ((unsigned long long)threadIdx.x << shift);
                pref[1] = (unsigned long long)((unsigned)kk - excl);
            }
        }
        __syncthreads();
        prefix = pref[0];
        kk = (int)pref[1];
        mask |= 0xFFULL << shift;
        __syncthreads();
    }
    return __longlong_as_double((long long)prefix);
}

// cdf = cumsum(p) in numpy's add.accumulate order (one sequential chain of f64 adds -- a parallel
// scan would round differently); cdf /= cdf[-1].  The chain runs on lane 0 over LDS-staged chunks so
// each step costs an f64 add, not a global-memory round trip (0.43 ms -> ~20 us at N = 3600).
constexpr int CDF_CHUNK = 2048;
__device__ void block_cdf(const double* p, double* cdf, int N) {
    __shared__ double s_buf[CDF_CHUNK];
    __shared__ double s_carry;
    __syncthreads();
    for (int base = 0; base < N; base += CDF_CHUNK) {
        const int n = (N - base) < CDF_CHUNK ? (N - base) : CDF_CHUNK;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s_buf[i] = p[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            double acc;
            int i = 0;
            if (base == 0) { acc = s_buf[0]; i = 1; } else acc = s_carry;
            // 16 values per step through registers: the LDS reads of a step are independent of the add chain, so
            // only the f64 add latency is serial (a read-add-write loop pays the LDS latency per element: 8x slower)
            for (; i + 16 <= n; i += 16) {
                double r[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) r[j] = s_buf[i + j];
#pragma unroll
                for (int j = 0; j < 16; ++j) { acc = acc + r[j]; r[j] = acc; }
#pragma unroll
                for (int j = 0; j < 16; ++j) s_buf[i + j] = r[j];
            }
            for (; i < n; ++i) { acc = acc + s_buf[i]; s_buf[i] = acc; }
            s_carry = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) cdf[base + i] = s_buf[i];
        __syncthreads();
    }
    const double last = s_carry;
    for (int i = threadIdx.x; i < N; i += blockDim.x) cdf[i] = cdf[i] / last;
    __syncthreads();
}

// sample_frames' weights (interface_searcher.py:345-352) + the cdf of np.random.choice
__global__ __launch_bounds__(ST) void sampler_prep_kernel(const double* __restrict__ P, const double* __restrict__ unvisited,
                                                          double* __restrict__ p, double* __restrict__ cdf, int N,
                                                          int num, double add, int* __restrict__ fallback,
                                                          SumProgram sp) {
    extern __shared__ double scratch[];
    __shared__ unsigned hist[256];
    __shared__ unsigned long long pref[2];
    __shared__ int s_cnt, s_nnz;
    __shared__ double s_min;
    for (int i = threadIdx.x; i < N; i += ST) p[i] = (P[i] + add) * unvisited[i];
    __syncthreads();
    const double vi = N * 0.75 + (1 + 0.75 * (1 - 1 - 1)) - 1;
    int lo = (int)floor(vi);
    const double g = vi - lo;
    lo = lo < 0 ? 0 : (lo > N - 1 ? N - 1 : lo);
    const double a_lo = block_select(p, N, lo, hist, pref);
    // next order statistic: a_lo again if it has duplicates past rank lo, else the smallest larger value
    if (threadIdx.x == 0) { s_cnt = 0; s_nnz = 0; s_min = INFINITY; }
    __syncthreads();
    int c_le = 0; double mn = INFINITY;
    for (int i = threadIdx.x; i < N; i += ST) { const double v = p[i]; c_le += v <= a_lo; if (v > a_lo && v < mn) mn = v; }
    atomicAdd(&s_cnt, c_le);
    // min over doubles >= 0 via the monotone bit pattern
    atomicMin(reinterpret_cast<unsigned long long*>(&s_min), (unsigned long long)__double_as_longlong(mn));
    __syncthreads();
    const double a_hi = (lo + 1 >= N || s_cnt > lo + 1) ? a_lo : s_min;
    const double thr = lerp75(a_lo, a_hi, g);
    int nnz = 0;
    for (int i = threadIdx.x; i < N; i += ST) { const double v = p[i] * (p[i] >= thr ? 1.0 : 0.0); p[i] = v; nnz += v != 0.0; }
    atomicAdd(&s_nnz, nnz);
    double s = block_np_sum(p, sp, scratch);
    const bool fb = (s == 0.0) || (s_nnz < num);
    if (fb) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += ST) p[i] = P[i] + add;
        s = block_np_sum(p, sp, scratch);
    }
    for (int i = threadIdx.x; i < N; i += ST) p[i] = p[i] / s;
    if (threadIdx.x == 0) *fallback = fb ? 1 : 0;
    block_cdf(p, cdf, N);
}

// pop_frames' weights (interface_searcher.py:369): p = score / score.sum()
// Also reports what numpy's choice() checks before drawing (mtrand.pyx, legacy RandomState.choice):
// info_nnz = count_nonzero(p > 0), info_sum = score.sum() (0 or NaN -> p holds NaN -> "probabilities contain NaN").
__global__ __launch_bounds__(ST) void pop_prep_kernel(const double* __restrict__ score, double* __restrict__ p,
                                                      double* __restrict__ cdf, int N, SumProgram sp,
                                                      int* __restrict__ info_nnz, double* __restrict__ info_sum) {
    extern __shared__ double scratch[];
    __shared__ int s_nnz;
    if (threadIdx.x == 0) s_nnz = 0;
    const double s = block_np_sum(score, sp, scratch);
    int nnz = 0;
    for (int i = threadIdx.x; i < N; i += ST) { const double v = score[i] / s; p[i] = v; nnz += v > 0.0; }
    atomicAdd(&s_nnz, nnz);
    block_cdf(p, cdf, N);
    if (threadIdx.x == 0) { *info_nnz = s_nnz; *info_sum = s; }
}

// choice()'s retry step: p[found] = 0; cdf = cumsum(p); cdf /= cdf[-1]
__global__ __launch_bounds__(ST) void exclude_kernel(double* __restrict__ p, double* __restrict__ cdf, int N,
                                                     const int* __restrict__ found, int m) {
    for (int i = threadIdx.x; i < m; i += ST) p[found[i]] = 0.0;
    block_cdf(p, cdf, N);
}

// cdf.searchsorted(x, side='right'): first i with cdf[i] > x (N if none)
__global__ void draw_kernel(const double* __restrict__ cdf, int N, const double* __restrict__ x, int k, int* __restrict__ idx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const double v = x[j];
    int lo = 0, hi = N;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= v) lo = mid + 1; else hi = mid; }
    idx[j] = lo;
}

__global__ void set_scores_kernel(double* __restrict__ score, const int* __restrict__ secs, const double* __restrict__ vals, int m) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < m; ++i) score[secs[i]] = vals[i];        // in order: later writes win
}

}  // namespace tstar

using namespace tstar;

struct tstar_searcher {
    int N = 0;
    double *score = nullptr, *unvisited = nullptr, *P = nullptr, *p = nullptr, *cdf = nullptr;
    double *d_t = nullptr, *d_c = nullptr, *d_vis_y = nullptr, *d_x = nullptr, *d_vals = nullptr;
    int *d_secs = nullptr, *d_vis_x = nullptr, *d_flag = nullptr, *d_idx = nullptr;
    double* d_info = nullptr;
    int cap = 0;                 // capacity of the small staging arrays
    SumProgram sp;
    size_t lds = 0;
};

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

extern "C" {

int tstar_searcher_destroy(tstar_searcher* s) {
    if (!s) return TSTAR_OK;
    void* ptrs[] = {s->score, s->unvisited, s->P, s->p, s->cdf, s->d_t, s->d_c, s->d_vis_y, s->d_x, s->d_vals,
                    s->d_secs, s->d_vis_x, s->d_flag, s->d_idx, s->d_info, s->sp.d_leaf_off, s->sp.d_leaf_len, s->sp.d_ops};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete s;
    return TSTAR_OK;
}

int tstar_searcher_create(tstar_searcher** out, int n_frames, double init_score, double init_p) {
    TSTAR_REQUIRE(out, "tstar_searcher_create: null argument");
    TSTAR_REQUIRE(n_frames >= 1 && n_frames <= (1 << 22), "tstar_searcher_create: n_frames must be in 1..4194304");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("tstar_searcher_create: no HIP device visible (this library has no CPU path)");
        return TSTAR_ERR_HIP;
    }
    tstar_searcher* s = new tstar_searcher();
    s->N = n_frames;
    const int N = n_frames;
    s->cap = N + 16;
    hipError_t e = hipSuccess;
    auto ad = [&](double** p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, n * sizeof(double)); };
    auto ai = [&](int** p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, n * sizeof(int)); };
    ad(&s->score, N); ad(&s->unvisited, N); ad(&s->P, N); ad(&s->p, N); ad(&s->cdf, N);
    ad(&s->d_t, s->cap + 8); ad(&s->d_c, s->cap + 8); ad(&s->d_vis_y, s->cap); ad(&s->d_x, s->cap); ad(&s->d_vals, s->cap);
    ai(&s->d_secs, s->cap); ai(&s->d_vis_x, s->cap); ai(&s->d_flag, 4); ai(&s->d_idx, s->cap);
    ad(&s->d_info, 2);
    std::vector<int> lo, ll; std::vector<signed char> ops;
    build_program(0, N, lo, ll, ops);
    s->sp.n_leaf = (int)lo.size(); s->sp.n_ops = (int)ops.size();
    ai(&s->sp.d_leaf_off, lo.size()); ai(&s->sp.d_leaf_len, ll.size());
    if (e == hipSuccess) e = hipMalloc(&s->sp.d_ops, ops.size());
    if (e == hipSuccess) e = hipMemcpy(s->sp.d_leaf_off, lo.data(), lo.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->sp.d_leaf_len, ll.data(), ll.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->sp.d_ops, ops.data(), ops.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error(std::string("tstar_searcher_create: allocation failed: ") + hipGetErrorString(e));
        tstar_searcher_destroy(s);
        return TSTAR_ERR_HIP;
    }
    s->lds = (size_t)(s->sp.n_leaf + 64) * sizeof(double);
    if (s->lds > 48 * 1024) {
        set_error("tstar_searcher_create: n_frames too large for the single-workgroup reduction");
        tstar_searcher_destroy(s);
        return TSTAR_ERR_ARG;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(ST), 0, 0, s->score, init_score, N);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(ST), 0, 0, s->unvisited, 1.0, N);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(ST), 0, 0, s->P, init_p, N);
    TSTAR_HIP_CHECK(hipDeviceSynchronize());
    *out = s;
    return TSTAR_OK;
}

static int launch_apply(tstar_searcher* s, const double* d_conf, int n, int window, int do_write, int do_compact, hipStream_t st) {
    // LDS working set of the window spread: score (N f64) + conf (n f64) + secs (n i32); global fallback beyond 144 KB
    const size_t need = (size_t)s->N * 8 + (size_t)n * 12;
    const int use_lds = need <= 144 * 1024;
    const size_t dyn = use_lds ? need : 0;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(apply_grid_kernel), 144 * 1024)) return rc;
    hipLaunchKernelGGL(apply_grid_kernel, dim3(1), dim3(ST), dyn, st, s->score, s->unvisited, s->d_secs, d_conf, n, s->N, window,
                       s->d_vis_x, s->d_vis_y, s->d_flag, use_lds, do_write, do_compact);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

int tstar_searcher_apply_grid(tstar_searcher* s, const int32_t* h_secs, const double* d_conf, int n,
                              int* h_n_visited, int32_t* h_vis_x, double* h_vis_y, void* stream) {
    TSTAR_REQUIRE(s && h_secs && d_conf && h_n_visited && h_vis_x && h_vis_y, "tstar_searcher_apply_grid: null argument");
    TSTAR_REQUIRE(n >= 1 && n <= s->cap, "tstar_searcher_apply_grid: bad n");
    for (int i = 0; i < n; ++i) TSTAR_REQUIRE(h_secs[i] >= 0 && h_secs[i] < s->N, "tstar_searcher_apply_grid: second out of range");
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_secs, h_secs, n * sizeof(int), hipMemcpyHostToDevice, st));
    RC(launch_apply(s, d_conf, n, 5, 1, 1, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_n_visited, s->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    const int nv = *h_n_visited;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_vis_x, s->d_vis_x, nv * sizeof(int), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_vis_y, s->d_vis_y, nv * sizeof(double), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    return TSTAR_OK;
}

int tstar_searcher_window_spread(tstar_searcher* s, const int32_t* h_secs, const double* h_conf, int n, int window,
                                 void* stream) {
    TSTAR_REQUIRE(s && h_secs && h_conf, "tstar_searcher_window_spread: null argument");
    TSTAR_REQUIRE(n >= 1 && n <= s->cap && window >= 0 && window <= 4096, "tstar_searcher_window_spread: bad n or window");
    for (int i = 0; i < n; ++i) TSTAR_REQUIRE(h_secs[i] >= 0 && h_secs[i] < s->N, "tstar_searcher_window_spread: second out of range");
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_secs, h_secs, n * sizeof(int), hipMemcpyHostToDevice, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_vals, h_conf, n * sizeof(double), hipMemcpyHostToDevice, st));
    RC(launch_apply(s, s->d_vals, n, window, 0, 0, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));     // host staging buffers are reused by the next call
    return TSTAR_OK;
}

int tstar_searcher_visited(tstar_searcher* s, int* h_n_visited, int32_t* h_vis_x, double* h_vis_y, void* stream) {
    TSTAR_REQUIRE(s && h_n_visited && h_vis_x && h_vis_y, "tstar_searcher_visited: null argument");
    hipStream_t st = (hipStream_t)stream;
    // n = 0 samples: no write-back, no spread (the percentile of an empty set is never used), compaction only
    RC(launch_apply(s, s->d_vals, 0, 0, 0, 1, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_n_visited, s->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    const int nv = *h_n_visited;
    if (nv > 0) {
        TSTAR_HIP_CHECK(hipMemcpyAsync(h_vis_x, s->d_vis_x, nv * sizeof(int), hipMemcpyDeviceToHost, st));
        TSTAR_HIP_CHECK(hipMemcpyAsync(h_vis_y, s->d_vis_y, nv * sizeof(double), hipMemcpyDeviceToHost, st));
        TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    }
    return TSTAR_OK;
}

int tstar_searcher_set_spline(tstar_searcher* s, const double* h_t, const double* h_c, int n_knots, int k, void* stream) {
    TSTAR_REQUIRE(s && h_t && h_c, "tstar_searcher_set_spline: null argument");
    TSTAR_REQUIRE(k >= 1 && k <= 5 && n_knots >= 2 * (k + 1) && n_knots <= s->cap + 8, "tstar_searcher_set_spline: bad spline");
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_t, h_t, n_knots * sizeof(double), hipMemcpyHostToDevice, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_c, h_c, n_knots * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(distribution_kernel, dim3(1), dim3(ST), s->lds, st, s->P, s->d_t, s->d_c, n_knots, k, s->N, s->sp);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

int tstar_searcher_sampler_prep(tstar_searcher* s, int num, double add, int* h_fallback, void* stream) {
    TSTAR_REQUIRE(s && h_fallback, "tstar_searcher_sampler_prep: null argument");
    TSTAR_REQUIRE(num >= 1 && num <= s->N, "tstar_searcher_sampler_prep: bad sample count");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sampler_prep_kernel, dim3(1), dim3(ST), s->lds, st, s->P, s->unvisited, s->p, s->cdf, s->N, num,
                       add, s->d_flag + 1, s->sp);
    TSTAR_HIP_CHECK(hipGetLastError());
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_fallback, s->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    return TSTAR_OK;
}

int tstar_searcher_pop_prep(tstar_searcher* s, int* h_nnz, double* h_sum, void* stream) {
    TSTAR_REQUIRE(s && h_nnz && h_sum, "tstar_searcher_pop_prep: null argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pop_prep_kernel, dim3(1), dim3(ST), s->lds, st, s->score, s->p, s->cdf, s->N, s->sp, s->d_flag + 2, s->d_info);
    TSTAR_HIP_CHECK(hipGetLastError());
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_nnz, s->d_flag + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_sum, s->d_info, sizeof(double), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    return TSTAR_OK;
}

int tstar_searcher_draw(tstar_searcher* s, const double* h_x, int k, int32_t* h_idx, void* stream) {
    TSTAR_REQUIRE(s && h_x && h_idx, "tstar_searcher_draw: null argument");
    TSTAR_REQUIRE(k >= 1 && k <= s->cap, "tstar_searcher_draw: bad k");
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_x, h_x, k * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(draw_kernel, dim3(cdiv(k, 256)), dim3(256), 0, st, s->cdf, s->N, s->d_x, k, s->d_idx);
    TSTAR_HIP_CHECK(hipGetLastError());
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_idx, s->d_idx, k * sizeof(int), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    return TSTAR_OK;
}

int tstar_searcher_exclude(tstar_searcher* s, const int32_t* h_found, int m, void* stream) {
    TSTAR_REQUIRE(s && h_found, "tstar_searcher_exclude: null argument");
    TSTAR_REQUIRE(m >= 1 && m <= s->cap, "tstar_searcher_exclude: bad m");
    for (int i = 0; i < m; ++i) TSTAR_REQUIRE(h_found[i] >= 0 && h_found[i] < s->N, "tstar_searcher_exclude: index out of range");
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_secs, h_found, m * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(exclude_kernel, dim3(1), dim3(ST), 0, st, s->p, s->cdf, s->N, s->d_secs, m);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

int tstar_searcher_set_scores(tstar_searcher* s, const int32_t* h_secs, const double* h_vals, int m, void* stream) {
    TSTAR_REQUIRE(s && h_secs && h_vals, "tstar_searcher_set_scores: null argument");
    TSTAR_REQUIRE(m >= 1 && m <= s->cap, "tstar_searcher_set_scores: bad m");
    for (int i = 0; i < m; ++i) TSTAR_REQUIRE(h_secs[i] >= 0 && h_secs[i] < s->N, "tstar_searcher_set_scores: second out of range");
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_secs, h_secs, m * sizeof(int), hipMemcpyHostToDevice, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(s->d_vals, h_vals, m * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(set_scores_kernel, dim3(1), dim3(64), 0, st, s->score, s->d_secs, s->d_vals, m);
    TSTAR_HIP_CHECK(hipGetLastError());
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));     // host staging buffers are reused by the next call
    return TSTAR_OK;
}

int tstar_searcher_read_state(tstar_searcher* s, double* h_out, void* stream) {
    TSTAR_REQUIRE(s && h_out, "tstar_searcher_read_state: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = (size_t)s->N * sizeof(double);
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_out, s->P, nb, hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_out + s->N, s->score, nb, hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_out + 2 * (size_t)s->N, s->unvisited, nb, hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    return TSTAR_OK;
}

int tstar_searcher_write(tstar_searcher* s, int which, const double* h_in, void* stream) {
    TSTAR_REQUIRE(s && h_in, "tstar_searcher_write: null argument");
    TSTAR_REQUIRE(which >= 0 && which <= 2, "tstar_searcher_write: which must be 0 (score), 1 (non_visiting) or 2 (P)");
    double* dst[] = {s->score, s->unvisited, s->P};
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(dst[which], h_in, s->N * sizeof(double), hipMemcpyHostToDevice, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));     // the host buffer may be reused by the caller
    return TSTAR_OK;
}

int tstar_searcher_read(tstar_searcher* s, int which, double* h_out, void* stream) {
    TSTAR_REQUIRE(s && h_out, "tstar_searcher_read: null argument");
    TSTAR_REQUIRE(which >= 0 && which <= 4, "tstar_searcher_read: which must be 0..4");
    const double* src[] = {s->score, s->unvisited, s->P, s->p, s->cdf};
    hipStream_t st = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_out, src[which], s->N * sizeof(double), hipMemcpyDeviceToHost, st));
    TSTAR_HIP_CHECK(hipStreamSynchronize(st));
    return TSTAR_OK;
}

}  // extern "C"
