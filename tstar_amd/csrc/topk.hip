// Downstream frame selection of the reference's evaluators: the top-k seconds of the saved
// keyframe_distribution, restricted to a clip (LVHaystackBench/val_qa_results.py:90-110):
//   dist = nan_to_num(float32(P)); dist.sum() == 0 -> ones; dist_clip = dist[start:end];
//   dist_clip.sum() == 0 -> ones; dist_clip /= dist_clip.sum(); topk = argsort(-dist_clip)[:k];
//   sorted ascending, + start.
// The division is done in float32 exactly as numpy does it (np.sum's pairwise order, then an f32
// divide): a near-flat P makes distinct values collide after the division,
// and those division-induced ties change which seconds are selected.  Ranking: k rounds of a
// block-wide arg-max over the NORMALISED values, ties -> lowest index (= np.argsort(kind='stable');
// the reference's default introsort leaves the order of equal keys to the numpy build and CPU), then
// an ascending sort of the k winners.  Latency-bound (N <= ~16 K); the workspace is a process-wide
// buffer grown on demand (no allocation per call).
#include "../../include/tstar_hip.h"
#include "common.h"
#include <math.h>
#include <mutex>

namespace tstar {

// numpy pairwise sum (loops_utils.h.src FLOAT_pairwise_sum): n < 8 sequential from 0.; n <= 128 eight
// accumulators; else split at n/2 rounded down to a multiple of 8.  One lane; recursion depth <= log2(N/128).
__device__ float np_pairwise_sum_f32(const float* a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
}

// a.sum() as numpy computes it: np.add.reduce feeds the inner loop one ufunc buffer (8192 elements) at a time and
// accumulates the chunk sums left to right
__device__ float np_sum_f32(const float* a, int n) {
    float s = np_pairwise_sum_f32(a, n < 8192 ? n : 8192);
    for (int c = 8192; c < n; c += 8192) s += np_pairwise_sum_f32(a + c, n - c < 8192 ? n - c : 8192);
    return s;
}

__global__ __launch_bounds__(1024) void topk_kernel(const double* __restrict__ P, int N, int start, int end, int k,
                                                    float* __restrict__ full, int* __restrict__ out) {
    __shared__ float s_v[1024];
    __shared__ int s_i[1024];
    __shared__ float s_sum;
    const int t = threadIdx.x, n = end - start;
    float* work = full + start;
    // dist = nan_to_num(float32(P), nan=0.0)  (+-inf -> +-FLT_MAX, as numpy)
    for (int i = t; i < N; i += 1024) {
        float v = (float)P[i];
        if (isnan(v)) v = 0.f;
        else if (isinf(v)) v = v > 0.f ? 3.402823466e+38f : -3.402823466e+38f;
        full[i] = v;
    }
    __syncthreads();
    if (t == 0) s_sum = np_sum_f32(full, N);
    __syncthreads();
    if (s_sum == 0.f) {                                   // if dist.sum() == 0: dist = ones_like(dist)
        __syncthreads();
        for (int i = t; i < N; i += 1024) full[i] = 1.f;
    }
    __syncthreads();
    if (t == 0) s_sum = np_sum_f32(work, n);
    __syncthreads();
    if (s_sum == 0.f) {                                   // if dist_clip.sum() == 0: ones
        __syncthreads();
        for (int i = t; i < n; i += 1024) work[i] = 1.f;
        __syncthreads();
        if (t == 0) s_sum = np_sum_f32(work, n);
        __syncthreads();
    }
    const float total = s_sum;
    for (int i = t; i < n; i += 1024) work[i] = work[i] / total;      // dist_clip /= dist_clip.sum()  (float32)
    __syncthreads();
    // argsort(-dist_clip)[:k]: NaN keys (0/0 cannot happen here; inf/inf can) sort last in numpy; they are never
    // selected before any finite value here because NaN compares false against everything
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = t; i < n; i += 1024) { const float v = work[i]; if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; } }
        s_v[t] = bv; s_i[t] = bi;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (t < o) {
                const float ov = s_v[t + o]; const int oi = s_i[t + o];
                if (ov > s_v[t] || (ov == s_v[t] && oi < s_i[t])) { s_v[t] = ov; s_i[t] = oi; }
            }
            __syncthreads();
        }
        if (t == 0) { out[r] = s_i[0]; if (s_i[0] < n) work[s_i[0]] = -INFINITY; }
        __syncthreads();
    }
    if (t == 0) {                                  // ascending order + clip offset (k is small)
        for (int a = 1; a < k; ++a) { int v = out[a], b = a - 1; while (b >= 0 && out[b] > v) { out[b + 1] = out[b]; --b; } out[b + 1] = v; }
        for (int a = 0; a < k; ++a) out[a] += start;
    }
}

// process-wide workspace (float [N] + int [k]), grown on demand; one top-k at a time per process
static std::mutex g_topk_mu;
static float* g_work = nullptr;
static int* g_out = nullptr;
static size_t g_work_n = 0, g_out_n = 0;
static int g_dev = -1;

}  // namespace tstar

using namespace tstar;
extern "C" int tstar_topk_seconds(const double* d_P, int N, int clip_start, int clip_end, int k, int32_t* h_out, void* stream) {
    TSTAR_REQUIRE(d_P && h_out, "tstar_topk_seconds: null argument");
    TSTAR_REQUIRE(N >= 1 && clip_start >= 0 && clip_end <= N && clip_start < clip_end, "tstar_topk_seconds: bad clip");
    TSTAR_REQUIRE(k >= 1 && k <= clip_end - clip_start && k <= 4096, "tstar_topk_seconds: k must be in 1..min(clip length, 4096)");
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(g_topk_mu);
    int dev = 0;
    TSTAR_HIP_CHECK(hipGetDevice(&dev));
    if (dev != g_dev || (size_t)N > g_work_n || (size_t)k > g_out_n) {
        if (g_work) (void)hipFree(g_work);
        if (g_out) (void)hipFree(g_out);
        g_work = nullptr; g_out = nullptr; g_work_n = g_out_n = 0;
        const size_t wn = round_up((size_t)N, 4096), on = round_up((size_t)k, 64);
        TSTAR_HIP_CHECK(hipMalloc(&g_work, wn * sizeof(float)));
        TSTAR_HIP_CHECK(hipMalloc(&g_out, on * sizeof(int)));
        g_work_n = wn; g_out_n = on; g_dev = dev;
    }
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), 0, s, d_P, N, clip_start, clip_end, k, g_work, g_out);
    TSTAR_HIP_CHECK(hipGetLastError());
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_out, g_out, (size_t)k * sizeof(int), hipMemcpyDeviceToHost, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}
