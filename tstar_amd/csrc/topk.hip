// Downstream frame selection of the reference's evaluators: the top-k seconds of the saved
// keyframe_distribution, restricted to a clip (LVHaystackBench/val_qa_results.py:90-110):
//   dist = nan_to_num(float32(P)); all-zero -> ones; dist_clip = dist[start:end]; all-zero -> ones;
//   dist_clip /= dist_clip.sum(); topk = argsort(-dist_clip)[:k]; sorted ascending, + start.
// Normalising by a positive sum does not change the order, so the kernel ranks the float32 clip
// directly: k rounds of a block-wide arg-max (ties -> lowest index; numpy's unstable argsort leaves
// tie order unspecified), then an ascending sort of the k winners.  Latency-bound (N <= ~16 K).
#include "../../include/tstar_hip.h"
#include "common.h"
#include <math.h>

namespace tstar {

__global__ __launch_bounds__(1024) void topk_kernel(const double* __restrict__ P, int N, int start, int end, int k,
                                                    float* __restrict__ work, int* __restrict__ out) {
    __shared__ float s_v[1024];
    __shared__ int s_i[1024];
    __shared__ int s_allzero;
    const int t = threadIdx.x, n = end - start;
    // dist = nan_to_num(float32(P)); if dist.sum() == 0: ones  (sum of non-negatives is 0 iff all are 0)
    if (t == 0) s_allzero = 1;
    __syncthreads();
    int nz = 0;
    for (int i = t; i < N; i += 1024) { float v = (float)P[i]; if (isnan(v)) v = 0.f; nz |= (v != 0.f); }
    if (nz) s_allzero = 0;
    __syncthreads();
    const int all0 = s_allzero;
    __syncthreads();
    if (t == 0) s_allzero = 1;
    __syncthreads();
    nz = 0;
    for (int i = t; i < n; i += 1024) {
        float v = (float)P[start + i];
        if (isnan(v)) v = 0.f;
        if (all0) v = 1.f;
        work[i] = v;
        nz |= (v != 0.f);
    }
    if (nz) s_allzero = 0;
    __syncthreads();
    if (s_allzero) for (int i = t; i < n; i += 1024) work[i] = 1.f;
    __syncthreads();
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
        if (t == 0) { out[r] = s_i[0]; work[s_i[0]] = -INFINITY; }
        __syncthreads();
    }
    if (t == 0) {                                  // ascending order + clip offset (k is small)
        for (int a = 1; a < k; ++a) { int v = out[a], b = a - 1; while (b >= 0 && out[b] > v) { out[b + 1] = out[b]; --b; } out[b + 1] = v; }
        for (int a = 0; a < k; ++a) out[a] += start;
    }
}

}  // namespace tstar

using namespace tstar;
extern "C" int tstar_topk_seconds(const double* d_P, int N, int clip_start, int clip_end, int k, int32_t* h_out, void* stream) {
    TSTAR_REQUIRE(d_P && h_out, "tstar_topk_seconds: null argument");
    TSTAR_REQUIRE(N >= 1 && clip_start >= 0 && clip_end <= N && clip_start < clip_end, "tstar_topk_seconds: bad clip");
    TSTAR_REQUIRE(k >= 1 && k <= clip_end - clip_start && k <= 4096, "tstar_topk_seconds: k must be in 1..min(clip length, 4096)");
    hipStream_t s = (hipStream_t)stream;
    float* work = nullptr; int* out = nullptr;
    TSTAR_HIP_CHECK(hipMalloc(&work, (size_t)(clip_end - clip_start) * sizeof(float)));
    TSTAR_HIP_CHECK(hipMalloc(&out, (size_t)k * sizeof(int)));
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), 0, s, d_P, N, clip_start, clip_end, k, work, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, out, (size_t)k * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(work); (void)hipFree(out);
    if (e != hipSuccess) { set_error(std::string("tstar_topk_seconds: ") + hipGetErrorString(e)); return TSTAR_ERR_HIP; }
    return TSTAR_OK;
}
