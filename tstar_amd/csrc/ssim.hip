// Search-quality metric of the reference's evaluator: pairwise SSIM between ground-truth and
// predicted keyframes (/root/reference/LVHaystackBench/val_tstar_results.py:48-95).
//
// Reference semantics, kept exactly (including its layout quirk): frames are HWC uint8, scaled by
// 1/255 in float32, and `ssim_torch` passes the HWC tensor to conv2d as if it were CHW -- so the
// "channel" axis is H (groups = H) and the 11x11 Gaussian window (sigma 1.5, zero padding 5) slides
// over the (W, 3) plane of every row: 11 taps along W and, because the colour axis is only 3 wide,
// at most 3 taps along it.  Five windowed sums per element (mu1, mu2, E[x^2], E[y^2], E[xy]) share
// their loads; SSIM = ((2 mu1 mu2 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)), mean over
// all H*W*3 elements.  float32 arithmetic like torch; the final mean is accumulated in float64 with a
// fixed (deterministic) order.  One block per (pair, row slab); HBM traffic is 2 frames per pair --
// the kernel is VALU-bound (330 FMAs per output element).
#include "../../include/tstar_hip.h"
#include "common.h"

namespace tstar {

constexpr int SS_ROWS = 4;   // image rows per block

__global__ __launch_bounds__(256) void ssim_pairs_kernel(const uint8_t* __restrict__ gt, const uint8_t* __restrict__ pred,
                                                         int P, int H, int W, const float* __restrict__ win /*[11][11]*/,
                                                         double* __restrict__ partial, int slabs) {
    __shared__ float s_win[121];
    __shared__ double s_red[256];
    const int pair = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int gi = pair / P, pj = pair % P;
    const uint8_t* a = gt + (size_t)gi * H * W * 3;
    const uint8_t* b = pred + (size_t)pj * H * W * 3;
    if (threadIdx.x < 121) s_win[threadIdx.x] = win[threadIdx.x];
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    double acc = 0.0;
    const int h0 = slab * SS_ROWS, h1 = (h0 + SS_ROWS < H) ? h0 + SS_ROWS : H;
    for (int idx = threadIdx.x; idx < (h1 - h0) * W; idx += blockDim.x) {
        const int h = h0 + idx / W, w = idx % W;
        const uint8_t* ra = a + (size_t)h * W * 3;
        const uint8_t* rb = b + (size_t)h * W * 3;
        float m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0}, s11[3] = {0, 0, 0}, s22[3] = {0, 0, 0}, s12[3] = {0, 0, 0};
        for (int dw = -5; dw <= 5; ++dw) {
            const int ww = w + dw;
            if (ww < 0 || ww >= W) continue;                      // zero padding
            float x[3], y[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { x[c] = (float)ra[ww * 3 + c] / 255.0f; y[c] = (float)rb[ww * 3 + c] / 255.0f; }
#pragma unroll
            for (int c = 0; c < 3; ++c)                           // output colour index
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {                  // input colour index, dc = cc - c in [-2, 2]
                    const float wgt = s_win[(dw + 5) * 11 + (cc - c + 5)];
                    m1[c] += wgt * x[cc];
                    m2[c] += wgt * y[cc];
                    s11[c] += wgt * (x[cc] * x[cc]);
                    s22[c] += wgt * (y[cc] * y[cc]);
                    s12[c] += wgt * (x[cc] * y[cc]);
                }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mu1_sq = m1[c] * m1[c], mu2_sq = m2[c] * m2[c], mu12 = m1[c] * m2[c];
            const float v1 = s11[c] - mu1_sq, v2 = s22[c] - mu2_sq, v12 = s12[c] - mu12;
            const float s = ((2.f * mu12 + C1) * (2.f * v12 + C2)) / ((mu1_sq + mu2_sq + C1) * (v1 + v2 + C2));
            acc += (double)s;
        }
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}

__global__ void ssim_finish_kernel(const double* __restrict__ partial, int slabs, double inv_n, double* __restrict__ out, int pairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    double s = 0.0;
    for (int i = 0; i < slabs; ++i) s += partial[(size_t)p * slabs + i];      // fixed order
    out[p] = s * inv_n;
}

}  // namespace tstar

using namespace tstar;
extern "C" int tstar_ssim_pairwise(const uint8_t* d_gt, int G, const uint8_t* d_pred, int P, int H, int W,
                                   const float* h_window, double* d_out, void* stream) {
    TSTAR_REQUIRE(d_gt && d_pred && h_window && d_out, "tstar_ssim_pairwise: null argument");
    TSTAR_REQUIRE(G >= 1 && P >= 1 && H >= 1 && W >= 1, "tstar_ssim_pairwise: empty input");
    hipStream_t s = (hipStream_t)stream;
    const int slabs = cdiv(H, SS_ROWS), pairs = G * P;
    float* d_win = nullptr; double* d_part = nullptr;
    TSTAR_HIP_CHECK(hipMalloc(&d_win, 121 * sizeof(float)));
    TSTAR_HIP_CHECK(hipMalloc(&d_part, (size_t)pairs * slabs * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d_win, h_window, 121 * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ssim_pairs_kernel, dim3(pairs * slabs), dim3(256), 0, s, d_gt, d_pred, P, H, W, d_win, d_part, slabs);
        hipLaunchKernelGGL(ssim_finish_kernel, dim3(cdiv(pairs, 64)), dim3(64), 0, s, d_part, slabs,
                           1.0 / ((double)H * W * 3), d_out, pairs);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_win); (void)hipFree(d_part);
    if (e != hipSuccess) { set_error(std::string("tstar_ssim_pairwise: ") + hipGetErrorString(e)); return TSTAR_ERR_HIP; }
    return TSTAR_OK;
}
