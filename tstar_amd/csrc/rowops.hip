// Row-wise (HBM-bound) pieces of the OWL-ViT forward: LayerNorm, CLS-row init,
// and the post-LN / CLS-merge / detection-LN fusion of
// OwlViTForObjectDetection.image_text_embedder (HF modeling_owlvit.py:1183-1191).
// One wave64 per row, 16-byte loads, shuffle reductions; no LDS.
#include "common.h"
#include "kernels.h"

namespace tstar {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// NV = D / 256 float4 per lane
template <int NV>
__device__ __forceinline__ void ln_regs(f32x4 (&x)[NV], const float* __restrict__ w, const float* __restrict__ b,
                                        int lane, float invD) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    const float mean = wave_sum(s) * invD;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { float d = x[i][e] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(v) * invD + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (i * 64 + lane) * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + (i * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[i][e] = (x[i][e] - mean) * rstd * wv[e] + bv[e];
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int D = NV * 256;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * D + (i * 64 + lane) * 4);
    ln_regs<NV>(v, w, b, lane, 1.0f / D);
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(y + (size_t)row * D + (i * 64 + lane) * 4) = v[i];
}

int layernorm_f32(const float* x, float* y, const float* w, const float* b, int rows, int D, hipStream_t s) {
    TSTAR_REQUIRE(rows > 0, "layernorm_f32: no rows");
    const int grid = cdiv(rows, 4);
    if (D == 768) hipLaunchKernelGGL(layernorm_kernel<3>, dim3(grid), dim3(256), 0, s, x, y, w, b, rows);
    else if (D == 512) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, s, x, y, w, b, rows);
    else { TSTAR_REQUIRE(false, "layernorm_f32: D must be 512 or 768"); }
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// token row 0 of every image: class embedding + position embedding[0]
// (OwlViTVisionEmbeddings.forward, modeling_owlvit.py:338-343)
__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                int ntok, int D) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(size_t)b * ntok * D + d] = cls[d] + pos[d];
}

int write_cls_rows(float* x, const float* cls, const float* pos, int B, int ntok, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3(B), dim3(256), 0, s, x, cls, pos, ntok, D);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// feats[b,p,:] = LN_det( LN_post(x[b,1+p,:]) * LN_post(x[b,0,:]) ), D = 768
__global__ __launch_bounds__(256) void merge_cls_ln_kernel(const float* __restrict__ x, float* __restrict__ feats,
                                                           const float* __restrict__ pw, const float* __restrict__ pb,
                                                           const float* __restrict__ dw, const float* __restrict__ db,
                                                           int B, int ntok) {
    constexpr int NV = 3, D = 768;
    const int lane = threadIdx.x & 63;
    const int np = ntok - 1;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // over B*np
    if (row >= B * np) return;
    const int b = row / np, p = row - b * np;
    const float* xr = x + ((size_t)b * ntok + 1 + p) * D;
    const float* xc = x + (size_t)b * ntok * D;
    f32x4 v[NV], c[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
        c[i] = *reinterpret_cast<const f32x4*>(xc + (i * 64 + lane) * 4);
    }
    ln_regs<NV>(v, pw, pb, lane, 1.0f / D);
    ln_regs<NV>(c, pw, pb, lane, 1.0f / D);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] *= c[i];
    ln_regs<NV>(v, dw, db, lane, 1.0f / D);
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(feats + (size_t)row * D + (i * 64 + lane) * 4) = v[i];
}

int merge_cls_ln(const float* x, float* feats, const float* post_w, const float* post_b,
                 const float* det_w, const float* det_b, int B, int ntok, int D, hipStream_t s) {
    TSTAR_REQUIRE(D == 768, "merge_cls_ln: D must be 768");
    const int rows = B * (ntok - 1);
    hipLaunchKernelGGL(merge_cls_ln_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, feats, post_w, post_b, det_w,
                       det_b, B, ntok);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

}  // namespace tstar
