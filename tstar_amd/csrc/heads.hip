// OWL-ViT detection heads epilogue + T* grid-cell aggregation.
//
// detect_rows_kernel (one wave64 per patch row) fuses, for every (image, patch):
//   * class head tail  (HF modeling_owlvit.py:1027-1045): L2-normalise the
//     dense0 output (+1e-6), dot with the pre-normalised query embeddings,
//     (sim + shift(feats)) * (ELU(scale(feats)) + 1), query mask;
//   * box head tail    (modeling_owlvit.py:997-999, 1133-1136): dense2 (768->4),
//     + box_bias, sigmoid;
//   * post-process     (image_processing_owlvit.py:151-177 as called from
//     /root/reference/TStar/interface_heuristic.py:242-243): max/argmax over
//     queries (first max wins), sigmoid, cxcywh -> xyxy * (W,H,W,H);
//   * the per-detection half of TStarSearcher.imageGridScoreFunction
//     (/root/reference/TStar/interface_searcher.py:133-148): class weight,
//     box centre -> grid cell.
// cell_reduce_kernel (one block per image) is the scatter-max + per-cell class
// set of interface_searcher.py:149-150 (max is order-independent, so the
// parallel reduction is bit-identical to the reference's sequential loop).
#include "common.h"
#include "kernels.h"
#include "heads.h"
#include <float.h>

namespace tstar {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void detect_rows_kernel(DetectRowsArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int p = row % a.np;

    // ---- feats row: 12 floats per lane
    f32x4 f[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) f[i] = *reinterpret_cast<const f32x4*>(a.feats + (size_t)row * 768 + (i * 64 + lane) * 4);
    float sh = 0.f, sc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.shift_w + (i * 64 + lane) * 4);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.scale_w + (i * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh += f[i][e] * w0[e]; sc += f[i][e] * w1[e]; }
    }
    sh = wsum(sh) + a.shift_b[0];
    sc = wsum(sc) + a.scale_b[0];
    sc = (sc > 0.f ? sc : expm1f(sc)) + 1.0f;          // ELU(x) + 1

    // ---- class embedding row: 8 floats per lane, L2 normalise with +1e-6
    f32x4 c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) c[i] = *reinterpret_cast<const f32x4*>(a.cls + (size_t)row * 512 + (i * 64 + lane) * 4);
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) n2 += c[i][e] * c[i][e];
    const float den = sqrtf(wsum(n2)) + 1e-6f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) c[i][e] = c[i][e] / den;

    const int set = a.image_set ? a.image_set[row / a.np] : 0;
    const int nq = a.setQ[set];
    const float* qn = a.qn + (size_t)set * 32 * 512;
    const uint8_t* qmask = a.qmask + set * 32;
    float best = -FLT_MAX;
    int label = 0;
    for (int q = 0; q < nq; ++q) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qn + (size_t)q * 512 + (i * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) d += c[i][e] * qv[e];
        }
        d = wsum(d);
        float lg = (d + sh) * sc;
        if (qmask[q] == 0) lg = -FLT_MAX;               // torch.finfo(float32).min
        if (a.logits && lane == 0) a.logits[(size_t)row * a.Q + q] = lg;
        if (lg > best) { best = lg; label = q; }        // ties -> lowest index (torch CPU max)
    }

    // ---- box head tail: 4 dots over the second GELU layer's output
    f32x4 hb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) hb[i] = *reinterpret_cast<const f32x4*>(a.boxh + (size_t)row * 768 + (i * 64 + lane) * 4);
    float bx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(a.box2_w + (size_t)k * 768 + (i * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) d += hb[i][e] * w[e];
        }
        d = wsum(d) + a.box2_b[k] + a.box_bias[p * 4 + k];
        bx[k] = 1.0f / (1.0f + expf(-d));
    }

    if (lane == 0) {
        const float score = 1.0f / (1.0f + expf(-best));
        const float fw = (float)a.img_w, fh = (float)a.img_h;
        const float x0 = (bx[0] - 0.5f * bx[2]) * fw, y0 = (bx[1] - 0.5f * bx[3]) * fh;
        const float x1 = (bx[0] + 0.5f * bx[2]) * fw, y1 = (bx[1] + 0.5f * bx[3]) * fh;
        a.scores[row] = score;
        a.labels[row] = label;
        f32x4 o; o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1;
        *reinterpret_cast<f32x4*>(a.xyxy + (size_t)row * 4) = o;
        if (a.cxcywh) {
            f32x4 bb; bb[0] = bx[0]; bb[1] = bx[1]; bb[2] = bx[2]; bb[3] = bx[3];
            *reinterpret_cast<f32x4*>(a.cxcywh + (size_t)row * 4) = bb;
        }
    }
}

int detect_rows(const DetectRowsArgs& a, hipStream_t s) {
    TSTAR_REQUIRE(a.rows > 0 && a.setQ, "detect_rows: empty problem");
    hipLaunchKernelGGL(detect_rows_kernel, dim3(cdiv(a.rows, 4)), dim3(256), 0, s, a);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// One block per image.  conf = float64(score) * float64(weight): the reference multiplies an np.float32 scalar by a
// Python float (interface_searcher.py:136-137), which is a float64 product under its pinned numpy 1.26 (exact, and
// equal to the float32 product, for the default weights 1.0 / 0.5; a user weight such as 0.7 needs the f64 form).
// conf is >= 0, so max over f64 == max over its bit pattern.
__global__ __launch_bounds__(256) void cell_reduce_kernel(const float* __restrict__ scores, const int* __restrict__ labels,
                                                          const float* __restrict__ xyxy, const double* __restrict__ qweight_all,
                                                          const int* __restrict__ image_set, int np, int img_w, int img_h, int grows, int gcols, float thr,
                                                          double* __restrict__ cell_conf, uint32_t* __restrict__ cell_mask,
                                                          int* __restrict__ n_kept) {
    extern __shared__ unsigned long long sm[];
    const int ncell = grows * gcols;
    unsigned long long* cbits = sm;
    uint32_t* cmask = reinterpret_cast<uint32_t*>(sm + ncell);
    __shared__ int kept;
    const int b = blockIdx.x;
    const double* qweight = qweight_all + (image_set ? image_set[b] : 0) * 32;
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) { cbits[i] = 0ull; cmask[i] = 0u; }
    if (threadIdx.x == 0) kept = 0;
    __syncthreads();
    // cell sizes are Python floats in the reference (interface_searcher.py:117-118)
    const double cw = (double)img_w / (double)gcols, ch = (double)img_h / (double)grows;
    for (int p = threadIdx.x; p < np; p += blockDim.x) {
        const size_t r = (size_t)b * np + p;
        const float s = scores[r];
        if (s > thr) {
            const int lab = labels[r];
            const double conf = (double)s * qweight[lab];
            const f32x4 bb = *reinterpret_cast<const f32x4*>(xyxy + r * 4);
            const float cx = (bb[0] + bb[2]) * 0.5f;          // f32 add, exact halving
            const float cy = (bb[1] + bb[3]) * 0.5f;
            int gx = (int)floor((double)cx / cw), gy = (int)floor((double)cy / ch);
            gx = gx < gcols - 1 ? gx : gcols - 1;
            gy = gy < grows - 1 ? gy : grows - 1;
            gx = gx < 0 ? 0 : gx; gy = gy < 0 ? 0 : gy;
            const int cell = gy * gcols + gx;
            atomicMax(&cbits[cell], (unsigned long long)__double_as_longlong(conf));
            atomicOr(&cmask[cell], 1u << lab);
            atomicAdd(&kept, 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) {
        cell_conf[(size_t)b * ncell + i] = __longlong_as_double((long long)cbits[i]);
        cell_mask[(size_t)b * ncell + i] = cmask[i];
    }
    if (threadIdx.x == 0 && n_kept) n_kept[b] = kept;
}

int cell_reduce(const float* scores, const int* labels, const float* xyxy, const double* qweight, const int* image_set,
                int B, int np, int img_w, int img_h, int grows, int gcols, float thr, double* cell_conf, uint32_t* cell_mask,
                int* n_kept, hipStream_t s) {
    TSTAR_REQUIRE(grows > 0 && gcols > 0 && grows * gcols <= 4096, "cell_reduce: grid must have 1..4096 cells");
    const size_t lds = (size_t)grows * gcols * (sizeof(unsigned long long) + sizeof(uint32_t));
    hipLaunchKernelGGL(cell_reduce_kernel, dim3(B), dim3(256), lds, s, scores, labels, xyxy, qweight, image_set, np, img_w, img_h,
                       grows, gcols, thr, cell_conf, cell_mask, n_kept);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// 1-px rectangles of every kept detection (score > thr) painted in place on u8 images [B,H,W,3]: the device form of
// OWLInterface.bbox_visualization (interface_heuristic.py:259-267) for the searcher's visual history.  Corner
// coordinates are rounded half-to-even and clamped exactly like the host painter (tstar_amd.interface_heuristic.
// draw_boxes); one colour, so the painting order does not matter.
__global__ __launch_bounds__(64) void draw_boxes_kernel(uint8_t* __restrict__ images, int H, int W, const float* __restrict__ xyxy,
                                                        const float* __restrict__ scores, int np, float thr) {
    const int p = blockIdx.x, b = blockIdx.y;
    if (!(scores[(size_t)b * np + p] > thr)) return;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(xyxy + ((size_t)b * np + p) * 4);
    auto pix = [](float v, int hi) { const double r = rint((double)v); return (int)(r < 0.0 ? 0.0 : (r > (double)hi ? (double)hi : r)); };
    const int xa = pix(bb[0], W - 1), ya = pix(bb[1], H - 1), xb = pix(bb[2], W - 1), yb = pix(bb[3], H - 1);
    if (xb < xa || yb < ya) return;
    uint8_t* img = images + (size_t)b * H * W * 3;
    auto put = [&](int x, int y) { uint8_t* q = img + ((size_t)y * W + x) * 3; q[0] = 255; q[1] = 64; q[2] = 64; };
    for (int x = xa + (int)threadIdx.x; x <= xb; x += 64) { put(x, ya); put(x, yb); }
    for (int y = ya + (int)threadIdx.x; y <= yb; y += 64) { put(xa, y); put(xb, y); }
}

int draw_boxes(uint8_t* images, int B, int H, int W, const float* xyxy, const float* scores, int np, float thr, hipStream_t s) {
    TSTAR_REQUIRE(B > 0 && H > 0 && W > 0 && np > 0, "draw_boxes: empty problem");
    hipLaunchKernelGGL(draw_boxes_kernel, dim3(np, B), dim3(64), 0, s, images, H, W, xyxy, scores, np, thr);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

}  // namespace tstar
