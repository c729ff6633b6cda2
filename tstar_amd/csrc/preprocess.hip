// Ingest kernels of the T* hot path (byte work, HBM-bound; no MFMA):
//
//  * Pillow-compatible 8-bit BICUBIC resampling to 768x768, exactly as the HF
//    OWL-ViT image processor applies it to the grid image
//    (/root/reference/TStar/interface_heuristic.py:234 -> HF
//    image_processing_pil_owlvit.py:109-119 -> PIL.Image.resize(BICUBIC);
//    algorithm = Pillow src/libImaging/Resample.c precompute_coeffs /
//    normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc /
//    ImagingResampleVertical_8bpc, restated in SURVEY.md Appendix A2):
//    horizontal pass first, u8 intermediate, 22-bit fixed-point coefficients.
//    The vertical pass is fused with rescale+normalise (a 3x256-entry LUT the
//    host computes with the reference's f64->f32 arithmetic) and with the
//    patch im2col, so the 7 MB fp32 CHW image is never materialised: the
//    output IS the A operand of the patch-embed GEMM.
//  * OpenCV-style INTER_LINEAR 8-bit resize (11-bit coefficients, the
//    HResizeLinear / VResizeLinear<uchar,int,short> fixed-point formulas) for
//    the three cv2.resize call sites of the searcher
//    (/root/reference/TStar/interface_searcher.py:362 -> 800x380, :186 ->
//    200x95, :403 -> 600x285), fused with the frame gather (decord get_batch,
//    :168-169, replaced by a resident decoded-frame store) and the grid tiling
//    (:187-188).  cv2 is not importable in the build container: this bilinear
//    is the build's own definition ("parity unpinned").
#include "common.h"
#include "heads.h"
#include <math.h>
#include <map>
#include <mutex>
#include <vector>

namespace tstar {

// ------------------------------------------------------------------ bicubic tables (host)
static inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

int build_bicubic_table(ResampleTable* t, int in_size, int out_size, hipStream_t s) {
    TSTAR_REQUIRE(in_size > 0 && out_size > 0, "bicubic table: sizes must be positive");
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    std::vector<int> bounds(out_size * 2), coefs((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x] * (double)(1 << 22);
            coefs[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    free_table(t);
    t->in_size = in_size; t->out_size = out_size; t->ksize = ksize;
    TSTAR_HIP_CHECK(hipMalloc(&t->d_bounds, bounds.size() * sizeof(int)));
    TSTAR_HIP_CHECK(hipMalloc(&t->d_coefs, coefs.size() * sizeof(int)));
    TSTAR_HIP_CHECK(hipMemcpy(t->d_bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice));
    TSTAR_HIP_CHECK(hipMemcpy(t->d_coefs, coefs.data(), coefs.size() * sizeof(int), hipMemcpyHostToDevice));
    (void)s;
    return TSTAR_OK;
}

void free_table(ResampleTable* t) {
    if (t->d_bounds) (void)hipFree(t->d_bounds);
    if (t->d_coefs) (void)hipFree(t->d_coefs);
    t->d_bounds = nullptr; t->d_coefs = nullptr;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[b,y,ox,:] = clip8(2^21 + sum_i in[b,y,xmin+i,:] * k[ox][i])
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         int H, int W, int OW, int ksize,
                                                         const int* __restrict__ bounds, const int* __restrict__ coefs,
                                                         size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % OW);
    const size_t by = gid / OW;                     // b*H + y
    const int xmin = bounds[ox * 2], n = bounds[ox * 2 + 1];
    const uint8_t* src = in + (by * W + xmin) * 3;
    const int* k = coefs + (size_t)ox * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int i = 0; i < n; ++i) {
        const int kk = k[i];
        s0 += src[i * 3 + 0] * kk;
        s1 += src[i * 3 + 1] * kk;
        s2 += src[i * 3 + 2] * kk;
    }
    uint8_t* dst = out + gid * 3;
    dst[0] = clip8(s0); dst[1] = clip8(s1); dst[2] = clip8(s2);
}

int resample_h_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, const ResampleTable& t, hipStream_t s) {
    TSTAR_REQUIRE(t.in_size == W, "resample_h_u8: table does not match the input width");
    const size_t total = (size_t)B * H * t.out_size;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, H, W,
                       t.out_size, t.ksize, t.d_bounds, t.d_coefs, total);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// in u8 [B,H,768,3]; vertical pass to 768 rows; LUT normalise; write the patch-embed
// A operand: row = b*576 + (y/32)*24 + x/32, col = c*1024 + (y%32)*32 + x%32.
// One thread per (b, y, x); x fastest -> 32 consecutive threads write 128 contiguous bytes per channel.
__global__ __launch_bounds__(256) void resample_v_patchify_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                                  uint8_t* __restrict__ out_u8, int H, int ksize,
                                                                  const int* __restrict__ bounds,
                                                                  const int* __restrict__ coefs,
                                                                  const float* __restrict__ lut, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = (int)(gid % 768);
    const int y = (int)((gid / 768) % 768);
    const size_t b = gid / (768 * 768);
    const int ymin = bounds[y * 2], n = bounds[y * 2 + 1];
    const uint8_t* src = in + ((b * H + ymin) * 768 + x) * 3;
    const int* k = coefs + (size_t)y * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int i = 0; i < n; ++i) {
        const int kk = k[i];
        const uint8_t* p = src + (size_t)i * 768 * 3;
        s0 += p[0] * kk; s1 += p[1] * kk; s2 += p[2] * kk;
    }
    const uint8_t v0 = clip8(s0), v1 = clip8(s1), v2 = clip8(s2);
    if (out_u8) {
        uint8_t* d = out_u8 + gid * 3;
        d[0] = v0; d[1] = v1; d[2] = v2;
    }
    const size_t row = b * 576 + (size_t)(y >> 5) * 24 + (x >> 5);
    float* o = out + row * 3072 + (y & 31) * 32 + (x & 31);
    o[0] = lut[v0];
    o[1024] = lut[256 + v1];
    o[2048] = lut[512 + v2];
}

int resample_v_normalize_patchify(const uint8_t* in, float* out, uint8_t* out_u8, int B, int H, const ResampleTable& t,
                                  const float* lut, hipStream_t s) {
    TSTAR_REQUIRE(t.in_size == H && t.out_size == 768, "resample_v: table must map H -> 768");
    const size_t total = (size_t)B * 768 * 768;
    hipLaunchKernelGGL(resample_v_patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out,
                       out_u8, H, t.ksize, t.d_bounds, t.d_coefs, lut, total);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// ------------------------------------------------------------------ OpenCV-style bilinear
// Per output index d of an axis: (first source index s0, second source index s1, w0, w1),
// w in 11-bit fixed point.  Computed on the host in the same float/double arithmetic as
// cv::resize's coefficient loop, cached per (src, dst) pair, resident on the device.
struct LinTab { int4* d = nullptr; int n = 0; };
static std::map<std::pair<int, int>, LinTab> g_lintabs;
static std::mutex g_lintab_mu;

static int cv_round_half_even(float v) { return (int)lrintf(v); }
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static int get_lintab(int src, int dst, const int4** out) {
    std::lock_guard<std::mutex> lk(g_lintab_mu);
    auto key = std::make_pair(src, dst);
    auto it = g_lintabs.find(key);
    if (it == g_lintabs.end()) {
        std::vector<int4> h(dst);
        const double scale = (double)src / dst;
        for (int d = 0; d < dst; ++d) {
            float f = (float)((d + 0.5) * scale - 0.5);
            int s = (int)floorf(f);
            f -= (float)s;
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= src - 1) { f = 0.f; s = src - 1; }
            const int s1 = s + 1 < src ? s + 1 : src - 1;
            const int w0 = sat_short(cv_round_half_even((1.f - f) * 2048.f));
            const int w1 = sat_short(cv_round_half_even(f * 2048.f));
            h[d] = make_int4(s, s1, w0, w1);
        }
        LinTab t; t.n = dst;
        TSTAR_HIP_CHECK(hipMalloc(&t.d, dst * sizeof(int4)));
        TSTAR_HIP_CHECK(hipMemcpy(t.d, h.data(), dst * sizeof(int4), hipMemcpyHostToDevice));
        it = g_lintabs.emplace(key, t).first;
    }
    *out = it->second.d;
    return TSTAR_OK;
}

// Source pixel fetch.  RGB frames: interleaved u8 [H, W, 3].  NV12 frames: u8 [H*3/2, W] = luma plane
// followed by the interleaved half-resolution UV plane; converted on the fly with the BT.601
// limited-range integer matrix (298/409/100/208/516, >> 8) and nearest chroma (each 2x2 block shares
// one U,V pair) -- the build's own definition (the reference receives RGB from decord/swscale and never
// sees NV12).  NV12 halves the bytes per resident frame (345,600 B vs 691,200 B at 360x640).
struct Rgb { int r, g, b; };
__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// unaligned wide load (the hardware serves it; one instruction instead of six byte loads)
struct __attribute__((packed)) PackedU64 { uint64_t v; };

struct SrcRGB {
    const uint8_t* p; int W, H;
    __device__ __forceinline__ Rgb at(int x, int y) const {
        const uint8_t* q = p + ((size_t)y * W + x) * 3;
        return Rgb{q[0], q[1], q[2]};
    }
    // pixels x and x + 1 of row y from ONE 8-byte load; caller guarantees x <= W - 3 (the load stays inside the row)
    static constexpr bool kWidePair = true;
    __device__ __forceinline__ void pair(int x, int y, Rgb& a, Rgb& b) const {
        const uint64_t v = reinterpret_cast<const PackedU64*>(p + ((size_t)y * W + x) * 3)->v;
        a = Rgb{(int)(v & 0xFF), (int)((v >> 8) & 0xFF), (int)((v >> 16) & 0xFF)};
        b = Rgb{(int)((v >> 24) & 0xFF), (int)((v >> 32) & 0xFF), (int)((v >> 40) & 0xFF)};
    }
    static __device__ __forceinline__ size_t frame_bytes(int H, int W) { return (size_t)H * W * 3; }
};
struct SrcNV12 {
    const uint8_t* p; int W, H;
    __device__ __forceinline__ Rgb at(int x, int y) const {
        const int c = (int)p[(size_t)y * W + x] - 16;
        const uint8_t* uv = p + (size_t)H * W + (size_t)(y >> 1) * W + (x & ~1);
        const int d = (int)uv[0] - 128, e = (int)uv[1] - 128;
        return Rgb{clip255((298 * c + 409 * e + 128) >> 8), clip255((298 * c - 100 * d - 208 * e + 128) >> 8),
                   clip255((298 * c + 516 * d + 128) >> 8)};
    }
    static constexpr bool kWidePair = false;     // measured: a two-pixel form (u16 luma + u32 chroma) is slower here
    __device__ __forceinline__ void pair(int, int, Rgb&, Rgb&) const {}
    static __device__ __forceinline__ size_t frame_bytes(int H, int W) { return (size_t)H * W * 3 / 2; }
};

// one bilinear sample (all three channels) at output taps tx, ty
template <class SRC>
__device__ __forceinline__ Rgb lin_sample(const SRC& im, const int4 tx, const int4 ty) {
    Rgb a, b, c, d;
    if (SRC::kWidePair && tx.y == tx.x + 1 && tx.x <= im.W - 3) {     // two adjacent source columns, away from the edge
        im.pair(tx.x, ty.x, a, b);
        im.pair(tx.x, ty.y, c, d);
    } else {
        a = im.at(tx.x, ty.x); b = im.at(tx.y, ty.x); c = im.at(tx.x, ty.y); d = im.at(tx.y, ty.y);
    }
    auto mix = [&](int p00, int p01, int p10, int p11) {
        const int h0 = p00 * tx.z + p01 * tx.w;
        const int h1 = p10 * tx.z + p11 * tx.w;
        return (((ty.z * (h0 >> 4)) >> 16) + ((ty.w * (h1 >> 4)) >> 16) + 2) >> 2;
    };
    return Rgb{mix(a.r, b.r, c.r, d.r), mix(a.g, b.g, c.g, d.g), mix(a.b, b.b, c.b, d.b)};
}

template <class SRC>
__global__ __launch_bounds__(256) void bilinear_gather_kernel(const uint8_t* __restrict__ video, int H, int W,
                                                              const int* __restrict__ idx, int ow, int oh,
                                                              const int4* __restrict__ tabx, const int4* __restrict__ taby,
                                                              uint8_t* __restrict__ out, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % ow);
    const int oy = (int)((gid / ow) % oh);
    const int i = (int)(gid / ((size_t)ow * oh));
    SRC im{video + (size_t)idx[i] * SRC::frame_bytes(H, W), W, H};
    const Rgb v = lin_sample(im, tabx[ox], taby[oy]);
    uint8_t* d = out + gid * 3;
    d[0] = (uint8_t)v.r; d[1] = (uint8_t)v.g; d[2] = (uint8_t)v.b;
}

int bilinear_gather_u8(const uint8_t* video, int H, int W, const int* d_idx, int n, int ow, int oh, uint8_t* out,
                       int nv12, hipStream_t s) {
    TSTAR_REQUIRE(n > 0 && ow > 0 && oh > 0, "bilinear_gather_u8: empty output");
    const int4 *tx, *ty;
    int rc = get_lintab(W, ow, &tx); if (rc) return rc;
    rc = get_lintab(H, oh, &ty); if (rc) return rc;
    const size_t total = (size_t)n * ow * oh;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (nv12) hipLaunchKernelGGL(bilinear_gather_kernel<SrcNV12>, grid, dim3(256), 0, s, video, H, W, d_idx, ow, oh, tx, ty, out, total);
    else hipLaunchKernelGGL(bilinear_gather_kernel<SrcRGB>, grid, dim3(256), 0, s, video, H, W, d_idx, ow, oh, tx, ty, out, total);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// frame -> (4cw x 4ch) -> (cw x ch), both bilinear with a u8 round trip in between
// (interface_searcher.py:362 then :186), written straight into its grid cell.
template <class SRC>
__global__ __launch_bounds__(256) void frames_to_grid_kernel(const uint8_t* __restrict__ video, int H, int W,
                                                             const int* __restrict__ idx, int cols, int cw, int ch,
                                                             const int4* __restrict__ t1x, const int4* __restrict__ t1y,
                                                             const int4* __restrict__ t2x, const int4* __restrict__ t2y,
                                                             uint8_t* __restrict__ grid, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % cw);
    const int oy = (int)((gid / cw) % ch);
    const int i = (int)(gid / ((size_t)cw * ch));
    SRC im{video + (size_t)idx[i] * SRC::frame_bytes(H, W), W, H};
    const int4 ax = t2x[ox], ay = t2y[oy];          // taps into the intermediate image
    const int4 x0 = t1x[ax.x], x1 = t1x[ax.y], y0 = t1y[ay.x], y1 = t1y[ay.y];
    const int gr = i / cols, gc = i % cols;
    uint8_t* d = grid + (((size_t)gr * ch + oy) * ((size_t)cols * cw) + (size_t)gc * cw + ox) * 3;
    const Rgb p00 = lin_sample(im, x0, y0), p01 = lin_sample(im, x1, y0);
    const Rgb p10 = lin_sample(im, x0, y1), p11 = lin_sample(im, x1, y1);
    auto mix = [&](int a, int b, int c, int e) {
        const int h0 = a * ax.z + b * ax.w;
        const int h1 = c * ax.z + e * ax.w;
        return (uint8_t)((((ay.z * (h0 >> 4)) >> 16) + ((ay.w * (h1 >> 4)) >> 16) + 2) >> 2);
    };
    d[0] = mix(p00.r, p01.r, p10.r, p11.r);
    d[1] = mix(p00.g, p01.g, p10.g, p11.g);
    d[2] = mix(p00.b, p01.b, p10.b, p11.b);
}

int frames_to_grid_u8(const uint8_t* video, int H, int W, const int* d_idx, int rows, int cols, int cw, int ch,
                      uint8_t* grid, int nv12, hipStream_t s) {
    TSTAR_REQUIRE(rows > 0 && cols > 0 && cw > 0 && ch > 0, "frames_to_grid_u8: empty grid");
    const int4 *t1x, *t1y, *t2x, *t2y;
    int rc = get_lintab(W, 4 * cw, &t1x); if (rc) return rc;
    rc = get_lintab(H, 4 * ch, &t1y); if (rc) return rc;
    rc = get_lintab(4 * cw, cw, &t2x); if (rc) return rc;
    rc = get_lintab(4 * ch, ch, &t2y); if (rc) return rc;
    const size_t total = (size_t)rows * cols * cw * ch;
    const dim3 g((unsigned)((total + 255) / 256));
    if (nv12) hipLaunchKernelGGL(frames_to_grid_kernel<SrcNV12>, g, dim3(256), 0, s, video, H, W, d_idx, cols, cw, ch, t1x, t1y, t2x, t2y, grid, total);
    else hipLaunchKernelGGL(frames_to_grid_kernel<SrcRGB>, g, dim3(256), 0, s, video, H, W, d_idx, cols, cw, ch, t1x, t1y, t2x, t2y, grid, total);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

// native-resolution NV12 -> RGB for the frames handed back to the caller (pop_frames)
__global__ __launch_bounds__(256) void nv12_to_rgb_kernel(const uint8_t* __restrict__ video, int H, int W,
                                                          const int* __restrict__ idx, uint8_t* __restrict__ out, size_t total) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = (int)(gid % W), y = (int)((gid / W) % H), i = (int)(gid / ((size_t)W * H));
    SrcNV12 im{video + (size_t)idx[i] * SrcNV12::frame_bytes(H, W), W, H};
    const Rgb v = im.at(x, y);
    uint8_t* d = out + gid * 3;
    d[0] = (uint8_t)v.r; d[1] = (uint8_t)v.g; d[2] = (uint8_t)v.b;
}

// planar I420 (Y plane, U plane, V plane: what raw 4:2:0 containers such as YUV4MPEG2 carry) -> NV12 (Y plane + interleaved
// UV plane) for n frames; 16 bytes of luma / 8 chroma pairs per lane.  Pure byte movement, HBM-bound.
__global__ __launch_bounds__(256) void i420_to_nv12_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, size_t total16) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total16) return;
    const size_t fb = (size_t)H * W * 3 / 2, per = fb / 16;            // 16-byte units per frame
    const size_t f = gid / per, u = gid % per;
    const uint8_t* src = in + f * fb;
    uint8_t* dst = out + f * fb;
    const size_t ybytes = (size_t)H * W;
    if (u * 16 < ybytes) {
        *reinterpret_cast<uint4*>(dst + u * 16) = *reinterpret_cast<const uint4*>(src + u * 16);
    } else {
        const size_t c0 = (u * 16 - ybytes) / 2;                        // first chroma sample of this unit
        const uint8_t* pu = src + ybytes + c0;
        const uint8_t* pv = src + ybytes + ybytes / 4 + c0;
        const uint2 uu = *reinterpret_cast<const uint2*>(pu), vv = *reinterpret_cast<const uint2*>(pv);
        const uint8_t* ub = reinterpret_cast<const uint8_t*>(&uu);
        const uint8_t* vb = reinterpret_cast<const uint8_t*>(&vv);
        uint8_t o[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[2 * i] = ub[i]; o[2 * i + 1] = vb[i]; }
        *reinterpret_cast<uint4*>(dst + u * 16) = *reinterpret_cast<const uint4*>(o);
    }
}

int i420_to_nv12_u8(const uint8_t* in, int n, int H, int W, uint8_t* out, hipStream_t s) {
    TSTAR_REQUIRE(n > 0 && H % 2 == 0 && W % 2 == 0 && ((size_t)H * W) % 64 == 0, "i420_to_nv12_u8: needs even dimensions with H * W a multiple of 64");
    const size_t total16 = (size_t)n * H * W * 3 / 2 / 16;
    hipLaunchKernelGGL(i420_to_nv12_kernel, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, in, out, H, W, total16);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

int nv12_to_rgb_u8(const uint8_t* video, int H, int W, const int* d_idx, int n, uint8_t* out, hipStream_t s) {
    TSTAR_REQUIRE(n > 0 && H % 2 == 0 && W % 2 == 0, "nv12_to_rgb_u8: NV12 needs even dimensions");
    const size_t total = (size_t)n * H * W;
    hipLaunchKernelGGL(nv12_to_rgb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, video, H, W, d_idx, out, total);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}

}  // namespace tstar
