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
#include <string.h>
#include "heads.h"
#include <math.h>
#include <map>
#include <tuple>
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

// 12 bytes of 4 consecutive RGB pixels -> three aligned dword stores
__device__ __forceinline__ void store_px4_bytes(uint8_t* d, const unsigned (&v)[4][3]) {
    uint3 o;
    o.x = v[0][0] | (v[0][1] << 8) | (v[0][2] << 16) | (v[1][0] << 24);
    o.y = v[1][1] | (v[1][2] << 8) | (v[2][0] << 16) | (v[2][1] << 24);
    o.z = v[2][2] | (v[3][0] << 8) | (v[3][1] << 16) | (v[3][2] << 24);
    *reinterpret_cast<uint3*>(d) = o;
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
        const int kk = k[i];                       // |k| <= 2^22 (1.0 in 22-bit fixed point) and a pixel <= 255: 24-bit operands, so the
        s0 += __mul24(src[i * 3 + 0], kk);         // full-rate v_mad_i32_i24 computes the same 32-bit products as the quarter-rate
        s1 += __mul24(src[i * 3 + 1], kk);         // v_mul_lo_u32 the compiler has to assume
        s2 += __mul24(src[i * 3 + 2], kk);
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
// Round 6: FOUR consecutive x per thread (12 source bytes = three dwords per tap and row instead of twelve byte loads, the products on the
// full-rate v_mad_i32_i24 -- |k| <= 2^22, pixels <= 255: the same 32-bit values --, one float4 store per channel: 4 | 32, so the four stay in
// one patch row), the normalisation LUT in LDS.  Same integers, same LUT entries: bit-exact (tests/test_gpu_detector.py::test_preprocess_*).
__global__ __launch_bounds__(256) void resample_v_patchify_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                                  uint8_t* __restrict__ out_u8, int H, int ksize,
                                                                  const int* __restrict__ bounds,
                                                                  const int* __restrict__ coefs,
                                                                  const float* __restrict__ lut, size_t total4) {
    __shared__ float slut[768];
    for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total4) return;
    const int x = (int)(gid % 192) * 4;
    const int y = (int)((gid / 192) % 768);
    const size_t b = gid / (192 * 768);
    const int ymin = bounds[y * 2], n = bounds[y * 2 + 1];
    const uint8_t* src = in + ((b * H + ymin) * 768 + x) * 3;          // 12 bytes per tap row, dword-aligned (x % 4 == 0, rows of 2304 bytes)
    const int* k = coefs + (size_t)y * ksize;
    int s[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[j][c] = 1 << 21;
    for (int i = 0; i < n; ++i) {
        const int kk = k[i];
        const unsigned* p = reinterpret_cast<const unsigned*>(src + (size_t)i * 768 * 3);
        const unsigned w[3] = {p[0], p[1], p[2]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int byte = 3 * j + c;
                s[j][c] += __mul24((int)((w[byte >> 2] >> ((byte & 3) * 8)) & 0xFFu), kk);
            }
    }
    unsigned v[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[j][c] = clip8(s[j][c]);
    if (out_u8) store_px4_bytes(out_u8 + ((b * 768 + y) * 768 + x) * 3, v);
    const size_t row = b * 576 + (size_t)(y >> 5) * 24 + (x >> 5);
    float* o = out + row * 3072 + (y & 31) * 32 + (x & 31);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = slut[c * 256 + v[j][c]];
        *reinterpret_cast<f32x4*>(o + c * 1024) = q;
    }
}

int resample_v_normalize_patchify(const uint8_t* in, float* out, uint8_t* out_u8, int B, int H, const ResampleTable& t,
                                  const float* lut, hipStream_t s) {
    TSTAR_REQUIRE(t.in_size == H && t.out_size == 768, "resample_v: table must map H -> 768");
    const size_t total4 = (size_t)B * 768 * 192;                        // four consecutive x per thread
    hipLaunchKernelGGL(resample_v_patchify_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, out,
                       out_u8, H, t.ksize, t.d_bounds, t.d_coefs, lut, total4);
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

// ------------------------------------------------------------------ RGB fast path of the two kernels below
// The generic kernels (further down) spend ~500 executed instructions per output pixel, most of them overhead: three runtime integer
// divisions for the index decode, 64-bit address arithmetic per tap, quarter-rate 32-bit multiplies (the compiler cannot
// know the operands are small), byte extraction by shift / mask, two dependent levels of table loads.  For interleaved
// RGB sources the same arithmetic (bit for bit: OpenCV's HResizeLinear / VResizeLinear fixed-point formulas) is restated
// around what the hardware does in one instruction:
//  * the host folds the index tables into one entry per output column / row (FUSED tables): per bilinear sample the BYTE
//    offset of an 8-byte window inside the source row that holds both taps (clamped to the row: the last window ends at
//    the row's last byte, so nothing is read outside the frame), a v_perm_b32 selector that drops the two taps' channel-0
//    bytes into the halves of a dword (channels 1 / 2: selector + 0x00010001 / 0x00020002), the two 11-bit weights packed
//    as u16 pairs, and row byte offsets for the vertical taps;
//  * horizontal mix p0 * w0 + p1 * w1 = v_perm_b32 + v_dot2_u32_u16 per (row, channel);
//  * vertical mix with v_mul_u32_u24 (full rate: operands are 12 and 15 bits) and SDWA word selects for the >> 16;
//  * one frame per blockIdx.y (frame base in SGPRs, 32-bit per-lane offsets), pixel index -> (row, column) by a
//    multiply-high with a host-computed reciprocal.
// frames_to_grid: 8 window loads + ~190 full-rate VALU instructions per output pixel (16 taps x 3 channels through five
// exact fixed-point mixes); bilinear_gather: 4 loads + ~55.  NV12 sources and degenerate sizes (W < 3) keep the generic kernels.
struct FusedTab { uint4* d = nullptr; int n = 0; };
static std::map<std::tuple<int, int, int, int, int>, FusedTab> g_fused;      // (kind, src, mid, dst, row pitch in bytes)

static void fused_x_sample(const int4 t, int W, unsigned* off, unsigned* sel, unsigned* w) {
    const int lim = 3 * W - 8;
    const int o = 3 * t.x < lim ? 3 * t.x : lim;                       // window start (bytes into the row)
    const unsigned oL = (unsigned)(3 * t.x - o), oR = (unsigned)(3 * t.y - o);
    *off = (unsigned)o;
    *sel = oL | 0x0C00u | (oR << 16) | 0x0C000000u;
    *w = (unsigned)t.z | ((unsigned)t.w << 16);
}

// kind 0: X table of a single resize (src -> dst): one uint4 {off, sel, w, 0} per column
// kind 1: Y table of a single resize: one uint4 {row0 bytes, row1 bytes, b0 << 12, b1 << 12} per row (row pitch = pitch bytes)
// kind 2: X table of the two-step resize src -> mid -> dst: two uint4 {offA, selA, wA, offB} {selB, wB, wFinal, 0}
// kind 3: Y table of the two-step resize: three uint4 {r0a, r0b, r1a, r1b} {A.b0, A.b1, B.b0, B.b1} {F.b0, F.b1, 0, 0} (weights << 12)
static int get_fused(int kind, int src, int mid, int dst, int pitch, const uint4** out) {
    std::vector<int4> h1, h2;
    {   // host copies of the plain tables (same arithmetic as get_lintab)
        auto build = [](int s_, int d_, std::vector<int4>& h) {
            h.resize(d_);
            const double scale = (double)s_ / d_;
            for (int d = 0; d < d_; ++d) {
                float f = (float)((d + 0.5) * scale - 0.5);
                int s = (int)floorf(f);
                f -= (float)s;
                if (s < 0) { f = 0.f; s = 0; }
                if (s >= s_ - 1) { f = 0.f; s = s_ - 1; }
                const int s1 = s + 1 < s_ ? s + 1 : s_ - 1;
                h[d] = make_int4(s, s1, sat_short(cv_round_half_even((1.f - f) * 2048.f)), sat_short(cv_round_half_even(f * 2048.f)));
            }
        };
        std::lock_guard<std::mutex> lk(g_lintab_mu);
        const auto key2 = std::make_tuple(kind, src, mid, dst, pitch);
        auto it = g_fused.find(key2);
        if (it == g_fused.end()) {
            std::vector<uint4> h;
            if (kind == 8) {
                // X table of a single resize as THREE ARRAYS (off[], sel[], w[], each padded to a multiple of 4 entries): a lane
                // of the 4-pixel kernel reads its four consecutive entries of one field as ONE 16-byte load, and the 64 lanes of a
                // wave read 1 KB contiguously.  The array-of-uint4 layout (kind 0) made each of a lane's four entry loads touch
                // a different 64-byte line per lane -- 64 lines per wave instruction, 256 per four pixels (round 4: 91.6 -> 74.0 us
                // for 180 verification frames, 2.37 -> 2.93 TB/s)
                build(src, dst, h1);
                const int np = (dst + 3) / 4 * 4;
                std::vector<unsigned> a(3 * (size_t)np, 0u);
                for (int d = 0; d < dst; ++d) { unsigned o, sl, w; fused_x_sample(h1[d], src, &o, &sl, &w); a[d] = o; a[np + d] = sl; a[2 * np + d] = w; }
                h.resize(a.size() / 4);
                memcpy(h.data(), a.data(), a.size() * 4);
            } else if (kind == 0 || kind == 1) {
                build(src, dst, h1);
                h.resize(dst);
                for (int d = 0; d < dst; ++d) {
                    if (kind == 0) { unsigned o, sl, w; fused_x_sample(h1[d], src, &o, &sl, &w); h[d] = make_uint4(o, sl, w, 0); }
                    else h[d] = make_uint4((unsigned)h1[d].x * pitch, (unsigned)h1[d].y * pitch, (unsigned)h1[d].z << 12, (unsigned)h1[d].w << 12);
                }
            } else {
                build(src, mid, h1);
                build(mid, dst, h2);
                const int per = kind == 2 ? 2 : 3;
                h.resize((size_t)per * dst);
                for (int d = 0; d < dst; ++d) {
                    const int4 a = h1[h2[d].x], b = h1[h2[d].y];
                    const unsigned wf = (unsigned)h2[d].z | ((unsigned)h2[d].w << 16);
                    if (kind == 2) {
                        unsigned oa, sa, wa, ob, sb, wb;
                        fused_x_sample(a, src, &oa, &sa, &wa);
                        fused_x_sample(b, src, &ob, &sb, &wb);
                        h[2 * d] = make_uint4(oa, sa, wa, ob);
                        h[2 * d + 1] = make_uint4(sb, wb, wf, 0);
                    } else {
                        h[3 * d] = make_uint4((unsigned)a.x * pitch, (unsigned)a.y * pitch, (unsigned)b.x * pitch, (unsigned)b.y * pitch);
                        h[3 * d + 1] = make_uint4((unsigned)a.z << 12, (unsigned)a.w << 12, (unsigned)b.z << 12, (unsigned)b.w << 12);
                        h[3 * d + 2] = make_uint4((unsigned)h2[d].z << 12, (unsigned)h2[d].w << 12, 0, 0);
                    }
                }
            }
            FusedTab t; t.n = (int)h.size();
            TSTAR_HIP_CHECK(hipMalloc(&t.d, h.size() * sizeof(uint4)));
            TSTAR_HIP_CHECK(hipMemcpy(t.d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
            it = g_fused.emplace(key2, t).first;
        }
        *out = it->second.d;
    }
    return TSTAR_OK;
}

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
// p0 * w0 + p1 * w1 for channel c of the two taps inside the 8-byte window {hi, lo}
__device__ __forceinline__ unsigned hmix(unsigned hi, unsigned lo, unsigned sel, unsigned w) {
    const unsigned pair = __builtin_amdgcn_perm(hi, lo, sel);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, pair), __builtin_bit_cast(u16x2_t, w), 0u, false);
}
// VResizeLinear<uchar, int, short>: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// (b * (S >> 4)) >> 16 == ((b << 12) * (S & ~15)) >> 32 exactly (both factors are below 2^24), which is ONE full-rate
// v_mul_hi_u32_u24 after one AND instead of shift + multiply + shift; the tables carry the weights pre-shifted (B = b << 12).
__device__ __forceinline__ unsigned mulhi24(unsigned a, unsigned b) { unsigned d; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ unsigned vmix(unsigned h0, unsigned h1, unsigned B0, unsigned B1) {
    return (mulhi24(B0, h0 & ~15u) + mulhi24(B1, h1 & ~15u) + 2u) >> 2;
}
__device__ __forceinline__ uint64_t load_window(const uint8_t* p) { return reinterpret_cast<const PackedU64*>(p)->v; }

// 12 output bytes of 4 consecutive pixels -> three aligned dword stores (rows are multiples of 4 pixels; byte stores of
// single channels were the resize kernel's limiter: 3 strided store instructions per pixel)
__device__ __forceinline__ void store_px4(uint8_t* d, const unsigned (&v)[4][3]) {
    uint3 o;
    o.x = v[0][0] | (v[0][1] << 8) | (v[0][2] << 16) | (v[1][0] << 24);
    o.y = v[1][1] | (v[1][2] << 8) | (v[2][0] << 16) | (v[2][1] << 24);
    o.z = v[2][2] | (v[3][0] << 8) | (v[3][1] << 16) | (v[3][2] << 24);
    *reinterpret_cast<uint3*>(d) = o;
}

// PX output pixels of one row per lane (4 when the output width allows it, else 1)
template <int PX>
__global__ __launch_bounds__(256) void bilinear_gather_rgb_kernel(const uint8_t* __restrict__ video, size_t frame_bytes, const int* __restrict__ idx,
                                                                  int ow, int owq, unsigned magic_owq, int nunits, const uint4* __restrict__ fx,
                                                                  const uint4* __restrict__ fy, uint8_t* __restrict__ out) {
    const unsigned u = blockIdx.x * 256u + threadIdx.x;               // unit = PX consecutive pixels of a row
    if (u >= (unsigned)nunits) return;
    const int i = blockIdx.y;
    const uint8_t* f = video + (size_t)idx[i] * frame_bytes;          // wave-uniform
    const unsigned oy = __umulhi(u, magic_owq), ox = (u - oy * (unsigned)owq) * PX;
    const uint4 y = fy[oy];
    const unsigned b0 = y.z, b1 = y.w;
    uint4 x[PX];
    uint64_t w0[PX], w1[PX];
    if constexpr (PX == 4) {
        // fx = three arrays of ow entries (get_fused kind 8): one coalesced 16-byte load per field
        const unsigned* fa = reinterpret_cast<const unsigned*>(fx);
        const uint4 xo = *reinterpret_cast<const uint4*>(fa + ox), xs = *reinterpret_cast<const uint4*>(fa + ow + ox),
                    xw = *reinterpret_cast<const uint4*>(fa + 2 * ow + ox);
        x[0] = make_uint4(xo.x, xs.x, xw.x, 0); x[1 % PX] = make_uint4(xo.y, xs.y, xw.y, 0);
        x[2 % PX] = make_uint4(xo.z, xs.z, xw.z, 0); x[3 % PX] = make_uint4(xo.w, xs.w, xw.w, 0);
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k) x[k] = fx[ox + k];
    }
#pragma unroll
    for (int k = 0; k < PX; ++k) { w0[k] = load_window(f + (size_t)(y.x + x[k].x)); w1[k] = load_window(f + (size_t)(y.y + x[k].x)); }
    unsigned v[PX][3];
#pragma unroll
    for (int k = 0; k < PX; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned sel = x[k].y + 0x00010001u * c;
            v[k][c] = vmix(hmix((unsigned)(w0[k] >> 32), (unsigned)w0[k], sel, x[k].z), hmix((unsigned)(w1[k] >> 32), (unsigned)w1[k], sel, x[k].z), b0, b1);
        }
    uint8_t* d = out + (((size_t)i * (nunits / owq) + oy) * ow + ox) * 3;
    if constexpr (PX == 4) store_px4(d, v);
    else { d[0] = (uint8_t)v[0][0]; d[1] = (uint8_t)v[0][1]; d[2] = (uint8_t)v[0][2]; }
}

// one output pixel of the grid: 8 (or, when the two intermediate rows share their middle source row, 6) window loads, the
// horizontal mixes of the four source rows at the two sample columns, the four samples of the intermediate image (u8 round
// trip, interface_searcher.py:362), then the 4 : 1 step (:186)
template <bool DUP>
__device__ __forceinline__ void grid_pixel(const uint8_t* f, const uint4 yr, const uint4 yw, const uint4 yf, const uint4 xa, const uint4 xb,
                                           unsigned (&v)[3]) {
    const unsigned rows[4] = {yr.x, yr.y, yr.z, yr.w};               // r0a, r0b | r1a, r1b   (DUP: r1a == r0b)
    uint64_t wa[4], wb[4];                                             // windows: four source rows x columns (A, B)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (DUP && r == 2) { wa[2] = wa[1]; wb[2] = wb[1]; continue; }
        wa[r] = load_window(f + (size_t)(rows[r] + xa.x));             // SGPR base + 32-bit lane offset
        wb[r] = load_window(f + (size_t)(rows[r] + xa.w));
    }
    const unsigned fxa = xb.z & 0xFFFFu, fxb = xb.z >> 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned sa = xa.y + 0x00010001u * c, sb = xb.x + 0x00010001u * c;
        unsigned ha[4], hb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (DUP && r == 2) { ha[2] = ha[1]; hb[2] = hb[1]; continue; }
            ha[r] = hmix((unsigned)(wa[r] >> 32), (unsigned)wa[r], sa, xa.z);
            hb[r] = hmix((unsigned)(wb[r] >> 32), (unsigned)wb[r], sb, xb.y);
        }
        const unsigned p00 = vmix(ha[0], ha[1], yw.x, yw.y), p01 = vmix(hb[0], hb[1], yw.x, yw.y);
        const unsigned p10 = vmix(ha[2], ha[3], yw.z, yw.w), p11 = vmix(hb[2], hb[3], yw.z, yw.w);
        const unsigned h0 = __umul24(p00, fxa) + __umul24(p01, fxb), h1 = __umul24(p10, fxa) + __umul24(p11, fxb);
        v[c] = vmix(h0, h1, yf.x, yf.y);
    }
}

template <int PX>
__global__ __launch_bounds__(256) void frames_to_grid_rgb_kernel(const uint8_t* __restrict__ video, size_t frame_bytes, const int* __restrict__ idx,
                                                                 int cols, int cw, int ch, int cwq, unsigned magic_cwq, const uint4* __restrict__ fx,
                                                                 const uint4* __restrict__ fy, uint8_t* __restrict__ grid) {
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= (unsigned)(cwq * ch)) return;
    const int i = blockIdx.y;
    const uint8_t* f = video + (size_t)idx[i] * frame_bytes;          // wave-uniform
    const unsigned oy = __umulhi(u, magic_cwq), ox = (u - oy * (unsigned)cwq) * PX;
    unsigned v[PX][3];
    // A wave covers 64 consecutive pixels of a 200-wide cell row: two waves in three lie inside ONE output row.  Their row
    // table entry then comes through the scalar cache (three s_load_dwordx4 instead of three vector loads per lane), and
    // whether the two intermediate rows share their middle source row -- they do on ~95 % of the rows of a 360 -> 380 -> 95
    // resize -- is a scalar branch that drops two of the eight window loads and a quarter of the horizontal mixes.
    const unsigned oy_u = __builtin_amdgcn_readfirstlane(oy);
    if (PX == 1 && __all(oy == oy_u)) {
        const uint4* q = fy + 3 * oy_u;
        const uint4 yr = q[0], yw = q[1], yf = q[2];
        const uint4 xa = fx[2 * ox], xb = fx[2 * ox + 1];
        if (yr.y == yr.z) grid_pixel<true>(f, yr, yw, yf, xa, xb, v[0]);
        else grid_pixel<false>(f, yr, yw, yf, xa, xb, v[0]);
    } else {
        const uint4 yr = fy[3 * oy], yw = fy[3 * oy + 1], yf = fy[3 * oy + 2];
#pragma unroll
        for (int k = 0; k < PX; ++k) grid_pixel<false>(f, yr, yw, yf, fx[2 * (ox + k)], fx[2 * (ox + k) + 1], v[k]);
    }
    const int gr = i / cols, gc = i - gr * cols;
    uint8_t* d = grid + (((size_t)gr * ch + oy) * ((size_t)cols * cw) + (size_t)gc * cw + ox) * 3;
    if constexpr (PX == 4) store_px4(d, v);
    else { d[0] = (uint8_t)v[0][0]; d[1] = (uint8_t)v[0][1]; d[2] = (uint8_t)v[0][2]; }
}

// ------------------------------------------------------------------ NV12 fast path of the grid kernel
// Same structure for NV12 frame stores (luma plane + interleaved half-resolution UV plane).  The generic kernel issues
// three byte loads per tap (48 per grid pixel); here a (source row, sample) pair is TWO 4-byte windows -- the luma bytes
// of both taps, and the one or two UV pairs they use -- and every tap is converted with the same BT.601 integer matrix
// (SrcNV12::at, bit for bit) before it is mixed: (Y, V) / (Y, U) / (U, V) pairs are dropped into 16-bit halves with
// v_perm_b32 and each channel is one v_dot2_i32_i16 with the constant folded into the accumulator:
//   R = clip((298 Y + 409 V - 56992) >> 8)    B = clip((298 Y + 516 U - 70688) >> 8)
//   G = clip((R_pre - 100 U - 617 V + 91776) >> 8)            [= 298 Y - 100 U - 208 V + 34784]
// X table per output column, three uint4: {oY_A | oC_A << 16, selYU_A.L, selYU_A.R, w_A} {oY_B | oC_B << 16, selYU_B.L,
// selYU_B.R, w_B} {wFinal, 0, 0, 0}; a selector reads {chroma window, luma window} as bytes 4-7 / 0-3:
// selYU = lumaByte | 0x0C00 | (4 + uvByte) << 16 | 0x0C000000.  Y table per output row, four uint4: luma row offsets,
// chroma row offsets (frame-relative bytes), the two pairs of vertical weights << 12, the final pair << 12.
static void fused_x_sample_nv12(const int4 t, int W, unsigned* off, unsigned* selL, unsigned* selR, unsigned* w) {
    const int oy = t.x < W - 4 ? t.x : W - 4;                          // 4-byte luma window holding both taps
    const int cx = t.x & ~1, oc = cx < W - 4 ? cx : W - 4;             // 4-byte chroma window holding both taps' UV pairs
    const unsigned bL = (unsigned)(t.x - oy), bR = (unsigned)(t.y - oy);
    const unsigned cL = (unsigned)((t.x & ~1) - oc), cR = (unsigned)((t.y & ~1) - oc);
    *off = (unsigned)oy | ((unsigned)oc << 16);
    *selL = bL | 0x0C00u | ((4u + cL) << 16) | 0x0C000000u;
    *selR = bR | 0x0C00u | ((4u + cR) << 16) | 0x0C000000u;
    *w = (unsigned)t.z | ((unsigned)t.w << 16);
}

// kind 4: NV12 X table of the two-step resize (3 uint4 per column); kind 5: NV12 Y table (4 uint4 per row);
// kind 6 / 7: X / Y table of a single resize src -> dst (1 uint4 per column {oY | oC << 16, selYU.L, selYU.R, w}; 2 uint4
// per row {luma row 0, luma row 1, chroma row 0, chroma row 1} {b0 << 12, b1 << 12, 0, 0})
static int get_fused_nv12(int kind, int src, int mid, int dst, int W, int H, const uint4** out) {
    auto build = [](int s_, int d_, std::vector<int4>& h) {
        h.resize(d_);
        const double scale = (double)s_ / d_;
        for (int d = 0; d < d_; ++d) {
            float f = (float)((d + 0.5) * scale - 0.5);
            int s = (int)floorf(f);
            f -= (float)s;
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= s_ - 1) { f = 0.f; s = s_ - 1; }
            const int s1 = s + 1 < s_ ? s + 1 : s_ - 1;
            h[d] = make_int4(s, s1, sat_short(cv_round_half_even((1.f - f) * 2048.f)), sat_short(cv_round_half_even(f * 2048.f)));
        }
    };
    std::lock_guard<std::mutex> lk(g_lintab_mu);
    const auto key = std::make_tuple(kind, src, mid, dst, W * 65536 + H);
    auto it = g_fused.find(key);
    if (it == g_fused.end()) {
        std::vector<int4> h1, h2;
        std::vector<uint4> h;
        if (kind == 6 || kind == 7) {
            build(src, dst, h1);
            const unsigned hw = (unsigned)H * W;
            h.resize((kind == 6 ? 1 : 2) * (size_t)dst);
            for (int d = 0; d < dst; ++d) {
                if (kind == 6) {
                    unsigned o, sl, sr, w;
                    fused_x_sample_nv12(h1[d], W, &o, &sl, &sr, &w);
                    h[d] = make_uint4(o, sl, sr, w);
                } else {
                    h[2 * d] = make_uint4((unsigned)h1[d].x * W, (unsigned)h1[d].y * W, hw + (unsigned)(h1[d].x >> 1) * W, hw + (unsigned)(h1[d].y >> 1) * W);
                    h[2 * d + 1] = make_uint4((unsigned)h1[d].z << 12, (unsigned)h1[d].w << 12, 0, 0);
                }
            }
        } else {
            build(src, mid, h1);
            build(mid, dst, h2);
        }
        if (kind == 4) {
            h.resize(3 * (size_t)dst);
            for (int d = 0; d < dst; ++d) {
                unsigned o, sl, sr, w;
                fused_x_sample_nv12(h1[h2[d].x], W, &o, &sl, &sr, &w);
                h[3 * d] = make_uint4(o, sl, sr, w);
                fused_x_sample_nv12(h1[h2[d].y], W, &o, &sl, &sr, &w);
                h[3 * d + 1] = make_uint4(o, sl, sr, w);
                h[3 * d + 2] = make_uint4((unsigned)h2[d].z | ((unsigned)h2[d].w << 16), 0, 0, 0);
            }
        } else if (kind == 5) {
            h.resize(4 * (size_t)dst);
            const unsigned hw = (unsigned)H * W;
            for (int d = 0; d < dst; ++d) {
                const int4 a = h1[h2[d].x], b = h1[h2[d].y];
                h[4 * d] = make_uint4((unsigned)a.x * W, (unsigned)a.y * W, (unsigned)b.x * W, (unsigned)b.y * W);
                h[4 * d + 1] = make_uint4(hw + (unsigned)(a.x >> 1) * W, hw + (unsigned)(a.y >> 1) * W, hw + (unsigned)(b.x >> 1) * W, hw + (unsigned)(b.y >> 1) * W);
                h[4 * d + 2] = make_uint4((unsigned)a.z << 12, (unsigned)a.w << 12, (unsigned)b.z << 12, (unsigned)b.w << 12);
                h[4 * d + 3] = make_uint4((unsigned)h2[d].z << 12, (unsigned)h2[d].w << 12, 0, 0);
            }
        }
        FusedTab t; t.n = (int)h.size();
        TSTAR_HIP_CHECK(hipMalloc(&t.d, h.size() * sizeof(uint4)));
        TSTAR_HIP_CHECK(hipMemcpy(t.d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
        it = g_fused.emplace(key, t).first;
    }
    *out = it->second.d;
    return TSTAR_OK;
}

struct __attribute__((packed)) PackedU32 { unsigned v; };
__device__ __forceinline__ unsigned load_window32(const uint8_t* p) { return reinterpret_cast<const PackedU32*>(p)->v; }
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(unsigned pair, unsigned coef, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2_t, pair), __builtin_bit_cast(i16x2_t, coef), acc, false);
}
__device__ __forceinline__ unsigned clip_s8(int v) { const int t = v >> 8; return (unsigned)(t < 0 ? 0 : (t > 255 ? 255 : t)); }   // v_med3_i32
// one tap of an NV12 frame -> RGB (SrcNV12::at), from the luma / chroma windows and the tap's (Y, U) selector
__device__ __forceinline__ void nv12_tap(unsigned lu, unsigned ch, unsigned sel_yu, unsigned (&rgb)[3]) {
    const unsigned yu = __builtin_amdgcn_perm(ch, lu, sel_yu);                 // Y | U << 16
    const unsigned yv = __builtin_amdgcn_perm(ch, lu, sel_yu + 0x00010000u);   // Y | V << 16
    const unsigned uv = (yu >> 16) | (yv & 0xFFFF0000u);                        // U | V << 16
    const int rp = sdot2(yv, 298u | (409u << 16), -56992);
    const int bp = sdot2(yu, 298u | (516u << 16), -70688);
    const int gp = sdot2(uv, (unsigned)(unsigned short)(-100) | ((unsigned)(unsigned short)(-617) << 16), rp + 91776);
    rgb[0] = clip_s8(rp); rgb[1] = clip_s8(gp); rgb[2] = clip_s8(bp);
}

template <bool DUP>
__device__ __forceinline__ void grid_pixel_nv12(const uint8_t* f, const uint4 yl, const uint4 yc, const uint4 yw, const uint4 yf, const uint4 xa,
                                                const uint4 xb, const unsigned wfx, unsigned (&v)[3]) {
    const unsigned lrow[4] = {yl.x, yl.y, yl.z, yl.w}, crow[4] = {yc.x, yc.y, yc.z, yc.w};
    const unsigned oYa = xa.x & 0xFFFFu, oCa = xa.x >> 16, oYb = xb.x & 0xFFFFu, oCb = xb.x >> 16;
    const unsigned wa0 = xa.w & 0xFFFFu, wa1 = xa.w >> 16, wb0 = xb.w & 0xFFFFu, wb1 = xb.w >> 16;
    unsigned ha[4][3], hb[4][3];                                       // horizontal mixes per source row, sample, channel
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (DUP && r == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { ha[2][c] = ha[1][c]; hb[2][c] = hb[1][c]; }
            continue;
        }
        const unsigned la = load_window32(f + (size_t)(lrow[r] + oYa)), ca = load_window32(f + (size_t)(crow[r] + oCa));
        const unsigned lb = load_window32(f + (size_t)(lrow[r] + oYb)), cb = load_window32(f + (size_t)(crow[r] + oCb));
        unsigned pl[3], pr[3];
        nv12_tap(la, ca, xa.y, pl); nv12_tap(la, ca, xa.z, pr);
#pragma unroll
        for (int c = 0; c < 3; ++c) ha[r][c] = __umul24(pl[c], wa0) + __umul24(pr[c], wa1);
        nv12_tap(lb, cb, xb.y, pl); nv12_tap(lb, cb, xb.z, pr);
#pragma unroll
        for (int c = 0; c < 3; ++c) hb[r][c] = __umul24(pl[c], wb0) + __umul24(pr[c], wb1);
    }
    const unsigned fxa = wfx & 0xFFFFu, fxb = wfx >> 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned p00 = vmix(ha[0][c], ha[1][c], yw.x, yw.y), p01 = vmix(hb[0][c], hb[1][c], yw.x, yw.y);
        const unsigned p10 = vmix(ha[2][c], ha[3][c], yw.z, yw.w), p11 = vmix(hb[2][c], hb[3][c], yw.z, yw.w);
        const unsigned h0 = __umul24(p00, fxa) + __umul24(p01, fxb), h1 = __umul24(p10, fxa) + __umul24(p11, fxb);
        v[c] = vmix(h0, h1, yf.x, yf.y);
    }
}

__global__ __launch_bounds__(256) void frames_to_grid_nv12_kernel(const uint8_t* __restrict__ video, size_t frame_bytes, const int* __restrict__ idx,
                                                                  int cols, int cw, int ch, unsigned magic_cw, const uint4* __restrict__ fx,
                                                                  const uint4* __restrict__ fy, uint8_t* __restrict__ grid) {
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= (unsigned)(cw * ch)) return;
    const int i = blockIdx.y;
    const uint8_t* f = video + (size_t)idx[i] * frame_bytes;          // wave-uniform
    const unsigned oy = __umulhi(u, magic_cw), ox = u - oy * (unsigned)cw;
    const uint4 xa = fx[3 * ox], xb = fx[3 * ox + 1], xf = fx[3 * ox + 2];
    unsigned v[3];
    const unsigned oy_u = __builtin_amdgcn_readfirstlane(oy);
    if (__all(oy == oy_u)) {                                           // the row entry through the scalar cache (see the RGB kernel)
        const uint4* q = fy + 4 * oy_u;
        const uint4 yl = q[0], yc = q[1], yw = q[2], yf = q[3];
        if (yl.y == yl.z) grid_pixel_nv12<true>(f, yl, yc, yw, yf, xa, xb, xf.x, v);
        else grid_pixel_nv12<false>(f, yl, yc, yw, yf, xa, xb, xf.x, v);
    } else {
        grid_pixel_nv12<false>(f, fy[4 * oy], fy[4 * oy + 1], fy[4 * oy + 2], fy[4 * oy + 3], xa, xb, xf.x, v);
    }
    const int gr = i / cols, gc = i - gr * cols;
    uint8_t* d = grid + (((size_t)gr * ch + oy) * ((size_t)cols * cw) + (size_t)gc * cw + ox) * 3;
    d[0] = (uint8_t)v[0]; d[1] = (uint8_t)v[1]; d[2] = (uint8_t)v[2];
}

// NV12 form of the resize kernel: per pixel two rows x (luma window, chroma window), four taps converted, mixed as above
template <int PX>
__global__ __launch_bounds__(256) void bilinear_gather_nv12_kernel(const uint8_t* __restrict__ video, size_t frame_bytes, const int* __restrict__ idx,
                                                                   int ow, int owq, unsigned magic_owq, int nunits, const uint4* __restrict__ fx,
                                                                   const uint4* __restrict__ fy, uint8_t* __restrict__ out) {
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= (unsigned)nunits) return;
    const int i = blockIdx.y;
    const uint8_t* f = video + (size_t)idx[i] * frame_bytes;          // wave-uniform
    const unsigned oy = __umulhi(u, magic_owq), ox = (u - oy * (unsigned)owq) * PX;
    uint4 yr, yw;
    const unsigned oy_u = __builtin_amdgcn_readfirstlane(oy);
    if (__all(oy == oy_u)) { const uint4* q = fy + 2 * oy_u; yr = q[0]; yw = q[1]; }
    else { yr = fy[2 * oy]; yw = fy[2 * oy + 1]; }
    unsigned v[PX][3];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const uint4 x = fx[ox + k];
        const unsigned oY = x.x & 0xFFFFu, oC = x.x >> 16, w0 = x.w & 0xFFFFu, w1 = x.w >> 16;
        unsigned h[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned lu = load_window32(f + (size_t)((r ? yr.y : yr.x) + oY)), cc = load_window32(f + (size_t)((r ? yr.w : yr.z) + oC));
            unsigned pl[3], pr[3];
            nv12_tap(lu, cc, x.y, pl); nv12_tap(lu, cc, x.z, pr);
#pragma unroll
            for (int c = 0; c < 3; ++c) h[r][c] = __umul24(pl[c], w0) + __umul24(pr[c], w1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[k][c] = vmix(h[0][c], h[1][c], yw.x, yw.y);
    }
    uint8_t* d = out + (((size_t)i * (nunits / owq) + oy) * ow + ox) * 3;
    if constexpr (PX == 4) store_px4(d, v);
    else { d[0] = (uint8_t)v[0][0]; d[1] = (uint8_t)v[0][1]; d[2] = (uint8_t)v[0][2]; }
}

// Round 6: the NV12 resize with every SOURCE pixel converted once.  The kernel above converts per tap -- four BT.601 conversions (~13
// VALU each) per output pixel -- although a 360x640 -> 285x600 resize reads only 1.35 source pixels per output pixel.  Here a block
// owns an 8 x 128 tile of the output: it converts the source region the tile's taps fall in (rows ty[first].s0 .. ty[last].s1, columns
// from tx[first].s0 rounded down to a multiple of 4) to packed RGB in LDS -- 4 luma bytes + the 2 UV pairs they share per lane and
// step, one ds_write_b128 -- and then mixes 4 output pixels per lane from LDS with the same fixed-point formulas (v_perm_b32 +
// v_dot2_u32_u16 horizontally, vmix vertically): same bits, ~65 instead of ~110 VALU per output pixel.  Needs W % 4 == 0 and
// ow % 4 == 0 (the launcher falls back to the per-tap kernel otherwise); the region's size is computed on the host per
// (H, W, oh, ow) and bounds the dynamic LDS.
constexpr int NV_TR = 8, NV_TC = 128;
__global__ __launch_bounds__(256) void bilinear_gather_nv12_lds_kernel(const uint8_t* __restrict__ video, size_t frame_bytes, const int* __restrict__ idx,
                                                                       int H, int W, int ow, int oh, int tiles_x, int pitch,
                                                                       const int4* __restrict__ tx, const int4* __restrict__ ty, uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned nv_region[];
    const int t = threadIdx.x;
    const int tile = blockIdx.x, tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int oy0 = tyi * NV_TR, ox0 = txi * NV_TC;
    const int oy1 = (oy0 + NV_TR < oh ? oy0 + NV_TR : oh) - 1, ox1 = (ox0 + NV_TC < ow ? ox0 + NV_TC : ow) - 1;
    const uint8_t* f = video + (size_t)idx[blockIdx.y] * frame_bytes;     // block-uniform
    const int ry0 = ty[oy0].x, ry1 = ty[oy1].y;
    const int rx0 = tx[ox0].x & ~3, rx1 = tx[ox1].y;
    const int nrows = ry1 - ry0 + 1, ncols4 = ((rx1 - rx0) >> 2) + 1;
    // ---- phase 1: source region -> packed RGB (r | g << 8 | b << 16) in LDS
    const int items = nrows * ncols4;
    const float inv = 1.0f / (float)ncols4;
    const size_t hw = (size_t)H * W;
    for (int it = t; it < items; it += 256) {
        const int r = (int)(((float)it + 0.5f) * inv);                    // it / ncols4 (exact: it < 2^13, the half keeps clear of the rounding)
        const int c4 = it - r * ncols4;
        const int sy = ry0 + r, sx = rx0 + 4 * c4;
        const unsigned lu = *reinterpret_cast<const unsigned*>(f + (size_t)sy * W + sx);
        const unsigned ch = *reinterpret_cast<const unsigned*>(f + hw + (size_t)(sy >> 1) * W + sx);
        uint4 o;
        unsigned rgb[3];
        nv12_tap(lu, ch, 0u | 0x0C00u | (4u << 16) | 0x0C000000u, rgb); o.x = rgb[0] | (rgb[1] << 8) | (rgb[2] << 16);
        nv12_tap(lu, ch, 1u | 0x0C00u | (4u << 16) | 0x0C000000u, rgb); o.y = rgb[0] | (rgb[1] << 8) | (rgb[2] << 16);
        nv12_tap(lu, ch, 2u | 0x0C00u | (6u << 16) | 0x0C000000u, rgb); o.z = rgb[0] | (rgb[1] << 8) | (rgb[2] << 16);
        nv12_tap(lu, ch, 3u | 0x0C00u | (6u << 16) | 0x0C000000u, rgb); o.w = rgb[0] | (rgb[1] << 8) | (rgb[2] << 16);
        *reinterpret_cast<uint4*>(nv_region + r * pitch + 4 * c4) = o;
    }
    __syncthreads();
    // ---- phase 2: 4 consecutive output pixels of one row per lane
    const int oy = oy0 + (t >> 5), ox = ox0 + 4 * (t & 31);
    if (oy > oy1 || ox > ox1) return;
    const int4 ye = ty[oy];
    const unsigned* row0 = nv_region + (ye.x - ry0) * pitch - rx0;
    const unsigned* row1 = nv_region + (ye.y - ry0) * pitch - rx0;
    const unsigned b0 = (unsigned)ye.z << 12, b1 = (unsigned)ye.w << 12;
    unsigned v[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int4 xe = tx[ox + k];
        const unsigned w = (unsigned)xe.z | ((unsigned)xe.w << 16);
        const unsigned p00 = row0[xe.x], p01 = row0[xe.y], p10 = row1[xe.x], p11 = row1[xe.y];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned sel = (unsigned)c | 0x0C00u | ((4u + c) << 16) | 0x0C000000u;      // byte c of the left tap | byte c of the right tap << 16
            v[k][c] = vmix(hmix(p01, p00, sel, w), hmix(p11, p10, sel, w), b0, b1);
        }
    }
    store_px4(out + (((size_t)blockIdx.y * oh + oy) * ow + ox) * 3, v);
}

// largest source region (rows, 4-pixel column groups) any 8 x 128 output tile of a (H, W) -> (oh, ow) resize needs
static void nv12_lds_region(int H, int W, int oh, int ow, int* max_rows, int* max_cols4) {
    auto entry = [](int src, int dst, int d, int* s0, int* s1) {
        const double scale = (double)src / dst;
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        if (s < 0) s = 0;
        if (s >= src - 1) s = src - 1;
        *s0 = s; *s1 = s + 1 < src ? s + 1 : src - 1;
    };
    int mr = 0, mc = 0;
    for (int y0 = 0; y0 < oh; y0 += NV_TR) {
        int a, b, c, d;
        entry(H, oh, y0, &a, &b);
        entry(H, oh, (y0 + NV_TR < oh ? y0 + NV_TR : oh) - 1, &c, &d);
        if (d - a + 1 > mr) mr = d - a + 1;
    }
    for (int x0 = 0; x0 < ow; x0 += NV_TC) {
        int a, b, c, d;
        entry(W, ow, x0, &a, &b);
        entry(W, ow, (x0 + NV_TC < ow ? x0 + NV_TC : ow) - 1, &c, &d);
        const int n = ((d - (a & ~3)) >> 2) + 1;
        if (n > mc) mc = n;
    }
    *max_rows = mr; *max_cols4 = mc;
}

// TSTAR_INGEST_GENERIC=1 forces the generic kernels on RGB sources too (before / after counter runs, tools/pmc_ingest_counters.sh)
static bool rgb_fast_ok(int W, long long npix, int div) {
    static const bool generic = [] { const char* e = getenv("TSTAR_INGEST_GENERIC"); return e && atoi(e) != 0; }();
    // div >= 2: magic_of(1) would be 2^32, which does not fit the 32-bit multiplier (the row / column split would read past the tap tables)
    return !generic && W >= 3 && div >= 2 && npix > 0 && npix * (long long)div < (1ll << 32) && npix < (1ll << 31);
}
static unsigned magic_of(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }   // d >= 2 (rgb_fast_ok)

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
    if (!nv12 && n <= 65535 && rgb_fast_ok(W, (long long)ow * oh, ow) && (size_t)H * W * 3 < (1ull << 31)) {
        const uint4 *fx, *fy;
        // 4 pixels per lane need dword-aligned 12-byte stores: a row of the output must be a multiple of 4 pixels
        const int px = (ow % 4 == 0 && ow >= 8 && (reinterpret_cast<size_t>(out) & 3) == 0) ? 4 : 1;   // ow / px >= 2 (magic_of)
        rc = get_fused(px == 4 ? 8 : 0, W, 0, ow, 3 * W, &fx); if (rc) return rc;
        rc = get_fused(1, H, 0, oh, 3 * W, &fy); if (rc) return rc;
        const int owq = ow / px, nunits = owq * oh;
        const dim3 g((unsigned)((nunits + 255) / 256), (unsigned)n);
        if (px == 4) hipLaunchKernelGGL(bilinear_gather_rgb_kernel<4>, g, dim3(256), 0, s, video, (size_t)H * W * 3, d_idx, ow, owq, magic_of(owq), nunits, fx, fy, out);
        else hipLaunchKernelGGL(bilinear_gather_rgb_kernel<1>, g, dim3(256), 0, s, video, (size_t)H * W * 3, d_idx, ow, owq, magic_of(owq), nunits, fx, fy, out);
        TSTAR_HIP_CHECK(hipGetLastError());
        return TSTAR_OK;
    }
    // TSTAR_NV12_LDS=0: the per-tap kernel everywhere (same-session A/Bs)
    static const bool nv12_lds = [] { const char* e = getenv("TSTAR_NV12_LDS"); return !(e && atoi(e) == 0); }();
    if (nv12 && nv12_lds && n <= 65535 && W % 4 == 0 && H % 2 == 0 && ow % 4 == 0 && ow >= 4 && (reinterpret_cast<size_t>(out) & 3) == 0 &&
        (reinterpret_cast<size_t>(video) & 3) == 0 && (size_t)H * W * 3 / 2 < (1ull << 31) && ((size_t)H * W * 3 / 2) % 4 == 0) {
        int mr, mc;
        nv12_lds_region(H, W, oh, ow, &mr, &mc);
        const int pitch = mc * 4 + 4;                                     // dwords; rows stay 16-byte aligned, consecutive rows shifted by 4 banks
        const size_t lds = (size_t)mr * pitch * 4;
        if (lds <= 64 * 1024 && mr * mc < 8192) {
            const int tiles_x = (ow + NV_TC - 1) / NV_TC, tiles_y = (oh + NV_TR - 1) / NV_TR;
            hipLaunchKernelGGL(bilinear_gather_nv12_lds_kernel, dim3((unsigned)(tiles_x * tiles_y), (unsigned)n), dim3(256), lds, s, video,
                               (size_t)H * W * 3 / 2, d_idx, H, W, ow, oh, tiles_x, pitch, tx, ty, out);
            TSTAR_HIP_CHECK(hipGetLastError());
            return TSTAR_OK;
        }
    }
    if (nv12 && n <= 65535 && W >= 4 && W <= 65535 && W % 2 == 0 && H % 2 == 0 && rgb_fast_ok(W, (long long)ow * oh, ow) && (size_t)H * W * 3 / 2 < (1ull << 31)) {
        const uint4 *fx, *fy;
        rc = get_fused_nv12(6, W, 0, ow, W, H, &fx); if (rc) return rc;
        rc = get_fused_nv12(7, H, 0, oh, W, H, &fy); if (rc) return rc;
        const int px = (ow % 4 == 0 && ow >= 8 && (reinterpret_cast<size_t>(out) & 3) == 0) ? 4 : 1;   // ow / px >= 2 (magic_of)
        const int owq = ow / px, nunits = owq * oh;
        const dim3 g((unsigned)((nunits + 255) / 256), (unsigned)n);
        if (px == 4) hipLaunchKernelGGL(bilinear_gather_nv12_kernel<4>, g, dim3(256), 0, s, video, (size_t)H * W * 3 / 2, d_idx, ow, owq, magic_of(owq), nunits, fx, fy, out);
        else hipLaunchKernelGGL(bilinear_gather_nv12_kernel<1>, g, dim3(256), 0, s, video, (size_t)H * W * 3 / 2, d_idx, ow, owq, magic_of(owq), nunits, fx, fy, out);
        TSTAR_HIP_CHECK(hipGetLastError());
        return TSTAR_OK;
    }
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
    if (!nv12 && rows * cols <= 65535 && rgb_fast_ok(W, (long long)cw * ch, cw) && (size_t)H * W * 3 < (1ull << 31)) {
        const uint4 *fx, *fy;
        rc = get_fused(2, W, 4 * cw, cw, 3 * W, &fx); if (rc) return rc;
        rc = get_fused(3, H, 4 * ch, ch, 3 * W, &fy); if (rc) return rc;
        // one pixel per lane here: four (12-byte stores) measured 63 us against 52 for the 256-frame grid -- 32 window loads
        // and five dependent mix levels per lane leave too few lanes in flight; the grid's stores are 8 % of its bytes anyway
        static const int px_env = [] { const char* e = getenv("TSTAR_GRID_PX"); return e ? atoi(e) : 1; }();
        const int px = (px_env == 4 && cw % 4 == 0 && cw >= 8 && (reinterpret_cast<size_t>(grid) & 3) == 0) ? 4 : 1;
        const int cwq = cw / px;
        const dim3 g((unsigned)((cwq * ch + 255) / 256), (unsigned)(rows * cols));
        if (px == 4) hipLaunchKernelGGL(frames_to_grid_rgb_kernel<4>, g, dim3(256), 0, s, video, (size_t)H * W * 3, d_idx, cols, cw, ch, cwq, magic_of(cwq), fx, fy, grid);
        else hipLaunchKernelGGL(frames_to_grid_rgb_kernel<1>, g, dim3(256), 0, s, video, (size_t)H * W * 3, d_idx, cols, cw, ch, cwq, magic_of(cwq), fx, fy, grid);
        TSTAR_HIP_CHECK(hipGetLastError());
        return TSTAR_OK;
    }
    if (nv12 && rows * cols <= 65535 && W >= 4 && W <= 65535 && W % 2 == 0 && H % 2 == 0 && rgb_fast_ok(W, (long long)cw * ch, cw) &&
        (size_t)H * W * 3 / 2 < (1ull << 31)) {
        const uint4 *fx, *fy;
        rc = get_fused_nv12(4, W, 4 * cw, cw, W, H, &fx); if (rc) return rc;
        rc = get_fused_nv12(5, H, 4 * ch, ch, W, H, &fy); if (rc) return rc;
        hipLaunchKernelGGL(frames_to_grid_nv12_kernel, dim3((unsigned)((cw * ch + 255) / 256), (unsigned)(rows * cols)), dim3(256), 0, s, video,
                           (size_t)H * W * 3 / 2, d_idx, cols, cw, ch, magic_of(cw), fx, fy, grid);
        TSTAR_HIP_CHECK(hipGetLastError());
        return TSTAR_OK;
    }
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
