// C ABI of libtstar_hip.so (include/tstar_hip.h): handle management and the
// OWL-ViT-B/32 forward orchestration over the hand-written gfx950 kernels.
#include "../../include/tstar_hip.h"
#include "common.h"
#include "heads.h"
#include "kernels.h"
#include "owl_weights.h"
#include <math.h>
#include <map>
#include <mutex>
#include <set>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace tstar {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

int ensure_dyn_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("ensure_dyn_lds: hipGetDevice failed"); return 2; }
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(kernel, dev);
    if (done.count(key)) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e)); return 2; }
    done.insert(key);
    return 0;
}

// ---------------------------------------------------------------- small text-tower kernels
// x[q*T + t, :] = tok_emb[ids[q,t], :] + pos_emb[t, :]   (OwlViTTextEmbeddings, modeling_owlvit.py:356-372)
__global__ void embed_tokens_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                    const float* __restrict__ pos, float* __restrict__ x, int T, int D) {
    const int r = blockIdx.x;
    const size_t id = (size_t)ids[r];
    const int t = r % T;
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(size_t)r * D + d] = tok[id * D + d] + pos[(size_t)t * D + d];
}
// y[q, :] = x[q*T + eos[q], :]  (EOS pooling, modeling_owlvit.py:651-658)
__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ eos, float* __restrict__ y,
                                   int T, int D) {
    const int q = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) y[(size_t)q * D + d] = x[((size_t)q * T + eos[q]) * D + d];
}
// Wb[i] = bfloat16(W[i]), round to nearest even (exact when W already holds bf16 values);
// optional second term Wlo[i] = bfloat16(W[i] - Wb[i]) (the difference is exact in f32)
__global__ void f32_to_bf16_kernel(const float* __restrict__ W, __bf16* __restrict__ Wb, __bf16* __restrict__ Wlo, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = W[i];
        const __bf16 hi = (__bf16)w;
        Wb[i] = hi;
        if (Wlo) Wlo[i] = (__bf16)(w - (float)hi);
    }
}
int convert_f32_to_bf16(const float* W, __bf16* Wb, __bf16* Wlo, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(1024), dim3(256), 0, s, W, Wb, Wlo, n);
    TSTAR_HIP_CHECK(hipGetLastError());
    return TSTAR_OK;
}
// out[q,:] = in[q,:] / (||in[q,:]|| + eps); one wave per row, D = 512
__global__ void l2norm_rows_kernel(const float* __restrict__ in, float* __restrict__ out, float eps) {
    const int q = blockIdx.x, lane = threadIdx.x;
    float v[8], s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = in[(size_t)q * 512 + i * 64 + lane]; s += v[i] * v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float den = sqrtf(s) + eps;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[(size_t)q * 512 + i * 64 + lane] = v[i] / den;
}
}  // namespace tstar

using namespace tstar;

struct tstar_owl {
    float* d_vision = nullptr;
    float* d_text = nullptr;
    VisionW vw{};
    TextW tw{};
    bool has_text = false, has_vision = false;
    float* d_lut = nullptr;
    int max_batch = 0;
    size_t mpad = 0;
    // activation workspaces (per chunk of `cap` images).  Lane 0 is the handle's own (max_batch images, allocated at creation; the
    // text tower runs in it).  Lane 1 is a SMALL second one, allocated on first use (tstar_owl_score_lane): a forward that runs in it
    // on another stream shares nothing mutable with a forward in lane 0, so the two may execute concurrently (the searcher's
    // speculative next-grid forward, B = 1, beside the verification batch of the iteration before).
    struct Lane {
        float *x = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr;
        uint8_t* tmp_u8 = nullptr; size_t tmp_u8_bytes = 0;
        int* d_image_set = nullptr; int image_set_cap = 0;
        int cap = 0;                                                 // images per forward chunk
    } lane[TSTAR_OWL_LANES];
    // query sets: TSTAR_OWL_MAX_SETS independent (question) slots, each up to 32 queries; every image of
    // a score call names the slot it is scored against (several (video, question) items batched together)
    int Q[TSTAR_OWL_MAX_SETS] = {0};
    float *q_raw = nullptr, *qn = nullptr;                           // [sets][32][512]
    double* qweight = nullptr;                                       // [sets][32] object2weight per query (float64, as the reference's Python floats)
    uint8_t* qmask = nullptr;                                        // [sets][32]
    int* d_setQ = nullptr;
    int *d_ids = nullptr, *d_eos = nullptr;
    uint8_t* d_kmask = nullptr;
    int seq_cap = TSTAR_OWL_MAX_QUERIES;                             // sequences the three staging buffers above hold
    std::map<int, ResampleTable> tabs;   // in_size -> table to 768
    // weights_mode 1 / 3 (BASELINE config 5, bf16 weights; two-term / exact three-term activations): bfloat16 copy of every
    // GEMM weight matrix; weights_mode 4 (f32x3): every f32 matrix as three exact bf16 planes in MFMA-fragment order
    int weights_mode = TSTAR_WEIGHTS_F32;
    std::unordered_map<const float*, __bf16*> wb;
    std::unordered_map<const float*, void*> wp;
    std::unordered_map<const float*, void*> wq;          // two-term mode, the N = 768 matrices: the bf16 plane once more in MFMA-fragment order
    const void* w2_of(const float* w) const {
        if (weights_mode != TSTAR_WEIGHTS_BF16) return nullptr;
        auto it = wq.find(w);
        return it == wq.end() ? nullptr : it->second;
    }
    const __bf16* bf16_of(const float* w) const {
        if (weights_mode != TSTAR_WEIGHTS_BF16 && weights_mode != TSTAR_WEIGHTS_BF16_EXACT) return nullptr;
        auto it = wb.find(w);
        return it == wb.end() ? nullptr : it->second;
    }
    const void* packed_of(const float* w) const {
        if (weights_mode != TSTAR_WEIGHTS_F32X3) return nullptr;
        auto it = wp.find(w);
        return it == wp.end() ? nullptr : it->second;
    }
};

static size_t padded(size_t n) { return (n + 63) / 64 * 64; }

template <class MapFn>
static int upload_blob(const float* h_blob, size_t n_expected_check, float** d_out, MapFn&& mapfn) {
    // pass 1: sizes
    size_t packed = 0, pad_total = 0;
    std::vector<std::pair<size_t, size_t>> ents;   // (packed offset, n)
    auto count = [&](size_t n) -> const float* { ents.push_back({packed, n}); packed += n; pad_total += padded(n); return nullptr; };
    mapfn(count);
    if (packed != n_expected_check) {
        set_error("weight blob has " + std::to_string(n_expected_check) + " floats, layout wants " + std::to_string(packed));
        return TSTAR_ERR_ARG;
    }
    float* d = nullptr;
    TSTAR_HIP_CHECK(hipMalloc(&d, pad_total * sizeof(float)));
    TSTAR_HIP_CHECK(hipMemset(d, 0, pad_total * sizeof(float)));
    size_t off = 0, i = 0;
    int rc = TSTAR_OK;
    auto place = [&](size_t n) -> const float* {
        const float* p = d + off;
        if (hipMemcpy(d + off, h_blob + ents[i].first, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = TSTAR_ERR_HIP;
        off += padded(n); ++i;
        return p;
    };
    mapfn(place);
    if (rc) { set_error("hipMemcpy of weights failed"); (void)hipFree(d); return rc; }
    *d_out = d;
    return TSTAR_OK;
}

static size_t vision_floats() {
    size_t n = 0; VisionW w; map_vision(w, [&](size_t k) -> const float* { n += k; return nullptr; }); return n;
}
static size_t text_floats() {
    size_t n = 0; TextW w; map_text(w, [&](size_t k) -> const float* { n += k; return nullptr; }); return n;
}

static int get_table(tstar_owl* h, int in_size, ResampleTable** out, hipStream_t s) {
    auto it = h->tabs.find(in_size);
    if (it == h->tabs.end()) {
        ResampleTable t;
        int rc = build_bicubic_table(&t, in_size, 768, s);
        if (rc) return rc;
        it = h->tabs.emplace(in_size, t).first;
    }
    *out = &it->second;
    return TSTAR_OK;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

static GemmArgs mk_gemm(const tstar_owl* h, const float* A, const float* W, float* C, const float* bias, const float* res,
                        int M, int N, int K, int lda, int ldc, int act) {
    GemmArgs g{};
    g.A = A; g.W = W; g.Wb = h ? h->bf16_of(W) : nullptr; g.Wp = h ? h->packed_of(W) : nullptr; g.Wq = h ? h->w2_of(W) : nullptr; g.C = C; g.bias = bias; g.res = res; g.pos = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.act = act; g.patch_np = 0; g.tile_cfg = -1; g.m_split = 0;
    g.a_terms = h && h->weights_mode == TSTAR_WEIGHTS_BF16 ? 2 : 0;      // bf16 weights: two-term activations unless the exact mode is asked for
    return g;
}

// TSTAR_X3_ATTN_F32=1: the f32x3 mode with the exact-f32 MFMA attention of rounds 1-4 (same-session A/Bs)
static bool x3_attention_f32() {
    static const bool v = getenv("TSTAR_X3_ATTN_F32") != nullptr;
    return v;
}

// CLIP pre-LN encoder stack shared by both towers; x [M,D] updated in place
static int run_encoder(tstar_owl* h, tstar_owl::Lane& L, const LayerW* layers, int nlayers, int B, int T, int D, int FF, int heads,
                       int mode, const uint8_t* key_mask, hipStream_t s) {
    const int M = B * T;
    for (int l = 0; l < nlayers; ++l) {
        const LayerW& w = layers[l];
        RC(layernorm_f32(L.x, L.xn, w.ln1_w, w.ln1_b, M, D, s));
        RC(gemm_f32(mk_gemm(h, L.xn, w.qkv_w, L.qkv, w.qkv_b, nullptr, M, 3 * D, D, D, 3 * D, ACT_NONE), s));
        // full attention in the bf16-WEIGHT modes runs on the bf16 matrix pipe too (operands as two bf16 terms); in the f32x3
        // mode with all operand bits (three exact terms, six products: its claim is an error no larger than the f32 path's)
        if (mode == 0 && (h->weights_mode == TSTAR_WEIGHTS_BF16 || h->weights_mode == TSTAR_WEIGHTS_BF16_EXACT)) RC(attention_split(L.qkv, L.att, B, T, heads, s));
        else if (mode == 0 && h->weights_mode == TSTAR_WEIGHTS_F32X3 && !x3_attention_f32()) RC(attention_x3(L.qkv, L.att, B, T, heads, s));
        else RC(attention_f32(L.qkv, L.att, B, T, heads, mode, key_mask, s));
        RC(gemm_f32(mk_gemm(h, L.att, w.out_w, L.x, w.out_b, L.x, M, D, D, D, D, ACT_NONE), s));
        RC(layernorm_f32(L.x, L.xn, w.ln2_w, w.ln2_b, M, D, s));
        RC(gemm_f32(mk_gemm(h, L.xn, w.fc1_w, L.hid, w.fc1_b, nullptr, M, FF, D, D, FF, ACT_QGELU), s));
        RC(gemm_f32(mk_gemm(h, L.hid, w.fc2_w, L.x, w.fc2_b, L.x, M, D, FF, FF, D, ACT_NONE), s));
    }
    return TSTAR_OK;
}

static int preprocess_chunk(tstar_owl* h, tstar_owl::Lane& L, const uint8_t* d_images, int B, int H, int W, uint8_t* out_u8,
                            float* out_patches, hipStream_t s) {
    ResampleTable *th, *tv;
    RC(get_table(h, W, &th, s));
    RC(get_table(h, H, &tv, s));
    const size_t need = (size_t)B * H * 768 * 3;
    if (need > L.tmp_u8_bytes) {
        TSTAR_HIP_CHECK(hipStreamSynchronize(s));
        if (L.tmp_u8) TSTAR_HIP_CHECK(hipFree(L.tmp_u8));
        L.tmp_u8 = nullptr; L.tmp_u8_bytes = 0;
        TSTAR_HIP_CHECK(hipMalloc(&L.tmp_u8, need));
        L.tmp_u8_bytes = need;
    }
    RC(resample_h_u8(d_images, L.tmp_u8, B, H, W, *th, s));
    RC(resample_v_normalize_patchify(L.tmp_u8, out_patches, out_u8, B, H, *tv, h->d_lut, s));
    return TSTAR_OK;
}

// One activation workspace for forward chunks of up to `cap` images: x, xn, att [Mp, 768], qkv [Mp, 2304], hid [Mp, 3072] with
// Mp = roundup(cap * 577, 128); zero-filled (rows past M are read by the last GEMM tile of a launch).
static hipError_t alloc_lane(tstar_owl::Lane& L, int cap) {
    const size_t mp = round_up((size_t)cap * V_NTOK, 128);
    hipError_t e = hipSuccess;
    auto alloc = [&](float** p, size_t n) { if (e == hipSuccess) { e = hipMalloc(p, n * sizeof(float)); if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(float)); } };
    alloc(&L.x, mp * V_D); alloc(&L.xn, mp * V_D); alloc(&L.qkv, mp * 3 * V_D); alloc(&L.att, mp * V_D); alloc(&L.hid, mp * V_FF);
    if (e == hipSuccess) L.cap = cap;
    return e;
}
static void free_lane(tstar_owl::Lane& L) {
    void* ptrs[] = {L.x, L.xn, L.qkv, L.att, L.hid, L.tmp_u8, L.d_image_set};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    L = tstar_owl::Lane{};
}

extern "C" {

const char* tstar_last_error(void) { return g_err.c_str(); }
int tstar_abi_version(void) { return 3; }
size_t tstar_owl_vision_blob_floats(void) { return vision_floats(); }
size_t tstar_owl_text_blob_floats(void) { return text_floats(); }

static int make_bf16_copies(tstar_owl* h, int mode) {
    struct Mat { const float* w; int n, k; };
    std::vector<Mat> mats;
    mats.push_back({h->vw.patch_w, V_D, V_PATCH_K});
    auto layer = [&](const LayerW& l, int d, int ff) {
        mats.push_back({l.qkv_w, 3 * d, d}); mats.push_back({l.out_w, d, d});
        mats.push_back({l.fc1_w, ff, d}); mats.push_back({l.fc2_w, d, ff});
    };
    for (int i = 0; i < V_LAYERS; ++i) layer(h->vw.layers[i], V_D, V_FF);
    mats.push_back({h->vw.cls_w, PROJ, V_D});
    mats.push_back({h->vw.box0_w, V_D, V_D});
    mats.push_back({h->vw.box1_w, V_D, V_D});
    if (h->has_text) {
        for (int i = 0; i < T_LAYERS; ++i) layer(h->tw.layers[i], T_D, T_FF);
        mats.push_back({h->tw.text_proj, PROJ, T_D});
    }
    for (auto& m : mats) {
        const size_t n = (size_t)m.n * m.k;
        int rc;
        if (mode == TSTAR_WEIGHTS_F32X3) {
            void* p = nullptr;
            TSTAR_HIP_CHECK(hipMalloc(&p, n * 6));
            h->wp[m.w] = p;
            rc = pack_weights_x3(m.w, p, m.n, m.k, 0);
        } else {
            __bf16* p = nullptr;
            TSTAR_HIP_CHECK(hipMalloc(&p, n * sizeof(__bf16)));
            h->wb[m.w] = p;
            rc = convert_f32_to_bf16(m.w, p, nullptr, n, 0);
            static const bool w2v_off = getenv("TSTAR_W2V_OFF") != nullptr;           // same-session A/Bs: every layer on the LDS tile
            if (!rc && mode == TSTAR_WEIGHTS_BF16 && m.n == 768 && m.k % 32 == 0 && !w2v_off) {       // per-shape dispatch (gemm_f32.hip launch_mode)
                void* q = nullptr;
                TSTAR_HIP_CHECK(hipMalloc(&q, n * sizeof(__bf16)));
                h->wq[m.w] = q;
                rc = pack_weights_w2(p, q, m.n, m.k, 0);
            }
        }
        if (rc) return rc;
    }
    TSTAR_HIP_CHECK(hipDeviceSynchronize());
    h->weights_mode = mode;
    return TSTAR_OK;
}

int tstar_owl_create(tstar_owl** out, const float* h_vision_blob, size_t n_vision, const float* h_text_blob,
                     size_t n_text, const float* h_norm_lut, int max_batch, int weights_mode) {
    TSTAR_REQUIRE(out && (h_vision_blob || h_text_blob), "tstar_owl_create: null argument");
    TSTAR_REQUIRE(!h_vision_blob || h_norm_lut, "tstar_owl_create: the vision tower needs the normalisation LUT");
    TSTAR_REQUIRE(h_vision_blob || weights_mode == TSTAR_WEIGHTS_F32, "tstar_owl_create: a text-only handle runs in float32");
    TSTAR_REQUIRE(max_batch >= 1 && max_batch <= 1024, "tstar_owl_create: max_batch must be in 1..1024");
    TSTAR_REQUIRE(weights_mode == TSTAR_WEIGHTS_F32 || weights_mode == TSTAR_WEIGHTS_BF16 || weights_mode == TSTAR_WEIGHTS_BF16_EXACT ||
                      weights_mode == TSTAR_WEIGHTS_F32X3,
                  "tstar_owl_create: weights_mode must be 0 (f32), 1 (bf16), 3 (bf16, exact three-term activations) or 4 (f32x3); 2 (f32 split) was retired in ABI 3");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("tstar_owl_create: no HIP device visible (this library has no CPU path)");
        return TSTAR_ERR_HIP;
    }
    tstar_owl* h = new tstar_owl();
    int rc = TSTAR_OK;
    if (h_vision_blob) {               // NULL: a text-only handle (CLIP text features for the YOLO-World backend)
        rc = upload_blob(h_vision_blob, n_vision, &h->d_vision, [&](auto&& take) { map_vision(h->vw, take); });
        if (rc) { delete h; return rc; }
        h->has_vision = true;
    }
    if (h_text_blob) {
        rc = upload_blob(h_text_blob, n_text, &h->d_text, [&](auto&& take) { map_text(h->tw, take); });
        if (rc) { tstar_owl_destroy(h); return rc; }
        h->has_text = true;
    }
    h->max_batch = max_batch;
    h->mpad = round_up((size_t)max_batch * V_NTOK, 128);
    hipError_t e = hipSuccess;
    auto alloc = [&](float** p, size_t n) { if (e == hipSuccess) { e = hipMalloc(p, n * sizeof(float)); if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(float)); } };
    e = alloc_lane(h->lane[0], max_batch);
    alloc(&h->d_lut, 768);
    constexpr int NSQ = TSTAR_OWL_MAX_SETS * TSTAR_OWL_MAX_QUERIES;
    alloc(&h->q_raw, (size_t)NSQ * PROJ); alloc(&h->qn, (size_t)NSQ * PROJ);
    if (e == hipSuccess) e = hipMalloc(&h->qweight, NSQ * sizeof(double));
    if (e == hipSuccess) e = hipMemset(h->qweight, 0, NSQ * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->qmask, NSQ);
    if (e == hipSuccess) e = hipMemset(h->qmask, 0, NSQ);
    if (e == hipSuccess) e = hipMalloc(&h->d_setQ, TSTAR_OWL_MAX_SETS * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->d_setQ, 0, TSTAR_OWL_MAX_SETS * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_ids, TSTAR_OWL_MAX_QUERIES * T_LEN * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_eos, TSTAR_OWL_MAX_QUERIES * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_kmask, TSTAR_OWL_MAX_QUERIES * T_LEN);
    if (e == hipSuccess && h_norm_lut) e = hipMemcpy(h->d_lut, h_norm_lut, 768 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error(std::string("tstar_owl_create: workspace allocation failed: ") + hipGetErrorString(e));
        tstar_owl_destroy(h);
        return TSTAR_ERR_HIP;
    }
    if (weights_mode != TSTAR_WEIGHTS_F32) {
        rc = make_bf16_copies(h, weights_mode);
        if (rc) { tstar_owl_destroy(h); return rc; }
    }
    *out = h;
    return TSTAR_OK;
}

int tstar_owl_destroy(tstar_owl* h) {
    if (!h) return TSTAR_OK;
    void* ptrs[] = {h->d_vision, h->d_text, h->d_lut, h->q_raw, h->qn, h->qweight, h->qmask, h->d_ids, h->d_eos, h->d_kmask, h->d_setQ};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& L : h->lane) free_lane(L);
    for (auto& kv : h->tabs) free_table(&kv.second);
    for (auto& kv : h->wb) if (kv.second) (void)hipFree(kv.second);
    for (auto& kv : h->wp) if (kv.second) (void)hipFree(kv.second);
    for (auto& kv : h->wq) if (kv.second) (void)hipFree(kv.second);
    delete h;
    return TSTAR_OK;
}

#define CHECK_SET(set, fn) TSTAR_REQUIRE((set) >= 0 && (set) < TSTAR_OWL_MAX_SETS, fn ": query_set must be in 0..63")

static int finish_queries(tstar_owl* h, int set, const uint8_t* h_mask, const double* h_w, int Q, hipStream_t s) {
    const size_t qo = (size_t)set * TSTAR_OWL_MAX_QUERIES;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(Q), dim3(64), 0, s, h->q_raw + qo * PROJ, h->qn + qo * PROJ, 1e-6f);
    TSTAR_HIP_CHECK(hipGetLastError());
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->qmask + qo, h_mask, Q, hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->qweight + qo, h_w, Q * sizeof(double), hipMemcpyHostToDevice, s));
    h->Q[set] = Q;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_setQ, h->Q, sizeof(h->Q), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}

int tstar_owl_set_queries(tstar_owl* h, int query_set, const int32_t* h_ids, const int32_t* h_am, const double* h_w, int Q,
                          void* stream) {
    TSTAR_REQUIRE(h && h_ids && h_am && h_w, "tstar_owl_set_queries: null argument");
    CHECK_SET(query_set, "tstar_owl_set_queries");
    TSTAR_REQUIRE(Q >= 1 && Q <= TSTAR_OWL_MAX_QUERIES, "tstar_owl_set_queries: Q must be in 1..32");
    if (!h->has_text) { set_error("tstar_owl_set_queries: handle was created without text weights"); return TSTAR_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    auto& L = h->lane[0];                                      // the text tower runs in the handle's own workspace
    std::vector<int> eos(Q);
    std::vector<uint8_t> km(Q * T_LEN), qm(Q);
    for (int q = 0; q < Q; ++q) {
        int best = 0;
        for (int t = 0; t < T_LEN; ++t) {
            const int id = h_ids[q * T_LEN + t];
            TSTAR_REQUIRE(id >= 0 && id < T_VOCAB, "tstar_owl_set_queries: token id out of range");
            if (id > h_ids[q * T_LEN + best]) best = t;        // argmax, first occurrence
            km[q * T_LEN + t] = h_am[q * T_LEN + t] != 0;
        }
        eos[q] = best;
        qm[q] = h_ids[q * T_LEN] > 0;                          // modeling_owlvit.py:1447
    }
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_ids, h_ids, Q * T_LEN * sizeof(int), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_eos, eos.data(), Q * sizeof(int), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_kmask, km.data(), Q * T_LEN, hipMemcpyHostToDevice, s));
    const int M = Q * T_LEN;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(M), dim3(128), 0, s, h->d_ids, h->tw.tok_emb, h->tw.tpos_emb, L.x,
                       T_LEN, T_D);
    TSTAR_HIP_CHECK(hipGetLastError());
    RC(run_encoder(h, L, h->tw.layers, T_LAYERS, Q, T_LEN, T_D, T_FF, T_HEADS, 1, h->d_kmask, s));
    RC(layernorm_f32(L.x, L.xn, h->tw.final_ln_w, h->tw.final_ln_b, M, T_D, s));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(Q), dim3(128), 0, s, L.xn, h->d_eos, L.att, T_LEN, T_D);
    TSTAR_HIP_CHECK(hipGetLastError());
    RC(gemm_f32(mk_gemm(h, L.att, h->tw.text_proj, L.hid, nullptr, nullptr, Q, PROJ, T_D, T_D, PROJ, ACT_NONE), s));
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(Q), dim3(64), 0, s, L.hid,
                       h->q_raw + (size_t)query_set * TSTAR_OWL_MAX_QUERIES * PROJ, 0.0f);
    TSTAR_HIP_CHECK(hipGetLastError());
    return finish_queries(h, query_set, qm.data(), h_w, Q, s);
}

int tstar_owl_set_queries_many(tstar_owl* h, int n_sets, const int32_t* h_sets, const int32_t* h_Q, const int32_t* h_ids, const int32_t* h_am,
                               const double* h_w, void* stream) {
    TSTAR_REQUIRE(h && h_sets && h_Q && h_ids && h_am && h_w, "tstar_owl_set_queries_many: null argument");
    TSTAR_REQUIRE(n_sets >= 1 && n_sets <= TSTAR_OWL_MAX_SETS, "tstar_owl_set_queries_many: n_sets must be in 1..64");
    if (!h->has_text) { set_error("tstar_owl_set_queries_many: handle was created without text weights"); return TSTAR_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    auto& L = h->lane[0];
    int total = 0;
    for (int i = 0; i < n_sets; ++i) {
        CHECK_SET(h_sets[i], "tstar_owl_set_queries_many");
        TSTAR_REQUIRE(h_Q[i] >= 1 && h_Q[i] <= TSTAR_OWL_MAX_QUERIES, "tstar_owl_set_queries_many: every Q must be in 1..32");
        total += h_Q[i];
    }
    for (int q = 0; q < total * T_LEN; ++q) TSTAR_REQUIRE(h_ids[q] >= 0 && h_ids[q] < T_VOCAB, "tstar_owl_set_queries_many: token id out of range");
    // sequences per text forward: what the activation workspace holds (mpad rows of >= T_D floats; T_LEN rows per sequence)
    const int cap = (int)(h->mpad / T_LEN);
    TSTAR_REQUIRE(cap >= TSTAR_OWL_MAX_QUERIES, "tstar_owl_set_queries_many: workspace too small");
    if (h->seq_cap < cap) {                                   // staging for ids / EOS positions / key masks of one forward
        TSTAR_HIP_CHECK(hipStreamSynchronize(s));
        if (h->d_ids) (void)hipFree(h->d_ids);
        if (h->d_eos) (void)hipFree(h->d_eos);
        if (h->d_kmask) (void)hipFree(h->d_kmask);
        h->d_ids = nullptr; h->d_eos = nullptr; h->d_kmask = nullptr; h->seq_cap = 0;
        TSTAR_HIP_CHECK(hipMalloc(&h->d_ids, (size_t)cap * T_LEN * sizeof(int)));
        TSTAR_HIP_CHECK(hipMalloc(&h->d_eos, (size_t)cap * sizeof(int)));
        TSTAR_HIP_CHECK(hipMalloc(&h->d_kmask, (size_t)cap * T_LEN));
        h->seq_cap = cap;
    }
    std::vector<int> eos(total);
    std::vector<uint8_t> km((size_t)total * T_LEN), qm(total);
    for (int q = 0; q < total; ++q) {
        int best = 0;
        for (int t = 0; t < T_LEN; ++t) {
            if (h_ids[q * T_LEN + t] > h_ids[q * T_LEN + best]) best = t;   // argmax, first occurrence
            km[(size_t)q * T_LEN + t] = h_am[q * T_LEN + t] != 0;
        }
        eos[q] = best;
        qm[q] = h_ids[q * T_LEN] > 0;
    }
    // ONE text forward per group of sets that fits the workspace: the same kernels as tstar_owl_set_queries on more rows (every
    // GEMM tile shape and the per-sequence causal attention give the same bits whatever the batch: results equal one-by-one calls)
    int i0 = 0, q0 = 0;
    while (i0 < n_sets) {
        int i1 = i0, nseq = 0;
        while (i1 < n_sets && nseq + h_Q[i1] <= cap) nseq += h_Q[i1++];
        const int M = nseq * T_LEN;
        TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_ids, h_ids + (size_t)q0 * T_LEN, (size_t)M * sizeof(int), hipMemcpyHostToDevice, s));
        TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_eos, eos.data() + q0, nseq * sizeof(int), hipMemcpyHostToDevice, s));
        TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_kmask, km.data() + (size_t)q0 * T_LEN, M, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(embed_tokens_kernel, dim3(M), dim3(128), 0, s, h->d_ids, h->tw.tok_emb, h->tw.tpos_emb, L.x, T_LEN, T_D);
        TSTAR_HIP_CHECK(hipGetLastError());
        RC(run_encoder(h, L, h->tw.layers, T_LAYERS, nseq, T_LEN, T_D, T_FF, T_HEADS, 1, h->d_kmask, s));
        RC(layernorm_f32(L.x, L.xn, h->tw.final_ln_w, h->tw.final_ln_b, M, T_D, s));
        hipLaunchKernelGGL(gather_rows_kernel, dim3(nseq), dim3(128), 0, s, L.xn, h->d_eos, L.att, T_LEN, T_D);
        TSTAR_HIP_CHECK(hipGetLastError());
        RC(gemm_f32(mk_gemm(h, L.att, h->tw.text_proj, L.hid, nullptr, nullptr, nseq, PROJ, T_D, T_D, PROJ, ACT_NONE), s));
        int off = 0;
        for (int i = i0; i < i1; ++i) {
            const int set = h_sets[i], Q = h_Q[i];
            const size_t qo = (size_t)set * TSTAR_OWL_MAX_QUERIES;
            hipLaunchKernelGGL(l2norm_rows_kernel, dim3(Q), dim3(64), 0, s, L.hid + (size_t)off * PROJ, h->q_raw + qo * PROJ, 0.0f);
            hipLaunchKernelGGL(l2norm_rows_kernel, dim3(Q), dim3(64), 0, s, h->q_raw + qo * PROJ, h->qn + qo * PROJ, 1e-6f);
            TSTAR_HIP_CHECK(hipGetLastError());
            TSTAR_HIP_CHECK(hipMemcpyAsync(h->qmask + qo, qm.data() + q0 + off, Q, hipMemcpyHostToDevice, s));
            TSTAR_HIP_CHECK(hipMemcpyAsync(h->qweight + qo, h_w + q0 + off, Q * sizeof(double), hipMemcpyHostToDevice, s));
            h->Q[set] = Q;
            off += Q;
        }
        // the staging buffers are reused by the next group: wait for this one
        TSTAR_HIP_CHECK(hipStreamSynchronize(s));
        q0 += nseq;
        i0 = i1;
    }
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->d_setQ, h->Q, sizeof(h->Q), hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}

int tstar_owl_set_query_embeds(tstar_owl* h, int query_set, const float* h_qe, const uint8_t* h_mask, const double* h_w,
                               int Q, void* stream) {
    TSTAR_REQUIRE(h && h_qe && h_mask && h_w, "tstar_owl_set_query_embeds: null argument");
    CHECK_SET(query_set, "tstar_owl_set_query_embeds");
    TSTAR_REQUIRE(Q >= 1 && Q <= TSTAR_OWL_MAX_QUERIES, "tstar_owl_set_query_embeds: Q must be in 1..32");
    hipStream_t s = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->q_raw + (size_t)query_set * TSTAR_OWL_MAX_QUERIES * PROJ, h_qe,
                                   (size_t)Q * PROJ * sizeof(float), hipMemcpyHostToDevice, s));
    return finish_queries(h, query_set, h_mask, h_w, Q, s);
}

int tstar_owl_set_class_weights(tstar_owl* h, int query_set, const double* h_w, int Q, void* stream) {
    TSTAR_REQUIRE(h && h_w, "tstar_owl_set_class_weights: null argument");
    CHECK_SET(query_set, "tstar_owl_set_class_weights");
    TSTAR_REQUIRE(Q == h->Q[query_set] && Q >= 1, "tstar_owl_set_class_weights: Q does not match the installed queries");
    hipStream_t s = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h->qweight + (size_t)query_set * TSTAR_OWL_MAX_QUERIES, h_w, Q * sizeof(double),
                                   hipMemcpyHostToDevice, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}

int tstar_owl_get_query_embeds(tstar_owl* h, int query_set, float* h_out, int Q, void* stream) {
    TSTAR_REQUIRE(h && h_out, "tstar_owl_get_query_embeds: null argument");
    CHECK_SET(query_set, "tstar_owl_get_query_embeds");
    TSTAR_REQUIRE(Q == h->Q[query_set], "tstar_owl_get_query_embeds: Q does not match the installed queries");
    hipStream_t s = (hipStream_t)stream;
    TSTAR_HIP_CHECK(hipMemcpyAsync(h_out, h->q_raw + (size_t)query_set * TSTAR_OWL_MAX_QUERIES * PROJ,
                                   (size_t)Q * PROJ * sizeof(float), hipMemcpyDeviceToHost, s));
    TSTAR_HIP_CHECK(hipStreamSynchronize(s));
    return TSTAR_OK;
}

int tstar_owl_score(tstar_owl* h, const uint8_t* d_images, int B, int H, int W, int grid_rows, int grid_cols,
                    const int32_t* h_image_query_set, float* d_scores, int32_t* d_labels, float* d_boxes_xyxy, double* d_cell_conf,
                    uint32_t* d_cell_mask, int32_t* d_n_kept, float* d_logits, float* d_boxes_cxcywh, void* stream) {
    return tstar_owl_score_lane(h, 0, d_images, B, H, W, grid_rows, grid_cols, h_image_query_set, d_scores, d_labels, d_boxes_xyxy, d_cell_conf,
                                d_cell_mask, d_n_kept, d_logits, d_boxes_cxcywh, stream);
}

int tstar_owl_score_lane(tstar_owl* h, int lane, const uint8_t* d_images, int B, int H, int W, int grid_rows, int grid_cols,
                         const int32_t* h_image_query_set, float* d_scores, int32_t* d_labels, float* d_boxes_xyxy, double* d_cell_conf,
                         uint32_t* d_cell_mask, int32_t* d_n_kept, float* d_logits, float* d_boxes_cxcywh, void* stream) {
    TSTAR_REQUIRE(h && d_images && d_scores && d_labels && d_boxes_xyxy && d_cell_conf && d_cell_mask,
                  "tstar_owl_score: null argument");
    TSTAR_REQUIRE(lane >= 0 && lane < TSTAR_OWL_LANES, "tstar_owl_score_lane: lane must be 0 or 1");
    TSTAR_REQUIRE(B >= 1 && H >= 1 && W >= 1, "tstar_owl_score: empty batch or image");
    TSTAR_REQUIRE(grid_rows >= 1 && grid_cols >= 1, "tstar_owl_score: grid must be at least 1x1");
    if (!h->has_vision) { set_error("tstar_owl_score: handle was created without vision weights (text-only)"); return TSTAR_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    auto& L = h->lane[lane];
    if (lane != 0) {
        // lane 1: allocated on first use (a one-off, like the resample tables) for forward chunks of min(max_batch, max(TSTAR_OWL_AUX_BATCH, B))
        // images, and grown when a larger batch arrives (the device is drained first)
        int need = B > TSTAR_OWL_AUX_BATCH ? B : TSTAR_OWL_AUX_BATCH;
        if (need > h->max_batch) need = h->max_batch;
        if (!L.x || L.cap < need) {
            if (L.x) { TSTAR_HIP_CHECK(hipDeviceSynchronize()); free_lane(L); }      // (rare: whichever stream used the smaller workspace last)
            const hipError_t e = alloc_lane(L, need);
            if (e != hipSuccess) {
                free_lane(L);
                set_error(std::string("tstar_owl_score_lane: workspace allocation failed: ") + hipGetErrorString(e));
                return TSTAR_ERR_HIP;
            }
            TSTAR_HIP_CHECK(hipDeviceSynchronize());      // the zero fill ran on the null stream
        }
    }
    int q_uniform = -1;                                   // the common Q when every image uses one set size
    for (int b = 0; b < B; ++b) {
        const int set = h_image_query_set ? h_image_query_set[b] : 0;
        CHECK_SET(set, "tstar_owl_score");
        if (h->Q[set] == 0) { set_error("tstar_owl_score: no queries installed in the requested query set (call tstar_owl_set_queries first)"); return TSTAR_ERR_STATE; }
        q_uniform = (b == 0 || q_uniform == h->Q[set]) ? h->Q[set] : 0;
    }
    TSTAR_REQUIRE(!d_logits || q_uniform > 0, "tstar_owl_score: raw logits need the same query count for every image");
    if (h_image_query_set) {
        if (B > L.image_set_cap) {
            TSTAR_HIP_CHECK(hipStreamSynchronize(s));
            if (L.d_image_set) TSTAR_HIP_CHECK(hipFree(L.d_image_set));
            L.d_image_set = nullptr; L.image_set_cap = 0;
            TSTAR_HIP_CHECK(hipMalloc(&L.d_image_set, (size_t)B * sizeof(int)));
            L.image_set_cap = B;
        }
        TSTAR_HIP_CHECK(hipMemcpyAsync(L.d_image_set, h_image_query_set, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
    }
    const int ncell = grid_rows * grid_cols;
    for (int b0 = 0; b0 < B; b0 += L.cap) {
        const int Bc = (B - b0) < L.cap ? (B - b0) : L.cap;
        const int M = Bc * V_NTOK, MP = Bc * V_NP;
        RC(preprocess_chunk(h, L, d_images + (size_t)b0 * H * W * 3, Bc, H, W, nullptr, L.hid, s));
        GemmArgs pg = mk_gemm(h, L.hid, h->vw.patch_w, L.x, nullptr, nullptr, MP, V_D, V_PATCH_K, V_PATCH_K, V_D, ACT_NONE);
        pg.pos = h->vw.pos_emb; pg.patch_np = V_NP;
        RC(gemm_f32(pg, s));
        RC(write_cls_rows(L.x, h->vw.class_emb, h->vw.pos_emb, Bc, V_NTOK, V_D, s));
        RC(layernorm_f32(L.x, L.x, h->vw.pre_ln_w, h->vw.pre_ln_b, M, V_D, s));
        RC(run_encoder(h, L, h->vw.layers, V_LAYERS, Bc, V_NTOK, V_D, V_FF, V_HEADS, 0, nullptr, s));
        float* feats = L.xn;
        RC(merge_cls_ln(L.x, feats, h->vw.post_ln_w, h->vw.post_ln_b, h->vw.det_ln_w, h->vw.det_ln_b, Bc, V_NTOK, V_D, s));
        float* cls = L.att;      // [MP, 512]
        float* bh1 = L.qkv;      // [MP, 768]
        float* bh2 = L.hid;      // [MP, 768]
        RC(gemm_f32(mk_gemm(h, feats, h->vw.cls_w, cls, h->vw.cls_b, nullptr, MP, PROJ, V_D, V_D, PROJ, ACT_NONE), s));
        RC(gemm_f32(mk_gemm(h, feats, h->vw.box0_w, bh1, h->vw.box0_b, nullptr, MP, V_D, V_D, V_D, V_D, ACT_GELU), s));
        RC(gemm_f32(mk_gemm(h, bh1, h->vw.box1_w, bh2, h->vw.box1_b, nullptr, MP, V_D, V_D, V_D, V_D, ACT_GELU), s));
        DetectRowsArgs a{};
        a.feats = feats; a.cls = cls; a.boxh = bh2; a.qn = h->qn; a.qmask = h->qmask;
        a.shift_w = h->vw.shift_w; a.shift_b = h->vw.shift_b; a.scale_w = h->vw.scale_w; a.scale_b = h->vw.scale_b;
        a.box2_w = h->vw.box2_w; a.box2_b = h->vw.box2_b; a.box_bias = h->vw.box_bias;
        a.scores = d_scores + (size_t)b0 * V_NP;
        a.labels = d_labels + (size_t)b0 * V_NP;
        a.xyxy = d_boxes_xyxy + (size_t)b0 * V_NP * 4;
        a.logits = d_logits ? d_logits + (size_t)b0 * V_NP * q_uniform : nullptr;
        a.image_set = h_image_query_set ? L.d_image_set + b0 : nullptr;
        a.setQ = h->d_setQ;
        a.cxcywh = d_boxes_cxcywh ? d_boxes_cxcywh + (size_t)b0 * V_NP * 4 : nullptr;
        a.rows = MP; a.np = V_NP; a.Q = q_uniform; a.img_w = W; a.img_h = H;
        RC(detect_rows(a, s));
        RC(cell_reduce(a.scores, a.labels, a.xyxy, h->qweight, a.image_set, Bc, V_NP, W, H, grid_rows, grid_cols, 0.005f,
                       d_cell_conf + (size_t)b0 * ncell, d_cell_mask + (size_t)b0 * ncell,
                       d_n_kept ? d_n_kept + b0 : nullptr, s));
    }
    return TSTAR_OK;
}

int tstar_owl_debug_preprocess(tstar_owl* h, const uint8_t* d_images, int B, int H, int W, uint8_t* d_out_u8,
                               float* d_out_patches, void* stream) {
    TSTAR_REQUIRE(h && d_images && d_out_patches, "tstar_owl_debug_preprocess: null argument");
    TSTAR_REQUIRE(B >= 1 && B <= h->max_batch, "tstar_owl_debug_preprocess: B must be in 1..max_batch");
    if (!h->has_vision) { set_error("tstar_owl_debug_preprocess: handle was created without vision weights (text-only)"); return TSTAR_ERR_STATE; }
    return preprocess_chunk(h, h->lane[0], d_images, B, H, W, d_out_u8, d_out_patches, (hipStream_t)stream);
}

int tstar_frames_to_grid(const uint8_t* d_video, int N, int H, int W, const int32_t* d_frame_idx, int grid_rows,
                         int grid_cols, uint8_t* d_grid, int nv12, void* stream) {
    TSTAR_REQUIRE(d_video && d_frame_idx && d_grid, "tstar_frames_to_grid: null argument");
    TSTAR_REQUIRE(N >= 1 && H >= 2 && W >= 2, "tstar_frames_to_grid: bad video shape");
    TSTAR_REQUIRE(!nv12 || (H % 2 == 0 && W % 2 == 0), "tstar_frames_to_grid: NV12 needs even dimensions");
    return frames_to_grid_u8(d_video, H, W, d_frame_idx, grid_rows, grid_cols, 200, 95, d_grid, nv12, (hipStream_t)stream);
}

int tstar_frames_resize(const uint8_t* d_video, int N, int H, int W, const int32_t* d_frame_idx, int n, int out_w,
                        int out_h, uint8_t* d_out, int nv12, void* stream) {
    TSTAR_REQUIRE(d_video && d_frame_idx && d_out, "tstar_frames_resize: null argument");
    TSTAR_REQUIRE(N >= 1 && H >= 2 && W >= 2, "tstar_frames_resize: bad video shape");
    TSTAR_REQUIRE(!nv12 || (H % 2 == 0 && W % 2 == 0), "tstar_frames_resize: NV12 needs even dimensions");
    return bilinear_gather_u8(d_video, H, W, d_frame_idx, n, out_w, out_h, d_out, nv12, (hipStream_t)stream);
}

int tstar_nv12_to_rgb(const uint8_t* d_video, int N, int H, int W, const int32_t* d_frame_idx, int n, uint8_t* d_out,
                      void* stream) {
    TSTAR_REQUIRE(d_video && d_frame_idx && d_out, "tstar_nv12_to_rgb: null argument");
    TSTAR_REQUIRE(N >= 1 && H >= 2 && W >= 2, "tstar_nv12_to_rgb: bad video shape");
    return nv12_to_rgb_u8(d_video, H, W, d_frame_idx, n, d_out, (hipStream_t)stream);
}

int tstar_i420_to_nv12(const uint8_t* d_i420, int n, int H, int W, uint8_t* d_nv12, void* stream) {
    TSTAR_REQUIRE(d_i420 && d_nv12 && d_i420 != d_nv12, "tstar_i420_to_nv12: null or aliased argument");
    return i420_to_nv12_u8(d_i420, n, H, W, d_nv12, (hipStream_t)stream);
}

int tstar_gemm_f32(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual, int M,
                   int N, int K, int act, void* stream) {
    TSTAR_REQUIRE(d_A && d_W && d_C, "tstar_gemm_f32: null argument");
    return gemm_f32(mk_gemm(nullptr, d_A, d_W, d_C, d_bias, d_residual, M, N, K, K, N, act), (hipStream_t)stream);
}

int tstar_gemm_f32_cfg(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual,
                       int M, int N, int K, int act, int tile_cfg, void* stream) {
    TSTAR_REQUIRE(d_A && d_W && d_C, "tstar_gemm_f32_cfg: null argument");
    GemmArgs g = mk_gemm(nullptr, d_A, d_W, d_C, d_bias, d_residual, M, N, K, K, N, act);
    g.tile_cfg = tile_cfg;
    return gemm_f32(g, (hipStream_t)stream);
}

static int gemm_converted(const char* fn, int a_terms, const float* d_A, const float* d_W, float* d_C, const float* d_bias,
                          const float* d_residual, int M, int N, int K, int act, int tile_cfg, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    __bf16* wb = nullptr;
    TSTAR_HIP_CHECK(hipMalloc(&wb, (size_t)N * K * sizeof(__bf16)));
    int rc = convert_f32_to_bf16(d_W, wb, nullptr, (size_t)N * K, s);
    void* wq = nullptr;
    if (!rc && a_terms == 2 && N % 256 == 0 && K % 32 == 0) {       // the fragment-packed plane of the two-term mode's VGPR-weight tile (tile_cfg 6, or N = 768)
        TSTAR_HIP_CHECK(hipMalloc(&wq, (size_t)N * K * sizeof(__bf16)));
        rc = pack_weights_w2(wb, wq, N, K, s);
    }
    if (!rc) {
        GemmArgs g = mk_gemm(nullptr, d_A, d_W, d_C, d_bias, d_residual, M, N, K, K, N, act);
        g.Wb = wb;
        g.Wq = wq;
        g.a_terms = a_terms;
        g.tile_cfg = tile_cfg;
        rc = gemm_f32(g, s);
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(wb);
    if (wq) (void)hipFree(wq);
    if (!rc && e != hipSuccess) { set_error(std::string(fn) + ": " + hipGetErrorString(e)); rc = TSTAR_ERR_HIP; }
    return rc;
}

int tstar_gemm_bf16w(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual, int M,
                     int N, int K, int act, int tile_cfg, void* stream) {
    TSTAR_REQUIRE(d_A && d_W && d_C, "tstar_gemm_bf16w: null argument");
    return gemm_converted("tstar_gemm_bf16w", 0, d_A, d_W, d_C, d_bias, d_residual, M, N, K, act, tile_cfg, stream);
}

int tstar_gemm_bf16w_pre(const float* d_A, const void* d_Wb, float* d_C, const float* d_bias, const float* d_residual, int M, int N,
                         int K, int act, int a_terms, int tile_cfg, void* stream) {
    TSTAR_REQUIRE(d_A && d_Wb && d_C, "tstar_gemm_bf16w_pre: null argument");
    TSTAR_REQUIRE(a_terms == 2 || a_terms == 3, "tstar_gemm_bf16w_pre: a_terms must be 2 or 3");
    GemmArgs g = mk_gemm(nullptr, d_A, reinterpret_cast<const float*>(d_Wb), d_C, d_bias, d_residual, M, N, K, K, N, act);
    g.Wb = static_cast<const __bf16*>(d_Wb);
    g.a_terms = a_terms;
    g.tile_cfg = tile_cfg;
    return gemm_f32(g, (hipStream_t)stream);
}

int tstar_gemm_bf16w2(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual, int M,
                      int N, int K, int act, int tile_cfg, void* stream) {
    TSTAR_REQUIRE(d_A && d_W && d_C, "tstar_gemm_bf16w2: null argument");
    return gemm_converted("tstar_gemm_bf16w2", 2, d_A, d_W, d_C, d_bias, d_residual, M, N, K, act, tile_cfg, stream);
}

int tstar_gemm_f32x3(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual, int M,
                     int N, int K, int act, int tile_cfg, void* stream) {
    TSTAR_REQUIRE(d_A && d_W && d_C, "tstar_gemm_f32x3: null argument");
    TSTAR_REQUIRE(M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0, "tstar_gemm_f32x3: N must be a multiple of 128, K of 32 (include/tstar_hip.h)");
    hipStream_t s = (hipStream_t)stream;
    void* wp = nullptr;
    TSTAR_HIP_CHECK(hipMalloc(&wp, (size_t)N * K * 6));
    int rc = pack_weights_x3(d_W, wp, N, K, s);
    if (!rc) {
        GemmArgs g = mk_gemm(nullptr, d_A, d_W, d_C, d_bias, d_residual, M, N, K, K, N, act);
        g.Wp = wp;
        g.tile_cfg = tile_cfg;
        rc = gemm_f32(g, s);
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(wp);
    if (!rc && e != hipSuccess) { set_error(std::string("tstar_gemm_f32x3: ") + hipGetErrorString(e)); rc = TSTAR_ERR_HIP; }
    return rc;
}

int tstar_pack_f32x3(const float* d_W, void* d_Wp, int N, int K, void* stream) {
    TSTAR_REQUIRE(d_W && d_Wp, "tstar_pack_f32x3: null argument");
    TSTAR_REQUIRE(N > 0 && K > 0, "tstar_pack_f32x3: empty matrix");
    return pack_weights_x3(d_W, d_Wp, N, K, (hipStream_t)stream);
}

int tstar_gemm_f32x3_pre(const float* d_A, const void* d_Wp, float* d_C, const float* d_bias, const float* d_residual, int M, int N,
                         int K, int act, int tile_cfg, void* stream) {
    TSTAR_REQUIRE(d_A && d_Wp && d_C, "tstar_gemm_f32x3_pre: null argument");
    TSTAR_REQUIRE(M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0, "tstar_gemm_f32x3_pre: N must be a multiple of 128, K of 32 (include/tstar_hip.h)");
    GemmArgs g = mk_gemm(nullptr, d_A, reinterpret_cast<const float*>(d_Wp), d_C, d_bias, d_residual, M, N, K, K, N, act);
    g.Wp = d_Wp;
    g.tile_cfg = tile_cfg;
    return gemm_f32(g, (hipStream_t)stream);
}

int tstar_layernorm_f32(const float* d_x, float* d_y, const float* d_w, const float* d_b, int rows, int D, void* stream) {
    TSTAR_REQUIRE(d_x && d_y && d_w && d_b, "tstar_layernorm_f32: null argument");
    return layernorm_f32(d_x, d_y, d_w, d_b, rows, D, (hipStream_t)stream);
}

int tstar_attention_f32(const float* d_qkv, float* d_out, int B, int T, int heads, int mode, const uint8_t* d_key_mask,
                        void* stream) {
    TSTAR_REQUIRE(d_qkv && d_out, "tstar_attention_f32: null argument");
    return attention_f32(d_qkv, d_out, B, T, heads, mode, d_key_mask, (hipStream_t)stream);
}

int tstar_draw_boxes(uint8_t* d_images, int B, int H, int W, const float* d_boxes_xyxy, const float* d_scores, void* stream) {
    TSTAR_REQUIRE(d_images && d_boxes_xyxy && d_scores, "tstar_draw_boxes: null argument");
    return draw_boxes(d_images, B, H, W, d_boxes_xyxy, d_scores, V_NP, 0.005f, (hipStream_t)stream);
}

int tstar_attention_x3(const float* d_qkv, float* d_out, int B, int T, int heads, void* stream) {
    TSTAR_REQUIRE(d_qkv && d_out, "tstar_attention_x3: null argument");
    return attention_x3(d_qkv, d_out, B, T, heads, (hipStream_t)stream);
}

int tstar_attention_split(const float* d_qkv, float* d_out, int B, int T, int heads, void* stream) {
    TSTAR_REQUIRE(d_qkv && d_out, "tstar_attention_split: null argument");
    return attention_split(d_qkv, d_out, B, T, heads, (hipStream_t)stream);
}

}  // extern "C"
