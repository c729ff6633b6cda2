// Internal launcher declarations (not part of the C ABI; see include/tstar_hip.h).
#pragma once
#include "common.h"

namespace tstar {

enum { ACT_NONE = 0, ACT_QGELU = 1, ACT_GELU = 2 };

struct GemmArgs {
    const float* A;      // [M, lda] row-major, K contiguous
    const float* W;      // [N, K] row-major (nn.Linear layout)
    const __bf16* Wb;    // optional bfloat16 copy of W: selects the bf16-weight tile (exact split of A)
    const void* Wp;      // optional fragment-packed three-plane copy of W (pack_weights_x3): selects the f32x3 tile
    const void* Wq;      // optional fragment-packed copy of Wb (pack_weights_w2): the two-term mode's wide tile may stream its weights global -> VGPR
    float* C;            // [*, ldc]
    const float* bias;   // [N] or null
    const float* res;    // residual, same layout as C, or null (may alias C)
    const float* pos;    // patch-embed epilogue: position embedding [np+1, N], or null
    int M, N, K, lda, ldc;
    int act;             // ACT_*
    int patch_np;        // patches per image (576) for the patch-embed epilogue
    int tile_cfg;        // (4 / 5: wide tile forced on / off; 6: the two-term wide tile with weights global -> VGPR forced)  -1 auto; 0 = 128x128, 1 = 64x128, 2 = 64x64 block tile, 3 = hybrid 128x128 + 64x128 tail (half/half); 16 + n = hybrid with n big row tiles (diagnostic)
    int m_split;         // hybrid launch: rows [0, m_split) use 128-row tiles (set by the launcher)
    int group_m;         // row panels per super-panel of the tile order (gemm_f32.hip tile_mn); 0 = the library's default
    int a_terms;         // bf16-weight tile (Wb set): 2 = activations as two round-to-nearest bf16 terms (2 MFMA
                         // products per algorithmic product); anything else = the exact three-term split
};
int gemm_f32(const GemmArgs& g, hipStream_t stream);

// Wb[i] = bfloat16(W[i]) (round to nearest even); n elements.  With Wlo != null also Wlo[i] = bfloat16(W[i] - Wb[i]).
int convert_f32_to_bf16(const float* W, __bf16* Wb, __bf16* Wlo, size_t n, hipStream_t s);

// two-term mode: Wb [N, K] bf16 -> the same values in MFMA-fragment order, 2 * N * K bytes (gemm_f32.hip)
int pack_weights_w2(const __bf16* Wb, void* Wq, int N, int K, hipStream_t s);

// f32x3 mode: W [N, K] f32 -> three exact bf16 planes in MFMA-fragment order, 6 * N * K bytes (gemm_f32.hip)
int pack_weights_x3(const float* W, void* Wp, int N, int K, hipStream_t s);

// y[r,:] = LayerNorm(x[r,:]) * w + b over D (eps 1e-5, biased variance); D % 256 == 0, D <= 1024
int layernorm_f32(const float* x, float* y, const float* w, const float* b, int rows, int D, hipStream_t s);

// x[b,0,:] = cls + pos[0] (token row 0 of every image); x is [B*ntok, D]
int write_cls_rows(float* x, const float* cls, const float* pos, int B, int ntok, int D, hipStream_t s);

// feats[b,p,:] = LN_det( LN_post(x[b,1+p,:]) * LN_post(x[b,0,:]) )
int merge_cls_ln(const float* x, float* feats, const float* post_w, const float* post_b,
                 const float* det_w, const float* det_b, int B, int ntok, int D, hipStream_t s);

// multi-head self-attention over packed qkv [B*T, 3*D] (q | k | v, heads of 64) -> out [B*T, D]
// mode 0: full attention; mode 1: causal + key padding mask (key_mask [B,T] u8, 0 = masked)
int attention_f32(const float* qkv, float* out, int B, int T, int heads, int mode,
                  const uint8_t* key_mask, hipStream_t s);

// the same (mode 0 only) on the bf16 matrix pipe: every f32 operand as two bf16 terms, 3 products per MFMA step
int attention_split(const float* qkv, float* out, int B, int T, int heads, hipStream_t s);

// the same (mode 0 only) with EXACT operands: every f32 operand as three bf16 terms, six products per MFMA step (the f32x3 mode)
int attention_x3(const float* qkv, float* out, int B, int T, int heads, hipStream_t s);

}  // namespace tstar
