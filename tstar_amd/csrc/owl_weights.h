// Device-side view of the OWL-ViT-B/32 parameter blobs.  Entry order mirrors
// tstar_amd/weights.py vision_spec()/text_spec() one to one; the host blob is packed,
// the device copy pads every entry to 64 floats so all rows stay 16-byte aligned.
#pragma once
#include <stddef.h>

namespace tstar {

constexpr int V_D = 768, V_FF = 3072, V_LAYERS = 12, V_HEADS = 12, V_NP = 576, V_NTOK = 577, V_PATCH_K = 3072;
constexpr int T_D = 512, T_FF = 2048, T_LAYERS = 12, T_HEADS = 8, T_LEN = 16, T_VOCAB = 49408, PROJ = 512;

struct LayerW {
    const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *out_w, *out_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};

struct VisionW {
    const float *patch_w, *class_emb, *pos_emb, *pre_ln_w, *pre_ln_b;
    LayerW layers[V_LAYERS];
    const float *post_ln_w, *post_ln_b, *det_ln_w, *det_ln_b;
    const float *cls_w, *cls_b, *shift_w, *shift_b, *scale_w, *scale_b;
    const float *box0_w, *box0_b, *box1_w, *box1_b, *box2_w, *box2_b, *box_bias;
};

struct TextW {
    const float *tok_emb, *tpos_emb;
    LayerW layers[T_LAYERS];
    const float *final_ln_w, *final_ln_b, *text_proj;
};

// `take(n)` returns the pointer for the next entry of n floats.
template <class Take>
void map_layer(LayerW& l, int d, int ff, Take&& take) {
    l.ln1_w = take((size_t)d); l.ln1_b = take((size_t)d);
    l.qkv_w = take((size_t)3 * d * d); l.qkv_b = take((size_t)3 * d);
    l.out_w = take((size_t)d * d); l.out_b = take((size_t)d);
    l.ln2_w = take((size_t)d); l.ln2_b = take((size_t)d);
    l.fc1_w = take((size_t)ff * d); l.fc1_b = take((size_t)ff);
    l.fc2_w = take((size_t)d * ff); l.fc2_b = take((size_t)d);
}

template <class Take>
void map_vision(VisionW& w, Take&& take) {
    w.patch_w = take((size_t)V_D * V_PATCH_K);
    w.class_emb = take(V_D);
    w.pos_emb = take((size_t)V_NTOK * V_D);
    w.pre_ln_w = take(V_D); w.pre_ln_b = take(V_D);
    for (int i = 0; i < V_LAYERS; ++i) map_layer(w.layers[i], V_D, V_FF, take);
    w.post_ln_w = take(V_D); w.post_ln_b = take(V_D);
    w.det_ln_w = take(V_D); w.det_ln_b = take(V_D);
    w.cls_w = take((size_t)PROJ * V_D); w.cls_b = take(PROJ);
    w.shift_w = take(V_D); w.shift_b = take(1);
    w.scale_w = take(V_D); w.scale_b = take(1);
    w.box0_w = take((size_t)V_D * V_D); w.box0_b = take(V_D);
    w.box1_w = take((size_t)V_D * V_D); w.box1_b = take(V_D);
    w.box2_w = take((size_t)4 * V_D); w.box2_b = take(4);
    w.box_bias = take((size_t)V_NP * 4);
}

template <class Take>
void map_text(TextW& w, Take&& take) {
    w.tok_emb = take((size_t)T_VOCAB * T_D);
    w.tpos_emb = take((size_t)T_LEN * T_D);
    for (int i = 0; i < T_LAYERS; ++i) map_layer(w.layers[i], T_D, T_FF, take);
    w.final_ln_w = take(T_D); w.final_ln_b = take(T_D);
    w.text_proj = take((size_t)PROJ * T_D);
}

}  // namespace tstar
