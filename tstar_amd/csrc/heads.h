// Launchers of heads.hip / preprocess.hip (internal).
#pragma once
#include "common.h"

namespace tstar {

struct DetectRowsArgs {
    const float* feats;    // [rows, 768]  image feats after detection LayerNorm
    const float* cls;      // [rows, 512]  class_head.dense0 output
    const float* boxh;     // [rows, 768]  box_head after dense1 + GELU
    const float* qn;       // [sets][32][512]  query embeds / (||q|| + 1e-6)
    const uint8_t* qmask;  // [sets][32]       0 = padded query
    const int* image_set;  // [B] query set of every image, or null (all images use set 0)
    const int* setQ;       // [sets] number of queries per set
    const float* shift_w; const float* shift_b;
    const float* scale_w; const float* scale_b;
    const float* box2_w;   // [4, 768]
    const float* box2_b;   // [4]
    const float* box_bias; // [np, 4]
    float* scores;         // [rows]
    int* labels;           // [rows]
    float* xyxy;           // [rows, 4] pixels of the passed image
    float* logits;         // [rows, Q] or null
    float* cxcywh;         // [rows, 4] or null
    int rows, np, Q, img_w, img_h;   // Q: common query count (row stride of `logits`), 0 if the sets differ
};
int detect_rows(const DetectRowsArgs& a, hipStream_t s);

int cell_reduce(const float* scores, const int* labels, const float* xyxy, const double* qweight, const int* image_set, int B, int np,
                int img_w, int img_h, int grows, int gcols, float thr, double* cell_conf, uint32_t* cell_mask,
                int* n_kept, hipStream_t s);

// paint the kept detections' boxes (score > thr) on u8 images [B,H,W,3] in place; xyxy [B,np,4], scores [B,np]
int draw_boxes(uint8_t* images, int B, int H, int W, const float* xyxy, const float* scores, int np, float thr, hipStream_t s);

// ---- preprocess.hip ----
// Pillow-compatible fixed-point resampling tables for one axis (host side).
struct ResampleTable {
    int in_size = 0, out_size = 0, ksize = 0;
    int* d_bounds = nullptr;   // [out, 2] (first input index, tap count)
    int* d_coefs = nullptr;    // [out, ksize] int32, 22 fractional bits
};
int build_bicubic_table(ResampleTable* t, int in_size, int out_size, hipStream_t s);
void free_table(ResampleTable* t);

// u8 [B,H,W,3] -> u8 [B,H,OW,3]   (Pillow 8bpc horizontal pass)
int resample_h_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, const ResampleTable& t, hipStream_t s);
// u8 [B,H,768,3] -> vertical pass -> LUT normalise -> im2col f32 [B*576, 3072]
int resample_v_normalize_patchify(const uint8_t* in, float* out, uint8_t* out_u8, int B, int H, const ResampleTable& t,
                                  const float* lut, hipStream_t s);

// OpenCV-style fixed-point bilinear resize (11-bit coefficients), gather by frame index.
// mode 0: frames[idx[i]] (H,W) -> out[i] (oh,ow)
int bilinear_gather_u8(const uint8_t* video, int H, int W, const int* d_idx, int n, int ow, int oh, uint8_t* out,
                       int nv12, hipStream_t s);
// frames[idx[i]] -> (4*ch x 4*cw) -> (ch x cw) -> tile (i / cols, i % cols) of grid [rows*ch, cols*cw, 3]
int frames_to_grid_u8(const uint8_t* video, int H, int W, const int* d_idx, int rows, int cols, int cw, int ch,
                      uint8_t* grid, int nv12, hipStream_t s);
// n planar I420 frames [H*3/2*W bytes each: Y, U, V planes] -> NV12 [n, H*3/2, W]
int i420_to_nv12_u8(const uint8_t* in, int n, int H, int W, uint8_t* out, hipStream_t s);
// frames[idx[i]] NV12 [H*3/2, W] -> RGB u8 [n,H,W,3] (BT.601 limited range, nearest chroma)
int nv12_to_rgb_u8(const uint8_t* video, int H, int W, const int* d_idx, int n, uint8_t* out, hipStream_t s);

}  // namespace tstar
