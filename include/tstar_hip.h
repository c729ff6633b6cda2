/* tstar_hip.h -- C ABI of libtstar_hip.so: the MI355X (gfx950) implementation of the
 * T* keyframe-search hot path.
 *
 * The reference (mll-lab-nu/TStar) has NO native plug-in ABI: its plug-in surface is two
 * duck-typed Python classes, HeuristicInterface/OWLInterface
 * (TStar/interface_heuristic.py:28-37, 200-280) and TStarSearcher
 * (TStar/interface_searcher.py:14-538).  tstar_amd/ keeps that Python surface and binds
 * the entry points below with ctypes (tstar_amd/_lib.py); INTEGRATION.md shows the
 * binding a maintainer of the reference would add.  Each entry point names the reference
 * code it replaces.
 *
 * Conventions
 *   - plain C types only; "d_" pointers are DEVICE pointers (HBM) owned by the caller,
 *     "h_" pointers are host pointers; nothing is retained past the call unless stated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued on it; no entry point synchronises unless stated.
 *   - return value 0 = OK; otherwise an error code (TSTAR_ERR_*), with a human-readable
 *     message from tstar_last_error() (thread-local).
 *   - integer/byte outputs are bit-exact restatements of the reference arithmetic;
 *     floating-point outputs match the CPU oracle within 1e-3 (observed ~1e-6).
 */
#ifndef TSTAR_HIP_H
#define TSTAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSTAR_OK 0
#define TSTAR_ERR_ARG 1
#define TSTAR_ERR_HIP 2
#define TSTAR_ERR_STATE 3

#define TSTAR_OWL_NPATCH 576   /* 24 x 24 patches of a 768 x 768 detector image */
#define TSTAR_OWL_QDIM 512
#define TSTAR_OWL_TEXT_LEN 16
#define TSTAR_OWL_MAX_QUERIES 32
#define TSTAR_OWL_MAX_SETS 64      /* independent query sets (questions) resident at once */

const char* tstar_last_error(void);
int tstar_abi_version(void);
/* number of float32 values expected in the vision / text weight blobs (layout:
 * tstar_amd/weights.py vision_spec()/text_spec(), mirrored in csrc/owl_weights.h) */
size_t tstar_owl_vision_blob_floats(void);
size_t tstar_owl_text_blob_floats(void);

/* ------------------------------------------------------------------ detector (D-rows) */
typedef struct tstar_owl tstar_owl;

/* Replaces OWLInterface.__init__/load_model_and_tokenizer + model.to(device)
 * (interface_heuristic.py:201-210).  Copies the float32 blobs to HBM and allocates the
 * activation workspace for `max_batch` detector images per pass (larger batches are
 * processed in chunks).  h_text_blob may be NULL (then only tstar_owl_set_query_embeds
 * can install queries).  h_norm_lut: 3*256 float32, the rescale+normalise value of every
 * (channel, u8) pair, computed by the host with the reference's arithmetic
 * (HF image_transforms.py rescale/normalize via image_processing_pil_owlvit.py).
 * weights_mode:
 *   TSTAR_WEIGHTS_F32 (0): float32 weights, exact-f32 MFMA (v_mfma_f32_32x32x2_f32) -- the reference's
 *     arithmetic and the mode every headline number is quoted in.
 *   TSTAR_WEIGHTS_BF16 (1) (BASELINE config 5, "bf16 ViT weights"): every GEMM weight matrix is also kept
 *     as bfloat16 (round to nearest even; exact if the blob already holds bf16 values) and the GEMMs run
 *     on the bf16 matrix pipe.  A bf16 weight is exact in ONE term; the float32 activations are carried as TWO
 *     round-to-nearest bf16 terms a_hi + a_lo (16 significand bits, |a - a_hi - a_lo| <= 2^-17 |a|):
 *     C += a_lo*w + a_hi*w, exact products, f32 accumulation -- 2 MFMA products per algorithmic product.
 *     Detector scores stay within 1e-3 of a float32 run on the same rounded weights (tests state the measured
 *     bound, ~1e-5).  The vision tower's attention runs on the bf16 pipe as well in modes 1 and 3 (f32-split
 *     operands: two bf16 terms each, 3 products, f32 accumulation).
 *   (2, TSTAR_WEIGHTS_F32_SPLIT of ABI 2 -- both operands as two bf16 terms, 16 significand bits -- is retired and
 *     refused; TSTAR_WEIGHTS_F32X3 below carries all 24 bits.)
 *   TSTAR_WEIGHTS_BF16_EXACT (3): bf16 weights with the activations split EXACTLY into three bf16 terms
 *     (8 + 8 + 8 significand bits, truncation split; 3 MFMA products): the f32-accumulated product of the f32
 *     activations with the bf16 weights, f32-roundoff class (round 1-2's bf16 mode, kept selectable).
 *   TSTAR_WEIGHTS_F32X3 (4) (ABI 3): float32 checkpoints on the bf16 matrix pipe WITHOUT dropping an operand bit: every
 *     weight and every activation is split exactly into three round-to-nearest bfloat16 terms (8 + 8 + 8 significand
 *     bits, signed remainders; weights once at creation, packed in MFMA-fragment order, 6 bytes per weight) and
 *     C += a0 w2 + a1 w1 + a0 w1 + a2 w0 + a1 w0 + a0 w0 per K = 16 step -- the six partial products with ka + kw <= 2,
 *     each exact, f32 accumulation.  The three products left out are <= 2^-24 |a w| each and zero-mean; measured
 *     against float64 the result is no further away than the exact-f32 MFMA tile's (tests assert it on every shape
 *     they run; the nine-product form has the same error to four digits).  1.3-1.45x the f32 tile's rate on the batch
 *     shapes.  Attention, LayerNorm and the heads' tails stay float32.  Opt-in; the headline is quoted in mode 0.
 * ABI note (tstar_abi_version() == 3): since ABI 2 mode 1 means TWO-term activations (it was the exact three-term split,
 * now mode 3), TSTAR_OWL_MAX_SETS went 32 -> 64, mode 2 was retired and mode 4 added. */
#define TSTAR_WEIGHTS_F32 0
#define TSTAR_WEIGHTS_BF16 1
#define TSTAR_WEIGHTS_BF16_EXACT 3
#define TSTAR_WEIGHTS_F32X3 4
int tstar_owl_create(tstar_owl** out, const float* h_vision_blob, size_t n_vision,
                     const float* h_text_blob, size_t n_text, const float* h_norm_lut, int max_batch,
                     int weights_mode);
int tstar_owl_destroy(tstar_owl* h);

/* Replaces the text half of processor(...)+model(...) that the reference recomputes on every
 * detector call (interface_heuristic.py:234,239 -> HF modeling_owlvit.py:945-958, 631-663):
 * runs the CLIP text tower once for Q queries (ids/mask int32 [Q,16]) and keeps the
 * L2-normalised query embeddings resident.  h_class_weight float64 [Q] = object2weight of each
 * query's name (interface_searcher.py:88-91,136; Python floats in the reference, and the
 * confidence score * weight is formed in float64 as under the reference's pinned numpy 1.26).  `query_set` (0..TSTAR_OWL_MAX_SETS-1) is the slot the queries are stored
 * in: several (video, question) items can be resident at once and every image of a tstar_owl_score call
 * names the slot it is scored against (the reference keeps exactly one query set, = slot 0).
 * Synchronises `stream`. */
int tstar_owl_set_queries(tstar_owl* h, int query_set, const int32_t* h_input_ids, const int32_t* h_attention_mask,
                          const double* h_class_weight, int Q, void* stream);
/* Several query sets in ONE text-tower forward (a lock-step group installs the questions of all its items at once): h_sets
 * [n_sets] slots, h_Q [n_sets] queries per set, ids / attention masks [sum Q][16] and class weights [sum Q] concatenated in
 * set order.  Same kernels as tstar_owl_set_queries on more rows; results are bit-identical to one call per set.  Synchronises. */
int tstar_owl_set_queries_many(tstar_owl* h, int n_sets, const int32_t* h_sets, const int32_t* h_Q, const int32_t* h_ids,
                               const int32_t* h_am, const double* h_w, void* stream);
/* Same, from precomputed L2-normalised embeddings float32 [Q,512] and query mask u8 [Q]. */
int tstar_owl_set_query_embeds(tstar_owl* h, int query_set, const float* h_query_embeds, const uint8_t* h_query_mask,
                               const double* h_class_weight, int Q, void* stream);
/* Replaces only the per-query class weights (TStarSearcher sets object2weight AFTER it has
 * reparameterised the heuristic, interface_searcher.py:87-91). */
int tstar_owl_set_class_weights(tstar_owl* h, int query_set, const double* h_class_weight, int Q, void* stream);
/* Copies the resident (L2-normalised, pre-class-head) query embeddings float32 [Q,512] to the host. */
int tstar_owl_get_query_embeds(tstar_owl* h, int query_set, float* h_out, int Q, void* stream);

/* Replaces OWLInterface.inference_detector (interface_heuristic.py:232-246: HF preprocess,
 * both towers' forward, post_process_grounded_object_detection(threshold=0.005)) AND the
 * detection->grid-cell loop of TStarSearcher.imageGridScoreFunction
 * (interface_searcher.py:129-150) for B images of identical size in one call.
 *   d_images      u8  [B,H,W,3] RGB (the grid image, or a verification frame)
 *   h_image_query_set  i32 [B] (host) query set of every image, or NULL (all images use set 0)
 *   d_scores      f32 [B,576]   sigmoid(max_q logit)            (dense: not thresholded)
 *   d_labels      i32 [B,576]   argmax_q logit
 *   d_boxes_xyxy  f32 [B,576,4] pixels of the passed image
 *   d_cell_conf   f64 [B,rows*cols]  max over detections with score > 0.005 of
 *                                    float64(score) * class_weight[label], row-major cells; 0 if none
 *   d_cell_mask   u32 [B,rows*cols]  bit q set <=> a kept detection with label q fell in the cell
 *   d_n_kept      i32 [B]       number of detections with score > 0.005 (may be NULL)
 *   d_logits      f32 [B,576,Q] raw logits (may be NULL; needs the same Q for every image)
 *   d_boxes_cxcywh f32 [B,576,4] pred_boxes (may be NULL)
 */
int tstar_owl_score(tstar_owl* h, const uint8_t* d_images, int B, int H, int W, int grid_rows, int grid_cols,
                    const int32_t* h_image_query_set, float* d_scores, int32_t* d_labels, float* d_boxes_xyxy, double* d_cell_conf,
                    uint32_t* d_cell_mask, int32_t* d_n_kept, float* d_logits, float* d_boxes_cxcywh, void* stream);
/* Round 6 (an added entry point; tstar_abi_version() stays 3).  The same call on workspace `lane` (0 .. TSTAR_OWL_LANES - 1).  tstar_owl_score is lane 0, the handle's own workspace of
 * max_batch images.  Lane 1 is a second, SMALL workspace (forward chunks of min(max_batch, max(TSTAR_OWL_AUX_BATCH, B)) images; allocated on
 * first use and grown when a larger batch arrives, which synchronises the device once): a call on lane 1 shares no mutable state with a call on lane 0, so the two may be
 * enqueued on DIFFERENT streams and execute concurrently -- TStarSearcher queues the NEXT iteration's grid forward (one image: 120-456
 * wave tiles per GEMM for 1024 SIMDs) on lane 1 beside the verification batch of the iteration before (interface_searcher.py:444-491:
 * the loop the reference runs strictly one call after another).  Calls on ONE lane must stay ordered (one stream, or events), as
 * before.  Results do not depend on the lane: same kernels, same tile choices, same bits. */
#define TSTAR_OWL_LANES 2
#define TSTAR_OWL_AUX_BATCH 4
int tstar_owl_score_lane(tstar_owl* h, int lane, const uint8_t* d_images, int B, int H, int W, int grid_rows, int grid_cols,
                         const int32_t* h_image_query_set, float* d_scores, int32_t* d_labels, float* d_boxes_xyxy,
                         double* d_cell_conf, uint32_t* d_cell_mask, int32_t* d_n_kept, float* d_logits,
                         float* d_boxes_cxcywh, void* stream);

/* Diagnostics for parity tests: the preprocessed 768x768 u8 image of the LAST chunk's image 0
 * (after bicubic) and its patch-embed A operand can be read back. */
int tstar_owl_debug_preprocess(tstar_owl* h, const uint8_t* d_images, int B, int H, int W,
                               uint8_t* d_out_u8 /* [B,768,768,3] */, float* d_out_patches /* [B*576,3072] */,
                               void* stream);

/* ------------------------------------------------------------------ second detector backend: YOLO-World (D13)
 * Replaces YoloWorldInterface (interface_heuristic.py:39-190; wired at TStarFramework.py:178-184) -- the mmdet test
 * pipeline (keep-ratio resize to 640, letterbox pad 114, /255, channel swap), model.test_step (YOLOv8 CSPDarknet,
 * text-guided PAFPN, BN-contrastive head, DFL decode, class-aware NMS) and the wrapper's `score > 0.12`, top-50
 * (:148-152) -- with f32 VALU kernels (no MFMA, BASELINE configs[3]).  The model source is NOT in the reference tree
 * (dangling symlink): the architecture is restated from the published design, parity against the real model is
 * UNPINNED; oracle/yolo_ref.py is the independent CPU statement the tests compare against.
 *
 * The network is handed over as data (tstar_amd/yolo_world.py build_program): a float32 blob (BatchNorm folded), a
 * table of ops [n_ops][24] int32 over NHWC buffers [n_bufs][3] = (H, W, C), the max-sigmoid attention layers
 * [n_guides][5] = (embed, heads, guide_fc weight / bias offsets, per-head bias offset) and the head levels
 * [n_levels][8] = (embedding buffer, DFL buffer, map size, stride, offset of (exp(logit_scale), bias), 0, 0, 0). */
typedef struct tstar_yolo tstar_yolo;
int tstar_yolo_create(tstar_yolo** out, const float* h_blob, size_t n_blob, const int32_t* h_ops, int n_ops, int op_words,
                      const int32_t* h_bufs, int n_bufs, const int32_t* h_guides, int n_guides, const int32_t* h_levels,
                      int n_levels, int input_buf, int max_batch);
int tstar_yolo_destroy(tstar_yolo* h);
int tstar_yolo_num_anchors(tstar_yolo* h);
/* model.reparameterize(texts) (interface_heuristic.py:93): the cached CLIP text features float32 [Q,512] of query set
 * `query_set` (as the text backbone returns them: L2-normalised) + the searcher's class weights float64 [Q].  Synchronises. */
int tstar_yolo_set_text_feats(tstar_yolo* h, int query_set, const float* h_text, const double* h_class_weight, int Q, void* stream);
int tstar_yolo_set_class_weights(tstar_yolo* h, int query_set, const double* h_class_weight, int Q, void* stream);
/* inference_detector (:136-168) for B equally sized images u8 [B,H,W,3] (+ the searcher's detection -> cell loop,
 * interface_searcher.py:129-150, when d_cell_conf is given):
 *   d_det_scores f32 [B,max_dets], d_det_labels i32 [B,max_dets] (-1 = empty), d_det_boxes f32 [B,max_dets,4] xyxy pixels of
 *   the passed image, descending score; d_n_det i32 [B]; d_cell_conf f64 / d_cell_mask u32 [B,rows*cols] (may both be NULL);
 *   d_dense_scores f32 [B,8400,Q] / d_dense_boxes f32 [B,8400,4] (diagnostics, may be NULL). */
int tstar_yolo_detect(tstar_yolo* h, const uint8_t* d_images, int B, int H, int W, int grid_rows, int grid_cols,
                      const int32_t* h_image_query_set, float score_threshold, int max_dets, float* d_det_scores,
                      int32_t* d_det_labels, float* d_det_boxes, int32_t* d_n_det, double* d_cell_conf, uint32_t* d_cell_mask,
                      float* d_dense_scores, float* d_dense_boxes, void* stream);

/* ------------------------------------------------------------------ ingest (S1-S3, S8) */
/* The resident decoded video d_video is u8 [N,H,W,3] RGB (nv12 = 0) or NV12 u8 [N, H*3/2, W] (nv12 = 1:
 * luma plane + interleaved half-resolution UV plane, converted on the fly, BT.601 limited range, nearest
 * chroma -- half the bytes per frame).
 * Replaces read_frame_batch + cv2.resize(800x380) + create_image_grid's cv2.resize(200x95) +
 * hstack/vstack (interface_searcher.py:157-169, 362, 186-188): gathers rows*cols frames by index
 * and writes the grid image u8 [rows*95, cols*200, 3]. */
int tstar_frames_to_grid(const uint8_t* d_video, int N, int H, int W, const int32_t* d_frame_idx,
                         int grid_rows, int grid_cols, uint8_t* d_grid, int nv12, void* stream);
/* Replaces read_frame_batch + cv2.resize (interface_searcher.py:402-403; any target size):
 * out u8 [n,out_h,out_w,3]. */
int tstar_frames_resize(const uint8_t* d_video, int N, int H, int W, const int32_t* d_frame_idx, int n,
                        int out_w, int out_h, uint8_t* d_out, int nv12, void* stream);
/* Decode front end for RAW 4:2:0 containers (compressed streams need rocDecode / FFmpeg, which this
 * build does not have): n planar I420 frames (Y, U, V planes back to back, H*W*3/2 bytes each) already copied to the
 * device -> the resident NV12 store layout u8 [n, H*3/2, W] the ingest kernels read.  Replaces the decode half of
 * read_frame_batch (interface_searcher.py:157-169) for such files. */
int tstar_i420_to_nv12(const uint8_t* d_i420, int n, int H, int W, uint8_t* d_nv12, void* stream);
/* Native-resolution RGB u8 [n,H,W,3] of NV12 frames (the keyframes pop_frames hands back, :379-380). */
int tstar_nv12_to_rgb(const uint8_t* d_video, int N, int H, int W, const int32_t* d_frame_idx, int n,
                      uint8_t* d_out, void* stream);

/* ------------------------------------------------------------------ searcher state (S-rows)
 * Device-resident float64 state of one TStarSearcher (interface_searcher.py:73-75):
 * score_distribution, non_visiting_frames, P, plus the sampler's working arrays.  All
 * kernels are single-workgroup and reproduce numpy's operation order (pairwise sum,
 * sequential cumsum, 'linear' percentile); see csrc/searcher.hip for the line map. */
typedef struct tstar_searcher tstar_searcher;
int tstar_searcher_create(tstar_searcher** out, int n_frames, double init_score, double init_p);
int tstar_searcher_destroy(tstar_searcher* s);
/* update_frame_distribution up to the spline fit (interface_searcher.py:302-313, 260-261):
 * for the n sampled seconds (draw order) and their cell confidences d_conf f64 [n] (device,
 * cell i <-> sample i): mark visited, write scores, top-25 % window spread; returns the
 * visited frames (ascending) and their scores to the host for the FITPACK fit.  Synchronises. */
int tstar_searcher_apply_grid(tstar_searcher* s, const int32_t* h_secs, const double* d_conf, int n,
                              int* h_n_visited, int32_t* h_vis_x, double* h_vis_y, void* stream);
/* update_top_25_with_window alone (interface_searcher.py:215-241) on the device score array: np.percentile(h_conf, 75),
 * then for every sample with conf >= threshold, in the given order and in place, score[f + o] = max(score[f + o],
 * score[f] / (|o| + 1)) for |o| <= window.  Synchronises. */
int tstar_searcher_window_spread(tstar_searcher* s, const int32_t* h_secs, const double* h_conf, int n, int window,
                                 void* stream);
/* the visited frames (non_visiting == 0, ascending) and their scores (interface_searcher.py:260-261).  Synchronises. */
int tstar_searcher_visited(tstar_searcher* s, int* h_n_visited, int32_t* h_vis_x, double* h_vis_y, void* stream);
/* spline_keyframe_distribution after the fit (interface_searcher.py:266-274): evaluates the
 * B-spline (t, c, k) from scipy's UnivariateSpline at 0..N-1 (FITPACK splev, ext=0), clamps at
 * 1/N, sigmoid, normalises -> P. */
int tstar_searcher_set_spline(tstar_searcher* s, const double* h_t, const double* h_c, int n_knots, int k, void* stream);
/* sample_frames' weights (interface_searcher.py:345-352) with add = num/N, and the cdf of
 * np.random.choice; *h_fallback = 1 if the unvisited mask was dropped.  Synchronises. */
int tstar_searcher_sampler_prep(tstar_searcher* s, int num, double add, int* h_fallback, void* stream);
/* pop_frames' weights (interface_searcher.py:369) and their cdf.  *h_nnz = count_nonzero(p > 0) and *h_sum =
 * score.sum(): what numpy's choice() validates before drawing ("probabilities contain NaN" when the sum is 0 or NaN,
 * "Fewer non-zero entries in p than size").  Synchronises. */
int tstar_searcher_pop_prep(tstar_searcher* s, int* h_nnz, double* h_sum, void* stream);
/* cdf.searchsorted(x, 'right') for k host-drawn uniforms (the MT19937 stream stays on the host,
 * numpy legacy RandomState.choice).  Synchronises. */
int tstar_searcher_draw(tstar_searcher* s, const double* h_x, int k, int32_t* h_idx, void* stream);
/* choice()'s retry step: p[found] = 0, cdf recomputed. */
int tstar_searcher_exclude(tstar_searcher* s, const int32_t* h_found, int m, void* stream);
/* verification overwrites (interface_searcher.py:407): score[secs[i]] = vals[i], in order. */
int tstar_searcher_set_scores(tstar_searcher* s, const int32_t* h_secs, const double* h_vals, int m, void* stream);
/* store_score_distribution (interface_searcher.py:207-213): P, score_distribution and non_visiting_frames
 * copied to h_out f64 [3, N] (in that order) with ONE synchronisation. */
int tstar_searcher_read_state(tstar_searcher* s, double* h_out, void* stream);
/* overwrite a state array from the host (the reference's attributes are plain numpy arrays a caller may assign):
 * 0 score_distribution, 1 non_visiting_frames, 2 P -- e.g. P computed on the host by the reference's own numpy/scipy
 * calls (bit-identical by construction; tstar_searcher_set_spline is the device-side alternative).  Synchronises. */
int tstar_searcher_write(tstar_searcher* s, int which, const double* h_in, void* stream);
/* copy a state array to the host: 0 score, 1 non_visiting, 2 P, 3 sampler p, 4 cdf.  Synchronises. */
int tstar_searcher_read(tstar_searcher* s, int which, double* h_out, void* stream);

/* ------------------------------------------------------------------ multi-GPU
 * The path shards over independent (video, question) items with no data-path collective; the ONE exchange is an
 * all-gather of every rank's final keyframe indices -- what the sequential loop of
 * LVHaystackBench/run_TStar_onDataset.py:195-205 accumulates in `results`.  One process per GPU; RCCL over xGMI.
 * Bootstrap like NCCL's: rank 0 calls tstar_comm_unique_id and hands the TSTAR_COMM_ID_BYTES bytes to every rank by
 * any out-of-band channel (a file, a TCP store, MPI, torch.distributed's store); then EVERY rank calls
 * tstar_comm_create (collective; binds the current HIP device).  RCCL is dlopen'ed on first use (the copy a host
 * such as PyTorch-ROCm already carries is reused); TSTAR_RCCL_LIB overrides the library name. */
#define TSTAR_COMM_ID_BYTES 128
typedef struct tstar_comm tstar_comm;
/* binds RCCL in THIS process without touching any other rank (0 = usable).  tstar_comm_create is collective, so hosts
 * call this on every rank and agree on the outcome first: a rank that cannot load RCCL must not leave the others
 * blocked inside ncclCommInitRank. */
int tstar_comm_available(void);
int tstar_comm_unique_id(void* h_id /* TSTAR_COMM_ID_BYTES bytes */);
int tstar_comm_create(tstar_comm** out, const void* h_id, int world, int rank);
int tstar_comm_destroy(tstar_comm* c);
/* d_recv int32 [world * count] = concatenation over ranks (rank order) of every rank's d_send int32 [count] (both on
 * the device; pad short rows with -1).  Enqueued on `stream`; does not synchronise. */
int tstar_allgather_i32(tstar_comm* c, const int32_t* d_send, int32_t* d_recv, int count, void* stream);

/* ------------------------------------------------------------------ downstream selection (8f)
 * Replaces the score-based branch of extract_frames (LVHaystackBench/val_qa_results.py:90-110): the k
 * highest-probability seconds of a per-second distribution d_P f64 [N] (device) inside
 * [clip_start, clip_end), returned in ascending order (host int32 [k]); NaN -> 0, all-zero -> uniform; the clip is
 * normalised in float32 (numpy's pairwise sum, then an f32 divide) BEFORE ranking, as the reference does, so
 * division-induced ties are reproduced; ties resolve to the lowest index.  Synchronises. */
int tstar_topk_seconds(const double* d_P, int N, int clip_start, int clip_end, int k, int32_t* h_out, void* stream);

/* Replaces pairwise_ssim / ssim_torch (LVHaystackBench/val_tstar_results.py:48-95): SSIM of every
 * (ground-truth, predicted) keyframe pair, frames u8 [G,H,W,3] / [P,H,W,3] on the device, 11x11 Gaussian
 * window (host float32 [121], sigma 1.5) -> d_out f64 [G,P].  Keeps the reference's HWC-as-CHW layout
 * quirk (the window slides over the (W, colour) plane of each row).  Synchronises. */
int tstar_ssim_pairwise(const uint8_t* d_gt, int G, const uint8_t* d_pred, int P, int H, int W,
                        const float* h_window, double* d_out, void* stream);

/* Replaces OWLInterface.bbox_visualization (interface_heuristic.py:259-267) for images that are already on the
 * device: paints the 1-px box of every kept detection (score > 0.005) of image b -- d_boxes_xyxy [B,576,4] and
 * d_scores [B,576] as written by tstar_owl_score -- onto d_images u8 [B,H,W,3] in place. */
int tstar_draw_boxes(uint8_t* d_images, int B, int H, int W, const float* d_boxes_xyxy, const float* d_scores, void* stream);

/* ------------------------------------------------------------------ kernel-level diagnostics
 * (used by tests/ and bench.py to check and time individual kernels) */
/* C[M,N] = act(A[M,K] * W[N,K]^T + bias) (+ residual); act: 0 none, 1 quick-gelu, 2 gelu(erf) */
int tstar_gemm_f32(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual,
                   int M, int N, int K, int act, void* stream);
/* same with the block tile forced: 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = hybrid (128x128 + 64x128 tail);
 * -1 = the launcher's choice; 16 + n = hybrid with the first n row tiles of 128 rows big (tile-policy sweeps) */
int tstar_gemm_f32_cfg(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual,
                       int M, int N, int K, int act, int tile_cfg, void* stream);
/* bf16-weight GEMM (diagnostic): W is rounded to bfloat16 on the device, A is split exactly; synchronises */
int tstar_gemm_bf16w(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual,
                     int M, int N, int K, int act, int tile_cfg, void* stream);
/* bf16-weight GEMM with two-term activations (diagnostic; tile_cfg 4 forces the 128x256 tile, 5 forbids it, 6 (round 6; needs N % 256 == 0 and
 * M >= 128) forces the 128x256 tile whose weight fragments stream global -> VGPR from a fragment-packed plane -- what the library picks by itself
 * for the N = 768 layers; every choice returns the same bits) */
int tstar_gemm_bf16w2(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual,
                      int M, int N, int K, int act, int tile_cfg, void* stream);
/* bf16-weight GEMM on weights that are ALREADY bfloat16 on the device (d_Wb: [N, K] bf16); a_terms 2 or 3; enqueues only
 * (microbenchmarks) */
int tstar_gemm_bf16w_pre(const float* d_A, const void* d_Wb, float* d_C, const float* d_bias, const float* d_residual,
                         int M, int N, int K, int act, int a_terms, int tile_cfg, void* stream);
/* f32x3 GEMM (diagnostic): W is packed into three exact bf16 planes on the device, A is split on the fly, six products
 * (TSTAR_WEIGHTS_F32X3); N % 128 == 0, K % 32 == 0; tile_cfg as above, 4 forces the 128x256 tile, 5 forbids it; synchronises */
int tstar_gemm_f32x3(const float* d_A, const float* d_W, float* d_C, const float* d_bias, const float* d_residual,
                     int M, int N, int K, int act, int tile_cfg, void* stream);
/* the same in two steps for microbenchmarks: pack once (d_Wp: 6 * N * K bytes on the device), then enqueue-only GEMMs on the packed planes */
int tstar_pack_f32x3(const float* d_W, void* d_Wp, int N, int K, void* stream);
int tstar_gemm_f32x3_pre(const float* d_A, const void* d_Wp, float* d_C, const float* d_bias, const float* d_residual,
                         int M, int N, int K, int act, int tile_cfg, void* stream);
int tstar_layernorm_f32(const float* d_x, float* d_y, const float* d_w, const float* d_b, int rows, int D, void* stream);
/* qkv [B*T, 3*heads*64] -> out [B*T, heads*64]; mode 0 full, 1 causal + key mask u8 [B,T] */
int tstar_attention_f32(const float* d_qkv, float* d_out, int B, int T, int heads, int mode,
                        const uint8_t* d_key_mask, void* stream);

/* full attention with f32-split operands on the bf16 matrix pipe (what the bf16-pipe weights modes use) */
int tstar_attention_split(const float* d_qkv, float* d_out, int B, int T, int heads, void* stream);
/* full attention with EXACT operands on the bf16 matrix pipe: Q, K, V and the probabilities as three bf16 terms each, six
 * products per MFMA step, f32 accumulation (what TSTAR_WEIGHTS_F32X3 uses for the vision tower; replaces the fp32
 * softmax(Q K^T / 8) V of HF modeling_owlvit.py:377-402 behind TStar/interface_heuristic.py:237-239) */
int tstar_attention_x3(const float* d_qkv, float* d_out, int B, int T, int heads, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream, for bench.py's roofline leg.
 * category 0 = gemm_f32_kernel, 1 = attention_f32_kernel, 2 = conv_valu_kernel (YOLO-World backend).  enable(n > 0) resets the counters and times ONE of every
 * n consecutive launches of each category, at a position that varies from block to block (n = 1: all; a fixed phase would alias with periodic launch patterns);
 * read() synchronises the recorded events and returns launches / total ms / total algorithmic flops. */
int tstar_prof_enable(int on);
int tstar_prof_read(int category, long long* launches, double* total_ms, double* total_flops);
/* algorithmic HBM bytes of the launches tstar_prof_read counted (operands read once, results written once) */
int tstar_prof_read_bytes(int category, double* total_bytes);
/* EVERY launch of the category since enable(), sampled or not: their number and their algorithmic flops (exact; with the
 * sampled launches' flops / ms this gives the category's time without the sampling error of a 1-in-n sample of durations) */
int tstar_prof_read_totals(int category, long long* launches_all, double* flops_all);
/* trace markers: enqueue an empty kernel named prof_mark_begin_kernel (which = 0) / prof_mark_end_kernel (1) on `stream`,
 * so that a rocprofv3 kernel trace can be cut to the bracketed region on the GPU's own timeline */
int tstar_prof_mark(int which, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSTAR_HIP_H */
