"""ORACLE (test infrastructure, never shipped, never on the product path).

numpy restatement of the byte arithmetic in front of the detector:

* ``pil_bicubic_resize`` -- Pillow's 8-bit BICUBIC ``Image.resize`` that the HF
  OWL-ViT image processor applies to the grid image
  (/root/reference/TStar/interface_heuristic.py:234 -> HF
  image_processing_pil_owlvit.py:109-119).  Pillow (reference pin 10.4.0,
  requirements.txt:94; this container 12.2.0) is third-party, un-vendored; the
  algorithm is Pillow's src/libImaging/Resample.c (precompute_coeffs,
  normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc/Vertical_8bpc):
  horizontal pass first, uint8 intermediate, 22-bit fixed-point coefficients.
  PINNED: tests/test_oracle_owl.py::test_bicubic_equals_pillow compares with PIL bit for bit, and
  tests/golden/g7_g8_detector.npz holds hashes / crops of the HF processor output.
* ``hf_rescale_normalize`` -- HF image_transforms.py:118-122 and :419-437.
* ``cv_bilinear_resize`` -- INTER_LINEAR 8-bit resize in the fixed-point form of
  OpenCV's generic C++ path (HResizeLinear / VResizeLinear<uchar,int,short>,
  11-bit coefficients) for the three ``cv2.resize`` call sites of
  /root/reference/TStar/interface_searcher.py:186,362,403.  PARITY UNPINNED:
  cv2 (pin opencv-python 4.10.0, requirements.txt:88) is not importable here and
  the reference has no test for it; this function DEFINES the build's bilinear
  (it is also what the cv2 stub uses when goldens are generated).  Known OpenCV
  behaviour it does not special-case (from upstream knowledge): ``cv::resize``
  turns INTER_LINEAR into the INTER_AREA fast path when BOTH axes decimate by
  exactly 2 (e.g. a 1600x760 source -> 800x380).  No branch is needed: at scale 2
  the linear taps are (2d, 2d+1) with weights 1024 / 1024 and the fixed-point
  formula below collapses to ``(a + b + c + d + 2) >> 2`` -- the area result
  (tests/test_oracle_owl.py::test_cv_linear_at_2x_equals_area).
"""
from __future__ import annotations

import math

import functools

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size: int, out_size: int):
    """(bounds [out,2] int, coefs [out,ksize] int32) of Pillow's 8bpc bicubic for one axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    coefs = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)          # C truncation toward zero (values are > -1)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(n):
            w = k[x] / ww if ww != 0.0 else k[x]
            v = w * float(1 << 22)
            coefs[xx, x] = int(-0.5 + v) if v < 0 else int(0.5 + v)
        bounds[xx] = (xmin, n)
    return bounds, coefs


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """Resample along axis 0 of a uint8 array [n, ...]."""
    bounds, coefs = pil_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for o in range(out_size):
        lo, n = bounds[o]
        acc = np.tensordot(coefs[o, :n], src[lo:lo + n], axes=(0, 0)) + (1 << 21)
        out[o] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return out


def pil_bicubic_resize(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]; horizontal pass first."""
    assert img.dtype == np.uint8 and img.ndim == 3
    t = _resample_axis0(np.ascontiguousarray(img.transpose(1, 0, 2)), out_w).transpose(1, 0, 2)
    return _resample_axis0(np.ascontiguousarray(t), out_h)


def hf_rescale_normalize(img_u8: np.ndarray) -> np.ndarray:
    """uint8 [H,W,3] -> float32 [3,H,W] as the HF (PIL/numpy) processor does."""
    x = (img_u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    mean = np.array(CLIP_MEAN, dtype=np.float32)
    std = np.array(CLIP_STD, dtype=np.float32)
    x = (x - mean) / std
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def owl_preprocess(img_u8: np.ndarray) -> np.ndarray:
    """Grid image / frame uint8 [H,W,3] -> pixel_values float32 [3,768,768]."""
    return hf_rescale_normalize(pil_bicubic_resize(img_u8, 768, 768))


def patchify(pixel_values: np.ndarray) -> np.ndarray:
    """float32 [3,768,768] -> im2col [576, 3072] (row = patch, col = c*1024 + py*32 + px)."""
    x = pixel_values.reshape(3, 24, 32, 24, 32).transpose(1, 3, 0, 2, 4)
    return np.ascontiguousarray(x.reshape(576, 3072))


# ----------------------------------------------------------------------------- bilinear
@functools.lru_cache(maxsize=64)
def _cv_linear_table_cached(src: int, dst: int):
    t = cv_linear_table.__wrapped__(src, dst)
    t.setflags(write=False)
    return t


def cv_linear_table(src: int, dst: int):
    """Per output index: (s0, s1, w0, w1); float arithmetic as in cv::resize's coefficient loop.  (Cached per (src, dst): the
    loop below is pure Python; callers get a read-only array.)"""
    return _cv_linear_table_cached(int(src), int(dst))


def _cv_linear_table_uncached(src: int, dst: int):
    scale = float(src) / float(dst)
    tab = np.zeros((dst, 4), dtype=np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            f, s = np.float32(0), 0
        if s >= src - 1:
            f, s = np.float32(0), src - 1
        s1 = min(s + 1, src - 1)
        w0 = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))   # cvRound: half to even
        w1 = int(np.rint(np.float32(f * np.float32(2048))))
        tab[d] = (s, s1, w0, w1)
    return tab


cv_linear_table.__wrapped__ = _cv_linear_table_uncached


def cv_bilinear_resize(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    tx = cv_linear_table(W, out_w)
    ty = cv_linear_table(H, out_h)
    s = img.astype(np.int32)       # every intermediate stays below 2^27: int32 gives the same integers as int64, faster
    tx = tx.astype(np.int32)
    ty = ty.astype(np.int32)
    # horizontal pass on every needed source row: int = S[s0]*w0 + S[s1]*w1
    hp = s[:, tx[:, 0], :] * tx[:, 2][None, :, None] + s[:, tx[:, 1], :] * tx[:, 3][None, :, None]
    r0 = hp[ty[:, 0]] >> 4
    r1 = hp[ty[:, 1]] >> 4
    b0 = ty[:, 2][:, None, None]
    b1 = ty[:, 3][:, None, None]
    out = (((b0 * r0) >> 16) + ((b1 * r1) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def frames_to_grid(frames, rows: int, cols: int) -> np.ndarray:
    """sample_frames' resize (800x380) + create_image_grid (200x95 + tiling),
    /root/reference/TStar/interface_searcher.py:362, 183-188."""
    if len(frames) != rows * cols:
        raise ValueError("Frame count does not match grid dimensions")
    small = [cv_bilinear_resize(cv_bilinear_resize(f, 800, 380), 200, 95) for f in frames]
    return np.vstack([np.hstack(small[i * cols:(i + 1) * cols]) for i in range(rows)])


def nv12_to_rgb(frame: np.ndarray) -> np.ndarray:
    """NV12 uint8 [H*3/2, W] -> RGB uint8 [H,W,3]: BT.601 limited-range integer matrix, nearest chroma.
    PARITY UNPINNED (the reference receives RGB from decord/swscale and never sees NV12): this is the
    build's own definition of the NV12 ingest variant, mirrored by SrcNV12 in
    tstar_amd/csrc/preprocess.hip."""
    H = frame.shape[0] * 2 // 3
    W = frame.shape[1]
    y = frame[:H].astype(np.int64) - 16
    uv = frame[H:].reshape(H // 2, W // 2, 2).astype(np.int64) - 128
    d = np.repeat(np.repeat(uv[..., 0], 2, 0), 2, 1)
    e = np.repeat(np.repeat(uv[..., 1], 2, 0), 2, 1)
    r = (298 * y + 409 * e + 128) >> 8
    g = (298 * y - 100 * d - 208 * e + 128) >> 8
    b = (298 * y + 516 * d + 128) >> 8
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)
