"""Search-quality metrics of the reference's evaluator
(/root/reference/LVHaystackBench/val_tstar_results.py): pairwise SSIM on the GPU
(tstar_ssim_pairwise), temporal precision / recall / F1 (+-threshold) and ANND on the host (they
touch K <= 32 numbers per item).  Function names and return shapes follow the reference.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import _lib


def gaussian_window(window_size: int = 11, sigma: float = 1.5) -> np.ndarray:
    """create_window's 2-D kernel (:48-60) in float32: outer product of the normalised 1-D Gaussian."""
    coords = np.arange(window_size, dtype=np.float32) - np.float32(window_size // 2)
    g = np.exp(-(coords ** 2) / np.float32(2 * sigma ** 2)).astype(np.float32)
    g = (g / g.sum(dtype=np.float32)).astype(np.float32)
    return np.ascontiguousarray(np.outer(g, g).astype(np.float32))


def pairwise_ssim(gt_frames: Sequence[np.ndarray], pred_frames: Sequence[np.ndarray]) -> np.ndarray:
    """(:80-95) SSIM matrix [num_gt, num_pred] (float64).  Frames: HxWx3 uint8 (numpy or cuda tensors)."""
    import torch
    if not torch.cuda.is_available():
        raise _lib.TStarHipError("pairwise_ssim needs a HIP device; tstar_amd has no CPU path")
    lib = _lib.load()

    def stack(fr):
        if hasattr(fr, "data_ptr"):
            t = fr
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(f, dtype=np.uint8) for f in fr])))
        return t.to(device="cuda", dtype=torch.uint8).contiguous()

    g, p = stack(gt_frames), stack(pred_frames)
    if g.dim() != 4 or p.dim() != 4 or g.shape[1:] != p.shape[1:] or g.shape[-1] != 3:
        raise ValueError("pairwise_ssim: frames must all be HxWx3 of one size")
    G, H, W, _ = g.shape
    P = p.shape[0]
    out = torch.empty((G, P), dtype=torch.float64, device="cuda")
    win = gaussian_window()
    _lib.check(lib.tstar_ssim_pairwise(g.data_ptr(), G, p.data_ptr(), P, H, W, win.ctypes.data, out.data_ptr(),
                                       _lib.stream_ptr()), "tstar_ssim_pairwise")
    return out.cpu().numpy()


def calculate_ssim_scores(list_gt_images: List[List[np.ndarray]], list_pred_images: List[List[np.ndarray]]) -> List[Tuple[float, float]]:
    """(:216-239) per video: (mean over predictions of the best SSIM, mean over ground truth of the best SSIM)."""
    out = []
    for gt_images, pred_images in zip(list_gt_images, list_pred_images):
        if not len(gt_images) or not len(pred_images):
            continue
        gt = [im for im in gt_images if np.asarray(im).size > 0]
        pr = [im for im in pred_images if np.asarray(im).size > 0]
        if not gt or not pr:
            continue
        m = pairwise_ssim(gt, pr)
        out.append((np.mean(np.max(m, axis=0)), np.mean(np.max(m, axis=1))))
    return out


def _temporal_pairs(list_gt, list_pred):
    """Per video with both lists non-empty: (distance of every ground-truth second to its nearest prediction, distance of every
    predicted second to its nearest ground-truth second) from ONE |gt - pred| matrix."""
    for gt, pred in zip(list_gt, list_pred):
        gt, pred = np.asarray(gt), np.asarray(pred)
        if gt.size and pred.size:
            dist = np.abs(np.subtract.outer(gt, pred))           # [len(gt), len(pred)]
            yield dist.min(axis=1), dist.min(axis=0)


def calculate_prf(list_gt: List[np.ndarray], list_pred: List[np.ndarray], threshold: int = 5) -> Tuple[float, float, float]:
    """Temporal precision / recall / F1 averaged over the videos, a second counting as matched when its nearest counterpart is
    within +-threshold (the evaluator's definition: LVHaystackBench/val_tstar_results.py:186-214)."""
    rows = []
    for gt_to_pred, pred_to_gt in _temporal_pairs(list_gt, list_pred):
        precision = np.count_nonzero(pred_to_gt <= threshold) / pred_to_gt.size
        recall = np.count_nonzero(gt_to_pred <= threshold) / gt_to_pred.size
        both = precision + recall
        rows.append((precision, recall, 2 * (precision * recall) / both if both > 0 else 0.0))
    if not rows:
        return 0.0, 0.0, 0.0
    p, r, f = (np.mean(col) for col in zip(*rows))
    return p, r, f


def calculate_annd(list_gt: List[np.ndarray], list_pred: List[np.ndarray]) -> List[Tuple[float, float]]:
    """Average nearest-neighbour distance per video, (prediction -> ground truth, ground truth -> prediction)
    (val_tstar_results.py:241-256)."""
    return [(np.mean(pred_to_gt), np.mean(gt_to_pred)) for gt_to_pred, pred_to_gt in _temporal_pairs(list_gt, list_pred)]
