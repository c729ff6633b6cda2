"""ORACLE (test infrastructure, never shipped, never on the product path).

torch restatement of the reference's search-quality metrics
(/root/reference/LVHaystackBench/val_tstar_results.py:48-95 SSIM, :186-256 P/R/F1 and ANND).
PINNED by tests/golden/g10_metrics.npz, produced by importing the reference module itself
(tests/golden/make_goldens.py, stubs for its absent cv2 / skimage imports).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def ssim_pair(img1: np.ndarray, img2: np.ndarray, window_size: int = 11) -> float:
    """ssim_torch on two HWC uint8 frames, with the reference's HWC-as-CHW conv2d call (:62-78)."""
    a = torch.tensor(img1, dtype=torch.float32) / 255.0
    b = torch.tensor(img2, dtype=torch.float32) / 255.0
    ch = a.size(0)
    coords = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    k = g.unsqueeze(1)
    window = (k @ k.T).expand(ch, 1, window_size, window_size)
    pad = window_size // 2
    conv = lambda x: F.conv2d(x.unsqueeze(0), window, padding=pad, groups=ch)
    mu1, mu2 = conv(a), conv(b)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = conv(a * a) - mu1_sq
    s2 = conv(b * b) - mu2_sq
    s12 = conv(a * b) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean().item()


def pairwise_ssim(gt_frames, pred_frames) -> np.ndarray:
    out = np.zeros((len(gt_frames), len(pred_frames)))
    for i, g in enumerate(gt_frames):
        for j, p in enumerate(pred_frames):
            out[i, j] = ssim_pair(g, p)
    return out


def prf(list_gt, list_pred, threshold=5):
    ps, rs, fs = [], [], []
    for gt, pred in zip(list_gt, list_pred):
        if gt.size == 0 or pred.size == 0:
            continue
        d_gt = np.min(np.abs(gt[:, None] - pred), axis=1)
        d_pr = np.min(np.abs(pred[:, None] - gt), axis=1)
        p = np.sum(d_pr <= threshold) / len(pred)
        r = np.sum(d_gt <= threshold) / len(gt)
        f = 2 * p * r / (p + r) if (p + r) > 0 else 0.0
        ps.append(p), rs.append(r), fs.append(f)
    return (np.mean(ps) if ps else 0.0, np.mean(rs) if rs else 0.0, np.mean(fs) if fs else 0.0)


def annd(list_gt, list_pred):
    out = []
    for gt, pred in zip(list_gt, list_pred):
        if gt.size == 0 or pred.size == 0:
            continue
        out.append((np.mean(np.min(np.abs(pred[:, None] - gt), axis=1)), np.mean(np.min(np.abs(gt[:, None] - pred), axis=1))))
    return out
