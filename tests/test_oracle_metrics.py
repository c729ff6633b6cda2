"""CPU: metrics oracle against goldens produced by the reference's val_tstar_results.py itself."""
import os

import numpy as np

from oracle import metrics_ref as M
from tstar_amd.video import synthetic_frames_numpy


def test_g10_metrics_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_metrics.npz"))
    n, H, W, seed = [int(v) for v in g["video"]]
    fg = list(synthetic_frames_numpy(g["gt_idx"], n, H, W, seed=seed))
    fp = list(synthetic_frames_numpy(g["pred_idx"], n, H, W, seed=seed))
    m = M.pairwise_ssim(fg, fp)
    assert np.array_equal(m, g["ssim"])                       # same torch ops -> bit-identical
    assert abs(m[0, 0] - 1.0) < 1e-6                          # identical frames
    lg = [np.array([3.0, 10.0, 17.0, 100.0]), np.array([]), np.array([5.0, 50.0])]
    lp = [np.array([2.0, 16.5, 60.0]), np.array([1.0]), np.array([44.0, 45.0, 46.0, 200.0])]
    assert np.array_equal(np.array(M.prf(lg, lp, 5)), g["prf"])
    assert np.array_equal(np.array(M.annd(lg, lp)), g["annd"])
    assert np.allclose([np.mean(m.max(0)), np.mean(m.max(1))], g["ssim_scores"][0], atol=0, rtol=0)


def test_host_metrics_match_goldens(golden_dir):
    """The product's host-side P/R/F1 and ANND (tstar_amd.metrics) reproduce the reference's numbers."""
    from tstar_amd import metrics as PM
    g = np.load(os.path.join(golden_dir, "g10_metrics.npz"))
    lg = [np.array([3.0, 10.0, 17.0, 100.0]), np.array([]), np.array([5.0, 50.0])]
    lp = [np.array([2.0, 16.5, 60.0]), np.array([1.0]), np.array([44.0, 45.0, 46.0, 200.0])]
    assert np.array_equal(np.array(PM.calculate_prf(lg, lp, threshold=5)), g["prf"])
    assert np.array_equal(np.array(PM.calculate_annd(lg, lp)), g["annd"])
    assert PM.calculate_prf([], []) == (0.0, 0.0, 0.0)
    w = PM.gaussian_window()
    assert w.shape == (11, 11) and abs(float(w.sum()) - 1.0) < 1e-6
