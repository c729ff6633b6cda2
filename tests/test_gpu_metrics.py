"""8f row 1: pairwise SSIM on the GPU vs the reference's golden values and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ssim_matches_reference_golden(golden_dir):
    from tstar_amd import metrics as PM
    from tstar_amd.video import synthetic_frames_numpy
    g = np.load(os.path.join(golden_dir, "g10_metrics.npz"))
    n, H, W, seed = [int(v) for v in g["video"]]
    fg = list(synthetic_frames_numpy(g["gt_idx"], n, H, W, seed=seed))
    fp = list(synthetic_frames_numpy(g["pred_idx"], n, H, W, seed=seed))
    m = PM.pairwise_ssim(fg, fp)
    assert m.shape == g["ssim"].shape
    assert np.abs(m - g["ssim"]).max() < 1e-5                 # float32 convolution, different summation order
    sc = PM.calculate_ssim_scores([fg], [fp])
    assert np.abs(np.array(sc) - g["ssim_scores"]).max() < 1e-5
    assert PM.calculate_ssim_scores([[]], [fp]) == []


def test_ssim_native_resolution_vs_oracle():
    from oracle import metrics_ref as M
    from tstar_amd import metrics as PM
    from tstar_amd.video import synthetic_frames_numpy
    fr = list(synthetic_frames_numpy([0, 7, 8], 32, 360, 640, seed=2))
    noisy = np.clip(fr[1].astype(np.int32) + np.random.RandomState(0).randint(-30, 30, fr[1].shape), 0, 255).astype(np.uint8)
    got = PM.pairwise_ssim(fr[:2], [fr[1], noisy, fr[2]])
    want = M.pairwise_ssim(fr[:2], [fr[1], noisy, fr[2]])
    assert np.abs(got - want).max() < 1e-5
    assert got[1, 0] > 0.999999 and got[1, 1] < got[1, 0]
    with pytest.raises(ValueError, match="one size"):
        PM.pairwise_ssim(fr[:1], [fr[0][:100]])
