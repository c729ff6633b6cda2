"""8f row 2: downstream top-k selection (GPU vs oracle) and the result wire format."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k,clip", [(3600, 8, None), (3600, 8, (100.7, 460.2)), (14400, 32, None), (40, 8, (3, 11)),
                                      (10, 10, None), (20000, 16, None), (33000, 8, (4000.5, 29000.2))])
def test_topk_matches_oracle(n, k, clip):
    from oracle import searcher_ref as S
    from tstar_amd.results import topk_seconds
    rs = np.random.RandomState(n + k)
    p = rs.random_sample(n) ** 3
    p[rs.random_sample(n) < 0.3] = p[0]           # ties
    p[5] = np.nan
    p /= np.nansum(p)
    assert np.array_equal(topk_seconds(p, k, clip), S.topk_seconds(p, k, clip))


def test_topk_g11_reference_picks(golden_dir):
    """tstar_topk_seconds vs the reference's own extract_frames picks (golden G11), including the near-flat real P
    where the float32 division creates ties, and vs the oracle (always identical: same lowest-index tie rule)."""
    import golden_util as GU
    from oracle import searcher_ref as S
    from tstar_amd.results import topk_seconds
    for name, dist, k, clip, ref in GU.g11_cases(golden_dir):
        dc, start = S.topk_normalised_clip(dist, clip)
        got = topk_seconds(dist, k, clip)
        assert np.array_equal(got, S.topk_seconds(dist, k, clip)), name
        GU.check_topk_against_reference(got, ref, dc, start)


def test_topk_float32_division_ties():
    """Values that are distinct before ``dist_clip /= dist_clip.sum()`` and equal after it (float32): the kernel
    must rank the NORMALISED values, as the reference does."""
    from oracle import searcher_ref as S
    from tstar_amd.results import topk_seconds
    rs = np.random.RandomState(3)
    n = 3600
    base = 0.5 + rs.random_sample(n) * 1e-7                      # near-flat, like a real P
    dc, _ = S.topk_normalised_clip(base, None)
    raw = np.nan_to_num(base.astype(np.float32))
    assert len(np.unique(dc)) < len(np.unique(raw)) or len(np.unique(dc)) < n // 2
    for k in (8, 32):
        assert np.array_equal(topk_seconds(base, k), S.topk_seconds(base, k))


def test_topk_degenerate_distributions():
    from oracle import searcher_ref as S
    from tstar_amd.results import topk_seconds
    z = np.zeros(50)
    assert np.array_equal(topk_seconds(z, 8), S.topk_seconds(z, 8)) and topk_seconds(z, 8).tolist() == list(range(8))
    c = np.zeros(50)
    c[40:] = 1.0                                   # clip region all zero -> uniform inside the clip
    assert np.array_equal(topk_seconds(c, 4, (10, 30)), S.topk_seconds(c, 4, (10, 30)))


def test_result_wire_format(tmp_path):
    from tstar_amd.interface_heuristic import OWLInterface
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.results import result_from_searcher, save_results, topk_seconds
    from tstar_amd.video import synthetic_video
    h = OWLInterface(synthetic_seed=0, max_batch=8)
    s = TStarSearcher(synthetic_video(96, seed=3), h, ["couch"], ["tv"], search_nframes=4, image_grid_shape=(4, 4),
                      search_budget=0.2, confidence_threshold=0.6, rng=np.random.RandomState(1), keep_visual_history=False)
    s.search()
    r = result_from_searcher(s)
    assert set(r) == {"video_path", "grounding_objects", "keyframe_timestamps", "keyframe_distribution"}
    assert r["grounding_objects"] == {"target_objects": ["couch"], "cue_objects": ["tv"]}
    assert r["keyframe_timestamps"] == sorted(r["keyframe_timestamps"]) and len(r["keyframe_distribution"]) == 96
    path = tmp_path / "out" / "results.json"
    save_results([r], str(path))
    back = json.load(open(path))
    assert back[0]["keyframe_distribution"] == r["keyframe_distribution"]
    sel = topk_seconds(back[0]["keyframe_distribution"], 8)
    assert len(sel) == 8 and list(sel) == sorted(sel)


def test_owl_inference_from_image_file(tmp_path):
    """OWLInterface.inference (interface_heuristic.py:217-230): a file on disk gives the detections of the same pixels
    passed through inference_detector."""
    from PIL import Image
    from tstar_amd.interface_heuristic import OWLInterface
    h = OWLInterface(synthetic_seed=0, max_batch=2)
    h.reparameterize_object_list(["couch"], ["tv"])
    img = np.random.RandomState(3).randint(0, 256, (285, 600, 3)).astype(np.uint8)
    path = tmp_path / "frame.png"
    Image.fromarray(img).save(path)
    a = h.inference(str(path))
    b = h.inference_detector([img])[0]
    assert np.array_equal(a.xyxy, b.xyxy) and np.array_equal(a.confidence, b.confidence)
    assert np.array_equal(a.class_id, b.class_id) and len(a.confidence) > 0


def test_device_box_painter_equals_host_painter():
    """The searcher's visual history paints boxes on the device (tstar_draw_boxes) and copies frames and detections
    to the host once per batch; the result must equal the host painter (OWLInterface.bbox_visualization) applied to the
    un-annotated frames with the same detections."""
    import torch
    from tstar_amd.interface_heuristic import OWLInterface
    h = OWLInterface(synthetic_seed=0, max_batch=4)
    h.reparameterize_object_list(["couch"], ["tv", "chair"])
    rs = np.random.RandomState(8)
    frames = rs.randint(0, 256, (3, 285, 600, 3)).astype(np.uint8)
    d = torch.from_numpy(frames).cuda()
    res = h.score_batch(d, 1, 1)
    imgs, dets = h.annotated_batch(d, res)
    assert imgs.shape == frames.shape and len(dets) == 3
    for k in range(3):
        ref_det = h._detections_from(res, k)
        assert np.array_equal(dets[k].xyxy, ref_det.xyxy) and np.array_equal(dets[k].class_id, ref_det.class_id)
        want = h.bbox_visualization([frames[k].copy()], [ref_det])[0]
        assert np.array_equal(imgs[k], want)
        assert (imgs[k] != frames[k]).any()                 # something was painted
    # a sub-range of a batch result
    d2 = torch.from_numpy(frames[1:3].copy()).cuda()
    imgs2, dets2 = h.annotated_batch(d2, res, 1, 2)
    assert np.array_equal(imgs2, imgs[1:3]) and np.array_equal(dets2[1].confidence, dets[2].confidence)
