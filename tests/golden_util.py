"""Helpers shared by tests/golden/make_goldens.py (runs the REFERENCE, in the build container only) and
the parity tests (run the oracle / the HIP path against the committed vectors)."""
import hashlib

import numpy as np


def sha(a) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def fake_detections(seed: int, call_index: int, img_h: int, img_w: int, n_classes: int, n_det: int = 96,
                    conf_scale: float = 0.9):
    """Deterministic fake detector output for searcher goldens: (xyxy f32 [n,4], class_id i64 [n],
    confidence f32 [n]).  Peaky confidences (mostly tiny, OWL-like), boxes inside the image, a few
    centred exactly on cell edges / image borders."""
    rs = np.random.RandomState((seed * 1000003 + call_index * 7919) % (2 ** 31 - 1))
    cx = rs.random_sample(n_det) * img_w
    cy = rs.random_sample(n_det) * img_h
    w = rs.random_sample(n_det) * img_w * 0.2 + 2
    h = rs.random_sample(n_det) * img_h * 0.2 + 2
    # edge cases: centres on multiples of 200 x 95 (cell edges) and on the image border
    cx[:4] = [200.0, img_w, 0.0, min(400.0, img_w)]
    cy[:4] = [95.0, img_h, 0.0, min(190.0, img_h)]
    xyxy = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], axis=1).astype(np.float32)
    conf = (rs.random_sample(n_det) ** 6 * conf_scale + 0.0051).astype(np.float32)
    cls = rs.randint(0, n_classes, n_det).astype(np.int64)
    return xyxy, cls, conf


class FakeDet:
    def __init__(self, xyxy, class_id, confidence):
        self.xyxy, self.class_id, self.confidence = xyxy, class_id, confidence

    def __len__(self):
        return len(self.xyxy)


class FakeHeuristic:
    """Duck-typed heuristic (the surface of /root/reference/TStar/interface_heuristic.py the
    searcher uses) that injects ``fake_detections``."""

    def __init__(self, seed: int, n_det: int = 96, conf_scale: float = 0.9):
        self.seed, self.n_det, self.calls, self.conf_scale = seed, n_det, 0, conf_scale
        self.texts = []
        self.detections_inbatch = []
        self.log = []

    def reparameterize_object_list(self, target_objects, cue_objects):
        self.texts = [[o.strip()] for o in list(target_objects) + list(cue_objects)] + [[" "]]

    def inference_detector(self, images, **kw):
        h, w = images[0].shape[:2]
        xyxy, cls, conf = fake_detections(self.seed, self.calls, h, w, len(self.texts), self.n_det, self.conf_scale)
        self.calls += 1
        self.log.append((h, w))
        self.detections_inbatch = [FakeDet(xyxy, cls, conf)]
        return self.detections_inbatch

    def bbox_visualization(self, images, detections_inbatch):
        return list(images)


def detector_test_image(seed: int, H: int, W: int) -> np.ndarray:
    """Blocky random uint8 test image [H,W,3] for the detector goldens (regenerated, not stored)."""
    rs = np.random.RandomState(seed)
    low = rs.randint(0, 256, (H // 8 + 1, W // 8 + 1, 3)).astype(np.float32)
    img = np.repeat(np.repeat(low, 8, 0), 8, 1)[:H, :W] + rs.randint(-20, 20, (H, W, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def g11_cases(golden_dir):
    """(name, distribution f64 [N], k, clip or None, reference picks) of tests/golden/g11_topk.npz."""
    import os
    g = np.load(os.path.join(golden_dir, "g11_topk.npz"), allow_pickle=False)
    for name in [str(n) for n in g["names"]]:
        clip = g[f"clip_{name}"]
        yield name, g[f"dist_{name}"], int(g[f"k_{name}"]), (None if clip[0] < 0 else (float(clip[0]), float(clip[1]))), g[f"secs_{name}"]


def check_topk_against_reference(picks, ref_picks, dist_clip, start):
    """``picks`` vs the reference's recorded picks for one G11 case: identical when the selection does not cut
    through a run of equal normalised values; otherwise the picked VALUES must be identical (which of the equal
    seconds numpy's default argsort returns is left to the numpy build / CPU) and ``picks`` must be the
    lowest-index choice among the tied seconds."""
    picks, ref_picks = np.asarray(picks, dtype=np.int64), np.asarray(ref_picks, dtype=np.int64)
    assert len(picks) == len(ref_picks)
    assert np.all(np.diff(picks) > 0), "ascending, no duplicates"
    v, rv = dist_clip[picks - start], dist_clip[ref_picks - start]
    assert np.array_equal(np.sort(v), np.sort(rv)), "picked values differ from the reference's"
    kth = v.min()
    cut = np.count_nonzero(dist_clip == kth) > np.count_nonzero(v == kth)          # a tie is cut at the k-th value
    if not cut:
        assert np.array_equal(picks, ref_picks)
    else:
        tied = np.nonzero(dist_clip == kth)[0] + start
        assert np.array_equal(picks[v == kth], tied[:np.count_nonzero(v == kth)]), "ties must resolve to the lowest index"
    return cut
