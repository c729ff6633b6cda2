"""ORACLE (test infrastructure, never shipped, never on the product path).

The whole reference path on the CPU, assembled from the restatements in this directory:
``search()`` (searcher_ref.SearcherRef) driving one detector call per grid image / per
verification frame exactly as /root/reference/TStar/interface_searcher.py:124-127,404 and
/root/reference/TStar/interface_heuristic.py:232-246 do -- including the reference's
inefficiencies when ``faithful=True`` (text tower recomputed on every call, batch-1
verification forwards).  Used by tests (teacher-forced parity) and by bench.py's
``cpu_baseline`` leg (kind "port").
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import numpy as np
import torch

from tstar_amd import weights as W

from . import owl_ref, resize_ref, searcher_ref


class CpuOwlDetector:
    def __init__(self, state_dict: Dict[str, np.ndarray], faithful: bool = True):
        self.wv = W.unpack_blob(W.pack_blob(state_dict, W.vision_spec()), W.vision_spec())
        self.wt = W.unpack_blob(W.pack_blob(state_dict, W.text_spec()), W.text_spec())
        self.faithful = faithful
        self.texts = None
        self._qe = None

    def reparameterize_object_list(self, target_objects, cue_objects, ids, mask):
        self.texts = [[o.strip()] for o in list(target_objects) + list(cue_objects)] + [[" "]]
        self.ids, self.mask = np.asarray(ids, np.int64), np.asarray(mask, np.int64)
        self._qe = None

    def query_embeds(self) -> np.ndarray:
        if self.faithful or self._qe is None:          # the reference runs the text tower on every call
            with torch.no_grad():
                self._qe = owl_ref.text_query_embeds(self.ids, self.mask, self.wt).numpy()
        return self._qe

    @staticmethod
    def preprocess(image: np.ndarray, library_speed: bool = False) -> np.ndarray:
        """uint8 [H,W,3] -> float32 [3,768,768].  ``library_speed``: use Pillow's own BICUBIC resize when it is
        importable (bit-identical to resize_ref.pil_bicubic_resize, tests/test_oracle_owl.py) -- for timing the
        CPU baseline at the speed of the library the reference calls, not of its numpy restatement."""
        if library_speed:
            try:
                from PIL import Image
                u8 = np.asarray(Image.fromarray(image).resize((768, 768), Image.BICUBIC))
                return resize_ref.hf_rescale_normalize(u8)
            except ImportError:
                pass
        return resize_ref.owl_preprocess(image)

    def inference_detector(self, image: np.ndarray, library_speed: bool = False):
        """One image (HxWx3 uint8) -> (xyxy f32 [n,4], class_id i64 [n], confidence f32 [n]), score > 0.005."""
        H, Wd = image.shape[:2]
        px = self.preprocess(image, library_speed)[None]
        out = owl_ref.detect(px, self.query_embeds(), self.wv, H, Wd, query_mask=self.ids[:, 0] > 0)
        s, l, b = out["kept"][0]
        return b, l, s, out


def make_score_fn(det: CpuOwlDetector, frame_fn: Callable[[Sequence[int]], np.ndarray], object2weight: Dict[str, float],
                  log: List[dict] | None = None):
    """score_fn for SearcherRef: frames come from ``frame_fn(secs) -> uint8 [n,H,W,3]``."""

    def score_fn(kind, secs, rows, cols):
        frames = frame_fn(secs)
        if kind == "grid":
            img = resize_ref.frames_to_grid(list(frames), rows, cols)
        else:
            img = resize_ref.cv_bilinear_resize(frames[0], 600, 285)
        xyxy, lab, conf, out = det.inference_detector(img)
        cm, names = searcher_ref.image_grid_score(xyxy, lab, conf, det.texts, object2weight, img.shape[0], img.shape[1],
                                                  rows, cols)
        if log is not None:
            log.append(dict(kind=kind, secs=list(secs), conf=cm.copy(), dense=out["dense"]))
        return cm, names

    return score_fn
