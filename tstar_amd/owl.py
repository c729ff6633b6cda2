"""Host wrapper of the HIP OWL-ViT-B/32 scorer (tstar_owl_* in include/tstar_hip.h).

PyTorch is used for device memory and streams only; every computation is a
hand-written gfx950 kernel behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from . import weights as W

# CLIP normalisation constants (transformers/utils/constants.py:5-6)
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def normalize_lut() -> np.ndarray:
    """f32 [3,256]: value of (channel, u8) after HF rescale + normalize.

    Restates transformers image_transforms.py rescale (:118-122: f64(u8) * (1/255)
    -> f32) and normalize (:419-437: (x - f32(mean)) / f32(std) in f32), the
    arithmetic OWLInterface.inference_detector reaches through
    ``self.processor(...)`` (/root/reference/TStar/interface_heuristic.py:234).
    """
    u = np.arange(256, dtype=np.uint8)
    x = (u.astype(np.float64) * (1 / 255)).astype(np.float32)
    mean = np.array(OPENAI_CLIP_MEAN, dtype=np.float32)
    std = np.array(OPENAI_CLIP_STD, dtype=np.float32)
    lut = (x[None, :] - mean[:, None]) / std[:, None]
    return np.ascontiguousarray(lut.astype(np.float32))


@dataclass
class ScoreResult:
    """Device tensors produced by one tstar_owl_score call."""
    scores: "object"       # f32 [B,576]
    labels: "object"       # i32 [B,576]
    boxes: "object"        # f32 [B,576,4] xyxy pixels
    cell_conf: "object"    # f64 [B,rows*cols]
    cell_mask: "object"    # i32 view of u32 [B,rows*cols]
    n_kept: "object"       # i32 [B]
    logits: "object" = None
    boxes_cxcywh: "object" = None


class OwlScorer:
    """One OWL-ViT-B/32 scorer resident on the current HIP device."""

    WEIGHTS_MODES = {"f32": 0, "bf16": 1, "bf16_exact": 3, "f32x3": 4}      # TSTAR_WEIGHTS_* of include/tstar_hip.h

    def __init__(self, vision_blob: Optional[np.ndarray], text_blob: Optional[np.ndarray] = None, max_batch: int = 32,
                 weights_mode: str = "f32"):
        """``vision_blob=None`` gives a text-only handle: ``set_queries`` / ``get_query_embeds`` work (the CLIP text
        features of the YOLO-World backend), ``score`` raises."""
        import torch
        if not torch.cuda.is_available():
            raise _lib.TStarHipError("OwlScorer needs a HIP device (torch.cuda.is_available() is False); "
                                     "tstar_amd has no CPU path")
        if weights_mode not in self.WEIGHTS_MODES:
            raise ValueError("weights_mode must be one of " + ", ".join(repr(k) for k in self.WEIGHTS_MODES))
        self._torch = torch
        self._lib = _lib.load()
        if vision_blob is None and text_blob is None:
            raise ValueError("OwlScorer needs vision weights, text weights, or both")
        if vision_blob is not None:
            vision_blob = np.ascontiguousarray(vision_blob, dtype=np.float32)
        if text_blob is not None:
            text_blob = np.ascontiguousarray(text_blob, dtype=np.float32)
        lut = normalize_lut()
        h = C.c_void_p()
        rc = self._lib.tstar_owl_create(
            C.byref(h), None if vision_blob is None else vision_blob.ctypes.data, 0 if vision_blob is None else vision_blob.size,
            None if text_blob is None else text_blob.ctypes.data, 0 if text_blob is None else text_blob.size,
            lut.ctypes.data, int(max_batch), self.WEIGHTS_MODES[weights_mode])
        _lib.check(rc, "tstar_owl_create")
        self._h = h
        self.max_batch = int(max_batch)
        self.Qs = {}                # query-set slot -> number of queries
        self._pending = {}          # slot -> (ids, mask, weights) recorded by set_queries(lazy=True), installed on first use
        self.device = torch.device("cuda", torch.cuda.current_device())

    @classmethod
    def synthetic(cls, seed: int = 0, max_batch: int = 32, with_text: bool = True):
        sd = W.synthetic_state_dict(seed, "both" if with_text else "vision")
        vb = W.pack_blob(sd, W.vision_spec())
        tb = W.pack_blob(sd, W.text_spec()) if with_text else None
        return cls(vb, tb, max_batch)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tstar_owl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries
    @property
    def Q(self) -> int:
        return self.Qs.get(0, 0)

    def set_queries(self, input_ids: np.ndarray, attention_mask: np.ndarray, class_weight: Sequence[float], slot: int = 0,
                    lazy: bool = False):
        """Run the text tower on the queries and install them in ``slot``.  ``lazy=True`` only records them (after the checks
        the library would make): the text tower runs when the slot is first USED -- scored against, read back, re-weighted.
        A searcher's constructor installs its question in slot 0 like the reference's does (interface_searcher.py:87), but a
        lock-step group scores every item against its own slot 1..63 and never touches slot 0; the solo path uses it at once."""
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        am = np.ascontiguousarray(attention_mask, dtype=np.int32)
        w = np.ascontiguousarray(class_weight, dtype=np.float64)
        Q = ids.shape[0]
        if ids.shape != (Q, W.T_LEN) or am.shape != ids.shape or w.shape != (Q,):
            raise ValueError("set_queries: ids/mask must be [Q,16] and class_weight [Q]")
        self._pending.pop(int(slot), None)
        if lazy:
            if not 1 <= Q <= 32:
                raise _lib.TStarHipError(f"tstar_owl_set_queries: Q must be in 1..32 (got {Q})")
            if ids.min() < 0 or ids.max() >= 49408:
                raise _lib.TStarHipError("tstar_owl_set_queries: token id out of range")
            self._pending[int(slot)] = (ids.copy(), am.copy(), w.copy())
            self.Qs[int(slot)] = Q
            return
        rc = self._lib.tstar_owl_set_queries(self._h, int(slot), ids.ctypes.data, am.ctypes.data, w.ctypes.data, Q,
                                             _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_set_queries")
        self.Qs[int(slot)] = Q

    def _flush(self, slots):
        """Install the recorded (lazy) queries of the slots about to be used."""
        for sl in {int(v) for v in slots}:
            p = self._pending.pop(sl, None)
            if p is not None:
                self.set_queries(p[0], p[1], p[2], slot=sl)

    def set_queries_many(self, entries):
        """``entries``: [(slot, input_ids [Q,16], attention_mask [Q,16], class_weight [Q])] -- the queries of several slots through
        ONE text-tower forward (tstar_owl_set_queries_many); bit-identical to one ``set_queries`` call per slot."""
        if not entries:
            return
        slots = np.ascontiguousarray([int(e[0]) for e in entries], dtype=np.int32)
        ids = [np.ascontiguousarray(e[1], dtype=np.int32) for e in entries]
        am = [np.ascontiguousarray(e[2], dtype=np.int32) for e in entries]
        w = [np.ascontiguousarray(e[3], dtype=np.float64) for e in entries]
        for i_, a_, w_ in zip(ids, am, w):
            if i_.ndim != 2 or i_.shape[1] != W.T_LEN or a_.shape != i_.shape or w_.shape != (i_.shape[0],):
                raise ValueError("set_queries_many: ids/mask must be [Q,16] and class_weight [Q] per entry")
        Qs = np.ascontiguousarray([i_.shape[0] for i_ in ids], dtype=np.int32)
        ids_c, am_c, w_c = np.concatenate(ids), np.concatenate(am), np.concatenate(w)
        rc = self._lib.tstar_owl_set_queries_many(self._h, len(entries), slots.ctypes.data, Qs.ctypes.data, ids_c.ctypes.data, am_c.ctypes.data,
                                                  w_c.ctypes.data, _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_set_queries_many")
        for sl, q in zip(slots, Qs):
            self._pending.pop(int(sl), None)
            self.Qs[int(sl)] = int(q)

    def set_query_embeds(self, embeds: np.ndarray, query_mask: Sequence[int], class_weight: Sequence[float], slot: int = 0):
        e = np.ascontiguousarray(embeds, dtype=np.float32)
        m = np.ascontiguousarray(query_mask, dtype=np.uint8)
        w = np.ascontiguousarray(class_weight, dtype=np.float64)
        Q = e.shape[0]
        if e.shape != (Q, W.PROJ) or m.shape != (Q,) or w.shape != (Q,):
            raise ValueError("set_query_embeds: embeds [Q,512], mask [Q], class_weight [Q]")
        self._pending.pop(int(slot), None)
        rc = self._lib.tstar_owl_set_query_embeds(self._h, int(slot), e.ctypes.data, m.ctypes.data, w.ctypes.data, Q,
                                                  _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_set_query_embeds")
        self.Qs[int(slot)] = Q

    def set_class_weights(self, class_weight: Sequence[float], slot: int = 0):
        w = np.ascontiguousarray(class_weight, dtype=np.float64)
        if w.shape != (self.Qs.get(int(slot), 0),):
            raise ValueError("set_class_weights: one weight per installed query")
        if int(slot) in self._pending:                       # not installed yet: the weights ride along
            p = self._pending[int(slot)]
            self._pending[int(slot)] = (p[0], p[1], w.copy())
            return
        rc = self._lib.tstar_owl_set_class_weights(self._h, int(slot), w.ctypes.data, len(w), _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_set_class_weights")

    def get_query_embeds(self, slot: int = 0) -> np.ndarray:
        self._flush([slot])
        q = self.Qs.get(int(slot), 0)
        out = np.empty((q, W.PROJ), dtype=np.float32)
        rc = self._lib.tstar_owl_get_query_embeds(self._h, int(slot), out.ctypes.data, q, _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_get_query_embeds")
        return out

    # ---- scoring
    def score(self, images, grid_rows: int, grid_cols: int, want_logits: bool = False,
              image_sets: Optional[Sequence[int]] = None, lane: int = 0) -> ScoreResult:
        """images: torch u8 cuda tensor [B,H,W,3] (contiguous); ``image_sets``: query-set slot per image
        (default: slot 0 for all).  ``lane``: activation workspace of the forward (tstar_owl_score_lane) -- 0 = the handle's own,
        1 = the small second one: a call on lane 1 enqueued on ANOTHER stream may run beside a call on lane 0 (same results)."""
        torch = self._torch
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_cuda:
            raise ValueError("score: images must be a cuda uint8 tensor [B,H,W,3]")
        if grid_rows < 1 or grid_cols < 1:
            raise ValueError("score: the grid must be at least 1x1")
        images = images.contiguous()
        B, H, Wd, _ = images.shape
        dev = images.device
        ncell = grid_rows * grid_cols
        r = ScoreResult(
            scores=torch.empty((B, W.NPATCH), dtype=torch.float32, device=dev),
            labels=torch.empty((B, W.NPATCH), dtype=torch.int32, device=dev),
            boxes=torch.empty((B, W.NPATCH, 4), dtype=torch.float32, device=dev),
            cell_conf=torch.empty((B, ncell), dtype=torch.float64, device=dev),
            cell_mask=torch.empty((B, ncell), dtype=torch.int32, device=dev),
            n_kept=torch.empty((B,), dtype=torch.int32, device=dev),
        )
        sets = None
        if image_sets is not None:
            sets = np.ascontiguousarray(image_sets, dtype=np.int32)
            if sets.shape != (B,):
                raise ValueError("score: image_sets needs one slot per image")
        if self._pending:
            self._flush(sets.tolist() if sets is not None else [0])
        if want_logits:
            qs = {self.Qs.get(int(v), 0) for v in (sets if sets is not None else [0])}
            if len(qs) != 1:
                raise ValueError("score: raw logits need the same query count for every image")
            r.logits = torch.empty((B, W.NPATCH, qs.pop()), dtype=torch.float32, device=dev)
            r.boxes_cxcywh = torch.empty((B, W.NPATCH, 4), dtype=torch.float32, device=dev)
        rc = self._lib.tstar_owl_score_lane(
            self._h, int(lane), images.data_ptr(), B, H, Wd, grid_rows, grid_cols,
            None if sets is None else sets.ctypes.data, r.scores.data_ptr(), r.labels.data_ptr(), r.boxes.data_ptr(), r.cell_conf.data_ptr(),
            r.cell_mask.data_ptr(), r.n_kept.data_ptr(),
            _lib.ptr(r.logits), _lib.ptr(r.boxes_cxcywh), _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_score")
        return r

    def debug_preprocess(self, images):
        torch = self._torch
        images = images.contiguous()
        B, H, Wd, _ = images.shape
        u8 = torch.empty((B, 768, 768, 3), dtype=torch.uint8, device=images.device)
        pat = torch.empty((B * W.NPATCH, 3 * W.PATCH * W.PATCH), dtype=torch.float32, device=images.device)
        rc = self._lib.tstar_owl_debug_preprocess(self._h, images.data_ptr(), B, H, Wd, u8.data_ptr(),
                                                  pat.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tstar_owl_debug_preprocess")
        return u8, pat
