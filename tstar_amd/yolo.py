"""Host wrapper of the HIP YOLO-World detector (tstar_yolo_* in include/tstar_hip.h).

PyTorch is used for device memory and streams only; the network is executed by the hand-written f32 VALU kernels of
csrc/yolo.hip from the layer program tstar_amd.yolo_world.build_program emits.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from . import yolo_world as Y


@dataclass
class YoloResult:
    """Device tensors of one tstar_yolo_detect call (detections are sorted by descending score, padded to max_dets)."""
    scores: "object"        # f32 [B,max_dets]   (0 beyond n_det)
    labels: "object"        # i32 [B,max_dets]   (-1 beyond n_det)
    boxes: "object"         # f32 [B,max_dets,4] xyxy pixels of the passed image
    n_kept: "object"        # i32 [B]
    cell_conf: "object" = None   # f64 [B,rows*cols]
    cell_mask: "object" = None   # i32 view of u32 [B,rows*cols]
    dense_scores: "object" = None
    dense_boxes: "object" = None


class YoloDetector:
    """One YOLO-World-v2 detector resident on the current HIP device."""

    def __init__(self, state_dict, scale: str = "l", max_batch: int = 16):
        import torch
        if not torch.cuda.is_available():
            raise _lib.TStarHipError("YoloDetector needs a HIP device (torch.cuda.is_available() is False); tstar_amd has no CPU path")
        self._torch = torch
        self._lib = _lib.load()
        prog = Y.build_program(state_dict, scale)
        self.arch = prog["arch"]
        self.conv_flops_per_image = Y.conv_flops(prog)
        blob = np.ascontiguousarray(prog["blob"], dtype=np.float32)
        ops = np.ascontiguousarray(prog["ops"], dtype=np.int32)
        bufs = np.ascontiguousarray(prog["bufs"], dtype=np.int32)
        guides = np.ascontiguousarray(prog["guides"], dtype=np.int32)
        levels = np.ascontiguousarray(prog["levels"], dtype=np.int32)
        h = C.c_void_p()
        rc = self._lib.tstar_yolo_create(C.byref(h), blob.ctypes.data, blob.size, ops.ctypes.data, ops.shape[0], ops.shape[1],
                                         bufs.ctypes.data, bufs.shape[0], guides.ctypes.data, guides.shape[0], levels.ctypes.data,
                                         levels.shape[0], int(prog["input_buf"]), int(max_batch))
        _lib.check(rc, "tstar_yolo_create")
        self._h = h
        self.max_batch = int(max_batch)
        self.n_anchor = int(self._lib.tstar_yolo_num_anchors(h))
        self.Qs = {}
        self.device = torch.device("cuda", torch.cuda.current_device())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tstar_yolo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_text_feats(self, text_feats: np.ndarray, class_weight: Sequence[float], slot: int = 0):
        t = np.ascontiguousarray(text_feats, dtype=np.float32)
        w = np.ascontiguousarray(class_weight, dtype=np.float64)
        if t.ndim != 2 or t.shape[1] != Y.TEXT_DIM or w.shape != (t.shape[0],):
            raise ValueError("set_text_feats: text_feats [Q,512] and class_weight [Q]")
        _lib.check(self._lib.tstar_yolo_set_text_feats(self._h, int(slot), t.ctypes.data, w.ctypes.data, t.shape[0], _lib.stream_ptr()),
                   "tstar_yolo_set_text_feats")
        self.Qs[int(slot)] = t.shape[0]

    def set_class_weights(self, class_weight: Sequence[float], slot: int = 0):
        w = np.ascontiguousarray(class_weight, dtype=np.float64)
        if w.shape != (self.Qs.get(int(slot), 0),):
            raise ValueError("set_class_weights: one weight per installed query")
        _lib.check(self._lib.tstar_yolo_set_class_weights(self._h, int(slot), w.ctypes.data, len(w), _lib.stream_ptr()),
                   "tstar_yolo_set_class_weights")

    def detect(self, images, grid_rows: int = 1, grid_cols: int = 1, score_threshold: float = 0.12, max_dets: int = 50,
               image_sets: Optional[Sequence[int]] = None, want_dense: bool = False, want_cells: bool = True) -> YoloResult:
        """images: torch u8 cuda tensor [B,H,W,3] (RGB, as the searcher hands them to the reference)."""
        torch = self._torch
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_cuda:
            raise ValueError("detect: images must be a cuda uint8 tensor [B,H,W,3]")
        images = images.contiguous()
        B, H, W, _ = images.shape
        dev = images.device
        r = YoloResult(scores=torch.empty((B, max_dets), dtype=torch.float32, device=dev),
                       labels=torch.empty((B, max_dets), dtype=torch.int32, device=dev),
                       boxes=torch.empty((B, max_dets, 4), dtype=torch.float32, device=dev),
                       n_kept=torch.empty((B,), dtype=torch.int32, device=dev))
        if want_cells:
            r.cell_conf = torch.empty((B, grid_rows * grid_cols), dtype=torch.float64, device=dev)
            r.cell_mask = torch.empty((B, grid_rows * grid_cols), dtype=torch.int32, device=dev)
        sets = None
        if image_sets is not None:
            sets = np.ascontiguousarray(image_sets, dtype=np.int32)
            if sets.shape != (B,):
                raise ValueError("detect: image_sets needs one slot per image")
        if want_dense:
            qs = {self.Qs.get(int(v), 0) for v in (sets if sets is not None else [0])}
            if len(qs) != 1:
                raise ValueError("detect: dense scores need the same query count for every image")
            r.dense_scores = torch.empty((B, self.n_anchor, qs.pop()), dtype=torch.float32, device=dev)
            r.dense_boxes = torch.empty((B, self.n_anchor, 4), dtype=torch.float32, device=dev)
        rc = self._lib.tstar_yolo_detect(self._h, images.data_ptr(), B, H, W, int(grid_rows), int(grid_cols),
                                         None if sets is None else sets.ctypes.data, float(score_threshold), int(max_dets),
                                         r.scores.data_ptr(), r.labels.data_ptr(), r.boxes.data_ptr(), r.n_kept.data_ptr(),
                                         _lib.ptr(r.cell_conf), _lib.ptr(r.cell_mask), _lib.ptr(r.dense_scores), _lib.ptr(r.dense_boxes),
                                         _lib.stream_ptr())
        _lib.check(rc, "tstar_yolo_detect")
        return r
