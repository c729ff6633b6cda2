"""Result wire format and downstream frame selection of the reference's dataset tools.

* ``make_result`` / ``save_results``: the per-item dict and the JSON file written by
  /root/reference/LVHaystackBench/run_TStar_onDataset.py:139-144, 204-211 (keys ``video_path``,
  ``grounding_objects``, ``keyframe_timestamps``, ``keyframe_distribution``), so the reference's
  evaluators (val_tstar_results.py, val_qa_results.py) read our output unchanged.
* ``topk_seconds``: the score-based selection of val_qa_results.py:90-110 on the GPU
  (tstar_topk_seconds).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib


def make_result(video_path: str, target_objects: Sequence[str], cue_objects: Sequence[str],
                time_stamps: Sequence[float], keyframe_distribution: Sequence[float]) -> Dict:
    ts = sorted(float(t) for t in time_stamps)                      # run_TStar_onDataset.py:128
    return {
        "video_path": video_path,
        "grounding_objects": {"target_objects": list(target_objects), "cue_objects": list(cue_objects)},
        "keyframe_timestamps": ts,
        "keyframe_distribution": [float(v) for v in keyframe_distribution],
    }


def result_from_searcher(searcher) -> Dict:
    """After ``searcher.search()``: the reference's result dict (P_history[-1] is the distribution)."""
    if not searcher.P_history:
        raise IndexError("P_history is empty (no search iteration ran)")       # as the reference's [-1]
    return make_result(str(searcher.video_path), searcher.target_objects, searcher.cue_objects,
                       searcher.last_time_stamps, searcher.P_history[-1])


def save_results(results: List[Dict], path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(results, f, indent=4, ensure_ascii=False)


def topk_seconds(distribution, num_frames: int = 8, clip: Optional[Sequence[float]] = None) -> np.ndarray:
    """Top-``num_frames`` seconds of ``distribution`` (list / numpy / cuda f64 tensor) inside ``clip``
    = (start_sec, end_sec), ascending -- val_qa_results.py:90-110 (ties -> lowest index)."""
    import torch
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.TStarHipError("topk_seconds needs a HIP device; tstar_amd has no CPU path")
    d = distribution if hasattr(distribution, "data_ptr") else torch.as_tensor(np.asarray(distribution, dtype=np.float64))
    d = d.to(device="cuda", dtype=torch.float64).contiguous()
    n = d.numel()
    start, end = (0, n) if clip is None else (max(0, int(clip[0])), min(n, int(clip[1])))
    k = min(int(num_frames), end - start)
    out = np.empty(k, dtype=np.int32)
    _lib.check(lib.tstar_topk_seconds(d.data_ptr(), n, start, end, k, out.ctypes.data, _lib.stream_ptr()), "tstar_topk_seconds")
    return out.astype(np.int64)
