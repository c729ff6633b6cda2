"""Lock-step search of several independent (video, question) items on ONE GPU.

The reference processes items one after another (LVHaystackBench/run_TStar_onDataset.py:195-205).
Items are independent, so their search iterations can advance together: iteration t of every active
item contributes its grid image to ONE detector batch and its verification frames to ONE verification
batch (each image scored against its own question's query set), while every item keeps its own
sampler stream, device state and sequential ``remaining_targets`` logic.  Results are bit-identical to
running the items one by one with the same per-item RNGs; the GPU sees larger GEMMs (M = sum of the
items' images x 577), fewer launches and fewer synchronisation points per item, and the items' host-side
FITPACK fits run side by side in worker processes (tstar_amd.spline_pool).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import spline_pool
from .interface_searcher import TStarSearcher

MAX_GROUP = 31          # TSTAR_OWL_MAX_SETS - 1 query-set slots (include/tstar_hip.h)


def search_lockstep(searchers: Sequence[TStarSearcher]) -> List[Tuple[np.ndarray, list]]:
    """Run ``search()`` of every searcher in lock-step; returns [(frames, time_stamps)] in input order.

    All searchers must share one tstar_amd ``OWLInterface`` (fast path), use the same grid shape and
    carry their own ``rng`` (with the process-global numpy generator the draw order would depend on the
    interleaving, unlike sequential runs).  At most 31 items at a time (query-set slots 1..31; slot 0 stays the heuristic's own)."""
    import torch
    if not searchers:
        return []
    if len(searchers) > MAX_GROUP:
        raise ValueError(f"search_lockstep: at most {MAX_GROUP} items per lock-step group")
    h = searchers[0].heuristic
    shape = tuple(searchers[0].image_grid_shape)
    for s in searchers:
        if s.heuristic is not h or not s._fast:
            raise ValueError("search_lockstep: all searchers must share one tstar_amd OWLInterface")
        if tuple(s.image_grid_shape) != shape:
            raise ValueError("search_lockstep: all searchers must use the same image_grid_shape")
        if s._rng is None:
            raise ValueError("search_lockstep: every searcher needs its own rng= (a seeded RandomState)")
    rows, cols = shape
    n = rows * cols
    for i, s in enumerate(searchers):
        s._slot = i + 1
    if hasattr(h, "install_queries_many"):                   # one text-tower forward for the whole group's questions
        texts = h.install_queries_many([(s._slot, s.target_objects, s.cue_objects, s.object2weight) for s in searchers])
        for s, t in zip(searchers, texts):
            s._texts = t
    else:
        for s in searchers:
            s._texts = h.install_queries(s._slot, s.target_objects, s.cue_objects, s.object2weight)

    def active():
        return [s for s in searchers if s.remaining_targets and s.search_budget > 0]

    act = active()
    while act:
        secs_l, grids = [], []
        for s in act:
            secs = s._sample_secs(n)
            s.search_budget -= n
            secs_l.append(secs)
            grids.append(s._device_grid(secs))
        res = h.score_batch(torch.stack(grids), rows, cols, image_sets=[s._slot for s in act])
        masks = res.cell_mask.cpu().numpy().astype(np.uint32)
        names_l, fits, cand_l = [], [], []
        for i, s in enumerate(act):
            s.device_images_scored += 1
            names = [s._names_from_mask(int(m)) for m in masks[i][:len(secs_l[i])]]
            names_l.append(names)
            if s.keep_visual_history:
                imgs, dets = h.annotated_batch(grids[i].unsqueeze(0), res, i, 1)
                s.image_grid_iters.append([imgs[0]])
                s.detect_annotot_iters.append([imgs[0]])
                s.detect_bbox_iters.append(dets)
            s.frames_scored += n
            s.detector_calls += 1
            fits.append(s._state.apply_grid(secs_l[i], res.cell_conf[i]))
            cand_l.append([j for j, nm in enumerate(names) if any(t in nm for t in s.remaining_targets)])
        # ONE verification batch for every item of the group (device work only) ...
        vres = vframes = None
        offs = np.cumsum([0] + [len(c) for c in cand_l])
        if offs[-1] > 0:
            vframes = torch.cat([s._device_verify_frames([secs_l[i][j] for j in cand_l[i]])
                                 for i, s in enumerate(act) if cand_l[i]])
            sets = [s._slot for i, s in enumerate(act) for _ in cand_l[i]]
            vres = h.score_batch(vframes, 1, 1, image_sets=sets)
        # ... while the host builds the sampling distributions (FITPACK fit + sigmoid, interface_searcher.py:262-274):
        # one worker process per item (tstar_amd.spline_pool), same scipy / numpy calls, bit-identical P
        for s, P in zip(act, spline_pool.distribution_many(fits, [a.total_frame_num for a in act], s=0.5)):
            s._state.write(2, P)
        for s in act:
            s.store_score_distribution()
        if vres is not None:
            vconf = vres.cell_conf[:, 0].cpu().numpy()
            vmask = vres.cell_mask[:, 0].cpu().numpy().astype(np.uint32)
            for i, s in enumerate(act):
                if cand_l[i]:
                    s.device_images_scored += len(cand_l[i])
                    s._verify_replay(cand_l[i], vconf[offs[i]:offs[i + 1]], vmask[offs[i]:offs[i + 1]], secs_l[i],
                                     names_l[i], vframes, vres, int(offs[i]))
        for s in act:
            s.iterations += 1
        act = active()
    out = []
    for s in searchers:
        frames, ts = s.pop_frames(video_path=s.video_path, num_samples=s.search_nframes)
        s.last_time_stamps = list(ts)
        out.append((frames, ts))
    return out
