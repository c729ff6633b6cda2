"""Lock-step search of several independent (video, question) items on ONE GPU.

The reference processes items one after another (LVHaystackBench/run_TStar_onDataset.py:195-205).
Items are independent, so their search iterations can advance together: iteration t of every active
item contributes its grid image to ONE detector batch and its verification frames to ONE verification
batch (each image scored against its own question's query set), while every item keeps its own
sampler stream, device state and sequential ``remaining_targets`` logic.  Results are bit-identical to
running the items one by one with the same per-item RNGs; the GPU sees larger GEMMs (M = sum of the
items' images x 577), fewer launches and fewer synchronisation points per item, and the items' host-side
FITPACK fits run side by side in worker processes (tstar_amd.spline_pool).

Two (or more) lock-step groups can be ALTERNATED on the one GPU (``search_lockstep_groups``): while the
detector works on one group's verification batch, the host does the other group's bookkeeping (replay of
its verification results, score write-back, histories, next iteration's samples) and queues that group's
grid forward behind it.  Every group runs exactly the statements it would run alone, in the same order, so
results do not change; what changes is that the detector stream only drains for the short step between a
group's grid forward and its verification batch.  For that the small per-item state kernels and their
read-backs (``tstar_searcher_*``) go to a side stream: a read-back on the detector stream would wait for
everything queued behind it by the other group.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

from . import spline_pool
from .interface_searcher import CELL_H, CELL_W, SAMPLER_WARNING, VERIFY_H, VERIFY_W, TStarSearcher

MAX_GROUP = 63          # TSTAR_OWL_MAX_SETS - 1 query-set slots (include/tstar_hip.h); slot 0 stays the heuristic's own

_SIDE = {}              # device index -> the side stream of the searcher-state kernels
_AUX = {}               # device index -> the stream of the speculative next-grid forwards (workspace lane 1)
_SPECULATE = os.environ.get("TSTAR_NO_SPECULATION") is None      # TSTAR_NO_SPECULATION=1: same-session A/Bs of speculate()
# Round 6: the speculative forward runs BESIDE the verification batch -- its own stream, the detector's second workspace (lane 1) --
# instead of behind it on the detector stream.  TSTAR_SPECULATE_BEHIND=1 restores round 5's placement (same-session A/Bs).
_BESIDE = os.environ.get("TSTAR_SPECULATE_BEHIND") is None
# ... and the next verification batch is queued behind the running one before its results are read (verify_ahead()).
# TSTAR_NO_VERIFY_AHEAD=1: off (same-session A/Bs).
_AHEAD = os.environ.get("TSTAR_NO_VERIFY_AHEAD") is None
_AHEAD_ALWAYS = False   # tests: queue the next batch early even when the running one has already finished (every path, deterministically)
_ONE_UPLOAD = os.environ.get("TSTAR_PER_ITEM_UPLOADS") is None   # TSTAR_PER_ITEM_UPLOADS=1: one index upload + torch.stack / cat per item (round 5; A/Bs)
AUX_IMAGES = 31         # grid images of one forward that may go to lane 1 (the workspace grows to the batch: include/tstar_hip.h)
# Alternating lock-step groups, OPT-IN (TSTAR_GROUP_GRIDS_BESIDE=1): a group's grid forward on the auxiliary stream / lane 1 beside the other
# group's verification batch instead of behind it, so that its cell masks are back -- and its own verification batch queued -- before the
# detector stream drains (the host's masks -> candidates -> frames step leaves the chip idle 1-3 ms per group iteration).  Measured on the
# bench's headline: +0.4 % (11.92 -> 11.96 k, 11.88 -> 11.94 k frames/s, same keyframes), while the overlapping launches inflate every
# per-launch duration the roofline leg times (frac 0.562 -> 0.527, "share of step" above 1): not worth blurring the evidence, so off by default.
_GROUP_BESIDE = os.environ.get("TSTAR_GROUP_GRIDS_BESIDE") is not None


def _side_stream(torch):
    d = torch.cuda.current_device()
    if d not in _SIDE:
        _SIDE[d] = torch.cuda.Stream(device=d)
    return _SIDE[d]


def _accepts_lane(fn) -> bool:
    """A wrapper installed over ``heuristic.score_batch`` (a recorder, a logger) with the pre-round-6 signature keeps working: the
    speculative forward then stays on the detector stream."""
    import inspect
    try:
        ps = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return "lane" in ps or any(p.kind == inspect.Parameter.VAR_KEYWORD for p in ps.values())


def _aux_stream(torch):
    d = torch.cuda.current_device()
    if d not in _AUX:
        _AUX[d] = torch.cuda.Stream(device=d)
    return _AUX[d]


class _Group:
    """One lock-step group: the statements of an iteration cut at the two points where the host has to wait for
    the detector (grid forward, verification batch)."""

    def __init__(self, searchers: Sequence[TStarSearcher], first_slot: int, torch):
        self.torch = torch
        self.ss = list(searchers)
        self.h = self.ss[0].heuristic
        self.rows, self.cols = tuple(self.ss[0].image_grid_shape)
        self.n = self.rows * self.cols
        for i, s in enumerate(self.ss):
            s._slot = first_slot + i
        self.main = torch.cuda.current_stream()
        self.side = _side_stream(torch)
        # speculative forwards beside the verification batch: only a detector with a second workspace (OWLInterface.aux_lane)
        self.aux = _aux_stream(torch) if (_BESIDE and getattr(self.h, "aux_lane", False) and _accepts_lane(self.h.score_batch)) else None
        self.pending = None            # the verification batch in flight (end() consumes it)
        self.act = []
        self.spec = None               # the NEXT iteration's samples / grid forward, queued speculatively (see speculate())
        self.spec_beside = False       # ... which runs beside the verification batch (auxiliary stream, lane 1), not behind it
        self.grids_beside = False      # EVERY grid forward of this group beside whatever the detector stream runs (alternating groups)
        self.ahead = None              # ... and its verification batch, queued behind the one in flight (see verify_ahead())
        self.vq = self.masks = None    # the verification batch of THIS iteration (verify_launch()) and the grid's cell masks
        self.solo = False              # one searcher driven through its own slot 0 and its public sample_frames hook

    def install(self):
        h = self.h
        self.side.wait_stream(self.main)          # state written on the caller's stream before the search (once: a wait per
                                                  # iteration would queue the samples behind the other group's batch)
        if self.aux is not None:
            self.aux.wait_stream(self.main)       # (the resident video may have been produced on the caller's stream just before)
        if self.solo:                             # the searcher's own question sits in slot 0 since its constructor
            self.act = self._active()
            return
        if hasattr(h, "install_queries_many"):                   # one text-tower forward for the whole group's questions
            texts = h.install_queries_many([(s._slot, s.target_objects, s.cue_objects, s.object2weight) for s in self.ss])
            for s, t in zip(self.ss, texts):
                s._texts = t
        else:
            for s in self.ss:
                s._texts = h.install_queries(s._slot, s.target_objects, s.cue_objects, s.object2weight)
        self.act = self._active()

    def _active(self):
        return [s for s in self.ss if s.remaining_targets and s.search_budget > 0]

    def _sets(self, items):
        return None if self.solo else [s._slot for s in items]

    def _draw(self, s):
        """The iteration's samples of one searcher.  A searcher running alone goes through its public ``sample_frames`` (the
        reference's call: wrappers / subclasses see exactly one call per executed iteration, at the reference's place)."""
        return [int(v) for v in s.sample_frames(self.n)[0]] if self.solo else s._sample_secs(self.n)

    def begin(self):
        """Samples of the iteration (state kernels, side stream), grid images and the grid forward (detector stream) -- or, when
        ``speculate()`` queued exactly this iteration behind the previous verification batch, nothing but taking it over."""
        torch = self.torch
        if self.spec is not None:
            items, secs_l, grids, res, ev, states, warned = self.spec
            if items != self.act:
                raise AssertionError("stale speculation: end() must have discarded it")
            self.spec = None
            # every speculated item is still searching (and no other is): the work is already queued
            if not self.solo:
                for w in warned:
                    if w:
                        print(SAMPLER_WARNING)        # the draw is adopted: the warning belongs to an executed iteration
                self.secs_l, self.grids, self.res, self.ev_grid = secs_l, grids, res, ev
                return
            # solo: the public hook sees the draw now, at the reference's place.  The generator is put back to where it was
            # BEFORE the speculative draw and the draw is handed to ``_sample_secs`` together with the state after it: a hook that
            # goes through the original method gets the speculated draw and leaves the generator where the sequential loop would;
            # a hook that replaces the method draws from the state the sequential loop would have given it.
            s, secs = items[0], secs_l[0]
            rng = s._rng if s._rng is not None else np.random
            post = rng.get_state()
            rng.set_state(states[0])
            s._prefetched_secs = (list(secs), post, warned[0])
            try:
                with torch.cuda.stream(self.side):    # (a replacing hook may run state kernels of its own)
                    got = [int(v) for v in s.sample_frames(self.n)[0]]
            finally:
                s._prefetched_secs = None             # never left behind for an unrelated later draw
            if got == secs:
                self.secs_l, self.grids, self.res, self.ev_grid = secs_l, grids, res, ev
                return
            # an override changed (or replaced) the draw: the iteration runs on ITS samples, as in the sequential loop; the
            # speculated forward is dropped (the image did go through the detector).  The budget was already charged.
            if self.ahead is not None:
                self._drop_ahead()                # (chosen from the dropped forward's masks)
            s.device_images_scored += 1
            cb = getattr(self.h, "_speculation_dropped", None)
            if cb is not None:
                cb(res)
            self._score_grids([got])
            return
        secs_l = []
        with torch.cuda.stream(self.side):
            for s in self.act:
                secs_l.append(self._draw(s))
                s.search_budget -= self.n
        self._score_grids(secs_l)

    def _score_grids(self, secs_l):
        """Grid images of the active items' samples and their forward: on the detector stream, or -- a group that alternates with
        others -- on the auxiliary stream in the detector's second workspace, beside the other group's verification batch."""
        torch = self.torch
        beside = self.grids_beside and self.aux is not None and len(self.act) <= AUX_IMAGES
        stream = self.aux if beside else self.main
        with torch.cuda.stream(stream):
            batch, grids = self._grid_batch(self.act, secs_l)
            self.secs_l, self.grids = secs_l, grids
            if beside:
                self.res = self.h.score_batch(batch, self.rows, self.cols, image_sets=self._sets(self.act), lane=1)
            else:
                self.res = self.h.score_batch(batch, self.rows, self.cols, image_sets=self._sets(self.act))
            self.ev_grid = torch.cuda.Event()
            self.ev_grid.record(stream)

    def _grid_batch(self, items, secs_l):
        """The items' grid images as ONE uint8 [n, rows * 95, cols * 200, 3] tensor (and its per-item views): one index upload, every item's
        gather / resize / tile kernel writes its slice."""
        torch = self.torch
        if not _ONE_UPLOAD:
            grids = [s._device_grid(secs) for s, secs in zip(items, secs_l)]
            return torch.stack(grids), grids
        dev = items[0].store.frames.device
        n = self.n
        for secs in secs_l:
            if len(secs) != n:
                raise ValueError("Frame count does not match grid dimensions")      # interface_searcher.py:183-184
        d_idx = torch.as_tensor([int(v) for secs in secs_l for v in secs], dtype=torch.int32, device=dev)
        batch = torch.empty((len(items), self.rows * CELL_H, self.cols * CELL_W, 3), dtype=torch.uint8, device=dev)
        for i, s in enumerate(items):
            s._device_grid_into(d_idx[i * n:(i + 1) * n], batch[i])
        return batch, [batch[i] for i in range(len(items))]

    def speculate(self):
        """Queue the NEXT iteration's grid forward behind the verification batch that ``middle()`` has just queued, BEFORE its
        results are known (review item 4b of round 4).  Valid because the next samples depend only on what is already final:
        ``sample_frames`` reads P and the unvisited mask (interface_searcher.py:345-358), both written by this iteration's grid
        stage, while verification only overwrites ``score_distribution[sec]`` and ``remaining_targets`` (:407, :416-419) -- so the
        draw is the one the sequential loop would make, from the same generator state.  If verification ends a search (its last
        target confirmed) the speculation of the whole group is discarded in ``end()``: generator states and budgets are
        restored, the queued forward's result is dropped (one wasted grid image, once per search).  Used when nothing else
        would keep the detector busy between a verification batch and the next grid forward: a single live group."""
        torch = self.torch
        items = [s for s in self.act if s.remaining_targets and s.search_budget > 0]
        if not items or len(items) != len(self.act):
            return                                # an item stops on its budget after this iteration: begin() draws for the rest
        states = [(s._rng if s._rng is not None else np.random).get_state() for s in items]
        secs_l, grids, warned = [], [], []
        with torch.cuda.stream(self.side):
            for s in items:
                w = []
                secs_l.append(s._sample_secs(self.n, _warned=w))   # NOT the public hook, and no warning printed: a discarded draw must stay invisible
                warned.append(bool(w and w[0]))
                s.search_budget -= self.n
        # Where the forward runs.  Behind the verification batch on the detector stream it would wait for it; the B = 1 forward of a search
        # running alone launches 120-456 wave tiles per GEMM for 1024 SIMDs and a verification batch of ~10 images as few as ~140 wide
        # blocks for 256 CUs -- side by side (another stream, the detector's second workspace: tstar_owl_score_lane) each fills what the
        # other leaves idle.  Same kernels, same tiles, same bits; lane 1 holds forward chunks of AUX_IMAGES images, so larger groups
        # keep the round-5 placement.
        beside = self.aux is not None and len(items) <= AUX_IMAGES
        stream = self.aux if beside else self.main
        with torch.cuda.stream(stream):
            batch, grids = self._grid_batch(items, secs_l)
            if beside:
                res = self.h.score_batch(batch, self.rows, self.cols, image_sets=self._sets(items), lane=1)
            else:
                res = self.h.score_batch(batch, self.rows, self.cols, image_sets=self._sets(items))
            ev = torch.cuda.Event()
            ev.record(stream)
        self.spec = (items, secs_l, grids, res, ev, states, warned)
        self.spec_beside = beside

    def _drop_speculation(self):
        if self.ahead is not None:
            self._drop_ahead()
        items, _, _, res, _, states, _ = self.spec
        self.spec = None
        for s, st in zip(items, states):
            (s._rng if s._rng is not None else np.random).set_state(st)
            s.search_budget += self.n
            s.device_images_scored += 1           # the image did go through the detector
        cb = getattr(self.h, "_speculation_dropped", None)       # observers of score_batch (a test recorder) drop it too
        if cb is not None:
            cb(res)

    def _verify_queue(self, act, secs_l, res, ev_grid):
        """Wait for a grid forward's cell masks and queue the verification batch they call for on the detector stream:
        -> (masks, (vres, vframes, event, candidates per item, row offsets, the remaining targets the candidates were chosen for))."""
        torch, h = self.torch, self.h
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev_grid)
            masks = res.cell_mask.cpu().numpy().astype(np.uint32)
        cand_l = []
        for i, s in enumerate(act):
            # candidates of the verification batch: cells whose mask lists a remaining target (the same test as
            # ``any(t in names ...)`` on the decoded names, which are only needed by the replay and are built in update())
            tb = 0
            for q, t in enumerate(s._texts):
                if t[0] in s.remaining_targets:
                    tb |= 1 << q
            cand_l.append(np.nonzero(masks[i][:len(secs_l[i])] & np.uint32(tb))[0].tolist())
        # ONE verification batch for every item of the group (device work only): it needs the cell masks only
        vres = vframes = ev = None
        offs = np.cumsum([0] + [len(c) for c in cand_l])
        if offs[-1] > 0:
            # ONE index upload and ONE output tensor for the whole group (each item's resize writes its slice): the host's share of the
            # step between a grid forward's masks and the first kernel of the verification batch is detector idle time
            if _ONE_UPLOAD:
                dev = act[0].store.frames.device
                d_idx = torch.as_tensor([secs_l[i][j] for i in range(len(act)) for j in cand_l[i]], dtype=torch.int32, device=dev)
                vframes = torch.empty((int(offs[-1]), VERIFY_H, VERIFY_W, 3), dtype=torch.uint8, device=dev)
                for i, s in enumerate(act):
                    if cand_l[i]:
                        s._device_resized_into(d_idx[int(offs[i]):int(offs[i + 1])], vframes[int(offs[i]):int(offs[i + 1])])
            else:
                vframes = torch.cat([s._device_verify_frames([secs_l[i][j] for j in cand_l[i]])
                                     for i, s in enumerate(act) if cand_l[i]])
            sets = None if self.solo else [s._slot for i, s in enumerate(act) for _ in cand_l[i]]
            vres = h.score_batch(vframes, 1, 1, image_sets=sets)
            ev = torch.cuda.Event()
            ev.record(self.main)
        return masks, (vres, vframes, ev, cand_l, offs, [list(s.remaining_targets) for s in act])

    def middle(self):
        """Wait for the grid forward; the verification batch (detector stream) and, under its shadow, the score write-back
        (side stream), the FITPACK fits, P and the histories."""
        self.verify_launch()
        self.update()

    def verify_launch(self):
        """The iteration's verification batch, queued before anything else of ``middle()``: the detector stream is empty until it
        arrives -- unless ``verify_ahead()`` already queued exactly this batch behind the previous one."""
        if self.ahead is not None:
            self.masks, self.vq = self.ahead
            self.ahead = None
            return
        self.masks, self.vq = self._verify_queue(self.act, self.secs_l, self.res, self.ev_grid)

    def verify_ahead(self):
        """Round 6, a search running alone: queue the NEXT iteration's verification batch as soon as its (speculative) grid forward is
        back -- behind the verification batch that is still running, BEFORE that batch's results are read -- so that the detector stream
        goes from one verification batch straight into the next and the host's replay / write-back / fit run in its shadow.  Valid
        because the candidates of a verification batch are the cells of the NEXT grid image that list a remaining target
        (interface_searcher.py:481-486): they depend on this iteration's verification only through ``remaining_targets``.  ``end()``
        keeps the batch if the replay left ``remaining_targets`` as they were when the candidates were chosen; if a target was
        found, the batch is dropped (the search is over, or ``verify_launch()`` queues the smaller batch of the sequential loop)."""
        if self.spec is None or self.ahead is not None:
            return
        items, secs_l, _, res, ev, _, _ = self.spec
        self.ahead = self._verify_queue(items, secs_l, res, ev)

    def verification_done(self) -> bool:
        """Has the verification batch in flight already finished (nothing would be gained by queueing the next one behind it)?"""
        ev = self.vq[2] if self.vq is not None else None
        return ev is None or ev.query()

    def _drop_ahead(self):
        _, (vres, _, _, cand_l, _, _) = self.ahead
        self.ahead = None
        for s, c in zip(self.spec[0] if self.spec is not None else self.act, cand_l):
            s.device_images_scored += len(c)      # the frames did go through the detector
        cb = getattr(self.h, "_speculation_dropped", None)
        if cb is not None and vres is not None:
            cb(vres)

    def update(self):
        """The grid scores go into the per-item state (write-back, window spread, visited list: side stream) while the host builds the
        sampling distributions; histories."""
        torch, h, act, res, secs_l, n, masks = self.torch, self.h, self.act, self.res, self.secs_l, self.n, self.masks
        names_l, fits = [], []
        with torch.cuda.stream(self.side):
            for i, s in enumerate(act):
                names_l.append([s._names_from_mask(int(m)) for m in masks[i][:len(secs_l[i])]])
                s.device_images_scored += 1
                if s.keep_visual_history:
                    imgs, dets = h.annotated_batch(self.grids[i].unsqueeze(0), res, i, 1)
                    if self.solo:
                        h.detections_inbatch = dets                 # refreshed per call, like the reference's (B.13)
                    s.image_grid_iters.append([imgs[0]])
                    s.detect_annotot_iters.append([imgs[0]])
                    s.detect_bbox_iters.append(dets)
                s.frames_scored += n
                s.detector_calls += 1
                fits.append(s._state.apply_grid(secs_l[i], res.cell_conf[i]))
        # ... while the host builds the sampling distributions (FITPACK fit + sigmoid, interface_searcher.py:262-274):
        # one worker process per item (tstar_amd.spline_pool), same scipy / numpy calls, bit-identical P
        with torch.cuda.stream(self.side):
            for s, P in zip(act, spline_pool.distribution_many(fits, [a.total_frame_num for a in act], s=0.5)):
                s._state.write(2, P)
            for s in act:
                if type(s).store_score_distribution is TStarSearcher.store_score_distribution:
                    s.store_score_distribution(_defer_lists=True)   # numpy -> list conversions: in end(), off the fit -> next-draw chain
                else:
                    s.store_score_distribution()                    # a subclass's own method, with the reference's signature
        vres, vframes, ev, cand_l, offs, _ = self.vq
        self.pending = (vres, vframes, ev, cand_l, offs, names_l)

    def end(self):
        """Wait for the verification batch; the sequential ``remaining_targets`` replay and its score overwrites."""
        for s in self.act:
            s._finalize_history()                 # (deferred by update(): the next forward is queued by now)
        if self.pending is None:
            if self.spec is not None:             # (defensive: a speculation always follows an update())
                self._drop_speculation()
            return
        torch = self.torch
        vres, vframes, ev, cand_l, offs, names_l = self.pending
        self.pending = None
        act, secs_l = self.act, self.secs_l
        if vres is not None:
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                vconf = vres.cell_conf[:, 0].cpu().numpy()
                vmask = vres.cell_mask[:, 0].cpu().numpy().astype(np.uint32)
                for i, s in enumerate(act):
                    if cand_l[i]:
                        s.device_images_scored += len(cand_l[i])
                        s._verify_replay(cand_l[i], vconf[offs[i]:offs[i + 1]], vmask[offs[i]:offs[i + 1]], secs_l[i],
                                         names_l[i], vframes, vres, int(offs[i]))
        for s in act:
            s.iterations += 1
        self.res = self.grids = None
        self.vq = None
        if self.spec is not None:
            # the speculation was made with this iteration's budget already spent, so "still active" is the same test it used
            still = [s for s in self.spec[0] if s.remaining_targets]
            if len(still) != len(self.spec[0]):
                self._drop_speculation()
            elif self.ahead is not None and any(list(s.remaining_targets) != snap for s, snap in zip(self.spec[0], self.ahead[1][5])):
                self._drop_ahead()                # a target was found and the search goes on: the next batch has fewer candidates
        self.act = self._active() if self.spec is None else list(self.spec[0])

    def finish(self):
        torch = self.torch
        out = []
        with torch.cuda.stream(self.side):
            for s in self.ss:
                s._finalize_history()
                frames, ts = s.pop_frames(video_path=s.video_path, num_samples=s.search_nframes)
                s.last_time_stamps = list(ts)
                out.append((frames, ts))
        self.main.wait_stream(self.side)          # whoever uses the searchers next on this stream sees the final state
        if self.aux is not None:
            self.main.wait_stream(self.aux)       # ... and finds lane 1 idle (a dropped speculative forward may still be running)
        return out


def search_lockstep_groups(groups: Sequence[Sequence[TStarSearcher]]) -> List[List[Tuple[np.ndarray, list]]]:
    """Run ``search()`` of every searcher, each inner list as one lock-step group, the groups alternating on the GPU
    (see the module docstring); returns, per group, [(frames, time_stamps)] in input order.

    All searchers must share one tstar_amd detector interface (fast path), the searchers of a group the same grid
    shape, and every searcher carries its own ``rng`` (with the process-global numpy generator the draw order would
    depend on the interleaving, unlike sequential runs).  At most 63 items in total (query-set slots 1..63)."""
    import torch
    groups = [list(g) for g in groups if len(g)]
    if not groups:
        return []
    if sum(len(g) for g in groups) > MAX_GROUP:
        raise ValueError(f"search_lockstep: at most {MAX_GROUP} items in flight (query-set slots)")
    h = groups[0][0].heuristic
    for g in groups:
        shape = tuple(g[0].image_grid_shape)
        for s in g:
            if s.heuristic is not h or not s._fast:
                raise ValueError("search_lockstep: all searchers must share one tstar_amd OWLInterface")
            if tuple(s.image_grid_shape) != shape:
                raise ValueError("search_lockstep: all searchers must use the same image_grid_shape")
            if s._rng is None:
                raise ValueError("search_lockstep: every searcher needs its own rng= (a seeded RandomState)")
    gs, slot = [], 1
    for g in groups:
        gs.append(_Group(g, slot, torch))
        slot += len(g)
    for g in gs:
        g.grids_beside = _GROUP_BESIDE and len(gs) > 1
        g.install()
    live = [g for g in gs if g.act]
    done = {}
    while live:
        for g in list(live):
            g.end()                    # nothing on the first pass; otherwise the other groups' work has covered the wait
            if not g.act:
                live.remove(g)
                done[id(g)] = g.finish()   # its final draws and the keyframes' device -> host copies run under the other groups' detector work
                continue
            g.begin()
            g.middle()
            if len(live) == 1 and _SPECULATE:
                g.speculate()          # nothing else would fill the detector between this group's verification and its next grid forward
    return [done[id(g)] if id(g) in done else g.finish() for g in gs]


def search_solo(searcher: TStarSearcher) -> Tuple[np.ndarray, list]:
    """``TStarSearcher.search()`` of ONE searcher on the fast path: the same statements in the same order as the sequential loop
    (interface_searcher.py:444-491 of the reference), cut at the two points where the host waits for the detector, with the state
    kernels on the side stream, the next iteration's grid forward queued speculatively BESIDE each verification batch (its own stream
    and detector workspace: ``_Group.speculate``) and, round 6, the next verification batch queued behind the running one as soon as
    that forward is back (``_Group.verify_ahead``).  The searcher keeps its own query-set slot 0 and its public ``sample_frames`` hook.

    What the detector sees: verification batch t | verification batch t + 1 | ... back to back on the caller's stream, grid forward
    t + 1 beside batch t on the auxiliary stream; what the host does meanwhile: fit(t) -> samples(t + 1) -> [wait for grid forward
    t + 1] -> queue batch t + 1 -> results of batch t, replay.  Every state-changing statement keeps its place in the reference's
    order: write-back / spread / P of iteration t + 1 come after the score overwrites of verification t."""
    import torch
    g = _Group([searcher], 0, torch)
    g.solo = True
    searcher._slot = 0
    g.install()
    if not g.act:
        return g.finish()[0]
    g.begin()
    g.verify_launch()
    while True:
        g.update()                     # write-back, fit, P, histories of iteration t (its verification batch is running)
        if _SPECULATE:
            g.speculate()              # samples and grid forward of iteration t + 1, beside the verification batch
            if _AHEAD and g.spec_beside and (_AHEAD_ALWAYS or not g.verification_done()):
                g.verify_ahead()       # ... and its verification batch behind the one in flight (a forward queued BEHIND the running
                                       # batch -- a detector without a second workspace -- would only make the host wait for both)
        g.end()                        # verification results of iteration t, replay
        if not g.act:
            break
        g.begin()
        g.verify_launch()
    return g.finish()[0]


def search_lockstep(searchers: Sequence[TStarSearcher]) -> List[Tuple[np.ndarray, list]]:
    """Run ``search()`` of every searcher in lock-step (one group); returns [(frames, time_stamps)] in input order."""
    if not searchers:
        return []
    return search_lockstep_groups([searchers])[0]
