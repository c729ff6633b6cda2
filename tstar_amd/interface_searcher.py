"""``TStarSearcher`` with the reference's constructor, attributes and ``search()`` contract
(/root/reference/TStar/interface_searcher.py:14-538), running on one MI355X:

* frames live in HBM (tstar_amd.video.FrameStore); gather + the cv2-style resizes + grid tiling
  are one HIP kernel (tstar_frames_to_grid), verification frames another (tstar_frames_resize);
* grid scoring and the detection -> grid-cell aggregation are tstar_owl_score;
* the float64 searcher state (score_distribution, non_visiting_frames, P) lives on the device and
  is updated by the tstar_searcher_* kernels; only the smoothing-spline distribution of the <= 1008
  visited frames (scipy.interpolate.UnivariateSpline + numpy's exp, the reference's own calls,
  :262-274 -- which makes P bit-identical to the reference's by construction) and the MT19937 draws
  (numpy legacy RandomState, as the reference's np.random.choice, :353,372) run on the host;
* verification (:382-420) is batched speculatively: every candidate frame of an iteration is
  scored in one launch, then the reference's sequential ``remaining_targets`` logic is replayed on
  the host, so results are identical to the one-call-per-frame loop.

Behavioural quirks of the reference that callers can observe are kept:
blank query, 0.005 threshold, verification overwriting ``score_distribution``, the sampler's
fallback branch, weighted-random (not top-k) final keyframes, ``Score_history``-based first
iteration, budget overshoot.  ``search_with_visualization`` is ``search`` (:493-538).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .spline_worker import spline_distribution      # FITPACK fit + sigmoid on the host, as the reference (:262-274)
from .video import FrameStore, open_video

CELL_W, CELL_H = 200, 95            # create_image_grid's hard-coded cell size (:186)
VERIFY_W, VERIFY_H = 200 * 3, 95 * 3  # verify_and_remove_target's resize (:403)
SAMPLER_WARNING = "Warning: Not enough non-zero entries, adjusting probability distribution."      # (:350)


class _DeviceState:
    """ctypes wrapper of the tstar_searcher_* entry points."""

    def __init__(self, n_frames: int, init_score: float, init_p: float):
        self.lib = _lib.load()
        self.N = n_frames
        h = C.c_void_p()
        _lib.check(self.lib.tstar_searcher_create(C.byref(h), n_frames, init_score, init_p), "tstar_searcher_create")
        self.h = h
        self._vx = np.empty(n_frames + 16, dtype=np.int32)
        self._vy = np.empty(n_frames + 16, dtype=np.float64)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tstar_searcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply_grid(self, secs, d_conf) -> Tuple[np.ndarray, np.ndarray]:
        s = np.ascontiguousarray(secs, dtype=np.int32)
        nv = C.c_int(0)
        _lib.check(self.lib.tstar_searcher_apply_grid(self.h, s.ctypes.data, d_conf.data_ptr(), len(s), C.byref(nv),
                                                      self._vx.ctypes.data, self._vy.ctypes.data, _lib.stream_ptr()),
                   "tstar_searcher_apply_grid")
        return self._vx[:nv.value].copy(), self._vy[:nv.value].copy()

    def set_spline(self, t, c, k):
        t = np.ascontiguousarray(t, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        if len(c) < len(t):
            c = np.concatenate([c, np.zeros(len(t) - len(c))])
        _lib.check(self.lib.tstar_searcher_set_spline(self.h, t.ctypes.data, c.ctypes.data, len(t), int(k),
                                                      _lib.stream_ptr()), "tstar_searcher_set_spline")

    def sampler_prep(self, num: int, add: float) -> bool:
        fb = C.c_int(0)
        _lib.check(self.lib.tstar_searcher_sampler_prep(self.h, num, add, C.byref(fb), _lib.stream_ptr()),
                   "tstar_searcher_sampler_prep")
        return bool(fb.value)

    def pop_prep(self) -> Tuple[int, float]:
        """-> (count_nonzero(p > 0), score.sum()) of the weights just prepared."""
        nnz, tot = C.c_int(0), C.c_double(0.0)
        _lib.check(self.lib.tstar_searcher_pop_prep(self.h, C.byref(nnz), C.byref(tot), _lib.stream_ptr()),
                   "tstar_searcher_pop_prep")
        return int(nnz.value), float(tot.value)

    def window_spread(self, secs, confs, window: int):
        s = np.ascontiguousarray(secs, dtype=np.int32)
        c = np.ascontiguousarray(confs, dtype=np.float64)
        if s.shape != c.shape or s.ndim != 1:
            raise ValueError("window_spread: one confidence per sampled frame")
        if len(s) == 0:
            return
        _lib.check(self.lib.tstar_searcher_window_spread(self.h, s.ctypes.data, c.ctypes.data, len(s), int(window),
                                                         _lib.stream_ptr()), "tstar_searcher_window_spread")

    def visited(self) -> Tuple[np.ndarray, np.ndarray]:
        nv = C.c_int(0)
        _lib.check(self.lib.tstar_searcher_visited(self.h, C.byref(nv), self._vx.ctypes.data, self._vy.ctypes.data,
                                                   _lib.stream_ptr()), "tstar_searcher_visited")
        return self._vx[:nv.value].copy(), self._vy[:nv.value].copy()

    def write(self, which: int, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.shape != (self.N,):
            raise ValueError(f"searcher state arrays have length {self.N}")
        _lib.check(self.lib.tstar_searcher_write(self.h, int(which), v.ctypes.data, _lib.stream_ptr()), "tstar_searcher_write")

    def draw(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(len(x), dtype=np.int32)
        _lib.check(self.lib.tstar_searcher_draw(self.h, x.ctypes.data, len(x), out.ctypes.data, _lib.stream_ptr()),
                   "tstar_searcher_draw")
        return out

    def exclude(self, found):
        f = np.ascontiguousarray(found, dtype=np.int32)
        _lib.check(self.lib.tstar_searcher_exclude(self.h, f.ctypes.data, len(f), _lib.stream_ptr()),
                   "tstar_searcher_exclude")

    def set_scores(self, secs, vals):
        s = np.ascontiguousarray(secs, dtype=np.int32)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        _lib.check(self.lib.tstar_searcher_set_scores(self.h, s.ctypes.data, v.ctypes.data, len(s), _lib.stream_ptr()),
                   "tstar_searcher_set_scores")

    def read_state(self) -> np.ndarray:
        """[3, N]: P, score_distribution, non_visiting_frames (one synchronisation)."""
        out = np.empty((3, self.N), dtype=np.float64)
        _lib.check(self.lib.tstar_searcher_read_state(self.h, out.ctypes.data, _lib.stream_ptr()), "tstar_searcher_read_state")
        return out

    def read(self, which: int) -> np.ndarray:
        out = np.empty(self.N, dtype=np.float64)
        _lib.check(self.lib.tstar_searcher_read(self.h, which, out.ctypes.data, _lib.stream_ptr()), "tstar_searcher_read")
        return out


class _ResizedFrames:
    """The second element of ``sample_frames``' return value: the sampled frames resized to (out_w, out_h) --
    ``[cv2.resize(frame, (800, 380)) for frame in frames]`` in the reference (:362) -- materialised on first
    access by the device resize kernel (tstar_frames_resize) and ONE device->host copy."""

    def __init__(self, searcher, secs, out_w: int, out_h: int):
        self.searcher, self.secs, self.out_w, self.out_h = searcher, [int(s) for s in secs], out_w, out_h
        self._host = None

    def _materialise(self) -> np.ndarray:
        if self._host is None:
            self._host = self.searcher._device_resized(self.secs, self.out_w, self.out_h).cpu().numpy()
        return self._host

    def __len__(self) -> int:
        return len(self.secs)

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


class TStarSearcher:
    """Keyframe search by object detection and dynamic sampling (drop-in for the reference class)."""

    def __init__(
        self,
        video_path,
        heuristic,
        target_objects: List[str],
        cue_objects: List[str],
        search_nframes: int = 8,
        image_grid_shape: Tuple[int, int] = (8, 8),
        search_budget: float = 0.1,
        output_dir: Optional[str] = None,
        confidence_threshold: float = 0.5,
        object2weight: Optional[dict] = None,
        *,
        rng: Optional[np.random.RandomState] = None,
        keep_visual_history: bool = True,
    ):
        """Arguments as the reference (:21-48).  ``video_path`` may also be a FrameStore or a
        ``synthetic://`` URL.  Extra keyword-only arguments: ``rng`` (a seeded legacy
        ``RandomState``; default = the process-global numpy generator the reference draws from)
        and ``keep_visual_history`` (False skips the device->host copies that only feed the
        visualisation attributes)."""
        self.video_path = video_path
        self.target_objects = target_objects
        self.cue_objects = cue_objects
        self.search_nframes = search_nframes
        self.image_grid_shape = image_grid_shape
        self.output_dir = output_dir
        self.confidence_threshold = confidence_threshold
        self.object2weight = object2weight if object2weight else {}
        self.fps = 1

        self.store: FrameStore = open_video(video_path)       # ValueError("Cannot open video file...") as :61-62
        self.raw_fps = self.store.raw_fps
        total_frames = int(self.store.raw_total_frames)
        self.duration = total_frames / self.raw_fps
        self.total_frame_num = int(self.duration * self.fps)
        if self.total_frame_num > self.store.num_seconds:
            raise ValueError("FrameStore holds fewer frames than duration * fps")
        self.remaining_targets = target_objects.copy()
        self.search_budget = min(1000, self.total_frame_num * search_budget)

        self._state = _DeviceState(self.total_frame_num, 1e-6, confidence_threshold * 0.3)
        self.P_history = []
        self.Score_history = []
        self.non_visiting_history = []
        self.image_grid_iters = []
        self.detect_annotot_iters = []
        self.detect_bbox_iters = []

        self.heuristic = heuristic
        self.heuristic.reparameterize_object_list(target_objects, cue_objects)
        for obj in target_objects:
            self.object2weight[obj] = 1.0
        for obj in cue_objects:
            self.object2weight[obj] = 0.5
        self._fast = hasattr(heuristic, "score_batch") and hasattr(heuristic, "set_class_weights")
        if self._fast:
            heuristic.set_class_weights(self.object2weight)
        self._texts = [list(t) for t in heuristic.texts]      # this searcher's class names (label -> name)

        self._rng = rng
        self.keep_visual_history = keep_visual_history
        self.frames_scored = 0        # g*g per grid call + 1 per verification call (BASELINE metric)
        self.detector_calls = 0       # calls the reference would have made
        self.device_images_scored = 0  # detector images actually pushed through the GPU
        self.iterations = 0

    # ---- state views (numpy copies of the device arrays, like the reference's attributes) -----
    # Reading gives a fresh host copy; ASSIGNING (as a caller of the reference may: they are plain attributes
    # there) uploads the array.  In-place edits of a copy do not reach the device -- assign it back.
    @property
    def score_distribution(self) -> np.ndarray:
        return self._state.read(0)

    @score_distribution.setter
    def score_distribution(self, values):
        self._state.write(0, values)

    @property
    def non_visiting_frames(self) -> np.ndarray:
        return self._state.read(1)

    @non_visiting_frames.setter
    def non_visiting_frames(self, values):
        self._state.write(1, values)

    @property
    def P(self) -> np.ndarray:
        return self._state.read(2)

    @P.setter
    def P(self, values):
        self._state.write(2, values)

    # ---- helpers ---------------------------------------------------------------------------------
    def _uniform(self, k: int) -> np.ndarray:
        return (self._rng if self._rng is not None else np.random).random_sample(k)

    def _choice(self, size: int, nnz: Optional[int] = None, total: Optional[float] = None) -> np.ndarray:
        """numpy legacy RandomState.choice(N, size, replace=False, p) over the device cdf
        (:353-358, :372): MT19937 doubles on the host, searchsorted on the device.  ``nnz`` / ``total``
        (count of p > 0, the sum p was normalised by) feed numpy's own argument checks, in numpy's order."""
        if total is not None and not (total == total and total != 0 and abs(total) != float("inf")):
            raise ValueError("probabilities contain NaN")            # p = score / 0 (or / nan)
        if size > self.total_frame_num:                  # numpy's own check (mtrand choice, replace=False)
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        if nnz is not None and nnz < size:
            raise ValueError("Fewer non-zero entries in p than size")
        found: List[int] = []
        while len(found) < size:
            x = self._uniform(size - len(found))
            if found:
                self._state.exclude(found)
            new = self._state.draw(x)
            _, first = np.unique(new, return_index=True)
            first.sort()
            found.extend(int(v) for v in new.take(first))
        return np.asarray(found[:size], dtype=np.int64)

    def _names_from_mask(self, mask: int) -> List[str]:
        return [self._texts[q][0] for q in range(len(self._texts)) if (mask >> q) & 1]

    def _d_idx(self, secs):
        import torch
        return torch.as_tensor([int(s) for s in secs], dtype=torch.int32, device=self.store.frames.device)

    # ---- detection -------------------------------------------------------------------------------
    def imageGridScoreFunction(self, images: List[np.ndarray], output_dir: Optional[str], image_grids: Tuple[int, int]):
        """Generic (host-image) path with the reference's signature and return types (:94-155): one detector call per image through
        ``heuristic.inference_detector``; ``(confidence maps float64 [n, rows, cols], per image the list of names per cell)``.
        The detections -> cells step is this package's own statement of it: array arithmetic per image (a scatter-max for the
        confidences, a stable grouping for the names) instead of a Python loop per box -- the same values: the box centre is a
        float32 sum halved (exact), floor-divided by the cell size in float64, and ``score * weight`` is a float64 product, as both
        are under the reference's pinned numpy 1.26 (and in ``cell_reduce_kernel``)."""
        n_images = len(images)
        if n_images == 0:
            return np.array([]), []
        rows, cols = image_grids
        cell_h, cell_w = images[0].shape[0] / rows, images[0].shape[1] / cols
        conf_maps = np.zeros((n_images, rows * cols))
        name_maps = []
        names = [t[0] for t in self.heuristic.texts]
        weight = np.array([float(self.object2weight.get(nm, 0.5)) for nm in names], dtype=np.float64)
        for k, image in enumerate(images):
            dets = self.heuristic.inference_detector(images=[image], use_amp=False)
            xyxy = np.concatenate([np.asarray(d.xyxy, dtype=np.float32).reshape(-1, 4) for d in dets] or [np.zeros((0, 4), np.float32)])
            label = np.concatenate([np.asarray(d.class_id, dtype=np.int64).reshape(-1) for d in dets] or [np.zeros(0, np.int64)])
            score = np.concatenate([np.asarray(d.confidence, dtype=np.float32).reshape(-1) for d in dets] or [np.zeros(0, np.float32)])
            cx = ((xyxy[:, 0] + xyxy[:, 2]) / np.float32(2)).astype(np.float64)
            cy = ((xyxy[:, 1] + xyxy[:, 3]) / np.float32(2)).astype(np.float64)
            gx = np.minimum(np.floor_divide(cx, cell_w).astype(np.int64), cols - 1)
            gy = np.minimum(np.floor_divide(cy, cell_h).astype(np.int64), rows - 1)
            cell = gy * cols + gx
            np.maximum.at(conf_maps[k], cell, score.astype(np.float64) * weight[label])
            per_cell: List[List[str]] = [[] for _ in range(rows * cols)]
            order = np.argsort(cell, kind="stable")                    # names of a cell keep the detections' order
            for c, q in zip(cell[order].tolist(), label[order].tolist()):
                per_cell[c].append(names[q])
            name_maps.append(per_cell)
        return conf_maps.reshape(n_images, rows, cols), name_maps

    def score_image_grids(self, images, image_grids):
        return self.imageGridScoreFunction(images, self.output_dir, image_grids)

    def read_frame_batch(self, video_path, frame_indices):
        """(:157-169) native-resolution frames by RAW frame index (floats accepted, as pop_frames
        passes them); the store is indexed by logical second, raw index i = int(sec * raw_fps)."""
        last = self.store.num_seconds - 1
        secs = [min(last, max(0, int(round(float(i) / self.raw_fps * self.fps)))) for i in frame_indices]
        return frame_indices, self.store.host_frames(secs)

    def create_image_grid(self, frames: List[np.ndarray], rows: int, cols: int) -> np.ndarray:
        """(:171-188) host frames (equal-size HxWx3 uint8) -> each resized to 200x95 and tiled row-major into one
        [95 rows, 200 cols, 3] image.  ``search()`` itself never calls this (its grid is built straight from the
        resident frame store by ``_device_grid``); kept for callers of the reference method: the frames make a round
        trip through the device so the resize is the same kernel."""
        import torch
        if len(frames) != rows * cols:
            raise ValueError("Frame count does not match grid dimensions")      # :183-184
        if isinstance(frames, _ResizedFrames) and frames.searcher is self and (rows, cols) == tuple(self.image_grid_shape):
            return self._device_grid(frames.secs).cpu().numpy()                  # same bytes, no host round trip
        st = np.ascontiguousarray(np.stack([np.asarray(f, dtype=np.uint8) for f in frames]))
        if st.ndim != 4 or st.shape[3] != 3:
            raise ValueError("create_image_grid expects HxWx3 uint8 frames of one size")
        n, H, Wd, _ = st.shape
        dev = self.store.frames.device
        d = torch.from_numpy(st).to(dev)
        out = torch.empty((n, CELL_H, CELL_W, 3), dtype=torch.uint8, device=dev)
        idx = torch.arange(n, dtype=torch.int32, device=dev)
        _lib.check(self._state.lib.tstar_frames_resize(d.data_ptr(), n, H, Wd, idx.data_ptr(), n, CELL_W, CELL_H,
                                                       out.data_ptr(), 0, _lib.stream_ptr()), "tstar_frames_resize")
        grid = out.view(rows, cols, CELL_H, CELL_W, 3).permute(0, 2, 1, 3, 4).reshape(rows * CELL_H, cols * CELL_W, 3)
        return grid.cpu().numpy()

    def _device_grid(self, secs):
        import torch
        rows, cols = self.image_grid_shape
        if len(secs) != rows * cols:
            raise ValueError("Frame count does not match grid dimensions")      # :183-184
        N, H, Wd, _ = self.store.shape
        grid = torch.empty((rows * CELL_H, cols * CELL_W, 3), dtype=torch.uint8, device=self.store.frames.device)
        idx = self._d_idx(secs)
        _lib.check(self._state.lib.tstar_frames_to_grid(self.store.frames.data_ptr(), N, H, Wd, idx.data_ptr(), rows,
                                                        cols, grid.data_ptr(), int(self.store.fmt == "nv12"),
                                                        _lib.stream_ptr()), "tstar_frames_to_grid")
        return grid

    def _device_grid_into(self, d_idx, grid):
        """``_device_grid`` with the sampled seconds already on the device (int32 [rows * cols]) into a caller-provided uint8
        [rows * 95, cols * 200, 3] tensor (a slice of a group's stacked grid batch: no per-item index upload, no torch.stack copy)."""
        rows, cols = self.image_grid_shape
        if int(d_idx.numel()) != rows * cols:
            raise ValueError("Frame count does not match grid dimensions")      # :183-184
        N, H, Wd, _ = self.store.shape
        _lib.check(self._state.lib.tstar_frames_to_grid(self.store.frames.data_ptr(), N, H, Wd, d_idx.data_ptr(), rows, cols,
                                                        grid.data_ptr(), int(self.store.fmt == "nv12"), _lib.stream_ptr()),
                   "tstar_frames_to_grid")

    def _device_resized_into(self, d_idx, out):
        """``_device_resized`` with the seconds already on the device (int32 [n]) into a caller-provided uint8 [n, h, w, 3] tensor."""
        N, H, Wd, _ = self.store.shape
        n, oh, ow, _ = out.shape
        _lib.check(self._state.lib.tstar_frames_resize(self.store.frames.data_ptr(), N, H, Wd, d_idx.data_ptr(), int(n), int(ow), int(oh),
                                                       out.data_ptr(), int(self.store.fmt == "nv12"), _lib.stream_ptr()),
                   "tstar_frames_resize")

    def _device_resized(self, secs, out_w: int, out_h: int):
        import torch
        N, H, Wd, _ = self.store.shape
        out = torch.empty((len(secs), out_h, out_w, 3), dtype=torch.uint8, device=self.store.frames.device)
        idx = self._d_idx(secs)
        _lib.check(self._state.lib.tstar_frames_resize(self.store.frames.data_ptr(), N, H, Wd, idx.data_ptr(), len(secs),
                                                       out_w, out_h, out.data_ptr(), int(self.store.fmt == "nv12"),
                                                       _lib.stream_ptr()),
                   "tstar_frames_resize")
        return out

    def _device_verify_frames(self, secs):
        return self._device_resized(secs, VERIFY_W, VERIFY_H)

    # ---- distribution ----------------------------------------------------------------------------
    def store_score_distribution(self, _defer_lists: bool = False):
        """(:207-213) append copies of P / score_distribution / non_visiting_frames to the three histories (lists of lists, like the
        reference's ``.tolist()``).  ``_defer_lists`` (the fast search loop): the state is read now -- one device -> host copy -- but the
        three 3.6 k-element numpy -> list conversions (~0.25 ms together) wait for ``_finalize_history()``, which the loop calls once the
        next iteration's forward is queued: they used to sit between the FITPACK fit and the next draw, on the critical chain of the late
        iterations.  The histories are complete before anything outside the loop can look at them."""
        st = self._state.read_state()
        if _defer_lists:
            self.P_history.append(st[0])
            self.Score_history.append(st[1])
            self.non_visiting_history.append(st[2])
            self._history_pending = True
            return
        self.P_history.append(st[0].tolist())
        self.Score_history.append(st[1].tolist())
        self.non_visiting_history.append(st[2].tolist())

    def _finalize_history(self):
        if getattr(self, "_history_pending", False):
            self._history_pending = False
            for h in (self.P_history, self.Score_history, self.non_visiting_history):
                for k in range(len(h) - 1, -1, -1):
                    if isinstance(h[k], np.ndarray):
                        h[k] = h[k].tolist()
                    else:
                        break

    def _update_from_device(self, secs: List[int], d_conf, overlap=None):
        """update_frame_distribution (:276-321) on the device state; d_conf f64 [rows*cols] (cell i <-> sample i).

        ``overlap``: optional callable run right after the score write-back / window spread and
        BEFORE the host-side FITPACK fit -- the searcher uses it to enqueue the iteration's
        verification batch so the GPU works while the host fits the spline (verification does
        not read P, and its score overwrites are applied after the histories are stored, exactly
        in the reference's order)."""
        vx, vy = self._state.apply_grid(secs, d_conf)
        # FITPACK fit, evaluation, sigmoid and normalisation on the host with the reference's own calls (:262-274).  With a
        # verification batch to enqueue the fit starts FIRST, in a worker process (tstar_amd.spline_pool: the same scipy
        # call, bit-identical P): it is the longest step of an iteration of a search running alone (15-60 ms at 4x4), and
        # the 3-5 ms the host needs to enqueue the batch then run beside it instead of before it
        if overlap is not None:
            from . import spline_pool
            wait = spline_pool.distribution_async([(vx, vy)], [self.total_frame_num])
            try:
                ctx = overlap()
            except BaseException:
                wait()                      # the pool stays locked until its reply is collected
                raise
            P = wait()[0]
        else:
            ctx = None
            P = spline_distribution(vx, vy, self.total_frame_num)
        self._state.write(2, P)
        self.store_score_distribution()
        return ctx

    def update_top_25_with_window(self, frame_confidences, sampled_frame_indices, window_size: int = 5):
        """(:215-241) raise the scores around the sampled frames whose confidence is in the top quartile of
        ``frame_confidences``: in the given order and in place, ``score[f + o] = max(score[f + o], score[f] / (|o| + 1))``
        for ``|o| <= window_size`` -- on the device score array (tstar_searcher_window_spread)."""
        self._state.window_spread([int(i) for i in sampled_frame_indices], [float(c) for c in frame_confidences],
                                  window_size)

    def spline_keyframe_distribution(self, non_visiting_frames, score_distribution, video_length: int) -> np.ndarray:
        """(:243-274) sampling distribution from the visited frames of the GIVEN arrays (it does not touch the
        searcher's state; ``update_frame_distribution`` assigns the result to ``P``)."""
        non_visiting_frames = np.asarray(non_visiting_frames)
        visited = np.nonzero(non_visiting_frames == 0)[0]
        return spline_distribution(visited, np.asarray(score_distribution, dtype=np.float64)[visited], int(video_length))

    def update_frame_distribution(self, sampled_frame_indices, confidence_maps, detected_objects_maps):
        """(:276-321) public form over host arrays: cell (i // cols, i % cols) of ``confidence_maps[0]`` belongs to the
        i-th sampled second; write-back, top-quartile window spread, spline distribution, history -- all on the
        device state (``search()`` feeds the device confidences directly through ``_update_from_device``)."""
        import torch
        confidence_map = np.asarray(confidence_maps[0], dtype=np.float64)
        detected_objects_map = detected_objects_maps[0]
        _, grid_cols = self.image_grid_shape
        secs = [int(s) for s in sampled_frame_indices]
        frame_confidences = [confidence_map[i // grid_cols, i % grid_cols] for i in range(len(secs))]
        frame_detected_objects = [detected_objects_map[i] for i in range(len(secs))]
        d_conf = torch.tensor(frame_confidences, dtype=torch.float64, device=self.store.frames.device)
        self._update_from_device(secs, d_conf)
        return frame_confidences, frame_detected_objects

    # ---- sampling --------------------------------------------------------------------------------
    def sample_frames(self, num_samples: int):
        """(:324-363) returns ``(sampled seconds in draw order, their frames resized to 800x380)`` like the
        reference.  The frames are a lazy sequence (``_ResizedFrames``): ``search()`` builds its grid straight from
        the resident store and never materialises them; a caller that indexes or iterates them gets uint8
        [380,800,3] arrays produced by the device resize kernel with one device->host copy."""
        secs = self._sample_secs(num_samples)
        return secs, _ResizedFrames(self, secs, CELL_W * 4, CELL_H * 4)

    def _sample_secs(self, num_samples: int, _warned: Optional[list] = None) -> List[int]:
        """The draw of ``sample_frames``.  ``_warned`` (lockstep's speculative draw): a list that receives the fallback-branch
        flag INSTEAD of the warning being printed -- the warning is printed when, and only if, the draw is adopted."""
        pre = getattr(self, "_prefetched_secs", None)
        if pre is not None:                     # drawn ahead by lockstep._Group.speculate() from the same generator state
            self._prefetched_secs = None
            secs, post_state, warned = pre
            (self._rng if self._rng is not None else np.random).set_state(post_state)    # where that draw left the generator
            if warned:
                print(SAMPLER_WARNING)
            return secs
        if num_samples > self.total_frame_num:
            num_samples = self.total_frame_num
        if not self.Score_history:
            interval = self.total_frame_num // num_samples
            secs = np.arange(0, self.total_frame_num, interval)[:num_samples]
            if len(secs) < num_samples:
                secs = np.append(secs, self.total_frame_num - 1)
        else:
            if self._state.sampler_prep(num_samples, num_samples / self.total_frame_num):
                if _warned is not None:
                    _warned.append(True)
                else:
                    print(SAMPLER_WARNING)
            secs = self._choice(num_samples)
        return [int(s) for s in secs]

    def pop_frames(self, video_path, num_samples: int):
        """(:365-380) weighted random sample of K seconds from score_distribution, sorted.  Raises what
        ``np.random.choice`` raises in the reference when fewer than K scores are non-zero (short videos under
        ``search_budget=1.0``: every frame visited, most cells empty)."""
        nnz, total = self._state.pop_prep()
        secs = self._choice(num_samples, nnz, total)
        secs.sort()
        time_stamps = [sec / self.fps for sec in secs]
        frames = self.store.host_frames(secs)
        return frames, time_stamps

    # ---- search ----------------------------------------------------------------------------------
    def _verify_launch(self, secs: List[int], names_per_frame: List[List[str]]):
        """Enqueue the speculative verification batch of the iteration (device work only)."""
        cands = [i for i, names in enumerate(names_per_frame) if any(t in names for t in self.remaining_targets)]
        if not cands:
            return None
        vframes = self._device_verify_frames([secs[i] for i in cands])
        res = self.heuristic.score_batch(vframes, 1, 1)
        self.device_images_scored += len(cands)
        return cands, vframes, res

    def _verify_finish(self, ctx, secs: List[int], names_per_frame: List[List[str]]):
        """verify_and_remove_target for every sampled frame of the iteration (:481-486, :382-420),
        replayed sequentially on the results of the batched launch."""
        if ctx is None:
            return
        cands, vframes, res = ctx
        vconf = res.cell_conf[:, 0].cpu().numpy()
        vmask = res.cell_mask[:, 0].cpu().numpy().astype(np.uint32)
        self._verify_replay(cands, vconf, vmask, secs, names_per_frame, vframes, res, 0)

    def _verify_replay(self, cands, vconf, vmask, secs, names_per_frame, vframes=None, res=None, res_offset=0):
        """Sequential ``remaining_targets`` logic over already-scored candidate frames: candidate j of this
        searcher is row ``j`` of vconf/vmask (and image ``res_offset + j`` of ``res`` / ``vframes``)."""
        slot = {i: j for j, i in enumerate(cands)}
        upd_s, upd_v = [], []
        hist = None            # (annotated frames, detections) of this searcher's candidates, fetched on first use
        for i, (sec, names) in enumerate(zip(secs, names_per_frame)):
            for target in list(self.remaining_targets):
                if target in names:
                    j = slot[i]
                    single_conf = vconf[j]
                    single_names = self._names_from_mask(int(vmask[j]))
                    upd_s.append(sec)
                    upd_v.append(single_conf)
                    self.frames_scored += 1
                    self.detector_calls += 1
                    if self.keep_visual_history and res is not None:
                        if hist is None:       # boxes painted on the device, one copy for all candidates
                            hist = self.heuristic.annotated_batch(vframes[res_offset:res_offset + len(cands)], res,
                                                                  res_offset, len(cands))
                        frame, det = hist[0][j], hist[1][j]
                        self.image_grid_iters.append([frame])          # the annotated array, aliased like the
                        self.detect_annotot_iters.append([frame])      # reference's in-place annotation (B.14)
                        self.detect_bbox_iters.append([det])
                    if target in single_names and single_conf > self.confidence_threshold:
                        self.remaining_targets.remove(target)
                        print(f"Found target '{target}' in frame {int(sec * self.raw_fps / self.fps)}, score {single_conf:.2f}")
                        break
        if upd_s:
            self._state.set_scores(upd_s, upd_v)

    def verify_and_remove_target(self, frame_sec: int, detected_objects: List[str], confidence_threshold: float) -> bool:
        """(:382-420) one frame: if a remaining target is among ``detected_objects``, re-score the frame alone at
        600x285, overwrite its score, and drop the target when it is confirmed above the threshold.  ``search()``
        batches these calls per iteration (``_verify_launch`` / ``_verify_replay``) with identical results."""
        for target in list(self.remaining_targets):
            if target in detected_objects:
                d_frame = self._device_verify_frames([int(frame_sec)])
                if self._fast:
                    res = self.heuristic.score_batch(d_frame, 1, 1)
                    single_conf = float(res.cell_conf[0, 0].item())
                    single_names = self._names_from_mask(int(res.cell_mask[0, 0].item()) & 0xFFFFFFFF)
                    det = self.heuristic._detections_from(res, 0) if self.keep_visual_history else None
                else:
                    conf_map, det_map = self.score_image_grids([d_frame[0].cpu().numpy()], (1, 1))
                    single_conf, single_names = conf_map[0, 0, 0], det_map[0][0]
                    det = None
                self._state.set_scores([int(frame_sec)], [single_conf])
                self.frames_scored += 1
                self.detector_calls += 1
                self.device_images_scored += 1
                if self.keep_visual_history:
                    frame = d_frame[0].cpu().numpy()
                    dets = [det] if det is not None else self.heuristic.detections_inbatch
                    self.image_grid_iters.append([frame])
                    self.detect_annotot_iters.append(self.heuristic.bbox_visualization([frame], dets))
                    self.detect_bbox_iters.append(dets)
                if target in single_names and single_conf > confidence_threshold:
                    self.remaining_targets.remove(target)
                    print(f"Found target '{target}' in frame {int(frame_sec * self.raw_fps / self.fps)}, score {single_conf:.2f}")
                    return True
        return False

    def _verify_generic(self, secs, names_per_frame):
        for sec, names in zip(secs, names_per_frame):
            for target in list(self.remaining_targets):
                if target in names:
                    frame = self._device_verify_frames([sec])[0].cpu().numpy()
                    conf_map, det_map = self.score_image_grids([frame], (1, 1))
                    single_conf = conf_map[0, 0, 0]
                    self._state.set_scores([sec], [single_conf])
                    self.frames_scored += 1
                    self.detector_calls += 1
                    self.device_images_scored += 1
                    if self.keep_visual_history:
                        self.image_grid_iters.append([frame])
                        self.detect_annotot_iters.append(self.heuristic.bbox_visualization(
                            images=[frame], detections_inbatch=self.heuristic.detections_inbatch))
                        self.detect_bbox_iters.append(self.heuristic.detections_inbatch)
                    if target in det_map[0][0] and single_conf > self.confidence_threshold:
                        self.remaining_targets.remove(target)
                        print(f"Found target '{target}' in frame {int(sec * self.raw_fps / self.fps)}, score {single_conf:.2f}")
                        break

    def search(self):
        """(:444-491) returns (keyframes uint8 [K,H,W,3], time_stamps list[float]).

        On the fast path (a tstar_amd detector interface) the loop below is executed by ``lockstep.search_solo``: the same
        statements in the same order, with the searcher-state kernels on a side stream and the next iteration's grid forward
        queued speculatively behind each verification batch (identical results: tests/test_gpu_searcher.py).
        ``TSTAR_SOLO_SEQUENTIAL=1`` runs the plain loop (same-session A/Bs)."""
        import torch
        if self._fast and os.environ.get("TSTAR_SOLO_SEQUENTIAL") is None:
            from .lockstep import search_solo
            return search_solo(self)
        while self.remaining_targets and self.search_budget > 0:
            rows, cols = self.image_grid_shape
            n = rows * cols
            secs, _lazy_frames = self.sample_frames(n)       # the reference's call (wrappers / subclasses see it)
            secs = [int(s_) for s_ in secs]
            self.search_budget -= n
            grid = self._device_grid(secs)
            if self._fast:
                res = self.heuristic.score_batch(grid.unsqueeze(0), rows, cols)
                self.device_images_scored += 1
                d_conf = res.cell_conf[0]
                masks = res.cell_mask[0].cpu().numpy().astype(np.uint32)
                names_per_frame = [self._names_from_mask(int(m)) for m in masks[:len(secs)]]
                if self.keep_visual_history:
                    imgs, dets = self.heuristic.annotated_batch(grid.unsqueeze(0), res, 0, 1)
                    self.heuristic.detections_inbatch = dets
                    self.image_grid_iters.append([imgs[0]])
                    self.detect_annotot_iters.append([imgs[0]])
                    self.detect_bbox_iters.append(dets)
            else:
                g = grid.cpu().numpy()
                conf_maps, name_maps = self.score_image_grids([g], self.image_grid_shape)
                self.device_images_scored += 1
                d_conf = torch.from_numpy(np.ascontiguousarray(conf_maps[0].reshape(-1))).to(grid.device)
                names_per_frame = [name_maps[0][i] for i in range(len(secs))]
                if self.keep_visual_history:
                    self.image_grid_iters.append([g])
                    self.detect_annotot_iters.append(self.heuristic.bbox_visualization(
                        images=[g], detections_inbatch=self.heuristic.detections_inbatch))
                    self.detect_bbox_iters.append(self.heuristic.detections_inbatch)
            self.frames_scored += n
            self.detector_calls += 1
            if self._fast:
                ctx = self._update_from_device(secs, d_conf, lambda: self._verify_launch(secs, names_per_frame))
                self._verify_finish(ctx, secs, names_per_frame)
            else:
                self._update_from_device(secs, d_conf)
                self._verify_generic(secs, names_per_frame)
            self.iterations += 1
        frames, time_stamps = self.pop_frames(video_path=self.video_path, num_samples=self.search_nframes)
        self.last_time_stamps = list(time_stamps)
        return frames, time_stamps

    search_with_visualization = search

    def plot_score_distribution(self, save_path: Optional[str] = None):
        """(:423-441)"""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        score = self.score_distribution
        time_axis = np.linspace(0, self.duration, len(score))
        plt.figure(figsize=(12, 6))
        plt.plot(time_axis, score, label="Score Distribution")
        plt.xlabel("Time (seconds)")
        plt.ylabel("Score")
        plt.title("Score Distribution Over Time")
        plt.grid(True)
        plt.legend()
        if save_path:
            plt.savefig(save_path, format="png", dpi=300)
            print(f"Plot saved to {save_path}")
        plt.close()
