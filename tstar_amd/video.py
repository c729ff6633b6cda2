"""Decoded-frame store resident in HBM.

The reference re-opens the file with decord on every read
(/root/reference/TStar/interface_searcher.py:157-169) and asks OpenCV for fps /
frame count (:60-65).  Here a video is decoded ONCE into a uint8 tensor
[N,H,W,3] (RGB) on the GPU and every later gather/resize is a HIP kernel
(tstar_frames_to_grid / tstar_frames_resize).  Decode itself (FFmpeg/VCN) is
outside the hot path (SURVEY.md 8f "next" row 3): raw 4:2:0 files (YUV4MPEG2) are
read and repacked on the device here (``load_y4m``); the compressed sequences Pillow
decodes in this image (animated GIF / WebP, AVIF sequences = AV1) go through
``load_pillow_sequence``; other compressed files need decord or cv2 on the host
(rocDecode / FFmpeg are not in this build), synthetic videos need nothing.
"""
from __future__ import annotations

import re
from typing import Optional

import numpy as np


def synthetic_lowres(n_frames: int, seed: int = 0) -> np.ndarray:
    """uint8 [N,10,17,3] noise lattice, one frozen RandomState stream."""
    return np.random.RandomState(seed).randint(0, 256, size=(n_frames, 10, 17, 3)).astype(np.uint8)


def _planted(n_frames: int, seed: int):
    """Seeded 5 % of frames get 1-2 solid rectangles: (frame, y0, y1, x0, x1, rgb) in 9x16 lattice units."""
    rs = np.random.RandomState(seed + 1)
    frames = np.nonzero(rs.random_sample(n_frames) < 0.05)[0]
    out = []
    for f in frames:
        for _ in range(1 + int(rs.randint(0, 2))):
            y0, x0 = int(rs.randint(0, 7)), int(rs.randint(0, 13))
            h, w = int(rs.randint(2, 4)), int(rs.randint(2, 5))
            rgb = tuple(int(v) for v in rs.randint(0, 256, 3))
            out.append((int(f), y0, min(y0 + h, 9), x0, min(x0 + w, 16), rgb))
    return out


def synthetic_frames_numpy(idx, n_frames: int, H: int = 360, W: int = 640, seed: int = 0) -> np.ndarray:
    """CPU statement of the synthetic video (integer arithmetic only): frames ``idx`` as uint8 [n,H,W,3].

    Frame = integer bilinear upsampling of the 10x17 lattice (cell size H/9 x W/16)
    + planted rectangles.  ``synthetic_video`` computes the same bytes on the GPU.
    """
    low = synthetic_lowres(n_frames, seed).astype(np.int32)
    cy, cx = H // 9, W // 16
    ys, xs = np.arange(H), np.arange(W)
    y0, fy = ys // cy, (ys % cy).astype(np.int32)
    x0, fx = xs // cx, (xs % cx).astype(np.int32)
    out = np.empty((len(idx), H, W, 3), dtype=np.uint8)
    rects = _planted(n_frames, seed)
    for k, i in enumerate(idx):
        L = low[int(i)]
        # separable form of the 4-term integer bilinear sum (identical integers, half the gathers)
        rows = L[:, x0] * (cx - fx)[None, :, None] + L[:, x0 + 1] * fx[None, :, None]          # [10, W, 3]
        fr = ((rows[y0] * (cy - fy)[:, None, None] + rows[y0 + 1] * fy[:, None, None]) // (cy * cx)).astype(np.uint8)
        for (f, ry0, ry1, rx0, rx1, rgb) in rects:
            if f == int(i):
                fr[ry0 * cy:ry1 * cy, rx0 * cx:rx1 * cx] = rgb
        out[k] = fr
    return out


class FrameStore:
    """A decoded video in HBM at the searcher's logical rate (1 frame per second):
    ``frames`` uint8 cuda tensor [N,H,W,3] with frames[s] = raw frame int(s * raw_fps)
    (the index map of interface_searcher.py:360), plus the raw stream's fps / frame count."""

    def __init__(self, frames, raw_fps: float, raw_total_frames: Optional[int] = None, name: str = "<frames>",
                 fmt: str = "rgb"):
        """``fmt``: "rgb" -> frames u8 [N,H,W,3]; "nv12" -> frames u8 [N, H*3/2, W] (luma plane + interleaved
        half-resolution UV plane), converted to RGB inside the ingest kernels (half the bytes per frame)."""
        if fmt not in ("rgb", "nv12"):
            raise ValueError("FrameStore fmt must be 'rgb' or 'nv12'")
        self.fmt = fmt
        self.frames = frames
        self.raw_fps = float(raw_fps)
        self.raw_total_frames = int(raw_total_frames if raw_total_frames is not None
                                    else round(frames.shape[0] * self.raw_fps))
        self.name = name

    @property
    def num_seconds(self) -> int:
        return int(self.frames.shape[0])

    @property
    def shape(self):
        """(N, H, W, 3) of the RGB view, whatever the storage format."""
        if self.fmt == "nv12":
            n, h32, w = self.frames.shape
            return (int(n), int(h32) * 2 // 3, int(w), 3)
        return tuple(self.frames.shape)

    def host_frames(self, secs) -> np.ndarray:
        """Native-resolution RGB frames of the given logical seconds -> uint8 numpy [n,H,W,3]."""
        import torch
        if self.fmt == "nv12":
            from . import _lib
            N, H, W, _ = self.shape
            idx = torch.as_tensor([int(i) for i in secs], dtype=torch.int32, device=self.frames.device)
            out = torch.empty((len(idx), H, W, 3), dtype=torch.uint8, device=self.frames.device)
            _lib.check(_lib.load().tstar_nv12_to_rgb(self.frames.data_ptr(), N, H, W, idx.data_ptr(), len(idx),
                                                      out.data_ptr(), _lib.stream_ptr()), "tstar_nv12_to_rgb")
            return out.cpu().numpy()
        ii = torch.as_tensor([int(i) for i in secs], dtype=torch.long, device=self.frames.device)
        return self.frames.index_select(0, ii).cpu().numpy()


def synthetic_video(n_frames: int = 3600, H: int = 360, W: int = 640, seed: int = 0, raw_fps: float = 1.0,
                    device: str = "cuda", chunk: int = 256) -> FrameStore:
    """Build the synthetic video directly in HBM (torch integer ops = tensor plumbing; the
    bytes equal ``synthetic_frames_numpy``)."""
    import torch
    if H % 9 or W % 16:
        raise ValueError("synthetic video needs H % 9 == 0 and W % 16 == 0")
    low = torch.from_numpy(synthetic_lowres(n_frames, seed)).to(device).to(torch.int32)
    cy, cx = H // 9, W // 16
    ys = torch.arange(H, device=device)
    xs = torch.arange(W, device=device)
    y0, fy = ys // cy, (ys % cy).to(torch.int32)
    x0, fx = xs // cx, (xs % cx).to(torch.int32)
    wy0, wy1 = (cy - fy)[None, :, None, None], fy[None, :, None, None]
    wx0, wx1 = (cx - fx)[None, None, :, None], fx[None, None, :, None]
    frames = torch.empty((n_frames, H, W, 3), dtype=torch.uint8, device=device)
    for s in range(0, n_frames, chunk):
        L = low[s:s + chunk]
        top = L[:, y0]
        bot = L[:, y0 + 1]
        v = top[:, :, x0] * (wy0 * wx0) + top[:, :, x0 + 1] * (wy0 * wx1) \
            + bot[:, :, x0] * (wy1 * wx0) + bot[:, :, x0 + 1] * (wy1 * wx1)
        frames[s:s + chunk] = torch.div(v, cy * cx, rounding_mode="floor").to(torch.uint8)
    for (f, ry0, ry1, rx0, rx1, rgb) in _planted(n_frames, seed):
        frames[f, ry0 * cy:ry1 * cy, rx0 * cx:rx1 * cx] = torch.tensor(rgb, dtype=torch.uint8, device=device)
    return FrameStore(frames, raw_fps, None, name=f"synthetic://n={n_frames},h={H},w={W},seed={seed}")


def synthetic_nv12_numpy(idx, n_frames: int, H: int = 360, W: int = 640, seed: int = 0) -> np.ndarray:
    """NV12 variant of the synthetic video (same lattice; channel 0 -> luma, channels 1,2 -> 4:2:0 chroma):
    uint8 [n, H*3/2, W].  Integer arithmetic only."""
    rgbish = synthetic_frames_numpy(idx, n_frames, H, W, seed)
    out = np.empty((len(idx), H * 3 // 2, W), dtype=np.uint8)
    out[:, :H, :] = 16 + (rgbish[..., 0].astype(np.int32) * 219 // 255).astype(np.uint8)         # limited-range luma
    uv = rgbish[:, ::2, ::2, 1:3].astype(np.int32)
    out[:, H:, :] = (16 + uv * 224 // 255).astype(np.uint8).reshape(len(idx), H // 2, W)          # interleaved U,V
    return out


def synthetic_video_nv12(n_frames: int = 3600, H: int = 360, W: int = 640, seed: int = 0, raw_fps: float = 1.0,
                         device: str = "cuda") -> FrameStore:
    """NV12 synthetic video in HBM (same bytes as ``synthetic_nv12_numpy``)."""
    import torch
    rgb = synthetic_video(n_frames, H, W, seed, raw_fps, device).frames
    out = torch.empty((n_frames, H * 3 // 2, W), dtype=torch.uint8, device=device)
    out[:, :H, :] = (16 + torch.div(rgb[..., 0].to(torch.int32) * 219, 255, rounding_mode="floor")).to(torch.uint8)
    uv = rgb[:, ::2, ::2, 1:3].to(torch.int32)
    out[:, H:, :] = (16 + torch.div(uv * 224, 255, rounding_mode="floor")).to(torch.uint8).reshape(n_frames, H // 2, W)
    return FrameStore(out, raw_fps, None, name=f"synthetic://n={n_frames},h={H},w={W},seed={seed},fmt=nv12", fmt="nv12")


def load_video_frames(video, num_frames: int = 8) -> np.ndarray:
    """The grounder's uniform frame loader (/root/reference/TStar/utilites.py:40-81): frames at raw
    indices floor(i * total / num_frames), RGB uint8 [n,H,W,3] -- served from the resident store (the
    stored frame nearest in time to each raw index when raw_fps != 1)."""
    import math
    st = open_video(video)
    total = st.raw_total_frames
    if total == 0:
        raise ValueError("Video has zero frames or could not retrieve frame count.")
    n = min(num_frames, total)
    step = total / n
    raw = [int(math.floor(i * step)) for i in range(n)]
    last = st.num_seconds - 1
    return st.host_frames([min(last, max(0, int(round(r / st.raw_fps)))) for r in raw])


def parse_y4m_header(path: str):
    """YUV4MPEG2 stream header -> dict(w, h, fps (float), fps_frac (num, den), chroma, data_offset, frame_bytes,
    frame_header_bytes, n_frames).  8-bit 4:2:0 only (C420, C420jpeg, C420mpeg2, C420paldv or no C tag)."""
    import os
    with open(path, "rb") as f:
        head = f.readline(4096)
        if not head.startswith(b"YUV4MPEG2 ") or not head.endswith(b"\n"):
            raise ValueError(f"Cannot open video file: {path} (not a YUV4MPEG2 stream)")
        tags = head[len(b"YUV4MPEG2 "):-1].split(b" ")
        kv = {t[:1].decode(): t[1:].decode() for t in tags if t}
        try:
            w, h = int(kv["W"]), int(kv["H"])
            num, den = (int(v) for v in kv.get("F", "25:1").split(":"))
        except (KeyError, ValueError):
            raise ValueError(f"Cannot open video file: {path} (malformed YUV4MPEG2 header)")
        chroma = kv.get("C", "420jpeg")
        if chroma not in ("420", "420jpeg", "420mpeg2", "420paldv"):
            raise ValueError(f"Cannot open video file: {path} (only 8-bit 4:2:0 YUV4MPEG2 is supported, got C{chroma})")
        if w % 2 or h % 2 or den <= 0 or num <= 0:
            raise ValueError(f"Cannot open video file: {path} (odd dimensions or bad frame rate)")
        off = f.tell()
        fh = f.readline(256)
        if not fh.startswith(b"FRAME"):
            raise ValueError(f"Cannot open video file: {path} (no FRAME marker)")
    fb = w * h * 3 // 2
    size = os.path.getsize(path)
    # every frame header of the streams we accept is the plain marker read above (parameters would change its length)
    n = (size - off) // (len(fh) + fb)
    return dict(w=w, h=h, fps=num / den, fps_frac=(num, den), chroma=chroma, data_offset=off, frame_bytes=fb,
                frame_header_bytes=len(fh), n_frames=int(n))


def load_y4m(path: str, device: str = "cuda", chunk: int = 64) -> FrameStore:
    """Decode front end for raw 4:2:0 video: a YUV4MPEG2 file is read ONCE, the frames the searcher can ever
    ask for (raw frame int(sec * fps) for every logical second, interface_searcher.py:360) are staged through two pinned
    host buffers, copied to HBM on a side stream and repacked I420 -> NV12 by a HIP kernel (tstar_i420_to_nv12) straight into
    the resident store, the host read of the next chunk overlapping the copy of the previous one.  Replaces the
    reopen-and-seek-per-call decord reader (:157-169) for such files; compressed streams would need rocDecode / FFmpeg,
    which this build does not have."""
    import torch
    from . import _lib
    hd = parse_y4m_header(path)
    w, h, fb = hd["w"], hd["h"], hd["frame_bytes"]
    if (w * h) % 64:          # before any store / pinned buffer is allocated (tstar_i420_to_nv12 moves 64-byte units)
        raise ValueError(f"Cannot open video file: {path} ({w}x{h}: the device repack needs W*H to be a multiple of 64)")
    n_sec = int(hd["n_frames"] / hd["fps"])
    if n_sec < 1:
        raise ValueError(f"Cannot open video file: {path} (shorter than one second)")
    want = [int(sec * hd["fps"]) for sec in range(n_sec)]
    lib = _lib.load()
    store = torch.empty((n_sec, h * 3 // 2, w), dtype=torch.uint8, device=device)
    pinned = [torch.empty((chunk, fb), dtype=torch.uint8).pin_memory() for _ in range(2)]
    staged = [torch.empty((chunk, fb), dtype=torch.uint8, device=device) for _ in range(2)]
    done = [None, None]
    side = torch.cuda.Stream()
    stride = hd["frame_header_bytes"] + fb
    with open(path, "rb") as f:
        for ci, s0 in enumerate(range(0, n_sec, chunk)):
            b = ci & 1
            if done[b] is not None:
                done[b].synchronize()                     # the pinned buffer's previous copy has left
            idx = want[s0:s0 + chunk]
            host = pinned[b].numpy()
            for j, fi in enumerate(idx):
                f.seek(hd["data_offset"] + fi * stride)
                if f.read(hd["frame_header_bytes"])[:5] != b"FRAME":       # per-frame parameters would shift every later frame
                    raise ValueError(f"Cannot open video file: {path} (no FRAME marker where frame {fi} should start: "
                                     f"frame headers of varying length are not supported)")
                got = f.readinto(memoryview(host[j]))
                if got != fb:
                    raise ValueError(f"Cannot open video file: {path} (truncated at frame {fi})")
            with torch.cuda.stream(side):
                staged[b][:len(idx)].copy_(pinned[b][:len(idx)], non_blocking=True)
                _lib.check(lib.tstar_i420_to_nv12(staged[b].data_ptr(), len(idx), h, w, store[s0:s0 + len(idx)].data_ptr(),
                                                  side.cuda_stream), "tstar_i420_to_nv12")
                done[b] = torch.cuda.Event()
                done[b].record(side)
    side.synchronize()
    return FrameStore(store, hd["fps"], hd["n_frames"], name=path, fmt="nv12")


def write_y4m(path: str, nv12_frames: np.ndarray, fps=(1, 1)) -> None:
    """uint8 [n, H*3/2, W] NV12 frames -> a YUV4MPEG2 file (planar I420 payload).  Test / example helper."""
    n, h32, w = nv12_frames.shape
    h = h32 * 2 // 3
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F{fps[0]}:{fps[1]} Ip A1:1 C420mpeg2\n".encode())
        for fr in nv12_frames:
            uv = fr[h:].reshape(h // 2, w // 2, 2)
            f.write(b"FRAME\n")
            f.write(fr[:h].tobytes())
            f.write(np.ascontiguousarray(uv[..., 0]).tobytes())
            f.write(np.ascontiguousarray(uv[..., 1]).tobytes())


_PIL_SEQUENCE_EXT = (".gif", ".webp", ".avif", ".avifs", ".apng", ".png")


def load_pillow_sequence(path: str, device: str = "cuda", chunk: int = 64) -> FrameStore:
    """Decode front end for the compressed multi-frame containers Pillow reads in this image -- animated GIF (LZW), WebP
    (VP8 / VP8L) and AVIF image sequences (AV1 through libavif) -- the only compressed "video" decoders available here
    (no FFmpeg, rocDecode, decord or cv2).  The stream's rate comes from the frame durations (frames / total duration);
    the frames the searcher can ever ask for (raw frame int(sec * fps) for every logical second,
    interface_searcher.py:360) are decoded once on the host, converted to RGB and copied to HBM in chunks -- the role
    decord's CPU reader plays in the reference (:157-169), without the reopen per call."""
    import torch
    from PIL import Image
    try:
        im = Image.open(path)
    except Exception as e:
        raise ValueError(f"Cannot open video file: {path} ({e})")
    with im:
        n = int(getattr(im, "n_frames", 1))
        w, h = im.size

        def frame(i):
            """RGB frame i and its duration in ms (WebP / AVIF report the duration only once the frame is decoded)."""
            im.seek(i)
            im.load()
            return np.asarray(im.convert("RGB"), dtype=np.uint8), float(im.info.get("duration", 0) or 0)

        def wanted(fps):
            """Raw frame of every logical second (non-decreasing; a stream slower than 1 fps names a frame repeatedly)."""
            n_sec = int(n / fps)
            return n_sec, [int(sec * fps) for sec in range(n_sec)]

        # One decode pass under the hypothesis that every frame lasts as long as the first (the usual case); the pass also
        # collects the real durations, and only a stream with varying durations is decoded a second time at its true rate.
        f0, d0 = frame(0)
        if n < 1 or d0 <= 0:
            raise ValueError(f"Cannot open video file: {path} (no frame durations: not an animated sequence)")
        fps = 1000.0 / d0
        for attempt in range(2):
            n_sec, want = wanted(fps)
            if n_sec < 1:
                raise ValueError(f"Cannot open video file: {path} (shorter than one second)")
            store = torch.empty((n_sec, h, w, 3), dtype=torch.uint8, device=device)
            host = np.empty((min(chunk, n_sec), h, w, 3), dtype=np.uint8)
            total_ms, filled, base, sec = 0.0, 0, 0, 0
            for i in range(n):
                fr, d = (f0, d0) if i == 0 else frame(i)
                total_ms += d
                while sec < n_sec and want[sec] == i:       # every second that maps to raw frame i, in order
                    host[filled] = fr
                    filled += 1
                    sec += 1
                    if filled == len(host) or sec == n_sec:
                        store[base:base + filled].copy_(torch.from_numpy(host[:filled]))
                        base, filled = base + filled, 0
            if base != n_sec:
                raise ValueError(f"Cannot open video file: {path} (decoded {base} of {n_sec} seconds)")
            true_fps = n * 1000.0 / total_ms
            if abs(true_fps - fps) <= 1e-9 * fps:
                break
            fps = true_fps                                  # varying durations: decode again at the stream's real rate
    return FrameStore(store, fps, n, name=path)


_SYN = re.compile(r"^synthetic://")


def open_video(video, device: str = "cuda") -> FrameStore:
    """``video``: a FrameStore, a ``synthetic://n=..,h=..,w=..,seed=..,fps=..`` URL, or a file path
    (decoded at native rate through decord, else cv2; both absent -> ValueError like the
    reference's ``Cannot open video file`` at interface_searcher.py:61-62)."""
    if isinstance(video, FrameStore):
        return video
    if isinstance(video, str) and _SYN.match(video):
        kv = dict(p.split("=") for p in video[len("synthetic://"):].split(",") if p)
        if kv.get("fmt", "rgb") == "nv12":
            return synthetic_video_nv12(int(kv.get("n", 3600)), int(kv.get("h", 360)), int(kv.get("w", 640)),
                                        int(kv.get("seed", 0)), float(kv.get("fps", 1.0)), device)
        return synthetic_video(int(kv.get("n", 3600)), int(kv.get("h", 360)), int(kv.get("w", 640)),
                               int(kv.get("seed", 0)), float(kv.get("fps", 1.0)), device)
    import torch
    if isinstance(video, str) and video.lower().endswith(".y4m"):
        import os
        if not os.path.isfile(video):
            raise ValueError(f"Cannot open video file: {video}")
        return load_y4m(video, device)
    if isinstance(video, str) and video.lower().endswith(_PIL_SEQUENCE_EXT):
        import os
        if not os.path.isfile(video):
            raise ValueError(f"Cannot open video file: {video}")
        return load_pillow_sequence(video, device)
    try:
        from decord import VideoReader, cpu  # type: ignore
        vr = VideoReader(video, ctx=cpu(0))
        fps = float(vr.get_avg_fps())
        total = len(vr)
        want = [int(sec * fps) for sec in range(int(total / fps))]
        parts = [torch.from_numpy(vr.get_batch(want[s:s + 256]).asnumpy()).to(device)
                 for s in range(0, len(want), 256)]
        return FrameStore(torch.cat(parts), fps, total, name=video)
    except ImportError:
        pass
    try:
        import cv2  # type: ignore
    except ImportError:
        raise ValueError(f"Cannot open video file: {video} (neither decord nor cv2 is importable; "
                         "pass a tstar_amd.video.FrameStore or a synthetic:// URL)")
    cap = cv2.VideoCapture(video)
    if not cap.isOpened():
        raise ValueError(f"Cannot open video file: {video}")
    fps = cap.get(cv2.CAP_PROP_FPS)
    total = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
    want = [int(sec * fps) for sec in range(int(total / fps))]
    out, i, sec = [], 0, 0
    while sec < len(want):
        ok, fr = cap.read()
        if not ok:
            break
        if want[sec] == i:
            dev = torch.from_numpy(fr[:, :, ::-1].copy()).to(device)
            while sec < len(want) and want[sec] == i:       # a stream slower than 1 fps names a frame repeatedly
                out.append(dev)
                sec += 1
        i += 1
    cap.release()
    return FrameStore(torch.stack(out), fps, total, name=video)
