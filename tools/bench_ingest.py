"""Ingest kernels against the HBM roofline ("report GB/s for ingest"): frame gather + the two bilinear
resizes + grid tiling (tstar_frames_to_grid) and the verification-size resize (tstar_frames_resize), from an RGB and
from an NV12 frame store resident in HBM.

Algorithmic bytes per launch = the source frames read once (3 H W per RGB frame, 1.5 H W per NV12 frame) + the output
written once; no intermediate (800x380, 200x95) ever exists in memory.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib
from tstar_amd.video import synthetic_video, synthetic_video_nv12

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
N, H, W = 1200, 360, 640
HBM_PEAK = 8000.0  # GB/s


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for fmt, store in (("rgb", synthetic_video(N, H, W, seed=0)), ("nv12", synthetic_video_nv12(N, H, W, seed=0))):
    src_frame = 3 * H * W if fmt == "rgb" else H * W * 3 // 2
    nv = int(fmt == "nv12")
    for g in (4, 16):
        n = g * g
        idx = (torch.randperm(N, device="cuda")[:n]).to(torch.int32)
        grid = torch.empty((95 * g, 200 * g, 3), dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: _lib.check(lib.tstar_frames_to_grid(store.frames.data_ptr(), N, H, W, idx.data_ptr(), g, g,
                                                                grid.data_ptr(), nv, s)))
        byts = n * src_frame + grid.numel()
        print(f"frames_to_grid  {fmt:4s} g={g:2d} ({n:3d} frames): {ms * 1e3:7.1f} us  {byts / 1e6:7.1f} MB  "
              f"{byts / ms / 1e6:7.1f} GB/s = {byts / ms / 1e6 / HBM_PEAK:.3f} of HBM peak")
    for n in (16, 64, 180):
        idx = (torch.randperm(N, device="cuda")[:n]).to(torch.int32)
        out = torch.empty((n, 285, 600, 3), dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: _lib.check(lib.tstar_frames_resize(store.frames.data_ptr(), N, H, W, idx.data_ptr(), n, 600, 285,
                                                               out.data_ptr(), nv, s)))
        byts = n * src_frame + out.numel()
        print(f"frames_resize   {fmt:4s} n={n:3d}            : {ms * 1e3:7.1f} us  {byts / 1e6:7.1f} MB  "
              f"{byts / ms / 1e6:7.1f} GB/s = {byts / ms / 1e6 / HBM_PEAK:.3f} of HBM peak")
