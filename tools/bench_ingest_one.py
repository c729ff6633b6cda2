"""One ingest launch shape, repeated (for counter passes): python tools/bench_ingest_one.py grid|resize [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
N, H, W = 1200, 360, 640
frames = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")    # one kernel: counter passes replay every launch
what = sys.argv[1] if len(sys.argv) > 1 else "grid"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if what == "grid":
    g = 16
    idx = (torch.randperm(N, device="cuda")[:g * g]).to(torch.int32)
    grid = torch.empty((95 * g, 200 * g, 3), dtype=torch.uint8, device="cuda")
    f = lambda: _lib.check(lib.tstar_frames_to_grid(frames.data_ptr(), N, H, W, idx.data_ptr(), g, g, grid.data_ptr(), 0, s))
    px = g * g * 200 * 95
else:
    n = 180
    idx = (torch.randperm(N, device="cuda")[:n]).to(torch.int32)
    out = torch.empty((n, 285, 600, 3), dtype=torch.uint8, device="cuda")
    f = lambda: _lib.check(lib.tstar_frames_resize(frames.data_ptr(), N, H, W, idx.data_ptr(), n, 600, 285, out.data_ptr(), 0, s))
    px = n * 600 * 285
for _ in range(reps):
    f()
torch.cuda.synchronize()
print(f"{what}: {reps} launches, {px} output pixels per launch")
