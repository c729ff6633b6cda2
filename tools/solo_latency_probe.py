"""Where the time of ONE search running alone goes (configs[1]: 3600 frames, 16x16 grid, K = 8): wall time of the
statement groups of `bench.py`'s single-search latency figure, and -- under `rocprofv3 --kernel-trace` -- marker kernels
around the search so that tools/rocpd_gaps.py --timed-region can report the GPU idle time inside it.

    python tools/solo_latency_probe.py [grid] [lockstep-of-one: 0 | 1] [weights mode: f32x3 (default, the bench's) | f32 | ...]
"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from tstar_amd import _lib
from tstar_amd.interface_heuristic import OWLInterface
from tstar_amd.lockstep import search_lockstep
from tstar_amd.video import synthetic_video

g = int(sys.argv[1]) if len(sys.argv) > 1 else 16
as_group = len(sys.argv) > 2 and sys.argv[2] == "1"
lib = _lib.load()
mode = sys.argv[3] if len(sys.argv) > 3 else "f32x3"
h = OWLInterface(synthetic_seed=0, max_batch=256, device="cuda:0", weights_dtype=mode)
print("weights mode", mode, "| speculation", "off" if os.environ.get("TSTAR_NO_SPECULATION") else "on", "| solo path",
      "sequential loop" if os.environ.get("TSTAR_SOLO_SEQUENTIAL") else "lockstep.search_solo")
store = synthetic_video(3600, 360, 640, seed=0)
item = lambda seed: dict(store=store, targets=bench.TARGETS, cues=bench.CUES, seed=seed)


def one(seed):
    s = bench.make_searcher(h, item(seed), g, 8)
    if as_group:
        return s, search_lockstep([s])[0][1]
    return s, s.search()[1]


for w in range(3):
    one(100 + w)
torch.cuda.synchronize()
ts = []
for r in range(3):
    t0 = time.perf_counter()
    s, _ = one(200 + r)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("solo search wall (s):", [round(t, 4) for t in ts], "detector images", s.device_images_scored, "iterations", s.iterations)
_lib.check(lib.tstar_prof_mark(0, _lib.stream_ptr()))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
s, _ = one(300)
torch.cuda.synchronize()
pr.disable()
print("profiled run wall (s):", round(time.perf_counter() - t0, 4))
_lib.check(lib.tstar_prof_mark(1, _lib.stream_ptr()))
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print(out.getvalue()[:6000])
