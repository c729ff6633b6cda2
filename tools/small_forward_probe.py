"""Back-to-back small grid forwards (the reference-default 4x4 grid image, B = 1 and B = 4): wall time per forward with HIP
events.  Under `rocprofv3 --kernel-trace` + tools/rocpd_gaps.py it shows how much of that wall time is idle GPU between the
~100 dependent launches of one forward -- the only thing a hipGraph could remove.

    python tools/small_forward_probe.py [forwards per batch size, default 40] [weights mode: f32 (default) | f32x3 | ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tstar_amd.interface_heuristic import OWLInterface

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
h = OWLInterface(synthetic_seed=0, max_batch=8, device="cuda:0", weights_dtype=mode)
print("weights mode", mode)
h.reparameterize_object_list(["couch"], ["tv"])
g = torch.Generator(device="cpu").manual_seed(0)
for B in (1, 4):
    x = torch.randint(0, 256, (B, 380, 800, 3), dtype=torch.uint8, generator=g).cuda()
    for _ in range(3):
        h.score_batch(x, 4, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        h.score_batch(x, 4, 4)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"B={B}: {ms:.3f} ms per forward ({B * 114.8 / ms:.1f} TFLOP/s of the 114.8 GFLOP/image)")
