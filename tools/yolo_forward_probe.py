"""Run the YOLO-World detector on one batch a few times (for rocprofv3 --kernel-trace: per-layer launch durations).
    python tools/yolo_forward_probe.py [B] [H] [W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tstar_amd import yolo_world as Y
from tstar_amd.yolo import YoloDetector

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 285
W = int(sys.argv[3]) if len(sys.argv) > 3 else 600
det = YoloDetector(Y.synthetic_state_dict(0, "l"), "l", max_batch=B)
rs = np.random.RandomState(0)
t = rs.standard_normal((4, 512)).astype(np.float32)
det.set_text_feats(t / np.linalg.norm(t, axis=1, keepdims=True), [1.0, 0.5, 0.5, 0.5])
img = torch.from_numpy(rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)).cuda()
for _ in range(2):
    det.detect(img, 1, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    det.detect(img, 1, 1)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"B={B} {H}x{W}: {ms:.2f} ms per batch, {ms / B:.3f} ms per image, {det.conv_flops_per_image * B / ms / 1e9:.1f} TFLOP/s (conv flops / wall)")
