"""YOLO-World forward time against the batch size (one detector, max_batch = the largest B): where the workgroup rounds of the
conv kernels quantise (halo kernel: 3 workgroups per CU = 768 slots; a 40x40 / 256-channel layer is 20 B workgroups).
    python tools/yolo_batch_sweep.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tstar_amd import yolo_world as Y
from tstar_amd.yolo import YoloDetector

Bs = [int(v) for v in sys.argv[1:]] or [4, 8, 12, 16, 19, 24, 32, 36, 38, 40, 48, 57, 64, 76, 80]
det = YoloDetector(Y.synthetic_state_dict(0, "l"), "l", max_batch=max(Bs))
rs = np.random.RandomState(0)
t = rs.standard_normal((4, 512)).astype(np.float32)
det.set_text_feats(t / np.linalg.norm(t, axis=1, keepdims=True), [1.0, 0.5, 0.5, 0.5])
full = torch.from_numpy(rs.randint(0, 256, (max(Bs), 285, 600, 3)).astype(np.uint8)).cuda()
for rnd in range(2):
    for B in Bs:
        img = full[:B]
        det.detect(img, 1, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 4
        for _ in range(n):
            det.detect(img, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        if rnd == 1:
            print(f"B={B:3d}: {ms:7.2f} ms per batch, {ms / B:.3f} ms per image, {det.conv_flops_per_image * B / ms / 1e9:6.1f} TFLOP/s (conv flops / wall)", flush=True)
