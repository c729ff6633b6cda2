"""Helper of tests/test_gpu_yolo.py::test_conv_kernels_bit_identical: one forward of a seeded YOLO-World-v2 model with
whatever TSTAR_YOLO_SW / TSTAR_YOLO_SW_P the parent set (the library reads them once per process); prints a SHA-256 of
the dense scores and boxes.

    python tests/yolo_conv_variant_probe.py <scale> <batch>
"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as GU  # noqa: E402
from tstar_amd import yolo_world as Y  # noqa: E402
from tstar_amd.yolo import YoloDetector  # noqa: E402

scale, B = sys.argv[1], int(sys.argv[2])
sd = Y.synthetic_state_dict(3, scale)
det = YoloDetector(sd, scale, max_batch=B)
rs = np.random.RandomState(11)
txt = rs.standard_normal((3, 512)).astype(np.float32)
txt /= np.linalg.norm(txt, axis=1, keepdims=True)
det.set_text_feats(txt, [1.0, 0.5, 0.5])
imgs = np.stack([GU.detector_test_image(90 + b, 285, 600) for b in range(B)])
r = det.detect(torch.from_numpy(imgs).cuda(), 1, 1, want_dense=True)
torch.cuda.synchronize()
h = hashlib.sha256()
h.update(r.dense_scores.cpu().numpy().tobytes())
h.update(r.dense_boxes.cpu().numpy().tobytes())
h.update(r.scores.cpu().numpy().tobytes())
print("SHA", h.hexdigest(), float(r.dense_scores.max()))
