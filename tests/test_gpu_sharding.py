"""Multi-GPU row on the one GPU of the test box: the library's RCCL entry points at world 1, and the
dataset driver (items sharded over ranks, lock-step groups, one all-gather) against one-by-one searches."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_allgather_world1():
    """tstar_comm_unique_id / tstar_comm_create / tstar_allgather_i32 / tstar_comm_destroy through the C ABI: RCCL is
    dlopen'ed (the copy torch already carries), a 1-rank communicator is created on the current device and the gather of
    a padded keyframe buffer returns it unchanged, on the caller's stream."""
    from tstar_amd import _lib
    lib = _lib.load()
    idbuf = C.create_string_buffer(128)
    _lib.check(lib.tstar_comm_unique_id(idbuf), "tstar_comm_unique_id")
    assert any(idbuf.raw)
    h = C.c_void_p()
    _lib.check(lib.tstar_comm_create(C.byref(h), idbuf.raw, 1, 0), "tstar_comm_create")
    try:
        send = torch.tensor([[3, 17, 99, -1], [5, 6, 7, 8]], dtype=torch.int32, device="cuda")
        recv = torch.full((1, 2, 4), -7, dtype=torch.int32, device="cuda")
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            _lib.check(lib.tstar_allgather_i32(h, send.data_ptr(), recv.data_ptr(), send.numel(), st.cuda_stream), "tstar_allgather_i32")
        st.synchronize()
        assert torch.equal(recv[0], send)
        assert lib.tstar_allgather_i32(h, send.data_ptr(), recv.data_ptr(), 0, None) == 1          # TSTAR_ERR_ARG
        assert lib.tstar_comm_create(C.byref(C.c_void_p()), idbuf.raw, 2, 2) == 1                   # rank out of range
    finally:
        lib.tstar_comm_destroy(h)


def test_dataset_driver_equals_one_by_one_searches(tmp_path):
    """examples/run_dataset.py (the reference's run_TStar_onDataset.py loop, sharded + lock-step) writes the reference's
    result JSON; every item's keyframes and distribution must equal a plain one-by-one search of that item with the same
    per-item sampler seed (results independent of grouping and rank count)."""
    from tstar_amd.interface_heuristic import initialize_heuristic
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.sharding import item_seed
    out = tmp_path / "res.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_dataset.py"), "--items", "5", "--nframes", "240",
                        "--lockstep", "3", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.load(open(out))
    assert len(res) == 5
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import run_dataset as RD
    h = initialize_heuristic("owl-vit", synthetic_seed=0, max_batch=64)
    for i, item in enumerate(res):
        assert set(item) >= {"video_path", "grounding_objects", "keyframe_timestamps", "keyframe_distribution"}
        t_, c_ = RD.QUESTIONS[i % 4]
        assert item["grounding_objects"] == {"target_objects": t_, "cue_objects": c_}
        s = TStarSearcher(item["video_path"], h, list(t_), list(c_), search_nframes=8, image_grid_shape=(4, 4),
                          search_budget=1000, confidence_threshold=0.6, rng=np.random.RandomState(item_seed(2025, i)),
                          keep_visual_history=False)
        _, ts = s.search()
        assert [float(t) for t in ts] == item["keyframe_timestamps"], i
        assert s.P_history[-1] == item["keyframe_distribution"]
