"""Error behaviour of the C ABI and the Python mirror: bad arguments return TSTAR_ERR_ARG / raise with a
message (like the reference raises ValueError), nothing crashes, nothing falls back to the CPU."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from tstar_amd import _lib
    from tstar_amd.interface_heuristic import OWLInterface
    return _lib, _lib.load(), OWLInterface(synthetic_seed=0, max_batch=2)


def test_owl_score_argument_validation(env):
    L, lib, h = env
    h.reparameterize_object_list(["couch"], [])
    img = torch.zeros((1, 95, 200, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError, match="cuda uint8"):
        h.scorer.score(img.float(), 1, 1)
    with pytest.raises(ValueError, match="cuda uint8"):
        h.scorer.score(img[0], 1, 1)
    with pytest.raises(ValueError, match="at least 1x1"):
        h.scorer.score(img, 0, 1)
    with pytest.raises(L.TStarHipError, match="1..4096 cells"):
        h.scorer.score(img, 100, 100)
    with pytest.raises(ValueError, match="one slot per image"):
        h.scorer.score(img, 1, 1, image_sets=[0, 0])
    with pytest.raises(L.TStarHipError, match="query_set must be in 0..63"):
        h.scorer.score(img, 1, 1, image_sets=[64])
    r = h.scorer.score(img, 1, 1)                                   # still healthy afterwards
    assert torch.isfinite(r.scores).all()


def test_query_limits(env):
    L, lib, h = env
    ids = np.zeros((33, 16), np.int32); ids[:, 0] = 49406; ids[:, 1] = 49407
    with pytest.raises(L.TStarHipError, match="Q must be in 1..32"):
        h.scorer.set_queries(ids, np.ones_like(ids), [1.0] * 33)
    bad = np.full((1, 16), 60000, np.int32)
    with pytest.raises(L.TStarHipError, match="token id out of range"):
        h.scorer.set_queries(bad, np.ones_like(bad), [1.0])
    with pytest.raises(ValueError, match=r"\[Q,16\]"):
        h.scorer.set_queries(np.zeros((2, 8), np.int32), np.zeros((2, 8), np.int32), [1.0, 1.0])
    # 32 queries (the maximum) work end to end
    names = [f"thing{i}" for i in range(31)]
    h.reparameterize_object_list(names[:1], names[1:])
    assert h.scorer.Q == 32
    r = h.scorer.score(torch.zeros((1, 95, 200, 3), dtype=torch.uint8, device="cuda"), 1, 1)
    assert int(r.labels.max()) < 32


def test_searcher_state_validation(env):
    L, lib, h = env
    from tstar_amd.interface_searcher import _DeviceState
    with pytest.raises(L.TStarHipError, match="n_frames must be in"):
        _DeviceState(0, 1e-6, 0.1)
    st = _DeviceState(50, 1e-6, 0.1)
    conf = torch.zeros(4, dtype=torch.float64, device="cuda")
    with pytest.raises(L.TStarHipError, match="second out of range"):
        st.apply_grid([0, 1, 2, 50], conf)
    with pytest.raises(L.TStarHipError, match="bad sample count"):
        st.sampler_prep(51, 0.1)
    with pytest.raises(L.TStarHipError, match="bad spline"):
        st.set_spline(np.zeros(4), np.zeros(4), 3)
    with pytest.raises(L.TStarHipError, match="index out of range"):
        st.exclude([60])


def test_searcher_python_level_errors(env):
    L, lib, h = env
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import FrameStore, synthetic_video
    st = synthetic_video(20, seed=1)
    s = TStarSearcher(st, h, ["a"], [], search_nframes=30, image_grid_shape=(2, 2), search_budget=0.5,
                      rng=np.random.RandomState(0), keep_visual_history=False)
    with pytest.raises(ValueError, match="larger sample than population"):       # numpy's message, as the reference
        s.pop_frames(None, 30)
    # grid larger than the video: the reference clamps the sample count and then fails in create_image_grid
    s2 = TStarSearcher(st, h, ["a"], [], image_grid_shape=(5, 5), rng=np.random.RandomState(0))
    with pytest.raises(ValueError, match="Frame count does not match grid dimensions"):
        s2.search()
    with pytest.raises(ValueError, match="fewer frames"):
        TStarSearcher(FrameStore(st.frames[:5], 1.0, raw_total_frames=20), h, ["a"], [])
    with pytest.raises(ValueError, match="fmt must be"):
        FrameStore(st.frames, 1.0, fmt="yuv")
    # no targets: the loop never runs, keyframes come from the flat initial scores (reference behaviour)
    s3 = TStarSearcher(st, h, [], ["tv"], search_nframes=3, image_grid_shape=(2, 2), rng=np.random.RandomState(1))
    fr, ts = s3.search()
    assert len(ts) == 3 and s3.iterations == 0 and s3.P_history == []


def test_kernel_entry_point_validation(env):
    L, lib, h = env
    d = torch.zeros(1024, device="cuda")
    assert lib.tstar_layernorm_f32(d.data_ptr(), d.data_ptr(), d.data_ptr(), d.data_ptr(), 1, 300, None) == 1
    assert b"512 or 768" in lib.tstar_last_error()
    assert lib.tstar_attention_f32(d.data_ptr(), d.data_ptr(), 1, 4, 1, 1, None, None) == 1      # mode 1 needs a mask
    assert lib.tstar_gemm_f32(d.data_ptr(), d.data_ptr(), d.data_ptr(), None, None, 0, 128, 32, 0, None) == 1
    assert lib.tstar_gemm_f32_cfg(d.data_ptr(), d.data_ptr(), d.data_ptr(), None, None, 8, 128, 32, 0, 9, None) == 1
    assert lib.tstar_topk_seconds(d.data_ptr(), 10, 5, 3, 1, d.data_ptr(), None) == 1
    assert lib.tstar_ssim_pairwise(None, 1, None, 1, 4, 4, None, None, None) == 1
    n = C.c_longlong(); ms = C.c_double(); fl = C.c_double()
    assert lib.tstar_prof_read(7, C.byref(n), C.byref(ms), C.byref(fl)) == 1


def test_round3_entry_points_validate_their_arguments(env):
    """The entries added in round 3: marker kernels, algorithmic-byte counters, prepared-weights bf16 GEMM, RCCL availability."""
    L, lib, h = env
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tstar_prof_mark(2, st) == 1 and b"0 (begin) or 1 (end)" in lib.tstar_last_error()
    assert lib.tstar_prof_mark(0, st) == 0 and lib.tstar_prof_mark(1, st) == 0
    by = C.c_double(-1.0)
    assert lib.tstar_prof_read_bytes(7, C.byref(by)) == 1
    L.check(lib.tstar_prof_enable(1))
    A = torch.randn(256, 64, device="cuda")
    Wb = torch.randn(128, 64, device="cuda").to(torch.bfloat16)
    Cc = torch.empty(256, 128, device="cuda")
    assert lib.tstar_gemm_bf16w_pre(A.data_ptr(), Wb.data_ptr(), Cc.data_ptr(), None, None, 256, 128, 64, 0, 4, -1, st) == 1     # a_terms 2 or 3
    assert lib.tstar_gemm_bf16w_pre(A.data_ptr(), Wb.data_ptr(), Cc.data_ptr(), None, None, 256, 100, 64, 0, 2, -1, st) == 1     # N % 128
    L.check(lib.tstar_gemm_bf16w_pre(A.data_ptr(), Wb.data_ptr(), Cc.data_ptr(), None, None, 256, 128, 64, 0, 2, -1, st))
    L.check(lib.tstar_gemm_bf16w_pre(A.data_ptr(), Wb.data_ptr(), Cc.data_ptr(), None, None, 256, 128, 64, 0, 3, -1, st))
    torch.cuda.synchronize()
    ref = A.double() @ Wb.double().t()
    assert (Cc.double() - ref).abs().max().item() < 1e-4
    n, ms, fl = C.c_longlong(0), C.c_double(0), C.c_double(0)
    L.check(lib.tstar_prof_read(0, C.byref(n), C.byref(ms), C.byref(fl)))
    L.check(lib.tstar_prof_read_bytes(0, C.byref(by)))
    na, fa = C.c_longlong(0), C.c_double(0)
    L.check(lib.tstar_prof_read_totals(0, C.byref(na), C.byref(fa)))
    assert lib.tstar_prof_read_totals(9, C.byref(na), C.byref(fa)) == 1
    assert na.value == 2 and fa.value == fl.value                                         # stride 1: every launch is sampled
    L.check(lib.tstar_prof_enable(0))
    assert n.value == 2 and fl.value == 2 * 2.0 * 256 * 128 * 64
    assert by.value == 2 * (4.0 * 256 * 64 + 2.0 * 128 * 64 + 4.0 * 256 * 128)            # A (f32) + W (bf16) + C, per launch
    assert lib.tstar_comm_available() == 0                                                # torch's RCCL is loadable on a GPU box
