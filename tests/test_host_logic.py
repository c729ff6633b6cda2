"""CPU: host-side logic, the C-ABI surface (symbols only; no compute without a GPU), sharding over
gloo with world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tstar_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tstar_hip.h")).read()
    declared = set(re.findall(r"\b(tstar_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.tstar_abi_version() == 3


def test_blob_layout_matches_library():
    from tstar_amd import _lib, weights as W
    lib = _lib.load()
    assert lib.tstar_owl_vision_blob_floats() == W.spec_size(W.vision_spec())
    assert lib.tstar_owl_text_blob_floats() == W.spec_size(W.text_spec())


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tstar_amd import _lib
    from tstar_amd.owl import OwlScorer
    from tstar_amd.interface_heuristic import OWLInterface
    with pytest.raises(_lib.TStarHipError):
        OwlScorer(np.zeros(4, np.float32))
    with pytest.raises(ValueError, match="GPU only"):
        OWLInterface(device="cpu", synthetic_seed=0)


def test_environment_knobs_are_validated_before_anything_is_loaded(monkeypatch):
    """TSTAR_WEIGHTS_DTYPE / TSTAR_MAX_BATCH / TSTAR_SYNTHETIC_SEED stand in for the keyword arguments an unchanged TStarFramework
    cannot pass (TStarFramework.py:171-187, 207); a bad value is reported by name, before weights are touched."""
    from tstar_amd.interface_heuristic import OWLInterface, _env_int
    monkeypatch.setenv("TSTAR_WEIGHTS_DTYPE", "fp8")
    with pytest.raises(ValueError, match="TSTAR_WEIGHTS_DTYPE"):
        OWLInterface()
    monkeypatch.setenv("TSTAR_WEIGHTS_DTYPE", "f32x3")
    monkeypatch.setenv("TSTAR_MAX_BATCH", "lots")
    with pytest.raises(ValueError, match="TSTAR_MAX_BATCH must be an integer"):
        OWLInterface()
    monkeypatch.setenv("TSTAR_MAX_BATCH", "")
    assert _env_int("TSTAR_MAX_BATCH", 32) == 32
    monkeypatch.setenv("TSTAR_MAX_BATCH", "48")
    assert _env_int("TSTAR_MAX_BATCH", 32) == 48


def test_heuristic_factory_names():
    """initialize_heuristic keeps the reference's type names (TStarFramework.py:171-187): 'owl-vit' and 'yolo-World' are
    built (the latter needs weights: nothing can be downloaded), unknown names raise NotImplementedError like the
    reference's else branch."""
    from tstar_amd.interface_heuristic import _yolo_scale_from_config, initialize_heuristic
    with pytest.raises(FileNotFoundError, match="no YOLO-World checkpoint"):
        initialize_heuristic("yolo-World")
    assert _yolo_scale_from_config("./YOLO-World/configs/pretrain/yolo_world_v2_xl_vlpan_bn_2e-3_100e_4x8gpus_obj365v1_goldg_train_lvis_minival.py") == "xl"
    assert _yolo_scale_from_config("yolo_world_v2_l_vlpan_bn.py") == "l" and _yolo_scale_from_config(None) == "l"
    with pytest.raises(NotImplementedError, match="not implemented"):
        initialize_heuristic("frcnn")


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py (its cpu_baseline / post-timer verification legs) may touch oracle/:
    nothing under the package, the examples or the tools does."""
    for top in ("tstar_amd", "examples", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(top, f)


def test_tokenizer_standin_layout():
    from tstar_amd.tokenizer import BOS, EOS, encode_queries, standin_word_id
    ids, am = encode_queries([["couch"], ["a photo of tv"], [" "]])
    assert ids.shape == (3, 16) and ids.dtype == np.int32
    assert ids[0, 0] == BOS and ids[0, 2] == EOS and ids[0, 3] == 0 and am[0].sum() == 3
    assert ids[1, 5] == EOS and am[1].sum() == 6
    assert ids[2, 0] == BOS and ids[2, 1] == EOS and am[2].sum() == 2       # the blank background query
    assert standin_word_id("Couch") == standin_word_id("couch") == ids[0, 1]


def test_synthetic_weights_are_reproducible_and_shaped():
    from tstar_amd import weights as W
    a = W.synthetic_state_dict(3, "text")
    b = W.synthetic_state_dict(3, "text")
    assert all(np.array_equal(a[k], b[k]) for k in a)
    blob = W.pack_blob(a, W.text_spec())
    assert blob.size == W.spec_size(W.text_spec())
    back = W.unpack_blob(blob, W.text_spec())
    assert back["tok_emb"].shape == (W.VOCAB, W.T_D)
    with pytest.raises(ValueError):
        W.unpack_blob(blob[:-1], W.text_spec())
    bb = W.compute_box_bias()
    assert bb.shape == (576, 4) and abs(bb[0, 0] - (-3.1332)) < 1e-3 and abs(bb[-1, 0] - 9.2103) < 1e-3


def test_synthetic_video_cpu_paths_agree():
    from tstar_amd import video
    idx = [0, 3, 17, 49]
    a = video.synthetic_frames_numpy(idx, 50, seed=9)
    b = video.synthetic_video(50, seed=9, device="cpu").frames[idx].numpy()
    assert np.array_equal(a, b)
    with pytest.raises(ValueError, match="Cannot open video file"):
        video.open_video("/nonexistent.mp4", device="cpu")


def test_shard_helpers():
    from tstar_amd import sharding as Sh
    assert Sh.shard_items(10, 4, 1) == [1, 5, 9]
    rows = [[i, i + 1] for i in range(10)]
    world = 4
    gathered = []
    for r in range(world):
        gathered += [rows[i] for i in Sh.shard_items(10, world, r)]
    assert Sh.interleave_by_item(gathered, 10, world) == rows
    assert Sh.gather_keyframes([[1, 2]], 1) == [[1, 2]]


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from tstar_amd.sharding import run_sharded, item_seed
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
res = run_sharded(5, lambda i: [item_seed(100, i), i * i, 7], 2, int(sys.argv[3]))
assert res == [[100 + i, i * i, 7] for i in range(5)], res
dist.destroy_process_group()
print("ok")
'''


def test_sharded_gather_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


_WORKER4 = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from tstar_amd.sharding import run_sharded, shard_items, item_seed
rank, world = int(sys.argv[3]), 4
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
# 10 items over 4 ranks: ranks 0,1 own 3 items, ranks 2,3 own 2 (uneven shards -> padded rows); K differs per item too
def search(i):
    return [item_seed(2025, i)] + list(range(i % 3 + 1))
res = run_sharded(10, search, world, rank)
assert res == [search(i) for i in range(10)], res
assert shard_items(10, world, rank) == list(range(rank, 10, world))
# fewer items than ranks: ranks 2 and 3 own nothing
res = run_sharded(2, search, world, rank)
assert res == [search(0), search(1)], res
# lock-step groups of 2 within each rank's shard (what configs[2] runs per GPU): same gathered rows, groups seen in shard order
seen = []
def group(ids):
    seen.append(list(ids))
    return [search(i) for i in ids]
res = run_sharded(10, None, world, rank, search_group=group, group_size=2)
assert res == [search(i) for i in range(10)], res
mine = shard_items(10, world, rank)
assert seen == [mine[k:k + 2] for k in range(0, len(mine), 2)], seen
# items that return ZERO keyframes keep their place (round-3 review: an all-(-1) row used to be dropped, shifting the rest
# of that rank's items); item 1 (rank 1's first) and item 6 (rank 2's last) are empty, then every item is empty
def search_some_empty(i):
    return [] if i in (1, 6) else search(i)
res = run_sharded(10, search_some_empty, world, rank)
assert res == [search_some_empty(i) for i in range(10)], res
res = run_sharded(10, lambda i: [], world, rank)
assert res == [[] for _ in range(10)], res
from tstar_amd import sharding
assert "gloo" in sharding.LAST_GATHER_PATH and "4 ranks" in sharding.LAST_GATHER_PATH
dist.destroy_process_group()
print("ok")
'''


def test_sharded_gather_world4_uneven_gloo(tmp_path):
    """BASELINE configs[2] shape on CPU: 10 items round-robin over 4 ranks (3/3/2/2), rows of different length, and
    the fewer-items-than-ranks case, through run_sharded's one all-gather (gloo here, RCCL on GPUs)."""
    script = tmp_path / "w4.py"
    script.write_text(_WORKER4)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(4)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_spline_pool_distribution_matches_reference_statements(monkeypatch):
    """lock-step groups get their sampling distributions P from the worker pool: bit-identical to the in-process
    statements (tstar_amd.spline_worker.spline_distribution = interface_searcher.py:262-274) and to the oracle,
    for items of DIFFERENT video lengths in one call."""
    from oracle import searcher_ref as S
    from tstar_amd import spline_pool
    from tstar_amd.spline_worker import spline_distribution
    probs, lens = [], [600, 1200, 350]
    for seed, N in enumerate(lens):
        rs = np.random.RandomState(seed)
        vis = np.sort(rs.choice(N, 40, replace=False))
        probs.append((vis.astype(np.int32), rs.rand(40) ** 4))
    pool = spline_pool.SplinePool(2)
    try:
        got = pool.fit_many(probs, s=0.5, n_frames=lens)
    finally:
        pool.close()
    for (x, y), N, P in zip(probs, lens, got):
        unv = np.ones(N)
        unv[x] = 0
        sc = np.zeros(N)
        sc[x] = y
        assert P.shape == (N,) and np.array_equal(P, spline_distribution(x, y, N)) and np.array_equal(P, S.spline_distribution(unv, sc))
    assert np.array_equal(spline_distribution(np.array([], int), np.array([]), 7), np.ones(7) / 7)
    monkeypatch.setenv("TSTAR_SPLINE_WORKERS", "0")
    monkeypatch.setattr(spline_pool, "_pool", None)
    monkeypatch.setattr(spline_pool, "_pool_failed", False)
    same = spline_pool.distribution_many(probs, lens)
    assert all(np.array_equal(a, b) for a, b in zip(same, got))
    with pytest.raises(ValueError):
        spline_pool.distribution_many(probs, lens[:2])
    # the asynchronous form (a search running alone starts its fit, enqueues the verification batch, then collects P):
    # without a pool it computes at wait(); with one, in a worker -- the same P either way, also for ONE problem and for
    # "nothing visited yet", and the caller's buffers may be overwritten between the two calls
    assert all(np.array_equal(a, b) for a, b in zip(spline_pool.distribution_async(probs, lens)(), got))
    monkeypatch.setenv("TSTAR_SPLINE_WORKERS", "2")
    monkeypatch.setattr(spline_pool, "_pool", None)
    try:
        x0, y0 = probs[0][0].copy(), probs[0][1].copy()
        wait = spline_pool.distribution_async([(x0, y0)], lens[:1])
        x0[:] = 0
        y0[:] = 0
        assert np.array_equal(wait()[0], got[0])
        assert all(np.array_equal(a, b) for a, b in zip(spline_pool.distribution_async(probs[:2], lens[:2])(), got[:2]))
        assert all(np.array_equal(a, b) for a, b in zip(spline_pool.distribution_async(probs, lens)(), got))       # 3 problems, 2 workers
        assert np.array_equal(spline_pool.distribution_async([(np.array([], int), np.array([]))], [7])()[0], np.ones(7) / 7)
        assert spline_pool.distribution_many(probs[:2], lens[:2])[1].shape == (lens[1],)                          # the pool is free again
    finally:
        if spline_pool._pool is not None:
            spline_pool._pool.close()
        monkeypatch.setattr(spline_pool, "_pool", None)
        monkeypatch.setattr(spline_pool, "_pool_failed", False)


def test_spline_pool_matches_in_process_fit(monkeypatch):
    """The lock-step host path fits each item's smoothing spline in a worker process: (t, c, k) must be
    bit-identical to the in-process ``UnivariateSpline(x, y, s=0.5)`` the reference calls
    (interface_searcher.py:265), in input order, for more problems than workers."""
    from scipy.interpolate import UnivariateSpline
    from tstar_amd import spline_pool
    N = 600
    x = np.arange(N, dtype=np.int32)
    probs = []
    for seed in range(5):
        rs = np.random.RandomState(seed)
        y = np.full(N, 0.1)
        idx = rs.choice(N, 25, replace=False)
        y[idx] = rs.rand(25)
        probs.append((x, y))
    pool = spline_pool.SplinePool(2)
    try:
        got = pool.fit_many(probs, s=0.5)
    finally:
        pool.close()
    for (xx, yy), (t2, c2, k2) in zip(probs, got):
        t, c, k = UnivariateSpline(xx, yy, s=0.5)._eval_args
        assert k2 == k and np.array_equal(t2, t)
        assert np.array_equal(c2[:len(c)], c) and not c2[len(c):].any()
    # disabled pool -> in-process path, same answers
    monkeypatch.setenv("TSTAR_SPLINE_WORKERS", "0")
    monkeypatch.setattr(spline_pool, "_pool", None)
    monkeypatch.setattr(spline_pool, "_pool_failed", False)
    assert spline_pool.get_pool() is None
    same = spline_pool.fit_many(probs[:2], s=0.5)
    assert np.array_equal(same[0][0], got[0][0]) and np.array_equal(same[1][1], got[1][1][:len(same[1][1])])


def test_spline_pool_reports_fit_errors(monkeypatch):
    """A fit the library rejects (x not increasing) surfaces as an error, not as a hang or a wrong curve;
    through the module-level entry the caller sees scipy's own exception."""
    from tstar_amd import spline_pool
    bad = (np.array([0.0, 2.0, 1.0, 3.0, 4.0]), np.zeros(5))
    good = (np.arange(8.0), np.arange(8.0) ** 2)
    pool = spline_pool.SplinePool(1)
    with pytest.raises(spline_pool.SplinePoolError, match="ValueError"):
        pool.fit_many([good, bad])
    assert len(pool) == 0                              # closed after a failure
    with pytest.raises(spline_pool.SplinePoolError, match="closed"):
        pool.fit_many([good])
    monkeypatch.setattr(spline_pool, "_pool", None)                # module state restored after the test
    monkeypatch.setattr(spline_pool, "_pool_failed", False)
    with pytest.raises(ValueError):
        spline_pool.fit_many([bad, bad])
    if spline_pool._pool is not None:
        spline_pool._pool.close()


def test_checkpoint_directory_round_trip(tmp_path):
    """An HF ``save_pretrained`` directory (model.safetensors, HF parameter names, no box_bias buffer, extra
    keys such as logit_scale) is found and packs into the C-ABI blobs exactly like the in-memory state dict --
    the path a real ``google/owlvit-base-patch32`` download takes (interface_heuristic.py:207-210)."""
    transformers = pytest.importorskip("transformers")
    pytest.importorskip("safetensors")
    import torch
    from tstar_amd import weights as W
    torch.manual_seed(0)
    m = transformers.OwlViTForObjectDetection(transformers.OwlViTConfig())
    d = tmp_path / "owlvit-base-patch32"
    m.save_pretrained(str(d), safe_serialization=True)
    ckpt = W.find_pretrained(str(d))
    assert ckpt is not None and ckpt.endswith("model.safetensors")
    sd = W.load_safetensors_state_dict(ckpt)
    assert "box_bias" not in sd                                     # a non-persistent buffer: recomputed
    vb = W.pack_blob(sd, W.vision_spec())
    tb = W.pack_blob(sd, W.text_spec())
    assert vb.size == W.spec_size(W.vision_spec()) and tb.size == W.spec_size(W.text_spec())
    wv = W.unpack_blob(vb, W.vision_spec())
    wt = W.unpack_blob(tb, W.text_spec())
    ref = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    l0 = "owlvit.vision_model.encoder.layers.0."
    qkv = np.concatenate([ref[l0 + f"self_attn.{n}_proj.weight"] for n in "qkv"])
    assert np.array_equal(wv[l0 + "qkv_w"], qkv)
    assert np.array_equal(wv["patch_w"].reshape(768, 3, 32, 32), ref["owlvit.vision_model.embeddings.patch_embedding.weight"])
    assert np.array_equal(wv["cls_w"], ref["class_head.dense0.weight"])
    assert np.array_equal(wt["text_proj"], ref["owlvit.text_projection.weight"])
    assert np.array_equal(wv["box_bias"], m.box_bias.numpy())       # restated buffer == HF's
    assert W.find_pretrained(str(tmp_path / "missing")) is None


def test_open_video_through_a_decord_like_reader(monkeypatch):
    """File paths are decoded once through decord (interface_searcher.py:157-169 reopens it per call): with a stub
    reader, the store holds raw frame int(sec * raw_fps) for every logical second and keeps raw fps / frame count."""
    import types
    from tstar_amd import video

    class _Batch:
        def __init__(self, a):
            self._a = a

        def asnumpy(self):
            return self._a

    class VideoReader:
        def __init__(self, path, ctx=None):
            assert path == "/data/clip.mp4"

        def get_avg_fps(self):
            return 29.97

        def __len__(self):
            return 305                       # 10 whole seconds and a bit

        def get_batch(self, idx):
            return _Batch(np.stack([np.full((6, 8, 3), i % 251, np.uint8) for i in idx]))

    stub = types.ModuleType("decord")
    stub.VideoReader = VideoReader
    stub.cpu = lambda i: None
    monkeypatch.setitem(sys.modules, "decord", stub)
    st = video.open_video("/data/clip.mp4", device="cpu")
    assert st.num_seconds == 10 and st.raw_fps == 29.97 and st.raw_total_frames == 305
    want = [int(s * 29.97) % 251 for s in range(10)]
    assert [int(st.frames[s, 0, 0, 0]) for s in range(10)] == want
    assert st.shape == (10, 6, 8, 3)


def test_comm_create_watchdog(monkeypatch):
    """Round 6 (review item 3): ``tstar_comm_create`` = ncclCommInitRank, a collective that can HANG (a peer that never arrives) instead
    of failing.  The watchdog runs it on a helper thread: a call that returns is passed through (handle / error text read on the thread
    that made the call), one that does not return within TSTAR_COMM_TIMEOUT_S is abandoned and reported, so that the gather falls back
    to torch.distributed instead of eating the job's time limit."""
    import threading
    import time
    from tstar_amd import _lib, sharding as SH
    lib = _lib.load()
    monkeypatch.setenv("TSTAR_COMM_TIMEOUT_S", "0.2")
    assert SH.comm_timeout_s() == 1.0                                     # floor of one second
    monkeypatch.setenv("TSTAR_COMM_TIMEOUT_S", "soon")
    assert SH.comm_timeout_s() == 90.0
    seen = {}

    def ok(hp, uid, world, rank):
        seen["thread"] = threading.current_thread().name
        seen["args"] = (bytes(uid), world, rank)
        return 0

    h, note = SH._create_with_watchdog(lib, b"x" * 128, 8, 3, 5.0, create=ok)
    assert h is not None and note == "" and seen["thread"] == "tstar-comm-create" and seen["args"] == (b"x" * 128, 8, 3)
    h, note = SH._create_with_watchdog(lib, b"x" * 128, 8, 3, 5.0, create=lambda *a: 2)
    assert h is None and note                                              # an error code: no handle, a reason
    release = threading.Event()
    monkeypatch.setattr(SH, "COMM_TIMED_OUT", False)
    t0 = time.perf_counter()
    h, note = SH._create_with_watchdog(lib, b"x" * 128, 8, 3, 0.3, create=lambda *a: (release.wait(20), 0)[1])
    assert h is None and "did not return within" in note and "8 ranks" in note and SH.COMM_TIMED_OUT is True
    assert time.perf_counter() - t0 < 5.0                                  # the caller got its thread back
    release.set()
    # the real entry point with a bad argument comes back through the same path with the library's own message
    h, note = SH._create_with_watchdog(lib, b"x" * 128, 2, 5, 5.0)
    assert h is None and "rank must be in 0..world-1" in note


def test_shard_interleave_property():
    """For every (n_items, world) the rank-major all-gather order maps back to item order, including worlds larger than the
    item count (empty shards) and ragged last rounds."""
    from hypothesis import given, settings, strategies as st
    from tstar_amd import sharding as Sh

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 70), st.integers(1, 16))
    def prop(n_items, world):
        rows = [[i, 3 * i + 1] for i in range(n_items)]
        shards = [Sh.shard_items(n_items, world, r) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(n_items))                   # a partition
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1    # balanced
        gathered = [rows[i] for s in shards for i in s]                          # what gather_keyframes returns (padding dropped)
        assert Sh.interleave_by_item(gathered, n_items, world) == rows
    prop()


def test_y4m_header_round_trip_and_rejection(tmp_path):
    """parse_y4m_header on streams written by write_y4m (random even sizes, frame rates, frame counts) and on malformed
    ones: wrong magic, odd dimensions, non-4:2:0 chroma, missing FRAME marker -- each a ValueError naming the file, the
    way the reference's reader reports a file it cannot open."""
    from hypothesis import given, settings, strategies as st
    from tstar_amd.video import parse_y4m_header, write_y4m

    @settings(max_examples=25, deadline=None)
    @given(st.integers(1, 12), st.integers(1, 10), st.integers(0, 4), st.sampled_from([(1, 1), (25, 1), (30000, 1001), (2, 3)]))
    def prop(hw, hh, n, fps):
        w, h = 2 * hw, 2 * hh
        rs = np.random.RandomState(w * 100 + h)
        p = str(tmp_path / f"v_{w}_{h}_{n}_{fps[0]}.y4m")
        write_y4m(p, rs.randint(0, 256, (max(n, 1), h * 3 // 2, w), dtype=np.uint8)[:max(n, 1)], fps=fps)
        hd = parse_y4m_header(p)
        assert (hd["w"], hd["h"], hd["n_frames"], hd["fps_frac"]) == (w, h, max(n, 1), fps)
        assert hd["frame_bytes"] == w * h * 3 // 2 and abs(hd["fps"] - fps[0] / fps[1]) < 1e-12
    prop()
    bad = {"magic": b"YUV4MPEG W4 H4 F1:1\nFRAME\n" + bytes(24), "odd": b"YUV4MPEG2 W5 H4 F1:1\nFRAME\n" + bytes(30),
           "chroma": b"YUV4MPEG2 W4 H4 F1:1 C444\nFRAME\n" + bytes(48), "noframe": b"YUV4MPEG2 W4 H4 F1:1\nXRAME\n" + bytes(24),
           "rate": b"YUV4MPEG2 W4 H4 F0:1\nFRAME\n" + bytes(24), "nosize": b"YUV4MPEG2 F1:1\nFRAME\n" + bytes(24)}
    for name, blob in bad.items():
        p = tmp_path / f"bad_{name}.y4m"
        p.write_bytes(blob)
        with pytest.raises(ValueError, match="Cannot open video file"):
            parse_y4m_header(str(p))


def test_real_checkpoint_directory_host_side(tmp_path):
    """D1 / D6 host side without a GPU: a checkpoint directory in HF's layout (HF-initialised model saved with
    save_pretrained + CLIP vocabulary files, tests/hf_checkpoint_util.py) is found, loaded from safetensors and packed into
    both tower blobs; queries take the real CLIP-BPE branch of tstar_amd/tokenizer.py: ids equal to transformers'
    CLIPTokenizer on the same files and to the restated algorithm (oracle/clip_bpe_ref.py), including HF's "!" = pad-id
    quirk, unicode, contractions and truncation to 16 with EOS kept."""
    import hf_checkpoint_util as H
    from oracle.clip_bpe_ref import ClipBpe
    from tstar_amd import weights as W
    from tstar_amd.tokenizer import encode_queries
    from transformers import CLIPTokenizer
    d = str(tmp_path / "ckpt")
    m = H.make_checkpoint_dir(d, seed=0)
    ck = W.find_pretrained(d)
    assert ck == os.path.join(d, "model.safetensors")
    sd = W.load_safetensors_state_dict(ck)
    bv, bt = W.pack_blob(sd, W.vision_spec()), W.pack_blob(sd, W.text_spec())
    wv = W.unpack_blob(bv, W.vision_spec())
    hf = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    assert np.array_equal(wv["box_bias"], m.box_bias.numpy())                  # the non-persistent buffer is recomputed identically
    k = "owlvit.vision_model.encoder.layers.3.mlp.fc1.weight"
    assert any(np.array_equal(v.reshape(-1), hf[k].reshape(-1)) for v in wv.values() if v.size == hf[k].size)
    names = ["couch", "tv", "park bench", " ", "Remote-Control 42!", "a photo of the dog's leash", "wow!! a!b  CAT",
             "the cat the dog the car the road the mug the desk the tv the chair the couch", "na\u00efve caf\u00e9 \u21165 \u00bd", "it's o'clock we're"]
    ids, am = encode_queries([[n] for n in names], d, allow_standin=False)
    tok = CLIPTokenizer.from_pretrained(d, local_files_only=True)
    want = tok(names, padding="max_length", max_length=16, truncation=True, return_tensors="np")
    assert np.array_equal(ids, want["input_ids"]) and np.array_equal(am, want["attention_mask"])
    bpe = ClipBpe(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
    ids2, am2 = bpe.encode_queries(names)
    assert np.array_equal(ids, ids2) and np.array_equal(am, am2)
    assert ids[0, 0] == 49406 and ids[0, 2] == 49407 and ids[7, 15] == 49407 and ids[4, -1] == 0 and (ids[3, :3] == [49406, 49407, 0]).all()
    # a different directory without vocabulary files does not inherit this one's tokenizer (per-path cache)
    with pytest.raises(RuntimeError, match="no CLIP tokenizer files"):
        encode_queries([["couch"]], str(tmp_path / "nowhere"), allow_standin=False)


def test_compressed_sequence_front_end(tmp_path):
    """SURVEY.md 8f-3 with what this image can decode: animated WebP (VP8L), GIF (LZW), APNG and AVIF sequences (AV1) through
    Pillow -- the only compressed multi-frame decoders present (no FFmpeg / rocDecode / decord / cv2).  The store holds raw
    frame int(sec * fps) for every logical second (interface_searcher.py:360); the rate comes from the frame durations,
    also when they vary (second decode pass at the true rate).  Lossless containers must reproduce the frames exactly."""
    from PIL import Image
    from tstar_amd import video as V
    rs = np.random.RandomState(0)
    frames = [np.kron(rs.randint(0, 256, (9, 16, 3)), np.ones((4, 4, 1))).astype(np.uint8) for _ in range(14)]
    pil = [Image.fromarray(f) for f in frames]
    for ext, kw, exact in ((".webp", dict(lossless=True), True), (".gif", {}, True), (".png", {}, True), (".avif", dict(quality=100, subsampling="4:4:4"), False)):
        p = str(tmp_path / ("clip" + ext))
        pil[0].save(p, save_all=True, append_images=pil[1:], duration=250, loop=0, **kw)
        st = V.open_video(p, device="cpu")
        assert st.raw_fps == 4.0 and st.raw_total_frames == 14 and st.num_seconds == 3 and st.shape == (3, 36, 64, 3), ext
        got = st.frames.numpy()
        for sec in range(3):
            err = np.abs(got[sec].astype(np.int32) - frames[4 * sec].astype(np.int32))
            assert (err.max() == 0) if exact else (err.max() <= 6 and err.mean() < 2.0), (ext, sec, err.max(), err.mean())      # AV1 is lossy even at q = 100
    # varying durations: 8 frames of 125 ms then 6 of 500 ms = 4 s -> 3.5 fps, seconds 0..3 are raw frames 0, 3, 7, 10
    p = str(tmp_path / "vary.webp")
    pil[0].save(p, save_all=True, append_images=pil[1:], duration=[125] * 8 + [500] * 6, loop=0, lossless=True)
    st = V.open_video(p, device="cpu")
    assert st.raw_fps == 3.5 and st.num_seconds == 4
    assert all(np.array_equal(st.frames.numpy()[s], frames[int(s * 3.5)]) for s in range(4))
    # slower than 1 fps: 5 frames of 2000 ms = 10 s at 0.5 fps -> every raw frame serves two logical seconds
    for ext, kw in ((".gif", {}), (".webp", dict(lossless=True))):
        p = str(tmp_path / ("slow" + ext))
        pil[0].save(p, save_all=True, append_images=pil[1:5], duration=2000, loop=0, **kw)
        st = V.load_pillow_sequence(p, device="cpu", chunk=3)          # chunk 3: a frame's two seconds straddle a copy
        assert st.raw_fps == 0.5 and st.num_seconds == 10 and st.raw_total_frames == 5, ext
        assert all(np.array_equal(st.frames.numpy()[s], frames[s // 2]) for s in range(10)), ext
    # a single still image is not a video; a missing file reads like the reference's error
    pil[0].save(str(tmp_path / "still.png"))
    with pytest.raises(ValueError, match="Cannot open video file"):
        V.open_video(str(tmp_path / "still.png"), device="cpu")
    with pytest.raises(ValueError, match="Cannot open video file"):
        V.open_video(str(tmp_path / "missing.webp"), device="cpu")


def _reference_default_fit_problems():
    """The 63 (visited frames, scores) fit problems of a reference-default search (N = 3600, 4x4 grid, 63 iterations): the oracle
    searcher driven by the golden generator's fake detector (the G1 case-0 configuration)."""
    import golden_util as GU
    from oracle import searcher_ref as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "g1_searcher_case0.npz"), allow_pickle=False)
    n, grid, seed, K, np_seed, calls, iters = [int(v) for v in g["meta"]]
    targets, cues = [str(t) for t in g["targets"]], [str(c) for c in g["cues"]]
    h = GU.FakeHeuristic(seed, conf_scale=float(g["conf_scale"]))
    h.reparameterize_object_list(targets, cues)
    o2w = {**{t: 1.0 for t in targets}, **{c: 0.5 for c in cues}}

    def score_fn(kind, secs, rows, cols):
        H, W = (95 * rows, 200 * cols) if kind == "grid" else (285, 600)
        det = h.inference_detector([np.zeros((H, W, 3), np.uint8)])[0]
        return S.image_grid_score(det.xyxy, det.class_id, det.confidence, h.texts, o2w, H, W, rows, cols)

    probs = []
    orig = S.spline_distribution

    def spy(unv, score):
        vis = np.nonzero(unv == 0)[0]
        probs.append((vis.astype(np.float64), score[vis].copy()))
        return orig(unv, score)
    S.spline_distribution = spy
    try:
        S.SearcherRef(n, 1.0, targets, cues, score_fn, np.random.RandomState(np_seed), search_nframes=K, image_grid_shape=(grid, grid),
                      search_budget=float(g["budget"]), confidence_threshold=float(g["thr"])).search()
    finally:
        S.spline_distribution = orig
    return probs


def test_native_curfit_bit_identical_to_scipy(golden_dir):
    """csrc/fitpack.cpp (round 4): FITPACK's curfit restated for the reference's one call, UnivariateSpline(x, y, s=0.5) --
    knots t, coefficients c and the residual fp BIT-IDENTICAL to scipy's on (a) all 63 fits of a reference-default search
    (16 .. 1008 points, up to ~800 knots, 3-11 smoothing-parameter iterations each), (b) golden G4's problems (the reference's
    own spline_keyframe_distribution inputs) incl. P itself, (c) seeded random problems: tiny and ragged sizes, other s,
    least-squares-polynomial (ier -2) and interpolation-limit cases; in the sequential, the AVX2 and the AVX-512 form of the
    skewed rotation pipeline (a form the CPU lacks falls back to the next narrower one)."""
    import warnings
    from scipy.interpolate import UnivariateSpline
    from tstar_amd import spline_worker as SW
    assert SW.curfit(np.arange(8.0), np.arange(8.0) ** 2 % 3, 0.5) is not None, "libtstar_fitpack.so missing: run python -m tstar_amd.build"
    lib = SW._native_fit()
    lib.tstar_curfit_pairing.restype = None

    def check(x, y, s, tag):
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            try:
                sp = UnivariateSpline(x, y, s=s)
            except Warning:
                assert SW.curfit(x, y, s) is None, tag          # FITPACK warning: the native path steps aside, scipy raises it
                return 0
        t, c, k = sp._eval_args
        for lanes, paired in ((1, 1), (4, 1), (8, 1), (4, 0), (8, 0)):     # paired: two batches per steady loop (round 5), same bits
            lib.tstar_curfit_pairing(paired)
            got = SW.curfit(x, y, s, lanes)
            lib.tstar_curfit_pairing(1)
            assert got is not None, (tag, lanes)
            tt, cc, kk, fp, ier = got
            assert kk == 3 and len(tt) == len(t) and np.array_equal(tt, t), (tag, lanes, paired, len(tt), len(t))
            assert np.array_equal(cc[:len(t) - 4], c[:len(t) - 4]) and fp == sp.get_residual(), (tag, lanes, paired)
        return len(t)

    probs = _reference_default_fit_problems()
    assert len(probs) == 63 and len(probs[-1][0]) == 1008
    knots = [check(x, y, 0.5, f"search fit {i}") for i, (x, y) in enumerate(probs)]
    assert max(knots) > 700
    g4 = np.load(os.path.join(golden_dir, "g4_spline.npz"), allow_pickle=False)
    for k in range(4):
        unv, sc, P = g4[f"unv{k}"], g4[f"score{k}"], g4[f"P{k}"]
        vis = np.nonzero(unv == 0)[0]
        check(vis.astype(np.float64), sc[vis], 0.5, f"g4 case {k}")
        got = SW.spline_distribution(vis, sc[vis], len(sc))
        assert np.abs(got - P).max() <= 4 * np.finfo(np.float64).eps * P.max()       # numpy exp's last-place freedom across CPUs (as the G4 test)
        os.environ["TSTAR_NATIVE_FIT"] = "0"
        try:
            SW._FIT = None
            ref = SW.spline_distribution(vis, sc[vis], len(sc))                         # scipy's fit on THIS machine
        finally:
            del os.environ["TSTAR_NATIVE_FIT"]
            SW._FIT = None
        assert np.array_equal(got, ref), k
    rs = np.random.RandomState(404)
    for trial in range(120):
        m = int(rs.choice([4, 5, 6, 7, 8, 9, 12, 16, 17, 18, 31, 33, 48, 100, 257, 400]))
        x = np.sort(rs.choice(3600, size=m, replace=False)).astype(np.float64)
        kind = trial % 4
        y = (rs.random_sample(m) ** 6 * 0.5 if kind == 0 else rs.random_sample(m) * (1e-3 if kind == 1 else 2.0) if kind < 3
             else np.linspace(0.1, 0.4, m) + 1e-6 * rs.standard_normal(m))
        s = float(rs.choice([0.5, 0.5, 0.05, 5.0, 1e-4]))
        check(x, y, s, f"random {trial} m={m} s={s}")


def test_spline_worker_count_follows_the_affinity_mask(monkeypatch):
    """The FITPACK workers of a rank are sized from the hardware threads the process may run on (its affinity mask, capped by the
    cgroup's CPU quota), ALWAYS divided by LOCAL_WORLD_SIZE (round-4 advisor: a narrow cpuset shared by all ranks cannot be told
    from a per-rank one from inside a rank; the conservative split never oversubscribes), unless the launcher declares a per-rank
    mask; the node total stays within half the usable threads."""
    from tstar_amd import spline_pool as SP

    def workers(mask, cpus, ranks, quota=None, per_rank=False):
        monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(mask)))
        monkeypatch.setattr(os, "cpu_count", lambda: cpus)
        monkeypatch.setattr(SP, "cgroup_cpu_quota", lambda: quota)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(ranks))
        monkeypatch.delenv("TSTAR_SPLINE_WORKERS", raising=False)
        if per_rank:
            monkeypatch.setenv("TSTAR_SPLINE_MASK_PER_RANK", "1")
        else:
            monkeypatch.delenv("TSTAR_SPLINE_MASK_PER_RANK", raising=False)
        return SP.default_workers()

    assert workers(256, 256, 1) == 16 and workers(256, 256, 8) == 16        # whole host: 256 / 8 / 2 = 16
    assert workers(32, 256, 8) == 2                                           # a 32-thread cpuset, shared for all we know: 32 / 8 / 2
    assert workers(32, 256, 8, per_rank=True) == 16                           # ... unless the launcher says it is this rank's own
    assert workers(64, 256, 8) == 4                                           # a 64-thread container shared by 8 ranks: 64 / 8 / 2
    assert workers(256, 256, 8, quota=32.0) == 2                              # a CPU quota without a cpuset counts too
    assert workers(8, 8, 2) == 2 and workers(2, 2, 8) == 1
    for mask, cpus, ranks in ((256, 256, 8), (64, 256, 8), (32, 256, 8), (8, 8, 2), (128, 128, 4)):
        assert workers(mask, cpus, ranks) * ranks <= max(ranks, mask // 2)
    monkeypatch.setenv("TSTAR_SPLINE_WORKERS", "3")
    assert SP.default_workers() == 3


def test_lockstep_only_uses_the_second_workspace_when_score_batch_takes_a_lane():
    """A wrapper installed over ``heuristic.score_batch`` with the pre-round-6 signature (a recorder, a logger) must keep working: the
    speculative forward then stays on the detector stream (lockstep._accepts_lane)."""
    from tstar_amd.lockstep import _accepts_lane
    assert _accepts_lane(lambda d, r, c, image_sets=None, lane=0: None)
    assert _accepts_lane(lambda d, r, c, image_sets=None, **kw: None)
    assert not _accepts_lane(lambda d, r, c, image_sets=None: None)
    assert not _accepts_lane(print) or True          # builtins without a signature are simply refused or accepted by **kwargs: no exception
    from tstar_amd.interface_heuristic import OWLInterface, YoloWorldInterface
    assert _accepts_lane(OWLInterface.score_batch) and OWLInterface.aux_lane is True
    assert not _accepts_lane(YoloWorldInterface.score_batch) and not getattr(YoloWorldInterface, "aux_lane", False)


def test_bench_lockstep_group_sizes():
    """bench.lockstep_group_sizes: every item in exactly one group, balanced sizes, a group count that is a multiple of the alternation
    depth whenever there are at least that many items (the driver's --steps 20: 10 + 10, not 7 + 7 | 6)."""
    import bench
    assert bench.lockstep_group_sizes(20, 8, 2) == [10, 10] and bench.lockstep_group_sizes(16, 8, 2) == [8, 8]
    assert bench.lockstep_group_sizes(5, 8, 2) == [3, 2] and bench.lockstep_group_sizes(1, 8, 2) == [1] and bench.lockstep_group_sizes(0, 8, 2) == []
    assert bench.lockstep_group_sizes(25, 8, 2) == [7, 6, 6, 6] and bench.lockstep_group_sizes(5, 4, 1) == [3, 2]
    for n in range(1, 140):
        for L in (1, 4, 8, 24, 31):
            for PL in (1, 2, 3):
                sz = bench.lockstep_group_sizes(n, L, PL)
                assert sum(sz) == n and min(sz) >= 1 and max(sz) - min(sz) <= 1, (n, L, PL, sz)
                assert max(sz) <= max(L + L // 2, 1) or max(sz) <= 31, (n, L, PL, sz)
                if n >= PL and PL > 1 and n > 1:
                    assert len(sz) % PL == 0 or len(sz) == n, (n, L, PL, sz)


def test_bench_traffic_is_only_quoted_for_the_same_launch_population(tmp_path):
    """bench.match_traffic (round 6, review item 2): `roofline.traffic` comes from committed PMC collections; a collection is quoted only for
    a run of the SAME launch population (the flags that shape the launches + its own algorithmic bytes per launch within 5 %), several
    collections of one round are told apart by population, a collection without a recorded population is refused, and an older round's file
    is never used once the newest round has one."""
    import json as J
    import bench
    d = tmp_path / "profiles"
    d.mkdir()
    pop16 = {"heuristic": "owl", "weights": "f32x3", "steps": 16, "lockstep": 8, "pipeline": 2, "max_batch": 512, "grid": 16, "nframes": 3600,
             "search_nframes": 8, "workload_kind": "single", "n_gpus": 1, "concurrency": 1}
    pop20 = dict(pop16, steps=20)
    (d / "r05_pmc_gemm_traffic_f32x3.json").write_text(J.dumps({"bytes_per_launch_corrected": 2.7e9}))                  # round 5: no population
    assert bench.match_traffic(pop16, 1.33e9, "pmc_gemm_traffic_f32x3", str(d))[0] is None
    assert "predates round 6" in bench.match_traffic(pop16, 1.33e9, "pmc_gemm_traffic_f32x3", str(d))[3]
    (d / "r06_pmc_gemm_traffic_f32x3.json").write_text(J.dumps({"bytes_per_launch_corrected": 2.8e9, "population": pop16, "algorithmic_bytes_per_launch": 1.35e9}))
    (d / "r06_pmc_gemm_traffic_f32x3_steps20.json").write_text(J.dumps({"bytes_per_launch_corrected": 3.1e9, "population": pop20, "algorithmic_bytes_per_launch": 1.50e9}))
    t, ratio, src, note = bench.match_traffic(pop16, 1.33e9, "pmc_gemm_traffic_f32x3", str(d))
    assert t == 2.8e9 and abs(ratio - 2.8 / 1.35) < 1e-12 and src.endswith("r06_pmc_gemm_traffic_f32x3.json") and "same launch population" in note
    t, ratio, src, note = bench.match_traffic(pop20, 1.49e9, "pmc_gemm_traffic_f32x3", str(d))
    assert t == 3.1e9 and src.endswith("_steps20.json")
    t, ratio, src, note = bench.match_traffic(dict(pop16, steps=3), 1.2e9, "pmc_gemm_traffic_f32x3", str(d))
    assert t is None and ratio is None and "steps (3 here, 16 there)" in note and "r06_" in src          # never the r05 file
    t, _, _, note = bench.match_traffic(pop16, 2.0e9, "pmc_gemm_traffic_f32x3", str(d))
    assert t is None and "algorithmic bytes per launch differ" in note
    assert bench.match_traffic(pop16, 1.3e9, "yolo_pmc_conv_traffic", str(d)) == (None, None, None, "no PMC collection under profiles/")


def test_bench_self_spawn_command(monkeypatch):
    """`python bench.py --gpus N` typed without a launcher (WORLD_SIZE unset) starts its own ranks: the command is the driver's
    (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1, the user's flags passed through), launcher variables of
    an enclosing job do not leak in, and with fewer visible GPUs than ranks the collectives fall back to gloo."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return subprocess.CompletedProcess(cmd, 0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.setenv("RANK", "7")
    monkeypatch.setenv("MASTER_PORT", "1")
    monkeypatch.delenv("TSTAR_BENCH_BACKEND", raising=False)
    assert bench.self_spawn(4) == 0
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert "RANK" not in env and "MASTER_PORT" not in env and env["MASTER_ADDR"] == "127.0.0.1"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    import torch
    if torch.cuda.device_count() < 4:
        assert env["TSTAR_BENCH_BACKEND"] == "gloo"
