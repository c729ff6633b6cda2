"""The drop-in surface exercised the way the reference's CALLERS use it (tests/caller_standin.py states which
reference lines each call shape comes from), plus the public searcher methods the reference exposes, each against
the golden vectors the reference itself produced (tests/golden/make_goldens.py)."""
import json
import os

import numpy as np
import pytest

import caller_standin as CS
import golden_util as GU

pytestmark = pytest.mark.gpu


def _ulp_close(a, b, ulps=4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


@pytest.fixture(scope="module")
def heuristic():
    from tstar_amd.interface_heuristic import initialize_heuristic
    return initialize_heuristic("owl-vit", synthetic_seed=0, max_batch=32)       # shared by every item, as :189-198


def test_dataset_loop_call_shapes(heuristic, tmp_path):
    """Two (video, question) items through the stand-in TStarFramework / dataset loop: keyword constructor, default
    visual history, global numpy RNG, in-place timestamp sort, P_history[-1], JSON dump -- and the same keyframes as
    a searcher driven with an explicit RandomState of the same seed."""
    from tstar_amd.interface_searcher import TStarSearcher
    args = dict(search_nframes=8, grid_rows=4, grid_cols=4, output_dir=str(tmp_path / "out"), confidence_threshold=0.6,
                search_budget=0.15)
    items = [{"video_path": "synthetic://n=420,seed=31", "targets": ["couch"], "cues": ["tv", "chair"]},
             {"video_path": "synthetic://n=300,seed=32", "targets": ["dog", "ball"], "cues": ["leash"]}]
    np.random.seed(2025)                                                  # the reference's only seed (val_qa_results.py:319)
    results = []
    for it in items:
        result, fw, searcher, frames = CS.run_item(TStarSearcher, heuristic, it, args)
        it2 = dict(it)
        it2.update(result)
        results.append(it2)
        n_iter = searcher.iterations
        assert n_iter == -(-int(min(1000, searcher.total_frame_num * 0.15)) // 16)
        assert fw.saved["iterations"] == len(searcher.image_grid_iters) >= n_iter          # + one entry per verification
        assert len(fw.saved["frames"]) == 8 and os.path.isfile(fw.saved["plot"])
        assert frames.shape[1:] == (360, 640, 3)
        assert isinstance(result["keyframe_timestamps"], list) and result["keyframe_timestamps"] == sorted(result["keyframe_timestamps"])
        assert len(result["keyframe_distribution"]) == searcher.total_frame_num
        assert searcher.image_grid_iters[0][0].shape == (380, 800, 3)
        assert len(searcher.detect_bbox_iters) == len(searcher.image_grid_iters)
        assert heuristic.texts[-1] == [" "] and [t[0] for t in heuristic.texts[:-1]] == it["targets"] + it["cues"]
    out = tmp_path / "results.json"
    CS.dump_results(results, str(out))
    back = json.load(open(out))
    assert [r["keyframe_timestamps"] for r in back] == [[float(t) for t in r["keyframe_timestamps"]] for r in results]
    assert set(back[0]) >= {"video_path", "grounding_objects", "keyframe_timestamps", "keyframe_distribution"}
    # item 0 again with an explicit generator of the same seed (history off): same keyframes
    s = TStarSearcher(video_path=items[0]["video_path"], target_objects=["couch"], cue_objects=["tv", "chair"], search_nframes=8,
                      image_grid_shape=(4, 4), output_dir=None, confidence_threshold=0.6, search_budget=0.15,
                      heuristic=heuristic, rng=np.random.RandomState(2025), keep_visual_history=False)
    _, ts = s.search()
    assert [float(t) for t in ts] == back[0]["keyframe_timestamps"]


def test_environment_selects_the_mode_under_an_unchanged_framework(monkeypatch, tmp_path):
    """Round 6 (review item 1a): an UNCHANGED ``TStarFramework`` builds the heuristic with ``initialize_heuristic(heuristic_type)`` and
    no keyword arguments (TStarFramework.py:171-187, 207), so the deployment's choices reach ``OWLInterface`` through the environment:
    ``TSTAR_WEIGHTS_DTYPE`` / ``TSTAR_MAX_BATCH`` / ``TSTAR_SYNTHETIC_SEED``.  The default-constructed heuristic then runs the bench's
    headline arithmetic (f32x3) through the stand-in framework's call shapes -- same keyframes as a heuristic built with the keywords;
    a keyword still wins over the environment."""
    from tstar_amd.interface_heuristic import initialize_heuristic
    from tstar_amd.interface_searcher import TStarSearcher
    for k in ("TSTAR_WEIGHTS_DTYPE", "TSTAR_MAX_BATCH", "TSTAR_SYNTHETIC_SEED"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(FileNotFoundError):
        initialize_heuristic("owl-vit")                                    # offline, no checkpoint, no seed: as before
    monkeypatch.setenv("TSTAR_WEIGHTS_DTYPE", "f32x3")
    monkeypatch.setenv("TSTAR_MAX_BATCH", "16")
    monkeypatch.setenv("TSTAR_SYNTHETIC_SEED", "0")
    h_env = initialize_heuristic("owl-vit")                                # the reference's call, no keywords
    assert h_env.weights_dtype == "f32x3" and h_env.scorer.max_batch == 16 and h_env.weights_source == "synthetic(seed=0)"
    h_kw = initialize_heuristic("owl-vit", weights_dtype="f32", max_batch=8)   # keywords win; the seed still comes from the environment
    assert h_kw.weights_dtype == "f32" and h_kw.scorer.max_batch == 8
    del h_kw
    args = dict(search_nframes=8, grid_rows=4, grid_cols=4, output_dir=str(tmp_path / "out"), confidence_threshold=0.6, search_budget=0.15)
    item = {"video_path": "synthetic://n=420,seed=31", "targets": ["couch"], "cues": ["tv", "chair"]}
    np.random.seed(2025)
    res_env, _, s_env, _ = CS.run_item(TStarSearcher, h_env, item, args)
    for k in ("TSTAR_WEIGHTS_DTYPE", "TSTAR_MAX_BATCH", "TSTAR_SYNTHETIC_SEED"):
        monkeypatch.delenv(k, raising=False)
    h_ref = initialize_heuristic("owl-vit", synthetic_seed=0, max_batch=16, weights_dtype="f32x3")
    np.random.seed(2025)
    res_ref, _, s_ref, _ = CS.run_item(TStarSearcher, h_ref, item, args)
    assert res_env["keyframe_timestamps"] == res_ref["keyframe_timestamps"]
    assert s_env.Score_history == s_ref.Score_history
    monkeypatch.setenv("TSTAR_WEIGHTS_DTYPE", "fp8")
    with pytest.raises(ValueError, match="TSTAR_WEIGHTS_DTYPE"):
        initialize_heuristic("owl-vit", synthetic_seed=0)


def test_update_top_25_with_window_vs_reference_g3(heuristic, golden_dir):
    """The public ``update_top_25_with_window`` (interface_searcher.py:215-241) on the device score array vs the
    reference's own before/after vectors (chained centres, array ends, a 12-frame video)."""
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import synthetic_video
    g = np.load(os.path.join(golden_dir, "g3_window.npz"), allow_pickle=False)
    for k in range(4):
        before, after = g[f"before{k}"], g[f"after{k}"]
        s = TStarSearcher(synthetic_video(len(before), seed=1), heuristic, ["couch"], [], image_grid_shape=(2, 2))
        s.score_distribution = before                                    # attribute assignment, as a reference caller may
        assert np.array_equal(s.score_distribution, before)
        s.update_top_25_with_window([float(c) for c in g[f"confs{k}"]], [int(i) for i in g[f"secs{k}"]])
        assert np.array_equal(s.score_distribution, after), k
        s.update_top_25_with_window(frame_confidences=[0.5], sampled_frame_indices=[0], window_size=2)   # keywords as :215-220
        exp = after.copy()
        for off in range(0, 3):
            if off < len(exp):
                exp[off] = max(exp[off], exp[0] / (off + 1))
        assert np.array_equal(s.score_distribution, exp)


def test_spline_keyframe_distribution_vs_reference_g4(heuristic, golden_dir):
    """The public ``spline_keyframe_distribution`` (:243-274) vs the reference's vectors: the host path (the reference's
    own numpy / scipy calls) within the last-place freedom of numpy's exp across CPUs -- and bit-equal to the oracle on
    THIS machine; the device evaluation (tstar_searcher_set_spline: FITPACK splev restated + device exp) within 4 ulp."""
    from oracle import searcher_ref as S
    from scipy.interpolate import UnivariateSpline
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import synthetic_video
    g = np.load(os.path.join(golden_dir, "g4_spline.npz"), allow_pickle=False)
    for k in range(4):
        unv, sc, P_ref = g[f"unv{k}"], g[f"score{k}"], g[f"P{k}"]
        N = len(unv)
        s = TStarSearcher(synthetic_video(min(N, 64), seed=1), heuristic, ["couch"], [], image_grid_shape=(2, 2))
        P = s.spline_keyframe_distribution(unv, sc, N)
        assert np.array_equal(P, S.spline_distribution(unv, sc))
        assert _ulp_close(P, P_ref, 2)
    # no visited frame -> uniform (:262-263)
    assert np.array_equal(s.spline_keyframe_distribution(np.ones(50), np.zeros(50), 50), np.ones(50) / 50)
    # device evaluation of the same fit
    unv, sc, P_ref = g["unv1"], g["score1"], g["P1"]
    N = len(unv)
    s = TStarSearcher(synthetic_video(N, seed=1), heuristic, ["couch"], [], image_grid_shape=(2, 2))
    s.non_visiting_frames, s.score_distribution = unv, sc
    vx, vy = s._state.visited()
    assert np.array_equal(vx, np.nonzero(unv == 0)[0]) and np.array_equal(vy, sc[unv == 0])
    t, c, kk = UnivariateSpline(vx, vy, s=0.5)._eval_args
    s._state.set_spline(t, c, kk)
    assert _ulp_close(s.P, P_ref, 4)


def test_sampler_and_pop_call_shapes_vs_reference_g5_g6(heuristic, golden_dir):
    """The golden generator's own call sequence on the reference class -- assign ``P`` / ``non_visiting_frames`` /
    ``score_distribution``, seed the GLOBAL numpy generator, ``secs, _ = sample_frames(16)``, ``_, ts = pop_frames(path, 8)``
    -- replayed on the drop-in: identical sampled seconds (incl. the fallback branch) and keyframe timestamps."""
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import synthetic_video
    g = np.load(os.path.join(golden_dir, "g5_g6_sampler.npz"), allow_pickle=False)
    for k in range(4):
        P, unv = g[f"P{k}"], g[f"unv{k}"]
        N = len(P)
        store = synthetic_video(N, seed=20 + k)
        sk = TStarSearcher(video_path=store, heuristic=heuristic, target_objects=["couch"], cue_objects=[], search_nframes=8,
                           image_grid_shape=(4, 4), search_budget=1000, confidence_threshold=0.6)
        sk.Score_history = [[0.0]]                       # not the first iteration
        sk.P, sk.non_visiting_frames = P.copy(), unv.copy()
        np.random.seed(int(g[f"seed{k}"]))
        secs, frames = sk.sample_frames(16)
        assert list(secs) == g[f"secs{k}"].tolist()
        assert len(frames) == 16
        sk.score_distribution = g[f"pop_score{k}"]
        np.random.seed(int(g[f"pop_seed{k}"]))
        kf, ts = sk.pop_frames(store, 8)
        assert [float(t) for t in ts] == g[f"pop_ts{k}"].tolist()
        assert kf.shape == (8, 360, 640, 3)
    # the resized frames sample_frames hands back are the oracle's 800x380 bilinear of the native frames
    from oracle import resize_ref as R
    from tstar_amd.video import synthetic_frames_numpy
    f0 = frames[0]
    assert f0.shape == (380, 800, 3)
    assert np.array_equal(f0, R.cv_bilinear_resize(synthetic_frames_numpy([secs[0]], N, seed=23)[0], 800, 380))


def test_pop_frames_raises_what_numpy_raises(heuristic):
    """Short video, every frame visited, most cells empty (the reference runner's default search_budget=1.0): fewer than K
    non-zero scores.  np.random.choice raises in the reference and the item is skipped (run_TStar_onDataset.py:200-202);
    the drop-in must raise the same exception, not return duplicate frame-0 keyframes."""
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import synthetic_video
    s = TStarSearcher(synthetic_video(24, seed=2), heuristic, ["couch"], [], search_nframes=8, image_grid_shape=(2, 2),
                      rng=np.random.RandomState(0))
    sc = np.zeros(24)
    sc[[3, 9, 20]] = [0.2, 0.1, 0.4]
    s.score_distribution = sc
    with pytest.raises(ValueError) as e1:
        s.pop_frames(None, 8)
    with pytest.raises(ValueError) as e2:
        np.random.RandomState(0).choice(24, size=8, replace=False, p=sc / sc.sum())
    assert str(e1.value) == str(e2.value) == "Fewer non-zero entries in p than size"
    _, ts = s.pop_frames(None, 3)                                         # exactly as many non-zeros as samples: fine
    assert [float(t) for t in ts] == [3.0, 9.0, 20.0]
    s.score_distribution = np.zeros(24)
    with pytest.raises(ValueError) as e3:
        s.pop_frames(None, 8)
    with np.errstate(invalid="ignore"), pytest.raises(ValueError) as e4:
        np.random.RandomState(0).choice(24, size=8, replace=False, p=np.zeros(24) / np.zeros(24).sum())
    assert str(e3.value) == str(e4.value)
    with pytest.raises(ValueError, match="Cannot take a larger sample than population"):
        s.score_distribution = np.ones(24)
        s.pop_frames(None, 25)


def test_real_checkpoint_without_tokenizer_refuses_standin_ids():
    """With real weights (here: a state dict handed in) and no CLIP vocab on disk the queries must NOT be encoded with
    the hash stand-in ids (they would feed garbage to the real text tower); synthetic weights may use it."""
    from tstar_amd import tokenizer, weights as W
    from tstar_amd.interface_heuristic import OWLInterface
    if tokenizer._hf_tokenizer("google/owlvit-base-patch32") is not None:
        pytest.skip("a real CLIP tokenizer is available on this machine")
    h = OWLInterface(state_dict=W.synthetic_state_dict(1), max_batch=1)
    assert h.allow_standin_tokenizer is False
    with pytest.raises(RuntimeError, match="stand-in tokenizer is disabled"):
        h.reparameterize_object_list(["couch"], ["tv"])
    h2 = OWLInterface(synthetic_seed=1, max_batch=1)
    assert h2.allow_standin_tokenizer is True
    h2.reparameterize_object_list(["couch"], ["tv"])
