"""CPU: the searcher oracle (oracle/searcher_ref.py) against golden vectors produced by the REFERENCE
itself (tests/golden/make_goldens.py imports /root/reference/TStar/interface_searcher.py unmodified)."""
import os

import numpy as np
import pytest

import golden_util as GU
from oracle import searcher_ref as S


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def oracle_replay(g):
    """Run SearcherRef on a G1 case with the same injected detections."""
    n, grid, seed, K, np_seed, calls, iters = [int(v) for v in g["meta"]]
    targets, cues = [str(t) for t in g["targets"]], [str(c) for c in g["cues"]]
    h = GU.FakeHeuristic(seed, conf_scale=float(g["conf_scale"]))
    h.reparameterize_object_list(targets, cues)
    o2w = {**{t: 1.0 for t in targets}, **{c: 0.5 for c in cues}}

    def score_fn(kind, secs, rows, cols):
        H, W = (95 * rows, 200 * cols) if kind == "grid" else (285, 600)
        det = h.inference_detector([np.zeros((H, W, 3), np.uint8)])[0]
        return S.image_grid_score(det.xyxy, det.class_id, det.confidence, h.texts, o2w, H, W, rows, cols)

    ref = S.SearcherRef(n, 1.0, targets, cues, score_fn, np.random.RandomState(np_seed), search_nframes=K,
                        image_grid_shape=(grid, grid), search_budget=float(g["budget"]),
                        confidence_threshold=float(g["thr"]))
    ts = ref.search()
    return ref, ts, h


@pytest.mark.parametrize("case", range(6))
def test_g1_trajectories_match_reference(golden_dir, case):
    g = _load(golden_dir, f"g1_searcher_case{case}.npz")
    ref, ts, h = oracle_replay(g)
    assert [it["secs"] for it in ref.trace] == g["secs"].tolist()
    assert ts == g["time_stamps"].tolist()
    assert h.calls == int(g["meta"][5])
    assert [GU.sha(x) for x in ref.Score_history] == g["score_sha"].tolist()
    assert [GU.sha(x) for x in ref.unvisited_history] == g["unvisited_sha"].tolist()
    assert [GU.sha(x) for x in ref.P_history] == g["P_sha"].tolist()
    assert np.array_equal(ref.score, g["score_final"])
    assert ref.remaining + ["<end>"] == g["remaining"].tolist()


def test_g2_grid_score(golden_dir):
    g = _load(golden_dir, "g2_grid_score.npz")
    texts = [["couch"], ["tv"], ["chair"], [" "]]
    o2w = {"couch": 1.0, "tv": 0.5, "chair": 0.5}
    for k in range(3):
        H, W, gr, call = [int(v) for v in g[f"shape{k}"]]
        xyxy, cls, conf = GU.fake_detections(0, call, H, W, 4)
        cm, names = S.image_grid_score(xyxy, cls, conf, texts, o2w, H, W, gr, gr)
        assert np.array_equal(cm, g[f"conf{k}"])
        assert ["|".join(n) for n in names] == g[f"names{k}"].tolist()


def test_g3_window_spread(golden_dir):
    g = _load(golden_dir, "g3_window.npz")
    for k in range(4):
        sc = g[f"before{k}"].copy()
        S.window_spread(sc, list(g[f"confs{k}"]), [int(i) for i in g[f"secs{k}"]])
        assert np.array_equal(sc, g[f"after{k}"])


def test_g4_spline_distribution(golden_dir):
    g = _load(golden_dir, "g4_spline.npz")
    for k in range(4):
        P = S.spline_distribution(g[f"unv{k}"], g[f"score{k}"])
        assert np.array_equal(P, g[f"P{k}"])
    assert np.array_equal(S.spline_distribution(np.ones(10), np.zeros(10)), np.ones(10) / 10)


def test_g5_g6_sampler_and_pop(golden_dir):
    g = _load(golden_dir, "g5_g6_sampler.npz")
    fallbacks = 0
    for k in range(4):
        P, unv = g[f"P{k}"], g[f"unv{k}"]
        p, fb = S.sampler_weights(P, unv, 16)
        fallbacks += fb
        secs = S.legacy_choice(np.random.RandomState(int(g[f"seed{k}"])), len(P), 16, p)
        assert secs.tolist() == g[f"secs{k}"].tolist()
        sc = g[f"pop_score{k}"]
        key = S.legacy_choice(np.random.RandomState(int(g[f"pop_seed{k}"])), len(sc), 8, sc / sc.sum())
        key.sort()
        assert [float(v) for v in key] == g[f"pop_ts{k}"].tolist()
    assert fallbacks >= 1          # the fixture set exercises interface_searcher.py:349-351


def test_restatements_equal_numpy():
    m, rs = S.MT19937(2025), np.random.RandomState(2025)
    assert all(m.random_sample() == rs.random_sample() for _ in range(1500))
    for seed in range(40):
        g = np.random.RandomState(seed + 999)
        p = g.random_sample(3600) ** 6
        p[g.random_sample(3600) < 0.6] = 0
        p /= p.sum()
        a = np.random.RandomState(seed).choice(3600, size=16, replace=False, p=p)
        assert np.array_equal(a, S.legacy_choice(np.random.RandomState(seed), 3600, 16, p))
    for n in [1, 2, 3, 4, 5, 16, 17, 100, 3600, 14400]:
        a = np.random.RandomState(n).random_sample(n)
        assert S.percentile75(a) == np.percentile(a, 75)
    with pytest.raises(ValueError, match="Fewer non-zero"):
        S.legacy_choice(np.random.RandomState(0), 4, 3, np.array([1.0, 0, 0, 0]))


def test_g11_topk_selection_matches_reference(golden_dir):
    """oracle.topk_seconds vs the picks of the reference's extract_frames (val_qa_results.py:90-110, imported
    unmodified by make_goldens.g11_topk): equal where no tie is cut, equal values (and lowest-index ties) where one is."""
    n_cut = 0
    for name, dist, k, clip, ref in GU.g11_cases(golden_dir):
        dc, start = S.topk_normalised_clip(dist, clip)
        got = S.topk_seconds(dist, k, clip)
        assert len(got) == min(k, len(dc)), name
        n_cut += GU.check_topk_against_reference(got, ref, dc, start)
    assert n_cut >= 3          # the flat / all-zero / plateau cases do cut through ties


@pytest.mark.parametrize("name", ["g9_end_to_end.npz", "g9b_end_to_end_3600.npz"])
def test_oracle_pipeline_free_running_vs_reference_end_to_end(golden_dir, name):
    """L3 on the CPU (round 4): the WHOLE oracle pipeline -- ingest restatement, OWL-ViT restatement, cell aggregation,
    searcher restatement -- run closed-loop against the reference's own end-to-end runs (reference searcher + reference
    OWLInterface on HF transformers; G9: 160 frames / 4 iterations / 48 detector calls, G9b: the 3600-frame video at the
    reference-default 4x4 grid, K = 8, 3 iterations / 29 calls).  Same sampled seconds in every iteration, every cell and
    verification confidence within 1e-5, same keyframes, same final score distribution: the oracle the GPU tests are
    teacher-forced through follows the reference end to end at the bench's video length."""
    from oracle import cpu_pipeline
    from tstar_amd import weights as W
    from tstar_amd.tokenizer import encode_queries
    from tstar_amd.video import synthetic_frames_numpy
    g = _load(golden_dir, name)
    N, grid, K, np_seed, vseed, ncalls = [int(v) for v in g["meta"]]
    budget = float(g["budget"]) if "budget" in g.files else 0.4
    targets, cues = ["couch"], ["tv", "chair"]
    det = cpu_pipeline.CpuOwlDetector(W.synthetic_state_dict(0), faithful=False)     # cached text tower: the same numbers
    texts = [[t] for t in targets + cues] + [[" "]]
    ids, am = encode_queries(texts)
    det.reparameterize_object_list(targets, cues, ids, am)
    log = []
    score_fn = cpu_pipeline.make_score_fn(det, lambda secs: synthetic_frames_numpy(list(secs), N, seed=vseed),
                                          {"couch": 1.0, "tv": 0.5, "chair": 0.5}, log)
    ref = S.SearcherRef(N, 1.0, targets, cues, score_fn, np.random.RandomState(np_seed), search_nframes=K,
                        image_grid_shape=(grid, grid), search_budget=budget, confidence_threshold=0.6)
    ts = ref.search()
    assert [it["secs"] for it in ref.trace] == g["secs"].tolist()
    assert len(log) == ncalls
    gc = np.stack([c["conf"] for c in log if c["kind"] == "grid"])
    vc = np.array([c["conf"][0, 0] for c in log if c["kind"] == "verify"])
    assert gc.shape == g["grid_conf"].shape and np.abs(gc - g["grid_conf"]).max() < 1e-5
    assert vc.shape == g["verify_conf"].shape and (vc.size == 0 or np.abs(vc - g["verify_conf"]).max() < 1e-5)
    assert ts == g["time_stamps"].tolist()
    assert np.abs(ref.score - g["score_final"]).max() < 1e-5
