"""Generate tests/golden/*.npz by running the REFERENCE itself (this container only).

/root/reference is imported unmodified; its absent third-party imports (cv2, decord, supervision,
mmengine, mmdet) are replaced by stub modules whose semantics are stated here:
  cv2.resize            = oracle.resize_ref.cv_bilinear_resize   (the build's own bilinear)
  cv2.VideoCapture      = fps / frame count of a procedural video registry
  decord.VideoReader    = frames of the procedural video (tstar_amd.video.synthetic_frames_numpy)
  supervision.Detections.from_transformers = plain container of boxes / scores / labels
No reference source is copied: only inputs and outputs are written.

    python tests/golden/make_goldens.py            (takes about 7 minutes on 8 cores)
    python tests/golden/make_goldens.py --only-g10
    python tests/golden/make_goldens.py --only-g11
    python tests/golden/make_goldens.py --only-g9b          (about one minute)
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import resize_ref  # noqa: E402
from tstar_amd.video import synthetic_frames_numpy  # noqa: E402
import golden_util as GU  # noqa: E402

VIDEOS = {}   # path -> dict(n=, fps=, seed=, h=, w=)
SEEKS = []    # frame indices the code under test asked cv2.VideoCapture for (G11)


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_COUNT, cv2.CAP_PROP_POS_FRAMES = 5, 7, 1
    cv2.COLOR_RGB2BGR = cv2.COLOR_BGR2RGB = 4

    class VideoCapture:
        def __init__(self, path):
            self.v = VIDEOS.get(path)

        def isOpened(self):
            return self.v is not None

        def get(self, prop):
            return float(self.v["fps"]) if prop == cv2.CAP_PROP_FPS else float(self.v["raw_total"])

        def release(self):
            pass

        def set(self, prop, value):                    # extract_frames seeks by frame index (val_qa_results.py:121)
            self.pos = int(value)
            SEEKS.append(int(value))

        def read(self):
            v = self.v
            return True, synthetic_frames_numpy([min(self.pos, v["n"] - 1)], v["n"], v["h"], v["w"], v["seed"])[0][:, :, ::-1]

    cv2.VideoCapture = VideoCapture
    cv2.resize = lambda img, size: resize_ref.cv_bilinear_resize(np.asarray(img), size[0], size[1])
    cv2.imwrite = lambda *a, **k: True
    cv2.cvtColor = lambda img, code: img[:, :, ::-1]
    sys.modules["cv2"] = cv2

    decord = types.ModuleType("decord")

    class _Batch:
        def __init__(self, a):
            self.a = a

        def asnumpy(self):
            return self.a

    class VideoReader:
        def __init__(self, path, ctx=None):
            self.v = VIDEOS[path]

        def get_batch(self, idx):
            v = self.v
            secs = [int(round(float(i) / v["fps"])) for i in idx]   # raw index -> logical second (fps = raw/1)
            return _Batch(synthetic_frames_numpy(secs, v["n"], v["h"], v["w"], v["seed"]))

    decord.VideoReader = VideoReader
    decord.cpu = lambda i=0: None
    sys.modules["decord"] = decord

    sv = types.ModuleType("supervision")

    class Detections:
        def __init__(self, xyxy, confidence, class_id, mask=None):
            self.xyxy, self.confidence, self.class_id, self.mask = xyxy, confidence, class_id, mask

        @classmethod
        def from_transformers(cls, transformers_results):
            r = transformers_results
            return cls(r["boxes"].cpu().numpy(), r["scores"].cpu().numpy(), r["labels"].cpu().numpy().astype(int))

        def __len__(self):
            return len(self.xyxy)

    class _Ann:
        def __init__(self, *a, **k):
            pass

        def annotate(self, image, detections, labels=None):
            return image

    sv.Detections = Detections
    sv.BoxAnnotator = sv.LabelAnnotator = sv.BoundingBoxAnnotator = _Ann
    draw = types.ModuleType("supervision.draw")
    color = types.ModuleType("supervision.draw.color")
    color.ColorPalette = types.SimpleNamespace(LEGACY=None)
    draw.color = color
    sv.draw = draw
    sys.modules.update({"supervision": sv, "supervision.draw": draw, "supervision.draw.color": color})

    for name, attrs in {"mmengine": {}, "mmengine.config": {"Config": object}, "mmengine.dataset": {"Compose": object},
                        "mmdet": {}, "mmdet.apis": {"init_detector": lambda *a, **k: None}}.items():
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    sys.path.insert(0, REF)


def register_video(n, seed, fps=1.0, h=360, w=640):
    path = f"/virtual/video_n{n}_s{seed}_f{fps}.mp4"
    VIDEOS[path] = dict(n=n, fps=fps, seed=seed, h=h, w=w, raw_total=int(round(n * fps)))
    return path


def g1_searcher(TStarSearcher):
    """Trajectories of the reference searcher with injected detections."""
    # cs = confidence scale of the fake detector: 0.5 never confirms a target at thr 0.6 (full budget),
    # 0.66 confirms late, 0.9 confirms in the first iterations
    cases = [dict(n=3600, g=4, seed=0, K=8, thr=0.6, budget=1000, cs=0.5, targets=["couch"], cues=["tv", "chair"]),
             dict(n=3600, g=8, seed=1, K=8, thr=0.6, budget=1000, cs=0.66, targets=["couch", "lamp"], cues=["tv"]),
             dict(n=100, g=4, seed=2, K=4, thr=0.5, budget=0.6, cs=0.45, targets=["a"], cues=[]),
             dict(n=14400, g=15, seed=3, K=32, thr=0.6, budget=1000, cs=0.5, targets=["couch"], cues=["tv", "chair"]),
             dict(n=3600, g=16, seed=4, K=8, thr=0.6, budget=1000, cs=0.5, targets=["couch"], cues=["tv", "chair"]),
             dict(n=777, g=4, seed=5, K=8, thr=0.3, budget=0.2, cs=0.9, targets=["couch", "dog"], cues=["tv"])]
    for ci, c in enumerate(cases):
        path = register_video(c["n"], 100 + ci)
        h = GU.FakeHeuristic(c["seed"], conf_scale=c["cs"])
        np.random.seed(2025 + ci)
        s = TStarSearcher(video_path=path, heuristic=h, target_objects=list(c["targets"]), cue_objects=list(c["cues"]),
                          search_nframes=c["K"], image_grid_shape=(c["g"], c["g"]), search_budget=c["budget"],
                          confidence_threshold=c["thr"])
        secs_log = []
        orig = s.sample_frames
        s.sample_frames = lambda num, _o=orig: (lambda r: (secs_log.append(list(r[0])), r)[1])(_o(num))
        frames, ts = s.search()
        np.savez_compressed(
            os.path.join(OUT, f"g1_searcher_case{ci}.npz"),
            meta=np.array([c["n"], c["g"], c["seed"], c["K"], 2025 + ci, h.calls, len(secs_log)], dtype=np.int64),
            thr=np.float64(c["thr"]), budget=np.float64(c["budget"]), video_seed=np.int64(100 + ci), conf_scale=np.float64(c["cs"]),
            targets=np.array(c["targets"]), cues=np.array(c["cues"] if c["cues"] else [""])[: len(c["cues"])],
            secs=np.array(secs_log, dtype=np.int64), time_stamps=np.array(ts, dtype=np.float64),
            frames_sha=np.array([GU.sha(np.asarray(frames))]),
            score_sha=np.array([GU.sha(np.asarray(x)) for x in s.Score_history]),
            unvisited_sha=np.array([GU.sha(np.asarray(x)) for x in s.non_visiting_history]),
            P_sha=np.array([GU.sha(np.asarray(x)) for x in s.P_history]),
            P_last=np.asarray(s.P_history[-1]), score_last=np.asarray(s.Score_history[-1]),
            score_final=np.asarray(s.score_distribution), remaining=np.array(s.remaining_targets + ["<end>"]),
            call_shapes=np.array(h.log, dtype=np.int64))
        print("g1 case", ci, "iters", len(secs_log), "calls", h.calls, "ts", ts[:4])


def g2_to_g6(TStarSearcher):
    path = register_video(3600, 7)
    h = GU.FakeHeuristic(0)
    s = TStarSearcher(video_path=path, heuristic=h, target_objects=["couch"], cue_objects=["tv", "chair"],
                      search_nframes=8, image_grid_shape=(4, 4), search_budget=1000, confidence_threshold=0.6)
    # G2 imageGridScoreFunction on 3 shapes / grids
    g2 = {}
    for k, (H, W, g) in enumerate([(380, 800, 4), (285, 600, 1), (1520, 3200, 16)]):
        h.calls = 50 + k
        img = np.zeros((H, W, 3), dtype=np.uint8)
        cm, names = s.imageGridScoreFunction([img], None, (g, g))
        xyxy, cls, conf = GU.fake_detections(0, 50 + k, H, W, 4)
        g2[f"conf{k}"] = cm[0]
        g2[f"names{k}"] = np.array(["|".join(n) for n in names[0]])
        g2[f"shape{k}"] = np.array([H, W, g, 50 + k])
    np.savez_compressed(os.path.join(OUT, "g2_grid_score.npz"), **g2)
    # G3 window spread
    rs = np.random.RandomState(11)
    g3 = {}
    for k in range(4):
        N = [3600, 40, 3600, 12][k]
        s.score_distribution = rs.random_sample(N) ** 4 * 0.3 + 1e-6
        secs = rs.choice(N, size=min(16, N), replace=False)
        if k == 2:
            secs[:6] = [0, 3, 5, N - 1, N - 4, 9]          # chained centres within +-5 and array ends
        confs = [float(s.score_distribution[i]) for i in secs]
        g3[f"before{k}"] = s.score_distribution.copy()
        g3[f"secs{k}"] = np.asarray(secs)
        g3[f"confs{k}"] = np.asarray(confs)
        s.update_top_25_with_window(confs, [int(i) for i in secs])
        g3[f"after{k}"] = s.score_distribution.copy()
    np.savez_compressed(os.path.join(OUT, "g3_window.npz"), **g3)
    # G4 spline distribution (incl. the noisy retry path and extrapolated ends)
    g4 = {}
    for k in range(4):
        N = [3600, 3600, 200, 14400][k]
        unv = np.ones(N)
        nvis = [16, 600, 30, 1000][k]
        vis = np.sort(rs.choice(np.arange(N // 10, N - N // 10), size=nvis, replace=False))
        unv[vis] = 0
        sc = np.zeros(N) + 1e-6
        sc[vis] = (rs.random_sample(nvis) ** 6 * 0.5) if k != 2 else np.linspace(0.1, 0.4, nvis)
        P = s.spline_keyframe_distribution(unv, sc, N)
        g4[f"unv{k}"], g4[f"score{k}"], g4[f"P{k}"] = unv, sc, P
    np.savez_compressed(os.path.join(OUT, "g4_spline.npz"), **g4)
    # G5 sampler (both branches) and G6 pop_frames
    g5 = {}
    for k in range(4):
        N = [3600, 3600, 64, 14400][k]
        path_k = register_video(N, 20 + k)
        sk = TStarSearcher(video_path=path_k, heuristic=h, target_objects=["couch"], cue_objects=[],
                           search_nframes=8, image_grid_shape=(4, 4), search_budget=1000, confidence_threshold=0.6)
        sk.Score_history = [[0.0]]                       # not the first iteration
        P = rs.random_sample(N)
        P = 1 / (1 + np.exp(-np.maximum(1 / N, P * 0.3)))
        P /= P.sum()
        unv = (rs.random_sample(N) > (0.95 if k == 2 else 0.2)).astype(np.float64)
        sk.P, sk.non_visiting_frames = P.copy(), unv.copy()
        np.random.seed(77 + k)
        secs, _ = sk.sample_frames(16)
        g5[f"P{k}"], g5[f"unv{k}"], g5[f"secs{k}"], g5[f"seed{k}"] = P, unv, np.asarray(secs), np.int64(77 + k)
        sk.score_distribution = rs.random_sample(N) ** 8 + 1e-6
        np.random.seed(177 + k)
        _, ts = sk.pop_frames(path_k, 8)
        g5[f"pop_score{k}"], g5[f"pop_ts{k}"], g5[f"pop_seed{k}"] = sk.score_distribution.copy(), np.asarray(ts), np.int64(177 + k)
    np.savez_compressed(os.path.join(OUT, "g5_g6_sampler.npz"), **g5)
    print("g2..g6 done")


def build_reference_owl():
    """The reference's OWLInterface (imported unmodified) over HF's OwlViTForObjectDetection with the seeded synthetic weights."""
    import torch
    from transformers import OwlViTConfig, OwlViTForObjectDetection
    from transformers.models.owlvit.image_processing_pil_owlvit import OwlViTImageProcessorPil
    import TStar.interface_heuristic as RH
    from tstar_amd import weights as W
    from tstar_amd.tokenizer import encode_queries
    sd = W.synthetic_state_dict(0)
    cfg = OwlViTConfig()
    cfg._attn_implementation = "eager"
    model = OwlViTForObjectDetection(cfg).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k != "box_bias"}, strict=False)
    ip = OwlViTImageProcessorPil()

    class Proc:
        """Tokenizer-free processor: stand-in ids (no CLIP vocab offline) + the real HF image processor."""
        def __call__(self, text=None, images=None, return_tensors="pt"):
            ids, am = encode_queries(text)
            pv = ip(images=images, return_tensors="pt")["pixel_values"]

            class BF(dict):
                def to(self, dev):
                    return self
            return BF(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(am.astype(np.int64)),
                      pixel_values=pv)

        def post_process_grounded_object_detection(self, outputs, target_sizes, threshold):
            return ip.post_process_object_detection(outputs, threshold=threshold, target_sizes=target_sizes)

    class RefOWL(RH.OWLInterface):
        def load_model_and_tokenizer(self, name):
            return Proc(), model

        def forward_model(self, inputs):
            out = super().forward_model(inputs)
            self.last_outputs = out
            return out

    owl = RefOWL("google/owlvit-base-patch32", device="cpu")
    owl.reparameterize_object_list(["couch"], ["tv", "chair"])
    return owl, ip, encode_queries


def g7_g8_g9(TStarSearcher):
    owl, ip, encode_queries = build_reference_owl()
    # G8 preprocess + G7 detector on two image shapes
    rs = np.random.RandomState(5)
    g7 = {}
    for k, (H, Wd) in enumerate([(380, 800), (285, 600)]):
        img = GU.detector_test_image(40 + k, H, Wd)
        det = owl.inference_detector([img])[0]
        o = owl.last_outputs
        pv = ip(images=img, return_tensors="np")["pixel_values"][0]
        g7[f"img_sha{k}"] = np.array([GU.sha(img)])
        g7[f"img_meta{k}"] = np.array([40 + k, H, Wd])
        g7[f"logits{k}"] = o.logits[0].numpy()
        g7[f"boxes{k}"] = o.pred_boxes[0].numpy()
        g7[f"det_xyxy{k}"], g7[f"det_conf{k}"], g7[f"det_cls{k}"] = det.xyxy, det.confidence, det.class_id.astype(np.int64)
        g7[f"pixel_sha{k}"] = np.array([GU.sha(pv)])
        g7[f"pixel_crop{k}"] = pv[:, 100:104, 200:204]
        g7[f"query_embeds{k}"] = o.text_embeds[0].numpy()
    ids, am = encode_queries(owl.texts)
    g7["ids"], g7["mask"] = ids, am
    np.savez_compressed(os.path.join(OUT, "g7_g8_detector.npz"), **g7)
    print("g7/g8 done")
    # G9 end-to-end: reference searcher + reference OWLInterface on the procedural video
    g9_run(TStarSearcher, owl, 160, 4, 4, 0.4, 5, 2025, "g9_end_to_end.npz")


def g9_run(TStarSearcher, owl, N, g, K, budget, vseed, np_seed, out_name):
    """One end-to-end run of the reference searcher driving the reference OWLInterface (HF transformers on the CPU)."""
    path = register_video(N, vseed)
    conf_log = []
    orig = TStarSearcher.imageGridScoreFunction

    def logged(self, images, output_dir, image_grids):
        cm, names = orig(self, images, output_dir, image_grids)
        conf_log.append((tuple(image_grids), cm[0].copy(), [list(n) for n in names[0]]))
        return cm, names

    TStarSearcher.imageGridScoreFunction = logged
    np.random.seed(np_seed)
    s = TStarSearcher(video_path=path, heuristic=owl, target_objects=["couch"], cue_objects=["tv", "chair"], search_nframes=K,
                      image_grid_shape=(g, g), search_budget=budget, confidence_threshold=0.6)
    secs_log = []
    orig_sf = s.sample_frames
    s.sample_frames = lambda num, _o=orig_sf: (lambda r: (secs_log.append(list(r[0])), r)[1])(_o(num))
    frames, ts = s.search()
    TStarSearcher.imageGridScoreFunction = orig
    grid_conf = np.stack([c for (gr, c, n) in conf_log if gr == (g, g)])
    ver_conf = np.array([c[0, 0] for (gr, c, n) in conf_log if gr == (1, 1)])
    np.savez_compressed(os.path.join(OUT, out_name), meta=np.array([N, g, K, np_seed, vseed, len(conf_log)], dtype=np.int64),
                        budget=np.float64(budget),
                        secs=np.array(secs_log, dtype=np.int64), time_stamps=np.asarray(ts, dtype=np.float64),
                        grid_conf=grid_conf, verify_conf=ver_conf,
                        grid_names=np.array(["|".join(sorted(set(x))) for (gr, c, n) in conf_log if gr == (g, g) for x in n]),
                        frames_sha=np.array([GU.sha(np.asarray(frames))]), score_final=np.asarray(s.score_distribution),
                        P_last=np.asarray(s.P_history[-1]))
    print(out_name, "done: calls", len(conf_log), "iterations", len(secs_log), "ts", ts)


def g9b_hour_video(TStarSearcher):
    """G9b (round 4, SURVEY 8c's 3600-frame end-to-end probe): the same reference pair on the 3600-frame procedural video at the
    reference's default 4x4 grid and K = 8, the budget cut to three iterations (0.0133 * 3600 = 47.9 -> 31.9 -> 15.9 -> < 0)."""
    owl, _, _ = build_reference_owl()
    g9_run(TStarSearcher, owl, 3600, 4, 8, 0.0133, 9, 4242, "g9b_end_to_end_3600.npz")


def g10_metrics():
    """Search-quality metrics of LVHaystackBench/val_tstar_results.py (imported unmodified)."""
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.metrics")
    skm.structural_similarity = lambda *a, **k: None            # imported by the reference, never called
    sk.metrics = skm
    sys.modules.update({"skimage": sk, "skimage.metrics": skm})
    sys.path.insert(0, os.path.join(REF, "LVHaystackBench"))
    import val_tstar_results as V
    gt_idx, pred_idx = [3, 10, 17], [3, 11, 40, 41]
    frames_gt = list(synthetic_frames_numpy(gt_idx, 64, 72, 128, seed=8))
    frames_pr = list(synthetic_frames_numpy(pred_idx, 64, 72, 128, seed=8))
    m = V.pairwise_ssim(frames_gt, frames_pr)
    scores = V.calculate_ssim_scores([frames_gt], [frames_pr])
    lg = [np.array([3.0, 10.0, 17.0, 100.0]), np.array([]), np.array([5.0, 50.0])]
    lp = [np.array([2.0, 16.5, 60.0]), np.array([1.0]), np.array([44.0, 45.0, 46.0, 200.0])]
    prf = V.calculate_prf(lg, lp, threshold=5)
    annd = V.calculate_annd(lg, lp)
    np.savez_compressed(os.path.join(OUT, "g10_metrics.npz"), gt_idx=np.array(gt_idx), pred_idx=np.array(pred_idx),
                        video=np.array([64, 72, 128, 8]), ssim=m, ssim_scores=np.array(scores, dtype=np.float64),
                        prf=np.array(prf, dtype=np.float64), annd=np.array(annd, dtype=np.float64))
    print("g10 done", m.round(4).tolist(), prf)


def g11_topk():
    """Downstream selection of LVHaystackBench/val_qa_results.py (imported unmodified; extract_frames :47-131): the
    seconds it seeks to for a saved keyframe_distribution -- flat, tied, clipped, zero and NaN cases.  The video is
    1 fps, so frame index = second.  Where the normalised float32 scores tie, WHICH of the equal seconds numpy's
    default argsort picks depends on the numpy build / CPU: the golden records the reference's picks on this machine
    and the normalised value of every pick; tests compare picks where no tie is cut and values otherwise."""
    g = types.ModuleType("TStar.interface_grounding")
    g.TStarUniversalGrounder = object                          # imported by the module, never used by extract_frames
    sys.modules["TStar.interface_grounding"] = g
    sys.path.insert(0, os.path.join(REF, "LVHaystackBench"))
    import val_qa_results as Q
    rs = np.random.RandomState(31)
    g1 = np.load(os.path.join(OUT, "g1_searcher_case0.npz"))
    P_real = g1["P_last"]                                      # a real, nearly flat P of a 63-iteration search
    N = len(P_real)
    peaky = rs.random_sample(N) ** 8
    plateau = np.round(rs.random_sample(N) * 6) / 6            # 7 distinct levels: ties everywhere
    withnan = peaky.copy()
    withnan[rs.choice(N, 300, replace=False)] = np.nan
    zeros_in_clip = peaky.copy()
    zeros_in_clip[500:900] = 0.0
    cases = [("real_P", P_real, 8, None), ("real_P_clip", P_real, 8, [100.0, 700.9]), ("real_P_k32", P_real, 32, None),
             ("peaky", peaky, 8, None), ("peaky_clip", peaky, 16, [3000.2, 3599.0]), ("plateau", plateau, 8, None),
             ("plateau_clip", plateau, 8, [10.0, 50.0]), ("flat", np.full(N, 0.25), 8, None), ("all_zero", np.zeros(N), 8, None),
             ("nan", withnan, 8, None), ("zero_clip", zeros_in_clip, 8, [500.0, 900.0]), ("short_clip", peaky, 8, [40.0, 45.0])]
    path = register_video(N, 61)
    out = {"names": np.array([c[0] for c in cases])}
    for name, dist, k, clip in cases:
        del SEEKS[:]
        item = {} if clip is None else {"vclip_interval_in_video": clip}
        frames = Q.extract_frames(path, item, frame_distribution=[float(v) for v in dist], num_frames=k, p_fps=1,
                                  duration_type="video" if clip is None else "clip")
        out[f"dist_{name}"] = np.asarray(dist, dtype=np.float64)
        out[f"k_{name}"] = np.int64(k)
        out[f"clip_{name}"] = np.array([-1.0, -1.0] if clip is None else clip)
        out[f"secs_{name}"] = np.array(SEEKS, dtype=np.int64)
        assert len(frames) == len(SEEKS)
        print("g11", name, SEEKS)
    np.savez_compressed(os.path.join(OUT, "g11_topk.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    if "--only-g10" in sys.argv:
        g10_metrics()
        return
    if "--only-g11" in sys.argv:
        g11_topk()
        return
    if "--only-g9b" in sys.argv:
        import matplotlib
        matplotlib.use("Agg")
        from TStar.interface_searcher import TStarSearcher
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                g9b_hour_video(TStarSearcher)
            finally:
                os.chdir(cwd)
        return
    import matplotlib
    matplotlib.use("Agg")
    from TStar.interface_searcher import TStarSearcher
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)                                   # the reference writes ./annotated_image.png per call
        try:
            g1_searcher(TStarSearcher)
            g2_to_g6(TStarSearcher)
            g7_g8_g9(TStarSearcher)
            g9b_hour_video(TStarSearcher)
            g10_metrics()
            g11_topk()
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
