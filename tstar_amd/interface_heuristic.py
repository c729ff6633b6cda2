"""Open-vocabulary detector plug-ins with the reference's duck-typed surface
(/root/reference/TStar/interface_heuristic.py): ``HeuristicInterface`` (:28-37) and
``OWLInterface`` (:200-280), backed by the HIP OWL-ViT-B/32 scorer instead of HF
transformers + torch.

Same names, argument meaning and behaviour as the reference where a caller can observe it:

* ``reparameterize_object_list(target_objects, cue_objects)`` builds
  ``texts = [[name.strip()], ..., [' ']]`` (:268-280, the trailing blank query included);
* ``inference_detector(images, **kw)`` scores ONLY ``images[0]`` (:234), keeps detections with
  score > 0.005 (:243) in patch order and returns ``[Detections]`` with ``.xyxy`` f32 [n,4]
  (pixels of the passed image), ``.confidence`` f32 [n], ``.class_id`` int64 [n]; it also
  refreshes ``self.detections_inbatch`` (:256);
* ``bbox_visualization(images, detections_inbatch)`` draws on the arrays it is given (:259-267).

Documented deviations: no ``./annotated_image.png`` is written per call (the reference's debug
block, :248-255, costs ~50 ms per call); the text tower runs once per
``reparameterize_object_list`` instead of once per detector call (it is constant per question).

``YoloWorldInterface`` (:39-190) is the second backend: the reference reaches YOLO-World through mmdet /
mmyolo and a repository that is NOT part of its tree, so the detector here is a from-scratch HIP
implementation of the published YOLO-World-v2 architecture (f32 VALU kernels, no MFMA -- BASELINE
configs[3]) whose parity against the real model is unpinned (oracle/yolo_ref.py states why); the
WRAPPER semantics the reference itself defines are kept: ``texts`` layout with the trailing blank
query, ``score > 0.12`` then top-50, only ``images[0]``, ``detections_inbatch``.

Extensions used by tstar_amd.TStarSearcher's batched fast path (a foreign heuristic without them
still works through ``inference_detector``): ``set_class_weights``, ``score_batch``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from . import weights as W
from . import _lib
from .tokenizer import encode_queries


@dataclass
class Detections:
    """Minimal stand-in for ``supervision.Detections`` (the fields the searcher reads,
    /root/reference/TStar/interface_searcher.py:134)."""
    xyxy: np.ndarray
    confidence: np.ndarray
    class_id: np.ndarray

    def __len__(self) -> int:
        return int(self.xyxy.shape[0])


def _env_int(name: str, default):
    v = os.environ.get(name)
    if v is None or v.strip() == "":
        return default
    try:
        return int(v)
    except ValueError:
        raise ValueError(f"{name} must be an integer, not {v!r}") from None


class HeuristicInterface:
    def __init__(self, heuristic_type: str = "owl-vit", **kwargs):
        """Base of the detector plug-ins (empty in the reference too, interface_heuristic.py:28-37)."""


class OWLInterface(HeuristicInterface):
    def __init__(self, model_name_or_path: str = "google/owlvit-base-patch32", device: str = "cuda",
                 max_batch: Optional[int] = None, synthetic_seed: Optional[int] = None, state_dict: Optional[Dict] = None,
                 weights_dtype: Optional[str] = None, allow_standin_tokenizer: Optional[bool] = None):
        """``device`` must be a HIP device (default "cuda" as in the reference, :201).

        Under an UNCHANGED ``TStarFramework`` the heuristic is built by ``initialize_heuristic(heuristic_type)`` without keyword
        arguments (TStarFramework.py:171-187, 207), so the three arguments a deployment chooses can also come from the environment
        -- read only when the keyword is not given: ``TSTAR_WEIGHTS_DTYPE`` (``weights_dtype``; default "f32"),
        ``TSTAR_MAX_BATCH`` (``max_batch``; default 32) and ``TSTAR_SYNTHETIC_SEED`` (``synthetic_seed``; unset = no synthetic
        weights: a missing checkpoint raises).

        Weights: ``state_dict`` (HF names) if given; else a local safetensors checkpoint of
        ``model_name_or_path`` if one exists on disk; else, only when ``synthetic_seed`` is not None,
        seeded synthetic weights (no checkpoint can be downloaded: there is no network).
        ``weights_dtype="bf16"`` rounds every weight matrix to bfloat16 (BASELINE config 5) and runs the
        GEMMs on the bf16 matrix pipe with the float32 activations carried as two round-to-nearest bf16
        terms (16 significand bits): products are exact and accumulate in float32; scores stay within
        ~1e-5 of a CPU float32 run on the same rounded weights (contract 1e-3).  ``"bf16_exact"`` splits
        the activations exactly into three terms instead (3 matrix products per algorithmic product):
        equal to that CPU run up to summation order.

        ``allow_standin_tokenizer``: the hash stand-in of tstar_amd.tokenizer is only meaningful with
        synthetic weights; default (None) = allowed exactly when the weights are the seeded synthetic
        ones.  With a real checkpoint and no CLIP vocab on disk, installing queries raises instead of
        feeding made-up ids to the real text tower."""
        import torch
        from .owl import OwlScorer
        if not str(device).startswith("cuda"):
            raise ValueError("tstar_amd.OWLInterface runs on the GPU only (device='cuda[:i]'); it has no CPU path")
        if weights_dtype is None:
            weights_dtype = os.environ.get("TSTAR_WEIGHTS_DTYPE") or "f32"
        if weights_dtype not in ("f32", "bf16", "bf16_exact", "f32x3"):
            raise ValueError(f"weights_dtype (or TSTAR_WEIGHTS_DTYPE) must be 'f32', 'bf16', 'bf16_exact' or 'f32x3', not {weights_dtype!r}")
        if max_batch is None:
            max_batch = _env_int("TSTAR_MAX_BATCH", 32)
        if synthetic_seed is None and state_dict is None:
            synthetic_seed = _env_int("TSTAR_SYNTHETIC_SEED", None)
        dev = torch.device(device)
        if dev.index is not None:
            torch.cuda.set_device(dev.index)
        if state_dict is None:
            ckpt = W.find_pretrained(model_name_or_path)
            if ckpt is not None:
                state_dict = W.load_safetensors_state_dict(ckpt)
                self.weights_source = ckpt
            elif synthetic_seed is not None:
                state_dict = W.synthetic_state_dict(int(synthetic_seed))
                self.weights_source = f"synthetic(seed={int(synthetic_seed)})"
            else:
                raise FileNotFoundError(
                    f"no local checkpoint for {model_name_or_path!r} (offline); pass synthetic_seed=<int> "
                    "for seeded synthetic OWL-ViT-B/32 weights or state_dict=<HF state dict>")
        else:
            self.weights_source = "state_dict"
        if allow_standin_tokenizer is None:
            allow_standin_tokenizer = self.weights_source.startswith("synthetic(")
        self.allow_standin_tokenizer = bool(allow_standin_tokenizer)
        if weights_dtype in ("bf16", "bf16_exact"):
            state_dict = W.round_weights_to_bf16(state_dict)
        self.weights_dtype = weights_dtype
        self.model_name_or_path = model_name_or_path
        self.scorer = OwlScorer(W.pack_blob(state_dict, W.vision_spec()), W.pack_blob(state_dict, W.text_spec()),
                                max_batch=max_batch, weights_mode=weights_dtype)
        self.device = device
        self.texts = ["couch", "table", "woman"]      # as the reference leaves it before reparameterisation (:203)
        self.detections_inbatch: List[Detections] = []
        self._class_weight: Optional[np.ndarray] = None
        self._ids = None

    # ---- reference surface -------------------------------------------------------------
    def reparameterize_object_list(self, target_objects: List[str], cue_objects: List[str]):
        combined = list(target_objects) + list(cue_objects)
        self.texts = [[obj.strip()] for obj in combined] + [[' ']]
        ids, am = encode_queries(self.texts, self.model_name_or_path, allow_standin=self.allow_standin_tokenizer)
        self._ids, self._am = ids, am
        # default weights = the searcher's own defaults (target 1.0, cue 0.5, unknown 0.5;
        # interface_searcher.py:88-91,136); a searcher overrides them via set_class_weights
        w = [1.0] * len(target_objects) + [0.5] * len(cue_objects) + [0.5]
        # recorded now, run through the text tower when slot 0 is first used (OwlScorer.set_queries): a searcher's constructor
        # calls this like the reference's, but a lock-step group never scores against slot 0
        self.scorer.set_queries(ids, am, w, lazy=True)
        self._class_weight = np.asarray(w, dtype=np.float64)

    def inference_detector(self, images, **kwargs) -> List[Detections]:
        import torch
        img = np.array(images[0], dtype=np.uint8, order="C")                  # only image 0, as the reference (own copy)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("inference_detector expects HxWx3 uint8 RGB images")
        d_img = torch.from_numpy(img).cuda().unsqueeze(0)
        r = self.scorer.score(d_img, 1, 1)
        dets = [self._detections_from(r, 0)]
        self.detections_inbatch = dets
        return dets

    def inference(self, image_path, use_amp: bool = False) -> Detections:
        """(:217-230) detector on an image FILE (decoded with Pillow, RGB) against the current ``texts``."""
        from PIL import Image
        with Image.open(image_path) as im:
            image = np.asarray(im.convert("RGB"), dtype=np.uint8)
        return self.inference_detector([image], use_amp=use_amp)[0]

    def bbox_visualization(self, images, detections_inbatch):
        out = []
        for image, det in zip(images, detections_inbatch):
            out.append(draw_boxes(image, det))
        return out

    # ---- fast-path extensions ----------------------------------------------------------
    def set_class_weights(self, object2weight: Dict[str, float]):
        """Install ``object2weight.get(name, 0.5)`` per query (interface_searcher.py:136)."""
        w = [float(object2weight.get(t[0], 0.5)) for t in self.texts]
        self.scorer.set_class_weights(w)
        self._class_weight = np.asarray(w, dtype=np.float64)

    aux_lane = True      # score_batch(..., lane=1) runs in a second workspace: safe to enqueue on another stream beside lane 0

    def score_batch(self, d_images, grid_rows: int, grid_cols: int, image_sets=None, lane: int = 0):
        """Batched scoring of device images u8 [B,H,W,3] -> tstar_amd.owl.ScoreResult (device tensors).
        ``image_sets``: query-set slot per image (see ``install_queries``); default slot 0.  ``lane``: see ``OwlScorer.score``."""
        return self.scorer.score(d_images, grid_rows, grid_cols, image_sets=image_sets, lane=lane)

    def install_queries(self, slot: int, target_objects: List[str], cue_objects: List[str],
                        object2weight: Optional[Dict[str, float]] = None) -> List[List[str]]:
        """Install a question's queries in slot 1..63 WITHOUT touching ``self.texts`` (slot 0 is what
        ``reparameterize_object_list`` manages).  Several (video, question) items can then be scored in
        one batch, each image against its own slot.  Returns the texts list of the slot."""
        texts, (slot, ids, am, weights) = self._query_entry(slot, target_objects, cue_objects, object2weight)
        self.scorer.set_queries(ids, am, weights, slot=slot)
        return texts

    def _query_entry(self, slot, target_objects, cue_objects, object2weight):
        """(texts of the slot, (slot, token ids, attention mask, class weights)) of one question: blank query appended, targets
        weigh 1.0 and cues 0.5 unless ``object2weight`` says otherwise (interface_searcher.py:88-91, 135-137)."""
        if not 1 <= int(slot) <= 63:
            raise ValueError("install_queries: slot must be in 1..63")
        texts = [[obj.strip()] for obj in list(target_objects) + list(cue_objects)] + [[' ']]
        ids, am = encode_queries(texts, self.model_name_or_path, allow_standin=self.allow_standin_tokenizer)
        o2w = dict(object2weight or {})
        for o in target_objects:
            o2w.setdefault(o, 1.0)
        for o in cue_objects:
            o2w.setdefault(o, 0.5)
        return texts, (int(slot), ids, am, [float(o2w.get(t[0], 0.5)) for t in texts])

    def install_queries_many(self, items) -> List[List[List[str]]]:
        """``install_queries`` for several slots at once -- ``items``: [(slot, target_objects, cue_objects, object2weight)] -- with
        the text tower run ONCE over all their queries (a lock-step group installs the questions of its items together).
        Returns the texts list of every slot; identical to one ``install_queries`` call per item."""
        built = [self._query_entry(*it) for it in items]
        self.scorer.set_queries_many([e for _, e in built])
        return [t for t, _ in built]

    def annotated_batch(self, d_images, r, start: int = 0, count: Optional[int] = None):
        """Device form of ``bbox_visualization`` + ``Detections`` for images that are already on the device (the
        searcher's visual history): paints the kept boxes of ``r`` (images ``start .. start+count`` of a
        ``score_batch`` result) onto ``d_images`` u8 [count,H,W,3] IN PLACE, then brings images and detections to the
        host with one copy each.  Returns (images uint8 [count,H,W,3], [Detections] * count)."""
        count = int(d_images.shape[0]) if count is None else int(count)
        if not d_images.is_contiguous() or d_images.shape[0] != count:
            raise ValueError("annotated_batch: d_images must be a contiguous [count,H,W,3] uint8 device tensor")
        boxes = r.boxes[start:start + count].contiguous()
        scores = r.scores[start:start + count].contiguous()
        lib = _lib.load()
        _lib.check(lib.tstar_draw_boxes(d_images.data_ptr(), count, int(d_images.shape[1]), int(d_images.shape[2]),
                                        boxes.data_ptr(), scores.data_ptr(), _lib.stream_ptr()), "tstar_draw_boxes")
        imgs = d_images.cpu().numpy()
        s, bx, lab = scores.cpu().numpy(), boxes.cpu().numpy(), r.labels[start:start + count].cpu().numpy()
        dets = []
        for k in range(count):
            keep = s[k] > np.float32(0.005)
            dets.append(Detections(xyxy=bx[k][keep], confidence=s[k][keep], class_id=lab[k][keep].astype(np.int64)))
        return imgs, dets

    def _detections_from(self, r, b: int) -> Detections:
        s = r.scores[b].cpu().numpy()
        keep = s > np.float32(0.005)
        return Detections(xyxy=r.boxes[b].cpu().numpy()[keep], confidence=s[keep],
                          class_id=r.labels[b].cpu().numpy()[keep].astype(np.int64))


def _yolo_scale_from_config(config_path: Optional[str]) -> str:
    """'yolo_world_v2_xl_vlpan_...' -> 'xl' (the config the reference wires, TStarFramework.py:181: the yolov8_x base scaled
    "from X to XL", widen 1.5); BASELINE configs[3] names the L model, which is the default when the name says nothing."""
    import re
    m = re.search(r"yolo_world(?:_v2)?_(s|m|l|xl|x)_", os.path.basename(str(config_path or "")))
    return m.group(1) if m else "l"


class YoloWorldInterface(HeuristicInterface):
    def __init__(self, config_path: Optional[str] = None, checkpoint_path: Optional[str] = None, device: str = "cuda:0", *,
                 scale: Optional[str] = None, synthetic_seed: Optional[int] = None, state_dict: Optional[Dict] = None,
                 text_state_dict: Optional[Dict] = None, max_batch: int = 16):
        """Arguments as the reference (:40-47: ``config_path``, ``checkpoint_path``, ``device``).  The mmengine config is
        only consulted for the model scale (its file name); weights come from ``state_dict`` (mmyolo / YOLO-World names),
        else ``checkpoint_path`` if that file exists (a torch checkpoint with a ``state_dict`` entry), else -- only when
        ``synthetic_seed`` is not None -- seeded synthetic parameters (there is no network to fetch a checkpoint).
        The CLIP text tower (HuggingCLIPLanguageBackbone in the real model) is the HIP text tower shared with the
        OWL-ViT backend, fed from ``text_state_dict`` (HF OWL-ViT / CLIP text names) or the checkpoint's
        ``backbone.text_model`` entries or synthetic weights."""
        import torch
        from .owl import OwlScorer
        from .yolo import YoloDetector
        from . import yolo_world as YW
        if not str(device).startswith("cuda"):
            raise ValueError("tstar_amd.YoloWorldInterface runs on the GPU only (device='cuda[:i]'); it has no CPU path")
        self.config_path, self.checkpoint_path, self.device = config_path, checkpoint_path, device
        self.scale = scale or _yolo_scale_from_config(config_path)
        if state_dict is None and synthetic_seed is None and not (checkpoint_path and os.path.isfile(checkpoint_path)):
            raise FileNotFoundError(
                f"no YOLO-World checkpoint at {checkpoint_path!r} (offline); pass synthetic_seed=<int> for seeded synthetic "
                f"YOLO-World-v2-{self.scale.upper()} weights or state_dict=<mmyolo state dict>")
        dev = torch.device(device)
        if dev.index is not None:
            torch.cuda.set_device(dev.index)
        if state_dict is None and checkpoint_path and os.path.isfile(checkpoint_path):
            # tensors only: the mmengine checkpoint's meta / optimizer objects are never needed, so nothing is unpickled
            # beyond what torch's safe loader admits
            ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            ck = ck.get("state_dict", ck)
            state_dict = {k: v.float().numpy() for k, v in ck.items() if hasattr(v, "numpy")}
            self.weights_source = checkpoint_path
        elif state_dict is not None:
            self.weights_source = "state_dict"
        else:
            state_dict = YW.synthetic_state_dict(int(synthetic_seed), self.scale)
            self.weights_source = f"synthetic(seed={int(synthetic_seed)})"
        if text_state_dict is None:
            pre = "backbone.text_model.model."
            clip = {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}
            if clip:          # HF CLIPTextModelWithProjection names -> the OWL-ViT text tower's names
                text_state_dict = {("owlvit." + k) if not k.startswith("text_projection") else "owlvit." + k: np.asarray(v, np.float32)
                                   for k, v in clip.items()}
                pos = "owlvit.text_model.embeddings.position_embedding.weight"
                if pos in text_state_dict:
                    text_state_dict[pos] = text_state_dict[pos][:W.T_LEN]          # 77 CLIP positions, queries use <= 16
            elif self.weights_source.startswith("synthetic("):
                text_state_dict = W.synthetic_state_dict(int(synthetic_seed), "text")
            else:
                raise FileNotFoundError("the YOLO-World state dict carries no CLIP text tower; pass text_state_dict=")
        self.allow_standin_tokenizer = self.weights_source.startswith("synthetic(")
        self.detector = YoloDetector(state_dict, self.scale, max_batch=max_batch)
        self.text_tower = OwlScorer(None, W.pack_blob(text_state_dict, W.text_spec()), max_batch=1)
        self.model_name_or_path = "openai/clip-vit-base-patch32"
        self.texts = []
        self.detections_inbatch: List[Detections] = []
        self._text_feats = None
        self._pending0 = None
        self.set_BBoxAnnotator()

    def set_BBoxAnnotator(self):
        """(:68-76) the reference builds supervision annotators; boxes are painted by ``draw_boxes`` here."""
        self.BOUNDING_BOX_ANNOTATOR = draw_boxes
        self.LABEL_ANNOTATOR = None

    # ---- reference surface -------------------------------------------------------------
    def _encode(self, texts, weights, slot):
        ids, am = encode_queries(texts, self.model_name_or_path, allow_standin=self.allow_standin_tokenizer)
        self.text_tower.set_queries(ids, am, weights, slot=0)
        feats = self.text_tower.get_query_embeds(0)              # text_embeds / ||text_embeds|| (the backbone's forward_text)
        self.detector.set_text_feats(feats, weights, slot=slot)
        return feats

    def reparameterize_object_list(self, target_objects: List[str], cue_objects: List[str]):
        """(:78-93) texts = [[name.strip()], ..., [' ']]; ``model.reparameterize(texts)`` caches the text features."""
        combined = list(target_objects) + list(cue_objects)
        self.texts = [[obj.strip()] for obj in combined] + [[' ']]
        w = [1.0] * len(target_objects) + [0.5] * len(cue_objects) + [0.5]
        # tokenised now (a missing vocabulary must be reported here), run through the CLIP text tower when slot 0 is first used
        ids, am = encode_queries(self.texts, self.model_name_or_path, allow_standin=self.allow_standin_tokenizer)
        self._pending0 = (ids, am, list(w))
        self._text_feats = None
        self._class_weight = np.asarray(w, dtype=np.float64)

    def _flush0(self):
        """Install the recorded queries of slot 0 (see ``reparameterize_object_list``)."""
        p = getattr(self, "_pending0", None)
        if p is not None:
            self._pending0 = None
            self.text_tower.set_queries(p[0], p[1], p[2], slot=0)
            self._text_feats = self.text_tower.get_query_embeds(0)
            self.detector.set_text_feats(self._text_feats, p[2], slot=0)

    def inference_detector(self, images, max_dets: int = 50, score_threshold: float = 0.12, use_amp: bool = False) -> List[Detections]:
        """(:136-168) only ``images[0]``; detections with score > ``score_threshold``, the ``max_dets`` best, descending."""
        import torch
        img = np.array(images[0], dtype=np.uint8, order="C")
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("inference_detector expects HxWx3 uint8 RGB images")
        self._flush0()
        r = self.detector.detect(torch.from_numpy(img).cuda().unsqueeze(0), 1, 1, score_threshold=score_threshold, max_dets=max_dets,
                                 want_cells=False)
        dets = [self._detections_from(r, 0)]
        self.detect_outputs_raw = r
        self.detections_inbatch = dets
        return dets

    def inference(self, image, max_dets: int = 100, score_threshold: float = 0.3, use_amp: bool = False) -> Detections:
        """(:96-134) detector on an image FILE.  mmdet's LoadImageFromFile hands BGR to the pipeline, whose preprocessor
        swaps to RGB; ``inference_detector`` receives RGB and (as in the reference) lets the same swap happen, so the file
        is decoded to BGR order here to end up with the channel order the model was trained on."""
        from PIL import Image
        with Image.open(image) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
        return self.inference_detector([np.ascontiguousarray(rgb[:, :, ::-1])], max_dets=max_dets, score_threshold=score_threshold)[0]

    def bbox_visualization(self, images, detections_inbatch):
        """(:170-190) annotated COPIES; like the reference, every entry annotates ``images[len(detections_inbatch) - 1]``."""
        out = []
        for detections in detections_inbatch:
            image = images[len(detections_inbatch) - 1]
            out.append(draw_boxes(image.copy(), detections))
        return out

    # ---- fast-path extensions (same contract as OWLInterface) -----------------------------
    def set_class_weights(self, object2weight: Dict[str, float]):
        w = [float(object2weight.get(t[0], 0.5)) for t in self.texts]
        if getattr(self, "_pending0", None) is not None:      # slot 0 not installed yet: the weights ride along
            self._pending0 = (self._pending0[0], self._pending0[1], w)
        else:
            self.detector.set_class_weights(w)
        self._class_weight = np.asarray(w, dtype=np.float64)

    def score_batch(self, d_images, grid_rows: int, grid_cols: int, image_sets=None):
        if getattr(self, "_pending0", None) is not None and (image_sets is None or 0 in [int(v) for v in image_sets]):
            self._flush0()
        return self.detector.detect(d_images, grid_rows, grid_cols, score_threshold=0.12, max_dets=50, image_sets=image_sets)

    def install_queries(self, slot: int, target_objects: List[str], cue_objects: List[str],
                        object2weight: Optional[Dict[str, float]] = None) -> List[List[str]]:
        if not 1 <= int(slot) <= 63:
            raise ValueError("install_queries: slot must be in 1..63")
        texts = [[obj.strip()] for obj in list(target_objects) + list(cue_objects)] + [[' ']]
        o2w = dict(object2weight or {})
        for o in target_objects:
            o2w.setdefault(o, 1.0)
        for o in cue_objects:
            o2w.setdefault(o, 0.5)
        self._encode(texts, [float(o2w.get(t[0], 0.5)) for t in texts], int(slot))
        return texts

    def install_queries_many(self, items) -> List[List[List[str]]]:
        """``install_queries`` for several slots with ONE run of the CLIP text tower over all their queries (the text tower keeps
        the same slot numbers; each slot's normalised text features then go to the detector's guide layers)."""
        out, entries, weights = [], [], []
        for slot, target_objects, cue_objects, object2weight in items:
            if not 1 <= int(slot) <= 63:
                raise ValueError("install_queries: slot must be in 1..63")
            texts = [[obj.strip()] for obj in list(target_objects) + list(cue_objects)] + [[' ']]
            o2w = dict(object2weight or {})
            for o in target_objects:
                o2w.setdefault(o, 1.0)
            for o in cue_objects:
                o2w.setdefault(o, 0.5)
            w = [float(o2w.get(t[0], 0.5)) for t in texts]
            ids, am = encode_queries(texts, self.model_name_or_path, allow_standin=self.allow_standin_tokenizer)
            entries.append((int(slot), ids, am, w))
            weights.append(w)
            out.append(texts)
        self.text_tower.set_queries_many(entries)
        for (slot, _, _, _), w in zip(entries, weights):
            self.detector.set_text_feats(self.text_tower.get_query_embeds(slot), w, slot=slot)
        return out

    def annotated_batch(self, d_images, r, start: int = 0, count: Optional[int] = None):
        """Images + detections of a ``score_batch`` result on the host, boxes painted (<= 50 per image: host painter)."""
        count = int(d_images.shape[0]) if count is None else int(count)
        imgs = d_images.cpu().numpy()
        dets = [self._detections_from(r, start + k) for k in range(count)]
        for k in range(count):
            draw_boxes(imgs[k], dets[k])
        return imgs, dets

    def _detections_from(self, r, b: int) -> Detections:
        n = int(r.n_kept[b].item())
        return Detections(xyxy=r.boxes[b, :n].cpu().numpy(), confidence=r.scores[b, :n].cpu().numpy(),
                          class_id=r.labels[b, :n].cpu().numpy().astype(np.int64))


def draw_boxes(image: np.ndarray, det: Detections, color=(255, 64, 64)) -> np.ndarray:
    """1-px rectangles painted IN PLACE on ``image`` (the reference's supervision BoxAnnotator also
    paints on the array it is given, Appendix B.14) and returned."""
    H, Wd = image.shape[:2]
    b = np.asarray(det.xyxy, dtype=np.float64).reshape(-1, 4)
    # round half to even (Python round / np.rint / the device painter's rint), clamp to the image
    xs = np.clip(np.rint(b[:, [0, 2]]), 0, Wd - 1).astype(np.int64)
    ys = np.clip(np.rint(b[:, [1, 3]]), 0, H - 1).astype(np.int64)
    for (xa, xb), (ya, yb) in zip(xs.tolist(), ys.tolist()):
        if xb < xa or yb < ya:
            continue
        image[ya, xa:xb + 1] = color
        image[yb, xa:xb + 1] = color
        image[ya:yb + 1, xa] = color
        image[ya:yb + 1, xb] = color
    return image


def initialize_heuristic(heuristic_type: str = "owl-vit", **kwargs) -> HeuristicInterface:
    """Factory with the reference's signature (TStarFramework.py:171-187)."""
    if heuristic_type == "owl-vit":
        return OWLInterface(model_name_or_path="google/owlvit-base-patch32", **kwargs)
    if heuristic_type == "yolo-World":
        # the paths the reference hard-codes (TStarFramework.py:181-182); the checkpoint is used when it exists
        config_path = "./YOLO-World/configs/pretrain/yolo_world_v2_xl_vlpan_bn_2e-3_100e_4x8gpus_obj365v1_goldg_train_lvis_minival.py"
        checkpoint_path = "./pretrained/YOLO-World/yolo_world_v2_xl_obj365v1_goldg_cc3mlite_pretrain-5daf1395.pth"
        kwargs.setdefault("config_path", config_path)
        kwargs.setdefault("checkpoint_path", checkpoint_path)
        return YoloWorldInterface(**kwargs)
    raise NotImplementedError(f"Heuristic type '{heuristic_type}' is not implemented.")
