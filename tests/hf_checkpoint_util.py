"""A local OWL-ViT checkpoint DIRECTORY of the real layout, made offline: HF's own ``OwlViTForObjectDetection(OwlViTConfig())``
(base-patch32 defaults, HF's ``_init_weights`` under ``torch.manual_seed``) saved with ``save_pretrained`` (config.json +
model.safetensors) next to a hand-made CLIP vocabulary (tests/clip_vocab_util.py).  What the reference loads with
``OwlViTForObjectDetection.from_pretrained(name)`` / ``OwlViTProcessor.from_pretrained(name)``
(/root/reference/TStar/interface_heuristic.py:207-210) when ``name`` is a directory."""
import os


def make_checkpoint_dir(dirpath: str, seed: int = 0):
    import torch
    import transformers
    from clip_vocab_util import write_clip_vocab
    torch.manual_seed(seed)
    cfg = transformers.OwlViTConfig()
    m = transformers.OwlViTForObjectDetection(cfg).eval()
    with torch.no_grad():
        # HF's init leaves the class head's scale / shift branches at unit scale: with random features the logits land in
        # the thousands and every sigmoid saturates (SURVEY.md 8c caveat c).  Shrinking these two Linear layers keeps scores
        # spread over (0, 1) so that a score comparison means something; everything else is HF's init as drawn.
        for lin in (m.class_head.logit_scale, m.class_head.logit_shift):
            lin.weight.mul_(0.01)
            lin.bias.mul_(0.01)
        # same for the box head: its three unit-variance Linear layers put the box logits in the tens of thousands (every
        # coordinate a saturated 0 / 1, and the few in transition ill-conditioned); 1/sqrt(fan_in) brings them to O(1)
        for lin in (m.box_head.dense0, m.box_head.dense1, m.box_head.dense2):
            lin.weight.mul_(lin.weight.shape[1] ** -0.5)
    os.makedirs(dirpath, exist_ok=True)
    m.save_pretrained(dirpath, safe_serialization=True)
    write_clip_vocab(dirpath)
    return m


def hf_detect(model, tokenizer, image, names, threshold=0.005):
    """The reference's detector call (interface_heuristic.py:232-246) on CPU with HF: tokenise, preprocess (Pillow bicubic +
    rescale + normalise: oracle/resize_ref.owl_preprocess is bit-equal to HF's PIL image processor, tests/test_oracle_owl.py),
    forward, post_process_object_detection semantics (sigmoid of the max logit, arg-max label, cxcywh -> xyxy in pixels of
    the passed image, score > threshold)."""
    import numpy as np
    import torch
    from oracle import resize_ref as R
    enc = tokenizer(names, padding="max_length", max_length=16, truncation=True, return_tensors="pt")
    px = torch.from_numpy(R.owl_preprocess(image)[None])
    with torch.no_grad():
        o = model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], pixel_values=px)
    logits, boxes = o.logits[0], o.pred_boxes[0]
    mx = logits.max(dim=-1)
    scores, labels = torch.sigmoid(mx.values), mx.indices
    cx, cy, w, h = boxes.unbind(-1)
    H, W = image.shape[:2]
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1) * torch.tensor([W, H, W, H], dtype=torch.float32)
    keep = scores > threshold
    return dict(ids=enc["input_ids"].numpy(), mask=enc["attention_mask"].numpy(), logits=logits.numpy(), dense_scores=scores.numpy(),
                scores=scores[keep].numpy(), labels=labels[keep].numpy(), xyxy=xyxy[keep].numpy(), dense_xyxy=xyxy.numpy(),
                dense_labels=labels.numpy(), text_embeds=o.text_embeds[0].numpy())
