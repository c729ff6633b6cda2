"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the OWL-ViT-B/32 detector arithmetic that the reference
reaches through ``OWLInterface.inference_detector``
(/root/reference/TStar/interface_heuristic.py:232-257).  The arithmetic itself
lives in the un-vendored third-party package ``transformers`` (reference pin
4.49.0, requirements.txt:160; this container has 5.15.0):

* vision tower      modeling_owlvit.py:731-756 (embeddings :334-344, layer
                    :488-509, attention :428-459 / eager :377-402, MLP :471-475)
* image embedder    modeling_owlvit.py:1183-1191 (post-LN, CLS merge, det-LN)
* class head        modeling_owlvit.py:1014-1047
* box head          modeling_owlvit.py:993-999, 1106-1137, bias :1072-1104
* text tower        modeling_owlvit.py:631-663, projection :945-958
* post-process      image_processing_owlvit.py:151-177

Pinned by: tests/test_oracle_owl.py (bit/1e-5 comparison with the installed HF
``OwlViTForObjectDetection`` when transformers is importable) and the golden
vectors tests/golden/g7_*.npz produced from HF by tests/golden/make_goldens.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Plain torch fp32 ops; weights come in as the dict
returned by ``tstar_amd.weights.unpack_blob``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from tstar_amd import weights as W


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a))


def _encoder_layer(x, w: Dict[str, np.ndarray], p: str, heads: int, mask=None):
    """Pre-LN CLIP layer; x [B,T,D]."""
    B, T, D = x.shape
    hd = D // heads
    h = F.layer_norm(x, (D,), _t(w[p + "ln1_w"]), _t(w[p + "ln1_b"]), W.LN_EPS)
    qkv = F.linear(h, _t(w[p + "qkv_w"]), _t(w[p + "qkv_b"]))
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, T, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    att = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    if mask is not None:
        att = att + mask
    att = torch.softmax(att, dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(o, _t(w[p + "out_w"]), _t(w[p + "out_b"]))
    h = F.layer_norm(x, (D,), _t(w[p + "ln2_w"]), _t(w[p + "ln2_b"]), W.LN_EPS)
    h = F.linear(h, _t(w[p + "fc1_w"]), _t(w[p + "fc1_b"]))
    h = h * torch.sigmoid(1.702 * h)                       # quick_gelu
    x = x + F.linear(h, _t(w[p + "fc2_w"]), _t(w[p + "fc2_b"]))
    return x


def vision_features(pixel_values: torch.Tensor, w: Dict[str, np.ndarray]) -> torch.Tensor:
    """pixel_values f32 [B,3,768,768] -> image feats [B,576,768] (after det-LN)."""
    B = pixel_values.shape[0]
    conv_w = _t(w["patch_w"]).view(W.V_D, 3, W.PATCH, W.PATCH)
    x = F.conv2d(pixel_values, conv_w, stride=W.PATCH)          # [B,768,24,24]
    x = x.flatten(2).transpose(1, 2)                           # [B,576,768]
    cls = _t(w["class_emb"]).expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + _t(w["pos_emb"])
    x = F.layer_norm(x, (W.V_D,), _t(w["pre_ln_w"]), _t(w["pre_ln_b"]), W.LN_EPS)
    for i in range(W.V_LAYERS):
        x = _encoder_layer(x, w, f"owlvit.vision_model.encoder.layers.{i}.", W.V_HEADS)
    e = F.layer_norm(x, (W.V_D,), _t(w["post_ln_w"]), _t(w["post_ln_b"]), W.LN_EPS)
    feats = e[:, 1:, :] * e[:, :1, :]
    feats = F.layer_norm(feats, (W.V_D,), _t(w["det_ln_w"]), _t(w["det_ln_b"]), W.LN_EPS)
    return feats


def text_query_embeds(input_ids: np.ndarray, attention_mask: np.ndarray, w: Dict[str, np.ndarray]) -> torch.Tensor:
    """ids int64 [Q,16], mask [Q,16] -> L2-normalised query embeds f32 [Q,512]."""
    ids = torch.from_numpy(np.asarray(input_ids, dtype=np.int64))
    am = torch.from_numpy(np.asarray(attention_mask, dtype=np.int64))
    Q, T = ids.shape
    x = _t(w["tok_emb"])[ids] + _t(w["tpos_emb"])[:T]
    neg = torch.finfo(torch.float32).min
    causal = torch.full((T, T), neg).triu(1)
    pad = torch.zeros(Q, 1, 1, T)
    pad = pad.masked_fill(am.view(Q, 1, 1, T) == 0, neg)
    mask = (causal.view(1, 1, T, T) + pad).clamp_min(neg)
    for i in range(W.T_LAYERS):
        x = _encoder_layer(x, w, f"owlvit.text_model.encoder.layers.{i}.", W.T_HEADS, mask)
    x = F.layer_norm(x, (W.T_D,), _t(w["final_ln_w"]), _t(w["final_ln_b"]), W.LN_EPS)
    pooled = x[torch.arange(Q), ids.to(torch.int).argmax(dim=-1)]
    t = F.linear(pooled, _t(w["text_proj"]))
    return t / torch.linalg.norm(t, ord=2, dim=-1, keepdim=True)


def heads(feats: torch.Tensor, query_embeds: torch.Tensor, w: Dict[str, np.ndarray],
          query_mask: np.ndarray | None = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """feats [B,576,768], query_embeds [Q,512] -> logits [B,576,Q], boxes [B,576,4] (cxcywh)."""
    c = F.linear(feats, _t(w["cls_w"]), _t(w["cls_b"]))
    c = c / (torch.linalg.norm(c, dim=-1, keepdim=True) + 1e-6)
    q = query_embeds / (torch.linalg.norm(query_embeds, dim=-1, keepdim=True) + 1e-6)
    logits = torch.einsum("bpd,qd->bpq", c, q)
    shift = F.linear(feats, _t(w["shift_w"]).view(1, -1), _t(w["shift_b"]))
    scale = F.linear(feats, _t(w["scale_w"]).view(1, -1), _t(w["scale_b"]))
    scale = F.elu(scale) + 1
    logits = (logits + shift) * scale
    if query_mask is not None:
        qm = torch.from_numpy(np.asarray(query_mask)).view(1, 1, -1)
        logits = torch.where(qm == 0, torch.finfo(torch.float32).min, logits)
    b = F.gelu(F.linear(feats, _t(w["box0_w"]), _t(w["box0_b"])))
    b = F.gelu(F.linear(b, _t(w["box1_w"]), _t(w["box1_b"])))
    b = F.linear(b, _t(w["box2_w"]), _t(w["box2_b"]))
    boxes = torch.sigmoid(b + _t(w["box_bias"]))
    return logits, boxes


def post_process(logits: torch.Tensor, boxes: torch.Tensor, height: int, width: int, threshold: float = 0.005):
    """Per image: (scores f32 [n], labels i64 [n], xyxy f32 [n,4]) kept in patch order.

    image_processing_owlvit.py:151-177 with threshold 0.005
    (/root/reference/TStar/interface_heuristic.py:242-243).  Also returns the
    dense (unfiltered) scores/labels/xyxy for kernel parity checks.
    """
    v, lab = torch.max(logits, dim=-1)
    score = torch.sigmoid(v)
    cx, cy, bw, bh = boxes.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], dim=-1)
    scale = torch.tensor([width, height, width, height], dtype=torch.int64)
    xyxy = xyxy * scale
    out = []
    for s, l, b in zip(score, lab, xyxy):
        keep = s > threshold
        out.append((s[keep].numpy(), l[keep].numpy(), b[keep].numpy()))
    return out, (score.numpy(), lab.numpy(), xyxy.numpy())


def detect(pixel_values: np.ndarray, query_embeds: np.ndarray, w: Dict[str, np.ndarray],
           height: int, width: int, query_mask=None):
    """Full detector on preprocessed pixels (f32 [B,3,768,768])."""
    with torch.no_grad():
        feats = vision_features(_t(pixel_values), w)
        logits, boxes = heads(feats, _t(query_embeds), w, query_mask)
        kept, dense = post_process(logits, boxes, height, width)
    return {"logits": logits.numpy(), "boxes": boxes.numpy(), "kept": kept, "dense": dense,
            "feats": feats.numpy()}
