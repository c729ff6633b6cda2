"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU statement of the YOLO-World-v2 detector behind the reference's ``YoloWorldInterface``
(/root/reference/TStar/interface_heuristic.py:39-190; wired at TStarFramework.py:178-184).

**PARITY UNPINNED -- SOURCE ABSENT.**  The reference does not contain YOLO-World: ``/root/reference/YOLO-World`` is a
dangling symlink to a repository cloned at install time at an unpinned HEAD (install.sh:12), and mmdet 3.3.0 /
mmyolo 0.6.0 / mmcv 2.1.0 / mmengine 0.10.6 (requirements.txt:73-76) are not installed; no reference test touches this
path.  Everything below restates the PUBLISHED architecture from upstream knowledge, in plain torch (NCHW, eval-mode
BatchNorm kept as its own step, nothing shared with tstar_amd's NHWC / folded-BN program):

  test pipeline   mmyolo YOLOv5KeepRatioResize(640) -> LetterResize(640, allow_scale_up=False, pad 114) ->
                  data_preprocessor(mean 0, std 255, bgr_to_rgb=True).  The reference hands RGB arrays to a
                  LoadImageFromNDArray pipeline that assumes BGR (:137), so the network sees the channels REVERSED;
                  kept.  The resize is cv2 INTER_AREA (ratio < 1) / INTER_LINEAR -- cv2 is not importable here, so the
                  resize itself is the build's own definition (box average over the exact source footprint in fixed
                  point for AREA; oracle/resize_ref.cv_bilinear_resize for LINEAR), stated in ``letterbox``.
  backbone        YOLOv8 CSPDarknet: stem conv3x3 s2, 4 stages of [conv3x3 s2, CSPLayerWithTwoConv], SPPF(5) at the end.
  neck            YOLOWorldPAFPN: top-down + bottom-up with MaxSigmoidCSPLayerWithTwoConv; attention
                  sigmoid(max_n <embed, guide_fc(text_n)> / sqrt(c_head) + bias_head) gates project_conv(x).
  head            YOLOWorldHeadModule(use_bn_head): cls tower -> 512-d embedding -> BatchNorm2d ->
                  <., normalize(text)> * exp(logit_scale) + bias; reg tower -> DFL (softmax over 16 bins) -> distances.
  post-process    mmyolo YOLOv5Head.predict_by_feat: sigmoid scores, point priors (offset 0.5) x stride, distance decode,
                  multi_label candidates score > 0.001, top nms_pre = 30000, un-letterbox, class-aware NMS (mmcv
                  batched_nms coordinate-offset trick, IoU 0.7), max_per_img = 300, clamp to the image;
  wrapper         score > 0.12 then top-50 (interface_heuristic.py:148-152).

What the tests check: tstar_amd's HIP path against THIS statement on the same seeded weights (self-consistency of two
independent implementations), never against the real model.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
REG_MAX = 16
STRIDES = (8, 16, 32)
IMG = 640


def _t(sd, k):
    return torch.from_numpy(np.asarray(sd[k], dtype=np.float32))


def conv_module(sd, name, x, stride=1, act=True):
    """mmcv ConvModule: Conv2d(bias=False, padding=k//2) -> BatchNorm2d(eval) -> SiLU."""
    w = _t(sd, name + ".conv.weight")
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = F.batch_norm(y, _t(sd, name + ".bn.running_mean"), _t(sd, name + ".bn.running_var"), _t(sd, name + ".bn.weight"),
                     _t(sd, name + ".bn.bias"), training=False, eps=BN_EPS)
    return F.silu(y) if act else y


def max_sigmoid_attn(sd, name, x, guide):
    """MaxSigmoidAttnBlock.forward (with_scale=False): x [B,C,H,W], guide [B,N,512]."""
    B, _, H, W = x.shape
    gw, gb = _t(sd, name + ".guide_fc.weight"), _t(sd, name + ".guide_fc.bias")
    heads = int(np.asarray(sd[name + ".bias"]).size)
    embed_c = gw.shape[0]
    hc = embed_c // heads
    g = F.linear(guide, gw, gb).reshape(B, -1, heads, hc)
    embed = conv_module(sd, name + ".embed_conv", x, act=False) if (name + ".embed_conv.conv.weight") in sd else x
    embed = embed.reshape(B, heads, hc, H, W)
    attn = torch.einsum("bmchw,bnmc->bmhwn", embed, g).max(dim=-1)[0]
    attn = attn / (hc ** 0.5) + _t(sd, name + ".bias")[None, :, None, None]
    attn = attn.sigmoid()
    y = conv_module(sd, name + ".project_conv", x, act=False)
    y = y.reshape(B, heads, -1, H, W) * attn.unsqueeze(2)
    return y.reshape(B, -1, H, W)


def csp_two_conv(sd, name, x, n, add_identity, guide=None):
    """CSPLayerWithTwoConv / MaxSigmoidCSPLayerWithTwoConv."""
    xm = conv_module(sd, name + ".main_conv", x)
    mid = xm.shape[1] // 2
    parts = list(xm.split((mid, mid), 1))
    for i in range(n):
        y = conv_module(sd, f"{name}.blocks.{i}.conv2", conv_module(sd, f"{name}.blocks.{i}.conv1", parts[-1]))
        parts.append(y + parts[-1] if add_identity else y)
    if guide is not None:
        parts.append(max_sigmoid_attn(sd, name + ".attn_block", parts[-1], guide))
    return conv_module(sd, name + ".final_conv", torch.cat(parts, 1))


def sppf(sd, name, x):
    x = conv_module(sd, name + ".conv1", x)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    return conv_module(sd, name + ".conv2", torch.cat([x, y1, y2, y3], 1))


def _count(sd, prefix):
    i = 0
    while f"{prefix}.{i}.conv1.conv.weight" in sd:
        i += 1
    return i


def backbone(sd, x):
    b = "backbone.image_model."
    x = conv_module(sd, b + "stem", x, stride=2)
    outs = []
    for si in range(1, 5):
        x = conv_module(sd, f"{b}stage{si}.0", x, stride=2)
        x = csp_two_conv(sd, f"{b}stage{si}.1", x, _count(sd, f"{b}stage{si}.1.blocks"), True)
        if f"{b}stage{si}.2.conv1.conv.weight" in sd:
            x = sppf(sd, f"{b}stage{si}.2", x)
        if si >= 2:
            outs.append(x)
    return outs


def neck(sd, feats, guide):
    """YOLOv8PAFPN data flow (upsample_feats_cat_first) with text-guided CSP layers."""
    n = _count(sd, "neck.top_down_layers.0.blocks")
    inner = [feats[2]]
    for li, idx in enumerate((2, 1)):
        up = F.interpolate(inner[0], scale_factor=2, mode="nearest")
        inner.insert(0, csp_two_conv(sd, f"neck.top_down_layers.{li}", torch.cat([up, feats[idx - 1]], 1), n, False, guide))
    outs = [inner[0]]
    for idx in (0, 1):
        dn = conv_module(sd, f"neck.downsample_layers.{idx}", outs[-1], stride=2)
        outs.append(csp_two_conv(sd, f"neck.bottom_up_layers.{idx}", torch.cat([dn, inner[idx + 1]], 1), n, False, guide))
    return outs


def head(sd, feats, text):
    """-> per level (cls_logit [B,K,H,W], bbox_dist [B,4,H,W] in stride units)."""
    h = "bbox_head.head_module."
    res = []
    for i, f in enumerate(feats):
        e = conv_module(sd, f"{h}cls_preds.{i}.1", conv_module(sd, f"{h}cls_preds.{i}.0", f))
        e = F.conv2d(e, _t(sd, f"{h}cls_preds.{i}.2.weight"), _t(sd, f"{h}cls_preds.{i}.2.bias"))
        c = f"{h}cls_contrasts.{i}."
        e = F.batch_norm(e, _t(sd, c + "norm.running_mean"), _t(sd, c + "norm.running_var"), _t(sd, c + "norm.weight"),
                         _t(sd, c + "norm.bias"), training=False, eps=BN_EPS)
        w = F.normalize(text, dim=-1, p=2)
        logit = torch.einsum("bchw,bkc->bkhw", e, w) * _t(sd, c + "logit_scale").exp() + _t(sd, c + "bias")
        r = conv_module(sd, f"{h}reg_preds.{i}.1", conv_module(sd, f"{h}reg_preds.{i}.0", f))
        r = F.conv2d(r, _t(sd, f"{h}reg_preds.{i}.2.weight"), _t(sd, f"{h}reg_preds.{i}.2.bias"))
        B, _, H, W = r.shape
        d = r.reshape(B, 4, REG_MAX, H * W).permute(0, 3, 1, 2).softmax(3).matmul(torch.arange(REG_MAX, dtype=torch.float32))
        res.append((logit, d.transpose(1, 2).reshape(B, 4, H, W)))
    return res


# ----------------------------------------------------------------------------- test pipeline
def area_resize(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """The build's definition of the INTER_AREA down-scale (cv2 absent: unpinned): every output pixel is the mean of its
    exact source footprint [o*s, (o+1)*s) per axis, computed separably with integer weights in units of 1/(out) source
    pixels (weights sum to `in` per axis), rounded half up once at the end.  Integer scale factors reduce to the plain
    box average.  uint8 [H,W,C] -> uint8 [out_h,out_w,C]."""
    H, W, _ = img.shape

    def taps(n_in, n_out):
        # output o covers source interval [o*n_in, (o+1)*n_in) in units of 1/n_out pixel; pixel p covers [p*n_out,(p+1)*n_out)
        t = []
        for o in range(n_out):
            lo, hi = o * n_in, (o + 1) * n_in
            p0, p1 = lo // n_out, (hi - 1) // n_out
            t.append([(p, min(hi, (p + 1) * n_out) - max(lo, p * n_out)) for p in range(p0, p1 + 1)])
        return t

    tx, ty = taps(W, out_w), taps(H, out_h)
    src = img.astype(np.int64)
    hp = np.zeros((H, out_w, img.shape[2]), dtype=np.int64)
    for o, t in enumerate(tx):
        for p, wgt in t:
            hp[:, o] += src[:, p] * wgt
    out = np.zeros((out_h, out_w, img.shape[2]), dtype=np.int64)
    for o, t in enumerate(ty):
        for p, wgt in t:
            out[o] += hp[p] * wgt
    den = W * H
    return ((2 * out + den) // (2 * den)).astype(np.uint8)


def letterbox(img: np.ndarray):
    """uint8 RGB [h,w,3] -> (float32 [3,640,640] as the network sees it, geometry dict)."""
    from . import resize_ref
    h, w = img.shape[:2]
    ratio = min(IMG / max(h, w), IMG / min(h, w))
    cur = img
    if ratio != 1:
        rw, rh = int(w * ratio), int(h * ratio)
        cur = area_resize(img, rw, rh) if ratio < 1 else resize_ref.cv_bilinear_resize(img, rw, rh)
    rh, rw = cur.shape[:2]
    sf = (rw / w, rh / h)
    r2 = min(min(IMG / rh, IMG / rw), 1.0)
    nh, nw = int(round(rh * r2)), int(round(rw * r2))
    if (nh, nw) != (rh, rw):
        cur = resize_ref.cv_bilinear_resize(cur, nw, nh)
        sf = (sf[0] * nw / rw, sf[1] * nh / rh)
    ph, pw = IMG - nh, IMG - nw
    top, left = int(round(ph // 2 - 0.1)), int(round(pw // 2 - 0.1))
    canvas = np.full((IMG, IMG, 3), 114, dtype=np.uint8)
    canvas[top:top + nh, left:left + nw] = cur
    x = canvas[:, :, ::-1].astype(np.float32) / np.float32(255.0)          # "bgr_to_rgb" applied to an RGB array
    return np.ascontiguousarray(x.transpose(2, 0, 1)), dict(scale_factor=sf, pad=(top, ph - top, left, pw - left), letterboxed=canvas)


# ----------------------------------------------------------------------------- post-process
def nms_class_aware(boxes: np.ndarray, scores: np.ndarray, labels: np.ndarray, iou_thr: float = 0.7, max_keep: Optional[int] = None,
                    n_use: Optional[int] = None) -> np.ndarray:
    """mmcv batched_nms (class-aware via the coordinate-offset trick, float32) + greedy NMS; returns kept indices in
    descending-score order.  Candidates must already be sorted by descending score.  ``max_keep``: stop once that many
    boxes are kept -- greedy NMS decides candidates in order, so the first ``max_keep`` survivors do not depend on
    anything after them (mmdet truncates to max_per_img AFTER the full pass; same result)."""
    if len(boxes) == 0:
        return np.zeros(0, dtype=np.int64)
    b = boxes.astype(np.float32)
    off = labels.astype(np.float32) * (b.max() + np.float32(1))
    b = b + off[:, None]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep: List[int] = []
    kb = np.zeros((0, 4), np.float32)
    ka = np.zeros(0, np.float32)
    thr = np.float32(iou_thr)
    for i in range(len(b) if n_use is None else min(n_use, len(b))):          # the offsets above come from ALL boxes
        if len(keep):
            iw = np.maximum(np.minimum(b[i, 2], kb[:, 2]) - np.maximum(b[i, 0], kb[:, 0]), np.float32(0))
            ih = np.maximum(np.minimum(b[i, 3], kb[:, 3]) - np.maximum(b[i, 1], kb[:, 1]), np.float32(0))
            inter = iw * ih
            with np.errstate(invalid="ignore", divide="ignore"):
                if np.any(inter / (area[i] + ka - inter) > thr):
                    continue
        keep.append(i)
        kb = np.concatenate([kb, b[i:i + 1]])
        ka = np.append(ka, area[i])
        if max_keep is not None and len(keep) >= max_keep:
            break
    return np.asarray(keep, dtype=np.int64)


def dense_decode(levels, geom: Dict):
    """One image.  levels: [(cls_logit [K,H,W], bbox_dist [4,H,W])] -> (scores f32 [A,K] after the sigmoid, boxes f32 [A,4]
    decoded from the point priors (offset 0.5) and un-letterboxed to the passed image's pixels, unclamped)."""
    pri, strd, cls, dist = [], [], [], []
    for (lg, d), s in zip(levels, STRIDES):
        K, H, W = lg.shape
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        pri.append(torch.stack([(xs.reshape(-1) + 0.5) * s, (ys.reshape(-1) + 0.5) * s], 1))
        strd.append(torch.full((H * W,), float(s)))
        cls.append(lg.permute(1, 2, 0).reshape(H * W, K))
        dist.append(d.permute(1, 2, 0).reshape(H * W, 4))
    pri, strd, scores, dist = torch.cat(pri), torch.cat(strd), torch.cat(cls).sigmoid(), torch.cat(dist)
    dd = dist * strd[:, None]
    boxes = torch.stack([pri[:, 0] - dd[:, 0], pri[:, 1] - dd[:, 1], pri[:, 0] + dd[:, 2], pri[:, 1] + dd[:, 3]], 1)
    top, _, left, _ = geom["pad"]
    sfw, sfh = geom["scale_factor"]
    boxes = (boxes - torch.tensor([left, top, left, top], dtype=torch.float32)) / torch.tensor([sfw, sfh, sfw, sfh], dtype=torch.float32)
    return scores.numpy(), boxes.numpy()


def select(sc: np.ndarray, boxes: np.ndarray, ori_hw, score_thr=0.001, nms_pre=30000, iou_thr=0.7, max_per_img=300,
           wrapper_thr=0.12, max_dets=50):
    """mmyolo predict_by_feat after the decode (multi_label) + the reference wrapper, on dense scores [A,K] / boxes [A,4]:
    candidates = every (anchor, class) pair with score > score_thr, descending score (equal scores in (anchor, class)
    order), the first nms_pre; class-aware NMS; max_per_img; clamp; then score > wrapper_thr and the max_dets best."""
    sc = np.asarray(sc, dtype=np.float32)
    boxes = np.asarray(boxes, dtype=np.float32)
    valid = np.argwhere(sc > np.float32(score_thr))
    vs = sc[valid[:, 0], valid[:, 1]]
    order = np.argsort(-vs, kind="stable")[:nms_pre]
    a_idx, labels, vs = valid[order, 0], valid[order, 1], vs[order]
    cand = boxes[a_idx]
    # survivors at or below the wrapper threshold are dropped afterwards and cannot influence earlier ones: stop there
    n_use = int(np.count_nonzero(vs > np.float32(wrapper_thr))) if wrapper_thr >= score_thr else len(vs)
    keep = nms_class_aware(cand, vs, labels, iou_thr, max_keep=max_per_img, n_use=n_use)
    kb = cand[keep].copy()
    kb[:, 0::2] = np.clip(kb[:, 0::2], 0, ori_hw[1])
    kb[:, 1::2] = np.clip(kb[:, 1::2], 0, ori_hw[0])
    ks, kl, ka = vs[keep], labels[keep], a_idx[keep]
    m = ks > np.float32(wrapper_thr)                                          # interface_heuristic.py:148-152
    kb, ks, kl, ka = kb[m], ks[m], kl[m], ka[m]
    if len(ks) > max_dets:
        t = np.argsort(-ks, kind="stable")[:max_dets]
        kb, ks, kl, ka = kb[t], ks[t], kl[t], ka[t]
    return dict(xyxy=kb.astype(np.float32), scores=ks.astype(np.float32), labels=kl.astype(np.int64), anchors=ka.astype(np.int64),
                n_candidates=int(len(valid)))


def post_process(levels, geom: Dict, ori_hw, **kw):
    sc, boxes = dense_decode(levels, geom)
    r = select(sc, boxes, ori_hw, **kw)
    r["dense_scores"], r["dense_boxes"] = sc, boxes
    return r


def detect(sd, images: Sequence[np.ndarray], text_feats: np.ndarray, **kw):
    """images: uint8 RGB [h,w,3] (equal sizes); text_feats f32 [K,512] (CLIP text embeddings, L2-normalised by the
    backbone).  Returns one post_process dict per image (plus 'logits' / 'dists' per level for tests)."""
    xs, geoms = zip(*[letterbox(im) for im in images])
    x = torch.from_numpy(np.stack(xs))
    B = x.shape[0]
    t = torch.from_numpy(np.asarray(text_feats, dtype=np.float32))[None].expand(B, -1, -1)
    with torch.no_grad():
        lv = head(sd, neck(sd, backbone(sd, x), t), t)
    out = []
    for b in range(B):
        r = post_process([(lg[b], d[b]) for lg, d in lv], geoms[b], images[b].shape[:2], **kw)
        r["levels"] = [(lg[b].numpy(), d[b].numpy()) for lg, d in lv]
        r["geom"] = geoms[b]
        out.append(r)
    return out
