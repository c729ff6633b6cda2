"""YOLO-World-v2 (image side) for the HIP detector: architecture tables, parameter sets and the layer program
the C library executes (tstar_yolo_* in include/tstar_hip.h).

The reference wires YOLO-World through mmdet/mmyolo and a repository cloned at install time
(/root/reference/TStar/TStarFramework.py:178-184, install.sh:12; config
``yolo_world_v2_xl_vlpan_bn_2e-3_100e_4x8gpus_obj365v1_goldg_train_lvis_minival.py``); NONE of that source is in
the reference tree (``/root/reference/YOLO-World`` is a dangling symlink), so this module restates the PUBLISHED
architecture from upstream knowledge -- YOLOv8 CSPDarknet backbone, text-guided PAFPN with max-sigmoid attention
(``MaxSigmoidCSPLayerWithTwoConv``), BN-contrastive head against CLIP text features, DFL box decode (reg_max 16) --
and its parity against the real model is UNPINNED.  What IS checked: the HIP path against the CPU
oracle (oracle/yolo_ref.py, an independent torch statement of the same architecture) on the same seeded weights.

Parameter names follow mmyolo / YOLO-World module paths (``backbone.image_model.stage2.1.blocks.0.conv1.conv.weight``,
``neck.top_down_layers.0.attn_block.guide_fc.weight``, ``bbox_head.head_module.cls_contrasts.1.norm.running_var`` ...)
so that a real checkpoint's ``state_dict`` can be handed to ``build_program`` unchanged.

The library is layout-agnostic: this module flattens the network into
  * one float32 blob (conv weights re-laid-out as [Cout][kh][kw][Cin], eval-mode BatchNorm folded into weight + bias
    in float64),
  * a table of ops (conv / max-pool / nearest-upsample-copy / max-sigmoid attention) over NHWC activation buffers --
    channel concatenation is free: producers write at a channel offset of the consumer's buffer,
  * per-level head descriptors (BN-contrastive classifier, DFL regressor, stride).
numpy only at import time.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

IMG_SIZE = 640
REG_MAX = 16
TEXT_DIM = 512
BN_EPS = 1e-3                       # mmyolo norm_cfg: BN eps 0.001, momentum 0.03
STRIDES = (8, 16, 32)
NUM_TRAINING_CLASSES = 80           # head_module.num_classes of the pretrain configs (sizes the cls tower)

# scale -> (deepen, widen, last_stage_out_channels)   (mmyolo yolov8_{s,m,l,x}; YOLO-World keeps them).  "xl" is the
# published yolo_world_v2_xl config -- the one the reference wires (TStarFramework.py:181-182) -- which scales the
# yolov8_x base "from X to XL": deepen 1.0, widen 1.5 (96-channel stem, 192/384/768/768 stages, 6/12/12 attention heads)
SCALES = {"s": (0.33, 0.5, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512), "x": (1.0, 1.25, 512), "xl": (1.0, 1.5, 512)}

# op codes of the layer program (mirrored in csrc/yolo.hip)
OP_CONV, OP_POOL5, OP_UPCOPY, OP_ATTN = 0, 1, 2, 3
ACT_NONE, ACT_SILU = 0, 1
MODE_PLAIN, MODE_RESIDUAL, MODE_ATTN_MUL = 0, 1, 2
OP_WORDS = 24


def make_divisible(x: float, widen: float, divisor: int = 8) -> int:
    return int(math.ceil(x * widen / divisor) * divisor)


def make_round(x: int, factor: float) -> int:
    return max(round(x * factor), 1) if x > 1 else x


def arch(scale: str = "l") -> Dict:
    """Channel / depth tables of YOLO-World-v2-<scale> (vlpan, bn head)."""
    if scale not in SCALES:
        raise ValueError(f"unknown YOLO-World scale {scale!r} (one of {sorted(SCALES)})")
    d, w, last = SCALES[scale]
    stage = [[64, 128, 3, True, False], [128, 256, 6, True, False], [256, 512, 6, True, False], [512, last, 3, True, True]]
    stages = [dict(cin=make_divisible(i, w), cout=make_divisible(o, w), n=make_round(n, d), add=a, spp=s) for i, o, n, a, s in stage]
    in_ch = [make_divisible(c, w) for c in (256, 512, last)]
    embed = [make_round(c, w) for c in (128, 256, last // 2)]
    heads = [make_round(c, w) for c in (4, 8, last // 2 // 32)]
    return dict(scale=scale, deepen=d, widen=w, stem=make_divisible(64, w), stages=stages, in_channels=in_ch, out_channels=list(in_ch),
                neck_blocks=make_round(3, d), embed=embed, heads=heads, reg_ch=max(16, in_ch[0] // 4, REG_MAX * 4),
                cls_ch=max(in_ch[0], NUM_TRAINING_CLASSES))


# ------------------------------------------------------------------------------------------ parameters
def _conv_module(sd, rs, name: str, cin: int, cout: int, k: int, gain: float = 1.0):
    """ConvModule = Conv2d(bias=False) + BatchNorm2d (+ SiLU): He-style weights, non-trivial BN statistics."""
    fan_in = cin * k * k
    sd[name + ".conv.weight"] = (rs.standard_normal((cout, cin, k, k)) * (gain * math.sqrt(2.0 / fan_in))).astype(np.float32)
    sd[name + ".bn.weight"] = (0.8 + 0.4 * rs.random_sample(cout)).astype(np.float32)
    sd[name + ".bn.bias"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    sd[name + ".bn.running_mean"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    sd[name + ".bn.running_var"] = (0.8 + 0.4 * rs.random_sample(cout)).astype(np.float32)


def _c2f(sd, rs, name, cin, cout, n, attn=None):
    mid = int(cout * 0.5)
    _conv_module(sd, rs, name + ".main_conv", cin, 2 * mid, 1)
    for i in range(n):
        _conv_module(sd, rs, f"{name}.blocks.{i}.conv1", mid, mid, 3)
        _conv_module(sd, rs, f"{name}.blocks.{i}.conv2", mid, mid, 3)
    extra = 1 if attn else 0
    _conv_module(sd, rs, name + ".final_conv", (2 + n + extra) * mid, cout, 1)
    if attn:
        embed, heads = attn
        a = name + ".attn_block"
        if embed != mid:
            _conv_module(sd, rs, a + ".embed_conv", mid, embed, 1)
        sd[a + ".guide_fc.weight"] = (rs.standard_normal((embed, TEXT_DIM)) * math.sqrt(1.0 / TEXT_DIM) * 4.0).astype(np.float32)
        sd[a + ".guide_fc.bias"] = (0.1 * rs.standard_normal(embed)).astype(np.float32)
        sd[a + ".bias"] = (0.3 * rs.standard_normal(heads)).astype(np.float32)
        _conv_module(sd, rs, a + ".project_conv", mid, mid, 3, gain=1.4)     # the sigmoid gate halves the scale


def synthetic_state_dict(seed: int = 0, scale: str = "l") -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic parameters with the mmyolo / YOLO-World names (no checkpoint can be downloaded).  One frozen
    ``RandomState`` stream, so the CPU oracle, the tests and the GPU box regenerate identical tensors.  The head's
    contrastive bias is placed so that a fraction of the 8400 x Q (anchor, class) pairs clears the wrapper's 0.12
    threshold and class-aware NMS has real work."""
    A = arch(scale)
    rs = np.random.RandomState(7919 + seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    b = "backbone.image_model."
    _conv_module(sd, rs, b + "stem", 3, A["stem"], 3)
    for si, st in enumerate(A["stages"], start=1):
        _conv_module(sd, rs, f"{b}stage{si}.0", st["cin"], st["cout"], 3)
        _c2f(sd, rs, f"{b}stage{si}.1", st["cout"], st["cout"], st["n"])
        if st["spp"]:
            _conv_module(sd, rs, f"{b}stage{si}.2.conv1", st["cout"], st["cout"] // 2, 1)
            _conv_module(sd, rs, f"{b}stage{si}.2.conv2", st["cout"] // 2 * 4, st["cout"], 1)
    ic, oc, nb = A["in_channels"], A["out_channels"], A["neck_blocks"]
    # top_down_layers[0] serves idx = 2 (output level 1), [1] serves idx = 1 (output level 0)
    for li, idx in enumerate((2, 1)):
        _c2f(sd, rs, f"neck.top_down_layers.{li}", ic[idx - 1] + (ic[idx] if idx == 2 else oc[idx]), oc[idx - 1], nb,
             attn=(A["embed"][idx - 1], A["heads"][idx - 1]))
    for idx in (0, 1):
        _conv_module(sd, rs, f"neck.downsample_layers.{idx}", oc[idx], oc[idx], 3)
        _c2f(sd, rs, f"neck.bottom_up_layers.{idx}", oc[idx] + oc[idx + 1], oc[idx + 1], nb, attn=(A["embed"][idx + 1], A["heads"][idx + 1]))
    h = "bbox_head.head_module."
    for i in range(3):
        _conv_module(sd, rs, f"{h}cls_preds.{i}.0", oc[i], A["cls_ch"], 3)
        _conv_module(sd, rs, f"{h}cls_preds.{i}.1", A["cls_ch"], A["cls_ch"], 3)
        sd[f"{h}cls_preds.{i}.2.weight"] = (rs.standard_normal((TEXT_DIM, A["cls_ch"], 1, 1)) * math.sqrt(1.0 / A["cls_ch"])).astype(np.float32)
        sd[f"{h}cls_preds.{i}.2.bias"] = (0.1 * rs.standard_normal(TEXT_DIM)).astype(np.float32)
        _conv_module(sd, rs, f"{h}reg_preds.{i}.0", oc[i], A["reg_ch"], 3)
        _conv_module(sd, rs, f"{h}reg_preds.{i}.1", A["reg_ch"], A["reg_ch"], 3)
        sd[f"{h}reg_preds.{i}.2.weight"] = (rs.standard_normal((4 * REG_MAX, A["reg_ch"], 1, 1)) * math.sqrt(2.0 / A["reg_ch"])).astype(np.float32)
        sd[f"{h}reg_preds.{i}.2.bias"] = (0.5 * rs.standard_normal(4 * REG_MAX)).astype(np.float32)
        c = f"{h}cls_contrasts.{i}."
        sd[c + "norm.weight"] = (0.8 + 0.4 * rs.random_sample(TEXT_DIM)).astype(np.float32)
        sd[c + "norm.bias"] = (0.1 * rs.standard_normal(TEXT_DIM)).astype(np.float32)
        sd[c + "norm.running_mean"] = (0.1 * rs.standard_normal(TEXT_DIM)).astype(np.float32)
        sd[c + "norm.running_var"] = (0.8 + 0.4 * rs.random_sample(TEXT_DIM)).astype(np.float32)
        sd[c + "bias"] = np.float32(-2.6 + 0.2 * i) * np.ones((), np.float32)
        sd[c + "logit_scale"] = np.float32(1.6 - 0.4 * i) * np.ones((), np.float32)
    return sd


# ------------------------------------------------------------------------------------------ layer program
class _Builder:
    def __init__(self, sd, A=None):
        self.sd = sd
        self.A = A or {"scale": "?"}
        self.blob: List[np.ndarray] = []
        self.n = 0
        self.ops: List[List[int]] = []
        self.bufs: List[Tuple[int, int, int]] = []          # (H, W, C) per image
        self.guides: List[Dict] = []                        # max-sigmoid attention layers (guide_fc + per-head bias)

    def put(self, a) -> int:
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        off = self.n
        pad = (-a.size) % 4                                  # keep every tensor 16-byte aligned
        self.blob.append(a)
        if pad:
            self.blob.append(np.zeros(pad, np.float32))
        self.n += a.size + pad
        return off

    def buf(self, H, W, C) -> int:
        self.bufs.append((H, W, C))
        return len(self.bufs) - 1

    def folded(self, name):
        """ConvModule -> (weight [Cout][kh][kw][Cin], bias [Cout]) with the eval-mode BatchNorm folded in float64."""
        if name + ".conv.weight" not in self.sd:
            raise ValueError(f"YOLO-World state dict has no tensor {name}.conv.weight: not a YOLO-World-v2-{self.A['scale'].upper()} "
                             f"model (wrong scale= / config name?)")
        w = np.asarray(self.sd[name + ".conv.weight"], dtype=np.float64)
        g, b = np.asarray(self.sd[name + ".bn.weight"], np.float64), np.asarray(self.sd[name + ".bn.bias"], np.float64)
        m, v = np.asarray(self.sd[name + ".bn.running_mean"], np.float64), np.asarray(self.sd[name + ".bn.running_var"], np.float64)
        s = g / np.sqrt(v + BN_EPS)
        return (w * s[:, None, None, None]).transpose(0, 2, 3, 1), b - m * s

    def conv(self, name, src, src_off, dst, dst_off, stride=1, act=ACT_SILU, mode=MODE_PLAIN, aux=-1, aux_off=0, raw=None):
        """Append a conv op; ``raw`` = (weight OIHW, bias or None) for plain nn.Conv2d layers."""
        if raw is None:
            w, b = self.folded(name)
        else:
            w = np.asarray(raw[0], np.float64).transpose(0, 2, 3, 1)
            b = None if raw[1] is None else np.asarray(raw[1], np.float64)
        cout, ks, _, cin = w.shape
        H, W, _ = self.bufs[src]
        Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
        what = name or "raw conv"
        if self.bufs[dst][:2] != (Ho, Wo):
            raise ValueError(f"YOLO-World layer {what}: output map {Ho}x{Wo} does not fit its buffer {self.bufs[dst][0]}x{self.bufs[dst][1]}")
        if src_off + cin > self.bufs[src][2] or dst_off + cout > self.bufs[dst][2]:
            raise ValueError(
                f"YOLO-World layer {what}: weight {tuple(w.shape[i] for i in (0, 3, 1, 2))} (Cout, Cin, k, k) expects {cin} input channels at "
                f"offset {src_off} of a {self.bufs[src][2]}-channel buffer and writes {cout} at offset {dst_off} of a {self.bufs[dst][2]}-channel "
                f"buffer: the state dict is not a YOLO-World-v2-{self.A['scale'].upper()} model (wrong scale= / config name?)")
        w_off = self.put(w)
        b_off = self.put(b) if b is not None else -1
        self.ops.append([OP_CONV, src, src_off, cin, dst, dst_off, cout, ks, stride, act, w_off, b_off, mode, aux, aux_off])

    def c2f(self, name, src, cin, cout, n, add, attn_id=None):
        """CSPLayerWithTwoConv / MaxSigmoidCSPLayerWithTwoConv: every branch writes into ONE concat buffer."""
        H, W, _ = self.bufs[src]
        mid = int(cout * 0.5)
        has_attn = (name + ".attn_block.guide_fc.weight") in self.sd
        cat = self.buf(H, W, (2 + n + (1 if has_attn else 0)) * mid)
        tmp = self.buf(H, W, mid)
        self.conv(name + ".main_conv", src, 0, cat, 0)
        for i in range(n):
            self.conv(f"{name}.blocks.{i}.conv1", cat, (1 + i) * mid, tmp, 0)
            self.conv(f"{name}.blocks.{i}.conv2", tmp, 0, cat, (2 + i) * mid, mode=MODE_RESIDUAL if add else MODE_PLAIN,
                      aux=cat if add else -1, aux_off=(1 + i) * mid)
        if has_attn:
            a = name + ".attn_block"
            gw, gb = self.sd[a + ".guide_fc.weight"], self.sd[a + ".guide_fc.bias"]
            embed = gw.shape[0]
            heads = int(np.asarray(self.sd[a + ".bias"]).size)
            last_off = (1 + n) * mid
            esrc, eoff = cat, last_off
            if (a + ".embed_conv.conv.weight") in self.sd:
                eb = self.buf(H, W, embed)
                self.conv(a + ".embed_conv", cat, last_off, eb, 0, act=ACT_NONE)
                esrc, eoff = eb, 0
            gid = len(self.guides)
            self.guides.append(dict(embed=embed, heads=heads, w_off=self.put(gw), b_off=self.put(gb), bias_off=self.put(self.sd[a + ".bias"])))
            ab = self.buf(H, W, heads)
            self.ops.append([OP_ATTN, esrc, eoff, embed, ab, 0, heads, gid])
            self.conv(a + ".project_conv", cat, last_off, cat, (2 + n) * mid, act=ACT_NONE, mode=MODE_ATTN_MUL, aux=ab, aux_off=0)
        out = self.buf(H, W, cout)
        self.conv(name + ".final_conv", cat, 0, out, 0)
        return out


def build_program(state_dict, scale: str = "l") -> Dict:
    """-> dict(blob f32, ops int32 [n, OP_WORDS], bufs int32 [m, 3], guides int32 [a, 5], levels int32 [3, 8], arch)."""
    A = arch(scale)
    B = _Builder(state_dict, A)
    S = IMG_SIZE
    b = "backbone.image_model."
    # the widths of the stem and of every stage's down-sampling conv identify the scale: name the first one that differs
    # (a narrower model would otherwise fit silently inside the wider model's buffers)
    want = [(b + "stem", A["stem"], 3)] + [(f"{b}stage{i}.0", st["cout"], st["cin"]) for i, st in enumerate(A["stages"], start=1)]
    for nm, cout, cin in want:
        t = state_dict.get(nm + ".conv.weight")
        if t is None or tuple(np.shape(t)[:2]) != (cout, cin):
            raise ValueError(f"YOLO-World tensor {nm}.conv.weight: expected (Cout, Cin) = ({cout}, {cin}) for YOLO-World-v2-{scale.upper()} "
                             f"(widen {A['widen']}), got {None if t is None else tuple(np.shape(t))}: wrong scale= / config name for this checkpoint?")
    x = B.buf(S, S, 3)
    cur = B.buf(S // 2, S // 2, A["stem"])
    B.conv(b + "stem", x, 0, cur, 0, stride=2)
    feats = []
    size = S // 2
    for si, st in enumerate(A["stages"], start=1):
        size //= 2
        d = B.buf(size, size, st["cout"])
        B.conv(f"{b}stage{si}.0", cur, 0, d, 0, stride=2)
        cur = B.c2f(f"{b}stage{si}.1", d, st["cout"], st["cout"], st["n"], st["add"])
        if st["spp"]:
            c = st["cout"]
            cat = B.buf(size, size, c // 2 * 4)
            B.conv(f"{b}stage{si}.2.conv1", cur, 0, cat, 0)
            for j in range(3):
                B.ops.append([OP_POOL5, cat, j * (c // 2), c // 2, cat, (j + 1) * (c // 2)])
            o = B.buf(size, size, c)
            B.conv(f"{b}stage{si}.2.conv2", cat, 0, o, 0)
            cur = o
        if si >= 2:
            feats.append(cur)
    ic, oc, nb = A["in_channels"], A["out_channels"], A["neck_blocks"]
    sizes = [S // s for s in STRIDES]

    def cat2(first, c_first, factor, second, c_second, size):
        """torch.cat([first (optionally upsampled x2), second], dim=1) as two copies into one buffer."""
        cb = B.buf(size, size, c_first + c_second)
        B.ops.append([OP_UPCOPY, first, 0, c_first, cb, 0, factor])
        B.ops.append([OP_UPCOPY, second, 0, c_second, cb, c_first, 1])
        return cb

    inner = [None, None, feats[2]]
    cb = cat2(feats[2], ic[2], 2, feats[1], ic[1], sizes[1])                       # upsample_feats_cat_first
    inner[1] = B.c2f("neck.top_down_layers.0", cb, ic[2] + ic[1], oc[1], nb, False)
    cb = cat2(inner[1], oc[1], 2, feats[0], ic[0], sizes[0])
    inner[0] = B.c2f("neck.top_down_layers.1", cb, oc[1] + ic[0], oc[0], nb, False)
    outs = [inner[0]]
    for idx in (0, 1):
        dn = B.buf(sizes[idx + 1], sizes[idx + 1], oc[idx])
        B.conv(f"neck.downsample_layers.{idx}", outs[-1], 0, dn, 0, stride=2)
        cb = cat2(dn, oc[idx], 1, inner[idx + 1], oc[idx + 1] if idx == 0 else ic[2], sizes[idx + 1])
        outs.append(B.c2f(f"neck.bottom_up_layers.{idx}", cb, 0, oc[idx + 1], nb, False))
    h = "bbox_head.head_module."
    levels = []
    for i in range(3):
        sz = sizes[i]
        t1, t2 = B.buf(sz, sz, A["cls_ch"]), B.buf(sz, sz, A["cls_ch"])
        e = B.buf(sz, sz, TEXT_DIM)
        B.conv(f"{h}cls_preds.{i}.0", outs[i], 0, t1, 0)
        B.conv(f"{h}cls_preds.{i}.1", t1, 0, t2, 0)
        # 1x1 conv (with bias) followed by the contrastive head's BatchNorm2d: folded into one affine map (float64)
        c = f"{h}cls_contrasts.{i}."
        g_, b_ = np.asarray(state_dict[c + "norm.weight"], np.float64), np.asarray(state_dict[c + "norm.bias"], np.float64)
        m_, v_ = np.asarray(state_dict[c + "norm.running_mean"], np.float64), np.asarray(state_dict[c + "norm.running_var"], np.float64)
        s_ = g_ / np.sqrt(v_ + BN_EPS)
        w2 = np.asarray(state_dict[f"{h}cls_preds.{i}.2.weight"], np.float64) * s_[:, None, None, None]
        b2 = (np.asarray(state_dict[f"{h}cls_preds.{i}.2.bias"], np.float64) - m_) * s_ + b_
        B.conv(None, t2, 0, e, 0, act=ACT_NONE, raw=(w2, b2))
        r1, r2 = B.buf(sz, sz, A["reg_ch"]), B.buf(sz, sz, A["reg_ch"])
        r = B.buf(sz, sz, 4 * REG_MAX)
        B.conv(f"{h}reg_preds.{i}.0", outs[i], 0, r1, 0)
        B.conv(f"{h}reg_preds.{i}.1", r1, 0, r2, 0)
        B.conv(None, r2, 0, r, 0, act=ACT_NONE, raw=(state_dict[f"{h}reg_preds.{i}.2.weight"], state_dict[f"{h}reg_preds.{i}.2.bias"]))
        ls = float(np.exp(np.float32(state_dict[c + "logit_scale"])))            # x * logit_scale.exp() + bias
        levels.append([e, r, sz, STRIDES[i], B.put([ls, float(np.asarray(state_dict[c + "bias"]))]), 0, 0, 0])
    ops = np.zeros((len(B.ops), OP_WORDS), dtype=np.int32)
    for i, o in enumerate(B.ops):
        ops[i, :len(o)] = o
    guides = np.array([[g["embed"], g["heads"], g["w_off"], g["b_off"], g["bias_off"]] for g in B.guides], dtype=np.int32)
    return dict(blob=np.concatenate(B.blob).astype(np.float32), ops=ops, bufs=np.array(B.bufs, dtype=np.int32), guides=guides,
                levels=np.array(levels, dtype=np.int32), arch=A, input_buf=x)


def conv_flops(prog) -> float:
    """Algorithmic multiply-add flops (x2) of every conv of one 640x640 image."""
    total = 0.0
    for o in prog["ops"]:
        if o[0] == OP_CONV:
            _, _, _, cin, dst, _, cout, ks, *_ = o
            H, W, _ = prog["bufs"][dst]
            total += 2.0 * H * W * cout * cin * ks * ks
    return total


# ------------------------------------------------------------------------------------------ letterbox geometry
def letterbox_geometry(h: int, w: int) -> Dict:
    """mmyolo test pipeline geometry for an h x w image (YOLOv5KeepRatioResize(640) + LetterResize(640, allow_scale_up=False,
    pad_val=114)): resized size, scale_factor (w, h), pad_param (top, bottom, left, right)."""
    ratio = min(IMG_SIZE / max(h, w), IMG_SIZE / min(h, w))
    rw, rh = (int(w * ratio), int(h * ratio)) if ratio != 1 else (w, h)
    sf = (rw / w, rh / h)
    r2 = min(min(IMG_SIZE / rh, IMG_SIZE / rw), 1.0)
    nh, nw = int(round(rh * r2)), int(round(rw * r2))
    sf = (sf[0] * (nw / rw), sf[1] * (nh / rh))
    ph, pw = IMG_SIZE - nh, IMG_SIZE - nw
    top, left = int(round(ph // 2 - 0.1)), int(round(pw // 2 - 0.1))
    return dict(ratio=ratio, resized=(nh, nw), first_resize=(rh, rw), scale_factor=sf, pad=(top, ph - top, left, pw - left),
                interp="area" if ratio < 1 else "bilinear")
