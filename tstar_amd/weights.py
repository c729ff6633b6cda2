"""OWL-ViT-B/32 parameter sets for the HIP scorer.

The reference loads ``google/owlvit-base-patch32`` through HF transformers
(/root/reference/TStar/interface_heuristic.py:207-210, TStarFramework.py:176).
The HIP library takes ONE flat little-endian float32 blob whose layout is
fixed by ``vision_spec()`` / ``text_spec()`` below (mirrored entry by entry in
``tstar_amd/csrc/owl_weights.h``).  This module

* builds that blob from a HF-style ``state_dict`` (names as in
  ``transformers/models/owlvit/modeling_owlvit.py``), either real weights found
  on disk (safetensors) or
* seeded synthetic weights: ``numpy.random.RandomState`` (frozen legacy stream,
  so CPU oracle, tests and the GPU box regenerate identical parameters) with
  the HF initialiser std's (modeling_owlvit.py ``_init_weights``; drawn from a
  unit-variance uniform instead of a normal, which keeps the std), small
  non-zero biases / LayerNorm perturbations so every parameter is exercised,
  and class-head shift/scale scaled by 0.01 so scores do not saturate
  (SURVEY.md 8c caveat (c)).

No torch import at module import time; numpy only.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

# ---- architecture constants (configuration_owlvit.py defaults = base-patch32) ----
V_D = 768          # vision hidden
V_FF = 3072
V_LAYERS = 12
V_HEADS = 12
IMG = 768
PATCH = 32
GRID = IMG // PATCH            # 24
NPATCH = GRID * GRID           # 576
NTOK = NPATCH + 1              # 577
T_D = 512          # text hidden
T_FF = 2048
T_LAYERS = 12
T_HEADS = 8
T_LEN = 16
VOCAB = 49408
PROJ = 512
LN_EPS = 1e-5

Spec = List[Tuple[str, Tuple[int, ...], Tuple[str, ...]]]


def _layer_spec(prefix: str, d: int, ff: int) -> Spec:
    p = prefix
    return [
        (p + "ln1_w", (d,), (p + "layer_norm1.weight",)),
        (p + "ln1_b", (d,), (p + "layer_norm1.bias",)),
        (p + "qkv_w", (3 * d, d), (p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight")),
        (p + "qkv_b", (3 * d,), (p + "self_attn.q_proj.bias", p + "self_attn.k_proj.bias", p + "self_attn.v_proj.bias")),
        (p + "out_w", (d, d), (p + "self_attn.out_proj.weight",)),
        (p + "out_b", (d,), (p + "self_attn.out_proj.bias",)),
        (p + "ln2_w", (d,), (p + "layer_norm2.weight",)),
        (p + "ln2_b", (d,), (p + "layer_norm2.bias",)),
        (p + "fc1_w", (ff, d), (p + "mlp.fc1.weight",)),
        (p + "fc1_b", (ff,), (p + "mlp.fc1.bias",)),
        (p + "fc2_w", (d, ff), (p + "mlp.fc2.weight",)),
        (p + "fc2_b", (d,), (p + "mlp.fc2.bias",)),
    ]


def vision_spec() -> Spec:
    """(blob entry name, shape, HF state_dict names concatenated along dim 0)."""
    vm = "owlvit.vision_model."
    s: Spec = [
        ("patch_w", (V_D, 3 * PATCH * PATCH), (vm + "embeddings.patch_embedding.weight",)),
        ("class_emb", (V_D,), (vm + "embeddings.class_embedding",)),
        ("pos_emb", (NTOK, V_D), (vm + "embeddings.position_embedding.weight",)),
        ("pre_ln_w", (V_D,), (vm + "pre_layernorm.weight",)),
        ("pre_ln_b", (V_D,), (vm + "pre_layernorm.bias",)),
    ]
    for i in range(V_LAYERS):
        s += _layer_spec(f"{vm}encoder.layers.{i}.", V_D, V_FF)
    s += [
        ("post_ln_w", (V_D,), (vm + "post_layernorm.weight",)),
        ("post_ln_b", (V_D,), (vm + "post_layernorm.bias",)),
        ("det_ln_w", (V_D,), ("layer_norm.weight",)),
        ("det_ln_b", (V_D,), ("layer_norm.bias",)),
        ("cls_w", (PROJ, V_D), ("class_head.dense0.weight",)),
        ("cls_b", (PROJ,), ("class_head.dense0.bias",)),
        ("shift_w", (V_D,), ("class_head.logit_shift.weight",)),
        ("shift_b", (1,), ("class_head.logit_shift.bias",)),
        ("scale_w", (V_D,), ("class_head.logit_scale.weight",)),
        ("scale_b", (1,), ("class_head.logit_scale.bias",)),
        ("box0_w", (V_D, V_D), ("box_head.dense0.weight",)),
        ("box0_b", (V_D,), ("box_head.dense0.bias",)),
        ("box1_w", (V_D, V_D), ("box_head.dense1.weight",)),
        ("box1_b", (V_D,), ("box_head.dense1.bias",)),
        ("box2_w", (4, V_D), ("box_head.dense2.weight",)),
        ("box2_b", (4,), ("box_head.dense2.bias",)),
        ("box_bias", (NPATCH, 4), ("box_bias",)),
    ]
    return s


def text_spec() -> Spec:
    tm = "owlvit.text_model."
    s: Spec = [
        ("tok_emb", (VOCAB, T_D), (tm + "embeddings.token_embedding.weight",)),
        ("tpos_emb", (T_LEN, T_D), (tm + "embeddings.position_embedding.weight",)),
    ]
    for i in range(T_LAYERS):
        s += _layer_spec(f"{tm}encoder.layers.{i}.", T_D, T_FF)
    s += [
        ("final_ln_w", (T_D,), (tm + "final_layer_norm.weight",)),
        ("final_ln_b", (T_D,), (tm + "final_layer_norm.bias",)),
        ("text_proj", (PROJ, T_D), ("owlvit.text_projection.weight",)),
    ]
    return s


def spec_size(spec: Spec) -> int:
    return int(sum(int(np.prod(shape)) for _, shape, _ in spec))


def compute_box_bias() -> np.ndarray:
    """box_bias buffer, restated from modeling_owlvit.py:1072-1104.

    xy = ((col+1)/24, (row+1)/24); bias = log(v+1e-4) - log1p(-v+1e-4); size
    entries use v = 1/24.  Row-major over the 24x24 patch grid.  Uses torch
    float32 ops (as HF does) so the buffer is bit-identical to the one HF
    builds at model init; torch is imported lazily.
    """
    import torch
    xs = torch.arange(1, GRID + 1, dtype=torch.float32)
    xx, yy = torch.meshgrid(xs, xs, indexing="xy")
    coords = torch.stack((xx, yy), dim=-1)
    coords[..., 0] /= GRID
    coords[..., 1] /= GRID
    coords = torch.clip(coords.view(-1, 2), 0.0, 1.0)
    cb = torch.log(coords + 1e-4) - torch.log1p(-coords + 1e-4)
    size = torch.full_like(cb, 1.0)
    size[..., 0] /= GRID
    size[..., 1] /= GRID
    sb = torch.log(size + 1e-4) - torch.log1p(-size + 1e-4)
    return torch.cat([cb, sb], dim=-1).numpy().astype(np.float32)


def _std_for(name: str, shape: Tuple[int, ...]) -> float:
    """HF ``_init_weights`` std per parameter (modeling_owlvit.py:532-565)."""
    vision = "vision_model" in name
    d = V_D if vision else T_D
    layers = V_LAYERS if vision else T_LAYERS
    if name.endswith("class_embedding"):
        return d ** -0.5
    if "embedding" in name:
        return 0.02
    if any(k in name for k in ("q_proj.weight", "k_proj.weight", "v_proj.weight")):
        return (d ** -0.5) * ((2 * layers) ** -0.5)
    if "out_proj.weight" in name:
        return d ** -0.5
    if "fc1.weight" in name:
        return (2 * d) ** -0.5
    if "fc2.weight" in name:
        return (d ** -0.5) * ((2 * layers) ** -0.5)
    if "text_projection" in name:
        return T_D ** -0.5
    if name.startswith("class_head") or name.startswith("box_head"):
        return 0.02
    return 0.02


def synthetic_state_dict(seed: int = 0, towers: str = "both") -> Dict[str, np.ndarray]:
    """Seeded synthetic OWL-ViT-B/32 parameters keyed by HF state_dict names.

    Drawn from ONE ``RandomState(seed)`` stream in spec order (vision first),
    so the text tower does not depend on whether the vision tower was built:
    each tower uses its own stream (seed, seed + 1).
    """
    out: Dict[str, np.ndarray] = {}

    def fill(spec: Spec, rs: np.random.RandomState) -> None:
        for _, _, hf_names in spec:
            for hf in hf_names:
                shape = _hf_shape(hf)
                n = int(np.prod(shape))
                if hf == "box_bias":
                    out[hf] = compute_box_bias()
                    continue
                # unit-variance uniform: (u - 0.5) * sqrt(12); the legacy
                # random_sample stream is frozen and ~30x faster than gauss
                x = ((rs.random_sample(n) - 0.5) * 3.4641016151377544).astype(np.float32).reshape(shape)
                if hf.endswith(".weight") and "norm" in hf.split(".")[-2]:
                    x = (1.0 + 0.1 * x).astype(np.float32)
                elif hf.endswith(".bias"):
                    x = (0.02 * x).astype(np.float32)
                else:
                    x = (np.float32(_std_for(hf, shape)) * x).astype(np.float32)
                if hf.startswith("class_head.logit_shift") or hf.startswith("class_head.logit_scale"):
                    x = (x * np.float32(0.01)).astype(np.float32)
                out[hf] = x

    if towers in ("both", "vision"):
        fill(vision_spec(), np.random.RandomState(seed))
    if towers in ("both", "text"):
        fill(text_spec(), np.random.RandomState(seed + 1))
    return out


_HF_SHAPES: Dict[str, Tuple[int, ...]] = {}


def _hf_shape(hf: str) -> Tuple[int, ...]:
    if not _HF_SHAPES:
        for spec in (vision_spec(), text_spec()):
            for _, shape, hf_names in spec:
                k = len(hf_names)
                for h in hf_names:
                    if h.endswith("patch_embedding.weight"):
                        _HF_SHAPES[h] = (V_D, 3, PATCH, PATCH)
                    elif h.startswith("class_head.logit_s") and h.endswith("weight"):
                        _HF_SHAPES[h] = (1, V_D)
                    elif k > 1:
                        _HF_SHAPES[h] = (shape[0] // k,) + tuple(shape[1:])
                    else:
                        _HF_SHAPES[h] = tuple(shape)
    return _HF_SHAPES[hf]


def to_bf16_values(a: np.ndarray) -> np.ndarray:
    """Round float32 values to the nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def round_weights_to_bf16(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """BASELINE config 5 ("bf16 ViT weights"): every matrix / embedding table is rounded to bf16;
    biases, LayerNorm parameters and the box_bias buffer stay float32."""
    out = {}
    for k, v in sd.items():
        v = np.asarray(v, dtype=np.float32)
        out[k] = to_bf16_values(v) if (v.ndim >= 2 and k != "box_bias") else v
    return out


def pack_blob(sd: Dict[str, np.ndarray], spec: Spec) -> np.ndarray:
    """Concatenate ``sd`` entries into the flat f32 blob the C ABI expects."""
    parts = []
    for name, shape, hf_names in spec:
        if name == "box_bias" and "box_bias" not in sd:
            arr = compute_box_bias()
        else:
            arr = np.concatenate([np.asarray(sd[h], dtype=np.float32).reshape(-1) for h in hf_names])
        if arr.size != int(np.prod(shape)):
            raise ValueError(f"weight {name}: expected {shape}, got {arr.size} elements")
        parts.append(arr.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def unpack_blob(blob: np.ndarray, spec: Spec) -> Dict[str, np.ndarray]:
    """Blob-entry-name -> array view (for the oracle)."""
    out = {}
    off = 0
    for name, shape, _ in spec:
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    if off != blob.size:
        raise ValueError(f"blob has {blob.size} floats, spec wants {off}")
    return out


def find_pretrained(model_name_or_path: str = "google/owlvit-base-patch32"):
    """Return a path to a local safetensors checkpoint, or None.

    No network: only a local directory / HF cache hit counts.
    """
    cands = []
    if os.path.isdir(model_name_or_path):
        cands.append(os.path.join(model_name_or_path, "model.safetensors"))
    hub = os.path.expanduser(os.environ.get("HF_HOME", "~/.cache/huggingface"))
    snap = os.path.join(hub, "hub", "models--" + model_name_or_path.replace("/", "--"), "snapshots")
    if os.path.isdir(snap):
        for d in sorted(os.listdir(snap)):
            cands.append(os.path.join(snap, d, "model.safetensors"))
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def load_safetensors_state_dict(path: str) -> Dict[str, np.ndarray]:
    from safetensors.numpy import load_file
    sd = load_file(path)
    return {k: np.asarray(v, dtype=np.float32) for k, v in sd.items()}
