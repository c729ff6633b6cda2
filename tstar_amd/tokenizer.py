"""Query tokenisation for the OWL-ViT text tower.

The reference tokenises through ``OwlViTProcessor`` (CLIP BPE,
/root/reference/TStar/interface_heuristic.py:208,234 -> HF processing_owlvit.py:101-129):
each query padded to 16 tokens, ``[BOS=49406, tokens.., EOS=49407, 0..]``.

* If a CLIP tokenizer is available locally (a checkpoint directory / HF cache with
  vocab.json + merges.txt) it is used: identical ids to the reference.
* Otherwise (no network, no vocab on disk -- the situation of this build and of the GPU box)
  a deterministic stand-in maps every whitespace-separated word to one id in [1000, 41000).
  It is only meaningful with the synthetic weights; the layout (BOS/EOS/pad/length 16) is the
  real one so every downstream code path (EOS pooling, causal+pad mask, query mask) is exercised.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

BOS, EOS, PAD, TEXT_LEN = 49406, 49407, 0, 16

_HF_TOK = {}            # model_name_or_path -> CLIPTokenizer or None (tried, not available)


def _hf_tokenizer(model_name_or_path: str):
    """The real CLIP BPE tokenizer of a LOCAL checkpoint directory / HF cache entry (vocab.json + merges.txt), or None."""
    key = str(model_name_or_path)
    if key in _HF_TOK:
        return _HF_TOK[key]
    tok = None
    try:
        import os
        os.environ.setdefault("HF_HUB_OFFLINE", "1")
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(model_name_or_path, local_files_only=True)
        # transformers can hand back an EMPTY tokenizer when no vocab is on disk: accept it only if
        # it really is the CLIP BPE vocabulary (BOS 49406 / EOS 49407 around a known word)
        probe = tok(["cat"], padding="max_length", max_length=TEXT_LEN, truncation=True)["input_ids"][0]
        if not (probe[0] == BOS and probe[2] == EOS and len(tok) >= 49408):
            tok = None
    except Exception:
        tok = None
    _HF_TOK[key] = tok
    return tok


def standin_word_id(word: str) -> int:
    """Salt-free, process-independent word -> id (Python's hash() is salted: never use it)."""
    h = 2166136261
    for ch in word.lower().encode("utf-8"):
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return 1000 + h % 40000


def encode_queries(texts: Sequence[Sequence[str]], model_name_or_path: str = "google/owlvit-base-patch32",
                   allow_standin: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """``texts`` as OWLInterface.texts (``[[name], ..., [' ']]``) -> (ids, mask) int32 [Q,16]."""
    names = [t[0] for t in texts]
    tok = _hf_tokenizer(model_name_or_path)
    if tok is not None:
        enc = tok(names, padding="max_length", max_length=TEXT_LEN, truncation=True, return_tensors="np")
        return enc["input_ids"].astype(np.int32), enc["attention_mask"].astype(np.int32)
    if not allow_standin:
        raise RuntimeError("no CLIP tokenizer files found locally and the stand-in tokenizer is disabled")
    ids = np.zeros((len(names), TEXT_LEN), dtype=np.int32)
    am = np.zeros((len(names), TEXT_LEN), dtype=np.int32)
    for i, n in enumerate(names):
        toks = [BOS] + [standin_word_id(w) for w in n.split()][:TEXT_LEN - 2] + [EOS]
        ids[i, :len(toks)] = toks
        am[i, :len(toks)] = 1
    return ids, am
