"""ORACLE (test infrastructure, never shipped, never on the product path).

CLIP byte-pair encoding restated from the published algorithm (openai/CLIP ``simple_tokenizer.py``; the tokenizer the
reference reaches through ``OwlViTProcessor`` at /root/reference/TStar/interface_heuristic.py:208,234 -> HF
``processing_owlvit.py`` -> ``CLIPTokenizer``): lower-case, collapse whitespace, split with CLIP's pattern, map the
UTF-8 bytes of every piece to printable code points, append ``</w>`` to the last symbol, merge adjacent pairs in rank
order, look the symbols up in ``vocab.json``; a query is ``[BOS, ids.., EOS]`` padded to 16 with the OWL-ViT
checkpoint's pad token ``"!"`` (id 0) and truncated keeping EOS last.

Pinned by: equality with ``transformers.CLIPTokenizer`` on the same vocabulary files (tests/test_host_logic.py).  The
real ``vocab.json`` / ``merges.txt`` are not on disk (no network), so the tests use a hand-made vocabulary of the real
shape: 256 byte symbols, 256 word-final byte symbols, merges, fillers up to 49406, BOS 49406, EOS 49407.
"""
from __future__ import annotations

import json
import re
import unicodedata
from typing import Dict, List, Tuple

BOS, EOS, PAD, TEXT_LEN = 49406, 49407, 0, 16


def bytes_to_unicode() -> Dict[int, str]:
    """byte -> printable code point: the printable Latin-1 ranges map to themselves, the rest to 256, 257, ..."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def _is_letter(ch: str) -> bool:
    return unicodedata.category(ch).startswith("L")


def _is_number(ch: str) -> bool:
    return unicodedata.category(ch).startswith("N")


def split_pieces(text: str) -> List[str]:
    """CLIP's pattern  's|'t|'re|'ve|'m|'ll|'d|[\\p{L}]+|[\\p{N}]|[^\\s\\p{L}\\p{N}]+  without the `regex` module."""
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch.isspace():
            i += 1
            continue
        m = re.match(r"'(?:s|t|re|ve|m|ll|d)", text[i:])
        if m:
            out.append(m.group(0))
            i += len(m.group(0))
        elif _is_letter(ch):
            j = i
            while j < n and _is_letter(text[j]):
                j += 1
            out.append(text[i:j])
            i = j
        elif _is_number(ch):
            out.append(ch)
            i += 1
        else:
            j = i
            while j < n and not text[j].isspace() and not _is_letter(text[j]) and not _is_number(text[j]):
                j += 1
            out.append(text[i:j])
            i = j
    return out


class ClipBpe:
    def __init__(self, vocab_path: str, merges_path: str):
        self.encoder: Dict[str, int] = json.load(open(vocab_path, encoding="utf-8"))
        lines = open(merges_path, encoding="utf-8").read().split("\n")
        if lines and lines[0].startswith("#version"):
            lines = lines[1:]
        self.ranks: Dict[Tuple[str, str], int] = {tuple(ln.split()): r for r, ln in enumerate(l for l in lines if l.strip())}
        self.b2u = bytes_to_unicode()

    def bpe(self, piece: str) -> List[str]:
        sym = [self.b2u[b] for b in piece.encode("utf-8")]
        sym[-1] += "</w>"
        while len(sym) > 1:
            best = min(((self.ranks.get((a, b), 1 << 30), k) for k, (a, b) in enumerate(zip(sym, sym[1:]))))
            if best[0] == 1 << 30:
                break
            a, b = sym[best[1]], sym[best[1] + 1]
            merged, k = [], 0
            while k < len(sym):                              # every occurrence of the pair, left to right
                if k + 1 < len(sym) and sym[k] == a and sym[k + 1] == b:
                    merged.append(a + b)
                    k += 2
                else:
                    merged.append(sym[k])
                    k += 1
            sym = merged
        return sym

    SPECIALS = ("<|startoftext|>", "<|endoftext|>", "!")

    def encode(self, text: str) -> List[int]:
        """HF quirk kept: the OWL-ViT checkpoint declares "!" as its pad token, and the tokenizer cuts every special token
        out of the raw text BEFORE the pattern split -- so a literal "!" becomes the pad id 0 (not "!</w>" = 256), "!!" two
        of them, and the text either side is tokenised as separate segments."""
        segs = re.split("(" + "|".join(re.escape(t) for t in self.SPECIALS) + ")", text)
        out: List[int] = []
        for seg in segs:
            if seg in self.SPECIALS:
                out.append(self.encoder[seg])
                continue
            seg = " ".join(seg.split()).strip().lower()
            out += [self.encoder[s] for p in split_pieces(seg) for s in self.bpe(p)]
        return out

    def encode_queries(self, names, length: int = TEXT_LEN):
        import numpy as np
        ids = np.full((len(names), length), PAD, dtype=np.int32)
        am = np.zeros((len(names), length), dtype=np.int32)
        for i, nm in enumerate(names):
            t = [BOS] + self.encode(nm)[:length - 2] + [EOS]
            ids[i, :len(t)] = t
            am[i, :len(t)] = 1
        return ids, am
