"""A hand-made CLIP-BPE vocabulary of the real files' SHAPE, for tests that exercise the real-checkpoint branch of
tstar_amd.tokenizer without network access: ``vocab.json`` = 256 byte symbols + 256 word-final byte symbols + one
symbol per merge + fillers up to index 49405 + <|startoftext|> 49406 + <|endoftext|> 49407; ``merges.txt`` = the merges
that build a small word list bottom-up (left to right); ``tokenizer_config.json`` as the OWL-ViT checkpoint's (pad "!")."""
import json
import os

WORDS = ["couch", "tv", "chair", "dog", "leash", "park", "bench", "red", "car", "road", "laptop", "mug", "desk", "cat",
         "remote", "control", "table", "woman", "a", "photo", "of", "the"]


def write_clip_vocab(dirpath: str, words=WORDS):
    from oracle.clip_bpe_ref import bytes_to_unicode
    b2u = bytes_to_unicode()
    base = [b2u[b] for b in sorted(b2u, key=lambda b: list(b2u).index(b))]
    vocab = base + [c + "</w>" for c in base]
    merges = []
    seen = set(vocab)
    for w in words:
        sym = [b2u[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        while len(sym) > 1:                       # build the word right to left: (.., x, y</w>) -> (.., xy</w>)
            a, b = sym[-2], sym[-1]
            if (a, b) not in merges:
                merges.append((a, b))
            if a + b not in seen:
                seen.add(a + b)
                vocab.append(a + b)
            sym[-2:] = [a + b]
    vocab += [f"<|filler{i}|>" for i in range(49406 - len(vocab))]
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    assert len(vocab) == 49408
    os.makedirs(dirpath, exist_ok=True)
    json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(dirpath, "vocab.json"), "w", encoding="utf-8"), ensure_ascii=False)
    with open(os.path.join(dirpath, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    json.dump({"tokenizer_class": "CLIPTokenizer", "pad_token": "!", "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>",
               "unk_token": "<|endoftext|>", "model_max_length": 16, "do_lower_case": True},
              open(os.path.join(dirpath, "tokenizer_config.json"), "w"))
    json.dump({"pad_token": "!", "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>"},
              open(os.path.join(dirpath, "special_tokens_map.json"), "w"))
    return os.path.join(dirpath, "vocab.json"), os.path.join(dirpath, "merges.txt")
