"""CPU: the YOLO-World host logic (tstar_amd/yolo_world.py: architecture tables, layer program, letterbox geometry) and
the CPU oracle's own building blocks (oracle/yolo_ref.py).  The oracle's parity against the real model is UNPINNED
(source absent from the reference tree); what can be pinned on CPU is internal consistency."""
import numpy as np
import pytest

from tstar_amd import yolo_world as Y


@pytest.mark.parametrize("scale,channels,embed,heads", [("l", [256, 512, 512], [128, 256, 256], [4, 8, 8]),
                                                        ("x", [320, 640, 640], [160, 320, 320], [5, 10, 10]),
                                                        ("xl", [384, 768, 768], [192, 384, 384], [6, 12, 12]),
                                                        ("s", [128, 256, 512], [64, 128, 256], [2, 4, 8])])
def test_architecture_tables(scale, channels, embed, heads):
    A = Y.arch(scale)
    assert A["in_channels"] == channels and A["embed"] == embed and A["heads"] == heads
    assert [s["cout"] for s in A["stages"]][1:] == channels
    assert A["reg_ch"] == max(16, channels[0] // 4, 64) and A["cls_ch"] == max(channels[0], 80)
    with pytest.raises(ValueError, match="unknown YOLO-World scale"):
        Y.arch("xxl")


def test_scale_mismatch_is_reported_by_name():
    """the reference wires the XL config (widen 1.5: 96-channel stem); building it as another scale names the
    first tensor that does not fit instead of dying in a bare assert."""
    A = Y.arch("xl")
    assert A["stem"] == 96 and [s["cout"] for s in A["stages"]] == [192, 384, 768, 768]
    sd = Y.synthetic_state_dict(0, "s")
    with pytest.raises(ValueError, match=r"backbone\.image_model\.stem.*YOLO-World-v2-M"):
        Y.build_program(sd, "m")
    prog = Y.build_program(Y.synthetic_state_dict(0, "xl"), "xl")
    assert prog["ops"].shape[0] > 60


def test_layer_program_is_well_formed():
    """The program the C library interprets: every conv reads / writes inside its buffers, channel offsets are 16-byte
    aligned, the output geometry matches, weights lie inside the blob; 8400 anchors; ~175 GFLOP per 640x640 image for L."""
    sd = Y.synthetic_state_dict(0, "l")
    prog = Y.build_program(sd, "l")
    ops, bufs, blob = prog["ops"], prog["bufs"], prog["blob"]
    assert ops.shape[1] == Y.OP_WORDS and blob.dtype == np.float32 and np.isfinite(blob).all()
    n_conv = 0
    for o in ops:
        if o[0] == Y.OP_CONV:
            _, src, soff, cin, dst, doff, cout, ks, stride, act, w_off, b_off, mode, aux, aux_off = o[:15]
            H, W, C = bufs[src]
            assert soff + cin <= C and doff + cout <= bufs[dst][2] and doff % 4 == 0 and cout % 4 == 0
            assert tuple(bufs[dst][:2]) == ((H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1)
            assert 0 <= w_off and w_off + cout * ks * ks * cin <= blob.size and (b_off < 0 or b_off + cout <= blob.size)
            assert mode == Y.MODE_PLAIN or 0 <= aux < len(bufs)
            n_conv += 1
    assert n_conv > 90
    assert sum(int(l[2]) ** 2 for l in prog["levels"]) == 8400
    assert [int(l[3]) for l in prog["levels"]] == [8, 16, 32]
    assert 160e9 < Y.conv_flops(prog) < 190e9
    assert len(prog["guides"]) == 4 and all(g[0] % g[1] == 0 for g in prog["guides"])
    # synthetic parameters are a frozen stream
    sd2 = Y.synthetic_state_dict(0, "l")
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)
    assert not np.array_equal(Y.synthetic_state_dict(1, "l")["backbone.image_model.stem.conv.weight"], sd["backbone.image_model.stem.conv.weight"])


def test_bn_folding_equals_eval_batchnorm():
    import torch
    import torch.nn.functional as F
    from oracle import yolo_ref as R
    sd = Y.synthetic_state_dict(3, "s")
    name = "backbone.image_model.stage1.0"
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 32, 12, 12)).astype(np.float32))
    ref = R.conv_module(sd, name, x, stride=2, act=False)
    b = Y._Builder(sd)
    w, bias = b.folded(name)                                         # [Cout][kh][kw][Cin], float64
    got = F.conv2d(x.double(), torch.from_numpy(w.transpose(0, 3, 1, 2).copy()), torch.from_numpy(bias), stride=2, padding=1)
    assert np.abs(got.numpy() - ref.numpy()).max() < 2e-5


@pytest.mark.parametrize("h,w", [(380, 800), (285, 600), (1520, 3200), (1425, 3000), (360, 640), (640, 640), (95, 200)])
def test_letterbox_geometry_matches_the_oracle(h, w):
    from oracle import yolo_ref as R
    img = np.random.RandomState(h).randint(0, 256, (h, w, 3)).astype(np.uint8)
    x, geom = R.letterbox(img)
    g = Y.letterbox_geometry(h, w)
    assert x.shape == (3, 640, 640) and geom["pad"] == g["pad"] and np.allclose(geom["scale_factor"], g["scale_factor"])
    assert g["resized"] == g["first_resize"]                        # the second (LetterResize) stage never rescales
    top, bottom, left, right = g["pad"]
    assert top + g["resized"][0] + bottom == 640 and left + g["resized"][1] + right == 640
    lb = geom["letterboxed"]
    if top:
        assert (lb[:top] == 114).all() and (lb[640 - bottom:] == 114).all()
    # channels reversed, / 255
    assert np.array_equal(x[0], lb[:, :, 2].astype(np.float32) / np.float32(255))


def test_area_resize_integer_scale_is_the_box_mean():
    from oracle import yolo_ref as R
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (40, 60, 3)).astype(np.uint8)
    out = R.area_resize(img, 12, 8)                                  # exactly 5x in both axes
    box = img.reshape(8, 5, 12, 5, 3).astype(np.int64).sum(axis=(1, 3))
    assert np.array_equal(out, ((2 * box + 25) // 50).astype(np.uint8))
    flat = np.full((38, 80, 3), 77, np.uint8)
    assert (R.area_resize(flat, 64, 30) == 77).all()                # non-integer scale: weights sum to the footprint


def test_nms_and_selection_semantics():
    from oracle import yolo_ref as R
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [0, 0, 10, 10], [50, 50, 60, 60], [0, 0, 10, 10.5]], np.float32)
    sc = np.zeros((5, 2), np.float32)
    sc[0, 0], sc[1, 0], sc[2, 1], sc[3, 0], sc[4, 0] = 0.9, 0.8, 0.7, 0.2, 0.1
    r = R.select(sc, boxes, (100, 100), wrapper_thr=0.12, max_dets=50)
    # anchor 1 (IoU 0.68 with anchor 0) survives, anchor 2 is another class, anchor 4 is below the wrapper threshold
    assert r["anchors"].tolist() == [0, 1, 2, 3] and r["labels"].tolist() == [0, 0, 1, 0]
    r = R.select(sc, boxes, (100, 100), wrapper_thr=0.12, max_dets=2)
    assert r["anchors"].tolist() == [0, 1]
    sc[1, 0] = 0.95                                                  # now anchor 1 leads; anchor 0 (IoU 0.68 < 0.7) still kept
    boxes[1] = [0.5, 0.5, 10.5, 10.5]                                # IoU 0.82 -> anchor 0 suppressed
    r = R.select(sc, boxes, (100, 100))
    assert r["anchors"].tolist() == [1, 2, 3]
    r = R.select(sc, boxes, (8, 8))                                  # clamp to the image
    assert r["xyxy"].max() <= 8
