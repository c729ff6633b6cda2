"""Ingest kernels: gather + cv-style bilinear + grid tiling (RGB and NV12 stores) vs the oracle, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from tstar_amd import _lib
    return _lib, _lib.load()


@pytest.mark.parametrize("g", [1, 4, 15])
def test_frames_to_grid_rgb(g):
    from oracle import resize_ref as R
    from tstar_amd.video import synthetic_frames_numpy, synthetic_video
    L, lib = _lib()
    N = 300
    st = synthetic_video(N, seed=4)
    secs = list(np.random.RandomState(g).choice(N, g * g, replace=False))
    idx = torch.tensor(secs, dtype=torch.int32, device="cuda")
    grid = torch.empty((95 * g, 200 * g, 3), dtype=torch.uint8, device="cuda")
    L.check(lib.tstar_frames_to_grid(st.frames.data_ptr(), N, 360, 640, idx.data_ptr(), g, g, grid.data_ptr(), 0, None))
    torch.cuda.synchronize()
    assert np.array_equal(grid.cpu().numpy(), R.frames_to_grid(list(synthetic_frames_numpy(secs, N, seed=4)), g, g))


@pytest.mark.parametrize("ow,oh", [(600, 285), (800, 380), (640, 360), (37, 11), (4, 3), (1, 5), (8, 2), (2, 1)])   # ow 4 / 1: one unit per row (magic_of(1))
def test_frames_resize_rgb(ow, oh):
    from oracle import resize_ref as R
    from tstar_amd.video import synthetic_frames_numpy, synthetic_video
    L, lib = _lib()
    N = 40
    st = synthetic_video(N, seed=6)
    secs = [5, 0, 39]
    idx = torch.tensor(secs, dtype=torch.int32, device="cuda")
    out = torch.empty((3, oh, ow, 3), dtype=torch.uint8, device="cuda")
    L.check(lib.tstar_frames_resize(st.frames.data_ptr(), N, 360, 640, idx.data_ptr(), 3, ow, oh, out.data_ptr(), 0, None))
    torch.cuda.synchronize()
    fr = synthetic_frames_numpy(secs, N, seed=6)
    for k in range(3):
        assert np.array_equal(out[k].cpu().numpy(), R.cv_bilinear_resize(fr[k], ow, oh))


def test_frames_resize_random_source_sizes():
    """Random noise frames of 20 seeded random source sizes (upscales, strong decimation, odd sizes, exact 2x / 4x / 1x of
    the target) to the grid-cell (200x95) and verification (600x285) sizes: bit-exact against the oracle's cv2.resize
    statement (at exactly 2x decimation, where cv2 switches to INTER_AREA, the two formulas coincide)."""
    from oracle import resize_ref as R
    L, lib = _lib()
    rs = np.random.RandomState(77)
    sizes = [(190, 400), (380, 800), (95, 200), (570, 1200), (285, 600), (2, 2), (2, 3), (1080, 1920)]
    while len(sizes) < 20:
        sizes.append((int(rs.randint(8, 1300)), int(rs.randint(8, 1300))))
    for H, W in sizes:
        frames = rs.randint(0, 256, (2, H, W, 3), dtype=np.uint8)
        d = torch.from_numpy(frames).cuda()
        idx = torch.tensor([1, 0], dtype=torch.int32, device="cuda")
        for ow, oh in ((200, 95), (600, 285)):
            out = torch.empty((2, oh, ow, 3), dtype=torch.uint8, device="cuda")
            L.check(lib.tstar_frames_resize(d.data_ptr(), 2, H, W, idx.data_ptr(), 2, ow, oh, out.data_ptr(), 0, None))
            torch.cuda.synchronize()
            for k, src in enumerate((1, 0)):
                assert np.array_equal(out[k].cpu().numpy(), R.cv_bilinear_resize(frames[src], ow, oh)), (H, W, ow, oh)


def test_nv12_store_matches_rgb_of_converted_frames():
    """NV12 ingest = the RGB ingest applied to the converted frames (conversion fused into the taps)."""
    from oracle import resize_ref as R
    from tstar_amd.video import synthetic_nv12_numpy, synthetic_video_nv12
    L, lib = _lib()
    N, g = 64, 4
    st = synthetic_video_nv12(N, seed=9)
    assert st.fmt == "nv12" and st.frames.shape == (N, 540, 640) and st.shape == (N, 360, 640, 3)
    secs = list(range(0, N, 4))
    nv = synthetic_nv12_numpy(secs, N, seed=9)
    assert np.array_equal(st.frames[secs].cpu().numpy(), nv)
    rgb = [R.nv12_to_rgb(f) for f in nv]
    assert np.array_equal(st.host_frames(secs[:3]), np.stack(rgb[:3]))
    idx = torch.tensor(secs, dtype=torch.int32, device="cuda")
    grid = torch.empty((95 * g, 200 * g, 3), dtype=torch.uint8, device="cuda")
    L.check(lib.tstar_frames_to_grid(st.frames.data_ptr(), N, 360, 640, idx.data_ptr(), g, g, grid.data_ptr(), 1, None))
    out = torch.empty((2, 285, 600, 3), dtype=torch.uint8, device="cuda")
    L.check(lib.tstar_frames_resize(st.frames.data_ptr(), N, 360, 640, idx.data_ptr(), 2, 600, 285, out.data_ptr(), 1, None))
    torch.cuda.synchronize()
    assert np.array_equal(grid.cpu().numpy(), R.frames_to_grid(rgb, g, g))
    assert np.array_equal(out[1].cpu().numpy(), R.cv_bilinear_resize(rgb[1], 600, 285))


@pytest.mark.parametrize("H,W,g", [(360, 640, 4), (72, 128, 3), (90, 160, 1), (480, 854, 2), (36, 4, 2), (1080, 1920, 2)])
def test_nv12_grid_random_full_range(H, W, g):
    """NV12 grid ingest on FULL-RANGE random bytes (the BT.601 conversion clips at both ends, which the synthetic video's
    limited-range content never reaches) and on sizes that exercise the window clamps at the right edge, the chroma-pair
    selection for odd / even taps and both forms of the row table (duplicate middle row or not): bit-exact against the RGB
    oracle applied to the converted frames.  1920-wide sources decimate: the two taps of a sample are still adjacent columns."""
    from oracle import resize_ref as R
    L, lib = _lib()
    rs = np.random.RandomState(H * 7 + W)
    n = g * g + 1
    nv = rs.randint(0, 256, (n, H * 3 // 2, W)).astype(np.uint8)
    d = torch.from_numpy(nv).cuda()
    order = list(rs.permutation(n)[:g * g])
    idx = torch.tensor(order, dtype=torch.int32, device="cuda")
    grid = torch.empty((95 * g, 200 * g, 3), dtype=torch.uint8, device="cuda")
    L.check(lib.tstar_frames_to_grid(d.data_ptr(), n, H, W, idx.data_ptr(), g, g, grid.data_ptr(), 1, None))
    torch.cuda.synchronize()
    rgb = [R.nv12_to_rgb(nv[i]) for i in order]
    assert np.array_equal(grid.cpu().numpy(), R.frames_to_grid(rgb, g, g)), (H, W, g)
    # the single-step resize on the same frames: 600x285 (4 pixels per lane) and an odd width (1 pixel per lane)
    m = min(2, len(order))
    for ow, oh in ((600, 285), (201, 97)):
        out = torch.empty((m, oh, ow, 3), dtype=torch.uint8, device="cuda")
        L.check(lib.tstar_frames_resize(d.data_ptr(), n, H, W, idx.data_ptr(), m, ow, oh, out.data_ptr(), 1, None))
        torch.cuda.synchronize()
        for k in range(m):
            assert np.array_equal(out[k].cpu().numpy(), R.cv_bilinear_resize(rgb[k], ow, oh)), (H, W, ow, oh, k)


def test_search_on_nv12_store_equals_search_on_converted_rgb_store():
    from tstar_amd.interface_heuristic import OWLInterface
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import FrameStore, load_video_frames, synthetic_video_nv12
    N = 96
    nv = synthetic_video_nv12(N, seed=2)
    rgb = FrameStore(torch.from_numpy(nv.host_frames(range(N))).cuda(), 1.0)
    h = OWLInterface(synthetic_seed=0, max_batch=8)
    res = []
    for st in (nv, rgb):
        s = TStarSearcher(st, h, ["couch"], ["tv"], search_nframes=4, image_grid_shape=(4, 4), search_budget=0.4,
                          confidence_threshold=0.6, rng=np.random.RandomState(3), keep_visual_history=False)
        fr, ts = s.search()
        res.append((fr, ts, s.score_distribution))
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][2], res[1][2])
    u = load_video_frames(nv, 8)                      # the grounder's uniform loader (utilites.py:40-81)
    assert u.shape == (8, 360, 640, 3) and np.array_equal(u[1], nv.host_frames([12])[0])


def test_ingest_rejects_bad_arguments():
    L, lib = _lib()
    d = torch.zeros(16, dtype=torch.uint8, device="cuda")
    i = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.tstar_frames_to_grid(d.data_ptr(), 1, 1, 4, i.data_ptr(), 1, 1, d.data_ptr(), 0, None) == 1
    assert lib.tstar_frames_resize(d.data_ptr(), 1, 3, 4, i.data_ptr(), 1, 2, 2, d.data_ptr(), 1, None) == 1   # odd H, NV12
    assert lib.tstar_frames_to_grid(None, 1, 2, 2, i.data_ptr(), 1, 1, d.data_ptr(), 0, None) == 1


def test_odd_resolution_and_non_unit_fps_store():
    """Arbitrary frame sizes (240x427) and a 29.97 fps stream: N = int(total / raw_fps), grid bit-exact."""
    from oracle import resize_ref as R
    from tstar_amd.interface_heuristic import OWLInterface
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import FrameStore
    L, lib = _lib()
    rs = np.random.RandomState(5)
    raw_fps, raw_total = 29.97, 1200                       # 40.04 s -> 40 logical seconds
    frames = rs.randint(0, 256, (40, 240, 427, 3), dtype=np.uint8)
    st = FrameStore(torch.from_numpy(frames).cuda(), raw_fps, raw_total)
    h = OWLInterface(synthetic_seed=0, max_batch=4)
    s = TStarSearcher(st, h, ["couch"], [], search_nframes=4, image_grid_shape=(2, 2), search_budget=0.5,
                      confidence_threshold=0.6, rng=np.random.RandomState(0), keep_visual_history=True)
    assert s.total_frame_num == int(raw_total / raw_fps * 1) == 40 and abs(s.duration - raw_total / raw_fps) < 1e-12
    assert s.search_budget == min(1000, 40 * 0.5)
    fr, ts = s.search()
    assert fr.shape == (4, 240, 427, 3) and s.iterations == 5          # budget 20 -> 5 iterations of 4
    secs0 = [0, 10, 20, 30]
    assert np.array_equal(s.image_grid_iters[0][0].shape, (190, 400, 3))
    grid = s._device_grid(secs0).cpu().numpy()
    assert np.array_equal(grid, R.frames_to_grid([frames[i] for i in secs0], 2, 2))
    assert np.array_equal(fr, frames[[int(t) for t in ts]])


def test_create_image_grid_reference_method():
    """TStarSearcher.create_image_grid (interface_searcher.py:171-188): host frames -> 200x95 cells tiled row-major;
    same bilinear as the oracle, and the reference's error for a wrong frame count."""
    from oracle import resize_ref as R
    from tstar_amd.interface_heuristic import OWLInterface
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.video import synthetic_video
    h = OWLInterface(synthetic_seed=0, max_batch=2)
    s = TStarSearcher(synthetic_video(40, seed=2), h, ["couch"], ["tv"], search_nframes=4, image_grid_shape=(2, 3),
                      search_budget=1.0, rng=np.random.RandomState(0), keep_visual_history=False)
    rs = np.random.RandomState(5)
    frames = [rs.randint(0, 256, (380, 800, 3)).astype(np.uint8) for _ in range(6)]
    grid = s.create_image_grid(frames, 2, 3)
    small = [R.cv_bilinear_resize(f, 200, 95) for f in frames]
    ref = np.vstack([np.hstack(small[r * 3:(r + 1) * 3]) for r in range(2)])
    assert grid.shape == (190, 600, 3) and np.array_equal(grid, ref)
    with pytest.raises(ValueError, match="Frame count does not match grid dimensions"):
        s.create_image_grid(frames[:5], 2, 3)


def test_y4m_decode_front_end(tmp_path):
    """SURVEY.md 8f-3, the part that is buildable here: a raw 4:2:0 file (YUV4MPEG2) is read once, staged through pinned
    buffers and repacked I420 -> NV12 on the device into the resident store; the searcher's index map (raw frame
    int(sec * fps), interface_searcher.py:360) selects the frames.  Bytes must equal the source frames', the ingest on the
    loaded store must equal the ingest on the same frames built directly, and the searcher opens the path by name."""
    from oracle import resize_ref as R
    from tstar_amd import video as V
    from tstar_amd.interface_searcher import TStarSearcher
    import golden_util as GU
    n_raw, H, W = 150, 72, 128
    raw = V.synthetic_nv12_numpy(list(range(n_raw)), n_raw, H, W, seed=9)
    path = str(tmp_path / "clip.y4m")
    V.write_y4m(path, raw, fps=(30000, 1001))                             # 29.97 fps -> 5 logical seconds
    hd = V.parse_y4m_header(path)
    assert hd["n_frames"] == n_raw and abs(hd["fps"] - 29.97) < 0.01
    st = V.open_video(path)
    assert st.fmt == "nv12" and st.num_seconds == int(n_raw / hd["fps"]) == 5 and st.raw_total_frames == n_raw
    want = [int(s * hd["fps"]) for s in range(5)]
    assert np.array_equal(st.frames.cpu().numpy(), raw[want])
    rgb = st.host_frames([0, 3])
    assert np.array_equal(rgb[1], R.nv12_to_rgb(raw[want[3]]))
    # chunking (more seconds than one staging chunk) and a 1 fps file
    raw2 = V.synthetic_nv12_numpy(list(range(70)), 70, H, W, seed=10)
    path2 = str(tmp_path / "long.y4m")
    V.write_y4m(path2, raw2, fps=(1, 1))
    st2 = V.load_y4m(path2, chunk=16)
    assert st2.num_seconds == 70 and np.array_equal(st2.frames.cpu().numpy(), raw2)
    h = GU.FakeHeuristic(0)
    s = TStarSearcher(video_path=path2, heuristic=h, target_objects=["a"], cue_objects=[], search_nframes=4,
                      image_grid_shape=(2, 2), search_budget=0.2, confidence_threshold=0.5, rng=np.random.RandomState(1))
    frames, ts = s.search()
    assert frames.shape == (4, H, W, 3) and s.total_frame_num == 70
    for bad, msg in ((b"RIFF....", "not a YUV4MPEG2"), (b"YUV4MPEG2 W128 H72 F1:1 C444\nFRAME\n", "only 8-bit 4:2:0")):
        pb = tmp_path / "bad.y4m"
        pb.write_bytes(bad)
        with pytest.raises(ValueError, match=msg):
            V.open_video(str(pb))
    with pytest.raises(ValueError, match="Cannot open video file"):
        V.open_video(str(tmp_path / "missing.y4m"))
    # sizes the device repack cannot take are refused before anything is allocated; a stream whose frame headers vary in
    # length is refused at the first frame that does not start with its marker
    odd = str(tmp_path / "odd.y4m")
    V.write_y4m(odd, np.zeros((2, 30 * 3 // 2, 42), np.uint8))
    with pytest.raises(ValueError, match="multiple of 64"):
        V.load_y4m(odd)
    blob = bytearray(open(path2, "rb").read())
    hd2 = V.parse_y4m_header(path2)
    at = hd2["data_offset"] + 3 * (hd2["frame_header_bytes"] + hd2["frame_bytes"])
    blob[at:at + 6] = b"FRAME "                                           # frame 3 now carries a (longer) parameterised header
    blob[at + 6:at + 6] = b"Ip\n"
    shifted = str(tmp_path / "shifted.y4m")
    open(shifted, "wb").write(bytes(blob) + bytes(hd2["frame_bytes"]))
    with pytest.raises(ValueError, match="no FRAME marker where frame 4"):
        V.load_y4m(shifted, chunk=16)


def test_compressed_webp_video_opens_by_name_and_searches(tmp_path):
    """A compressed (lossless animated WebP) file path handed to TStarSearcher like the reference hands an .mp4 to decord:
    decoded once by the Pillow front end into the resident store, then the ordinary device path."""
    from PIL import Image
    from tstar_amd import video as V
    from tstar_amd.interface_searcher import TStarSearcher
    import golden_util as GU
    frames = V.synthetic_frames_numpy(list(range(40)), 40, 72, 128, seed=4)
    pil = [Image.fromarray(f) for f in frames]
    path = str(tmp_path / "clip.webp")
    pil[0].save(path, save_all=True, append_images=pil[1:], duration=500, loop=0, lossless=True)      # 2 fps -> 20 logical seconds
    st = V.open_video(path)
    assert st.fmt == "rgb" and st.num_seconds == 20 and st.raw_fps == 2.0 and st.frames.is_cuda
    assert np.array_equal(st.frames.cpu().numpy(), frames[::2])
    s = TStarSearcher(video_path=path, heuristic=GU.FakeHeuristic(0), target_objects=["a"], cue_objects=[], search_nframes=4,
                      image_grid_shape=(2, 2), search_budget=0.5, confidence_threshold=0.5, rng=np.random.RandomState(1))
    out, ts = s.search()
    assert out.shape == (4, 72, 128, 3) and s.total_frame_num == 20 and s.raw_fps == 2.0
    assert all(np.array_equal(out[k], frames[int(t * 2.0)]) for k, t in enumerate(ts))
