"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
committed goldens.  Integer outputs (pixels, masks, counts) must be bit-exact; the float line
endpoints are compared bit-exactly too (same f32 operation order); derived f64/f32 lengths, meters
and angles within the north star's 1e-4."""
import hashlib
import os

import numpy as np
import pytest

import fixtures as fx
from oracle import oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-4


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_trait_sequence(v, frame, max_gap=15, anchors=None, grayscale=True):
    """The call order of the reference's own GPU test (vision-gpu/src/lib.rs:562-622)."""
    import squad_mortar_helper_amd as smh
    st = smh.VisionState(grayscale_map=grayscale, max_gap=max_gap)
    return st.process(v, frame, ocr_labels=anchors)


def test_native_library_is_loaded_and_device_is_gfx950(vision):
    import torch
    assert torch.cuda.is_available()
    maps = open("/proc/self/maps").read()
    assert "libsmh_vision_hip.so" in maps


def test_colour_predicate_exhaustive_2_pow_24(vision):
    """is_any_map_marker_color on the device == oracle for every RGB colour (hsv f32 rounding,
    division, the fmod identities and the integer pre-filter are all covered)."""
    dev = vision.debug_marker_table()
    ref = o.marker_table()
    assert np.array_equal(dev, ref)
    assert int(np.unpackbits(ref.view(np.uint8)).sum()) == 888998


@pytest.mark.parametrize("stem", fx.OPEN_STEMS)
def test_samples_match_goldens_and_oracle(vision, stem):
    frame, e, g = fx.load_fixture(stem)
    anchors = e.get("anchors") or None
    res = run_trait_sequence(vision, frame, anchors=anchors)
    assert res is not None and list(res.roi) == e["map_rect"]
    assert vision.red_pixels() == e["red_pixels"]
    lsd = vision.lsd_image()
    assert np.array_equal(np.flatnonzero(lsd.reshape(-1) == 255).astype(np.uint32), g["mask_idx"])   # marker pixel coords
    assert set(np.unique(lsd)) <= {0, 255} and sha(lsd) == e["sha_lsd"]
    assert res.markers.shape == g["lines"].shape and np.array_equal(res.markers, g["lines"])
    assert vision.lsd_stats(15, exact=True) == (e["rounds"], e["steps"])      # identical ray trajectories
    assert vision.lsd_stats(15, exact=False)[0] == e["rounds"]                  # sector culling keeps the visit order
    assert sha(res.map) == e["sha_ui_gray"]
    assert sha(vision.ocr_preprocess()) == e["sha_ocr"]
    assert sha(vision.find_scales_preprocess(0)) == e["sha_scales0"]
    # the reference's GPU test calls find_marker_lines(22)
    assert np.array_equal(vision.find_marker_lines(22), g["lines_gap22"])
    # debug views (vision-cpu/src/lib.rs:451-460)
    import squad_mortar_helper_amd as smh
    iso = vision.get_debug_view(smh.DebugView.LSD_PREPROCESS)
    assert sha(iso[..., :3]) == e["sha_isolated"] and (iso[..., 3] == 255).all()
    assert sha(vision.get_debug_view(smh.DebugView.CROPPED_BRQ)[..., :3]) == e["sha_brq"]
    assert np.array_equal(vision.get_debug_view(smh.DebugView.LSD_INPUT)[..., 0], lsd)
    if anchors:
        assert res.meters_to_px_ratio == e["mpx"]


@pytest.mark.parametrize("stem", fx.FULL_STEMS)
def test_colour_ui_map(vision, stem):
    frame, e, _ = fx.load_fixture(stem)
    vision.load_frame(frame)
    ui, roi = vision.crop_to_map(False)
    assert sha(ui) == e["sha_ui_color"]


def test_closed_map_returns_none_and_later_calls_error(vision):
    import squad_mortar_helper_amd as smh
    for stem in ("a_point_png", "line_angle_png"):
        frame, e, _ = fx.load_fixture(stem)
        vision.load_frame(frame)
        assert vision.crop_to_map(True) is None
        assert vision.red_pixels() == e["red_pixels"]
        with pytest.raises(smh.VisionError) as ei:
            vision.find_marker_lines(15)
        assert ei.value.code == -5
    with pytest.raises(smh.VisionError):
        vision.load_frame(np.zeros((44, 43, 4), np.uint8))          # convolution.png geometry


def test_button_threshold_edge(vision):
    """ratio < 0.65 => None; the count is compared as f32 exactly like the reference."""
    from squad_mortar_helper_amd import synth
    import squad_mortar_helper_amd as smh
    W, H = 1024, 768
    frame, _ = synth.make_frame(W, H, 1, n_lines=0)
    bx, by, bw, bh = smh.button_bounds(W, H)
    n = bw * bh
    k_open = int(np.ceil(0.65 * n))
    for k in (k_open - 1, k_open, k_open + 1):
        f = frame.copy()
        flat = f[by:by + bh, bx:bx + bw].reshape(-1, 4)
        flat[:, :3] = (49, 67, 217)
        flat[k:, :3] = (0, 0, 0)
        f[by:by + bh, bx:bx + bw] = flat.reshape(bh, bw, 4)
        vision.load_frame(f)
        got = vision.crop_to_map(True)
        want = o.crop_to_map(f, True)
        assert (got is None) == (want is None)
        assert vision.red_pixels() == k == o.button_red_pixels(f)
    # per-channel tolerance is inclusive at 25
    f = frame.copy()
    f[by:by + bh, bx:bx + bw, :3] = (49 + 25, 67 - 25, 217 + 25)
    vision.load_frame(f)
    assert vision.crop_to_map(True) is not None and vision.red_pixels() == n
    f[by:by + bh, bx:bx + bw, 2] = 217 + 26
    vision.load_frame(f)
    assert vision.crop_to_map(True) is None and vision.red_pixels() == 0


def test_lazy_ui_map_heightmap_switch_and_per_call_times(vision):
    """crop_to_map without a destination returns once the button test is known; smhv_ui_map hands out the image from pinned
    memory -- the same bytes as the eager form, still readable while the NEXT frame is processed (two buffers take turns).
    VisionState follows src/vision/mod.rs:121-124,219-223: WITHOUT a heightmap (`heightmaps::is_set()` false, the default) both
    branches run; WITH one the scales branch is skipped, meters_to_px_ratio is None and the markers closure runs on the calling
    thread.  lazy_map=False is the sequence the trait allows (the image by value from crop_to_map).  Every trait call leaves its wall
    time in the context's table (the reference's Timeshares entry per call, mod.rs:54-66)."""
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = 1920, 1080
    f0, i0 = synth.make_frame(W, H, 40, n_lines=2)
    f1, i1 = synth.make_frame(W, H, 41, n_lines=3)
    r0 = o.process_frame(f0, stages=0xF, anchors=i0["anchors"], scales_start_y=i0["scales_start_y"], want_images=True)
    r1 = o.process_frame(f1, stages=0xF, anchors=i1["anchors"], scales_start_y=i1["scales_start_y"], want_images=True)
    vision.load_frame(f0)
    eager = vision.crop_to_map(True)
    assert np.array_equal(eager[0], r0["ui_map"])
    vision.load_frame(f0)
    lazy = vision.crop_to_map(True, lazy=True)
    assert lazy is not None and lazy[0] is None and lazy[1] == eager[1]
    m0 = vision.ui_map()
    assert np.array_equal(m0, r0["ui_map"])
    vision.load_frame(f1)                                            # the next frame: frame 0's map is still there
    assert vision.crop_to_map(True, lazy=True) is not None
    m1 = vision.ui_map()
    assert np.array_equal(m1, r1["ui_map"]) and np.array_equal(m0, r0["ui_map"])
    colour = vision.crop_to_map(False, lazy=True)
    assert colour is not None and np.array_equal(vision.ui_map()[..., :3], f1[eager[1][1]:eager[1][1] + eager[1][3], eager[1][0]:eager[1][0] + eager[1][2], 2::-1])
    # a closed map has no ui_map
    closed = f0.copy()
    bx, by, bw, bh = smh.button_bounds(W, H)
    closed[by:by + bh, bx:bx + bw] = 0
    vision.load_frame(closed)
    assert vision.crop_to_map(True, lazy=True) is None
    with pytest.raises(smh.VisionError):
        vision.ui_map()
    # the caller contract with and without a heightmap
    vision.trait_times(reset=True)
    st = smh.VisionState()                                           # no heightmap: both branches (mod.rs:124 `Some(closure)`)
    assert st.heightmap_is_set is False
    res = st.process(vision, f0, ocr_labels=i0["anchors"])
    assert np.array_equal(res.map, r0["ui_map"]) and np.array_equal(res.markers, r0["lines"]) and res.meters_to_px_ratio == r0["mpx"]
    assert res.meters_to_px_ratio is not None
    tt = vision.trait_times()
    for k in ("load_frame", "crop_to_map", "find_minimap", "find_marker_lines", "ocr_preprocess", "find_scales_preprocess", "calc_meters_to_px_ratio", "ui_map"):
        assert tt[k][1] == 1 and tt[k][0] > 0.0, (k, tt[k])
    st2 = smh.VisionState(heightmap_is_set=True)                     # a heightmap is selected: `None` => (markers(), Ok(None))
    res2 = st2.process(vision, f0, ocr_labels=i0["anchors"])
    assert np.array_equal(res2.map, r0["ui_map"]) and np.array_equal(res2.markers, r0["lines"]) and res2.meters_to_px_ratio is None
    tt = vision.trait_times()
    assert tt["ocr_preprocess"][1] == 1 and tt["find_scales_preprocess"][1] == 1 and tt["calc_meters_to_px_ratio"][1] == 1   # (not called again)
    assert tt["find_marker_lines"][1] == 2
    assert st2._workers is None                                      # (no join: the markers closure ran on this thread)
    # the trait-shaped sequence: crop_to_map hands the image over by value, smhv_ui_map is never called
    st3 = smh.VisionState(lazy_map=False)
    res3 = st3.process(vision, f1, ocr_labels=i1["anchors"])
    assert np.array_equal(res3.map, r1["ui_map"]) and np.array_equal(res3.markers, r1["lines"]) and res3.meters_to_px_ratio == r1["mpx"]
    assert vision.trait_times()["ui_map"][1] == 2
    # VisionResults.map is the caller's own: the next frames do not touch it
    keep = res.map
    for f in (f1, f0, f1):
        st.process(vision, f, ocr_labels=i1["anchors"])
    assert np.array_equal(keep, r0["ui_map"])
    st3.close()
    st.close(); st2.close()


@pytest.mark.parametrize("size", [(1920, 1080), (2560, 1440), (1024, 768), (1280, 1024), (1600, 1024), (3840, 2160)])
def test_synthetic_frames_all_stages(vision, size):
    from squad_mortar_helper_amd import synth
    W, H = size
    for idx in (0, 1):
        frame, info = synth.make_frame(W, H, idx, n_lines=2 + idx)
        ref = o.process_frame(frame, stages=0xF, anchors=info["anchors"], scales_start_y=info["scales_start_y"], want_images=True)
        res = run_trait_sequence(vision, frame, anchors=info["anchors"])
        assert np.array_equal(res.map, ref["ui_map"])
        assert np.array_equal(vision.lsd_image(), ref["lsd"])
        assert np.array_equal(res.markers, ref["lines"])
        assert np.array_equal(vision.ocr_preprocess(), ref["ocr"])
        sc = vision.find_scales_preprocess(info["scales_start_y"])
        assert np.array_equal(sc[info["scales_start_y"]:], ref["scales"][info["scales_start_y"]:])
        assert res.meters_to_px_ratio == ref["mpx"]


def test_find_longest_line_random_points(vision):
    """Vision::find_longest_line on arbitrary (also fractional / off-mask / border) points."""
    frame, e, g = fx.load_fixture("points_intersect_png")
    run_trait_sequence(vision, frame)
    lsd = vision.lsd_image()
    h, w = lsd.shape
    rng = np.random.default_rng(3)
    ys, xs = np.nonzero(lsd == 255)
    pts = [(float(xs[i]), float(ys[i])) for i in rng.integers(0, len(xs), 12)]
    pts += [(float(xs[i]) + 0.5, float(ys[i]) + 0.25) for i in rng.integers(0, len(xs), 6)]
    pts += [(0.0, 0.0), (w - 1.0, h - 1.0), (w / 2.0, 0.0), (0.0, h / 2.0), (float(rng.uniform(0, w - 1)), float(rng.uniform(0, h - 1)))]
    for gap in (15.0, 22.0, 0.0, 3.5):
        for (x, y) in pts[:: (1 if gap == 15.0 else 4)]:
            line, ln = vision.find_longest_line((x, y), gap)
            rl, rn = o.find_longest_line(lsd, x, y, gap)
            assert np.array_equal(line, rl) and ln == rn, (x, y, gap, line, rl)


def test_ocr_neighbourhood_and_scales_semantics(vision):
    """Hand-made BRQ content: white glyph pixels, grey edge pixels inside/outside the 7x7 reach, the
    asymmetric `min(x+3, w-3)` clamp at the right/bottom border, near-black luma (scales)."""
    from squad_mortar_helper_amd import synth
    import squad_mortar_helper_amd as smh
    W, H = 1280, 1024
    frame, info = synth.make_frame(W, H, 2, n_lines=0)
    x, y, rw, rh = smh.map_bounds(W, H)
    qw, qh = rw // 2, rh // 2
    ox, oy = x + qw, y + qh
    rng = np.random.default_rng(11)
    brq = frame[oy:oy + qh, ox:ox + qw]
    brq[..., :3] = rng.integers(100, 200, (qh, qw, 1))            # grey field: many edge candidates (>=130), none white
    for _ in range(60):
        cx, cy = int(rng.integers(0, qw)), int(rng.integers(0, qh))
        brq[cy, cx, :3] = int(rng.integers(200, 256))             # white glyph pixels (r==g==b>=200)
    brq[qh - 1, qw - 1, :3] = 255                                 # white pixels the clamp makes unreachable as neighbours
    brq[qh - 2, qw - 2, :3] = 255
    brq[5, qw - 1, :3] = 230
    brq[0, 0, :3] = 210
    brq[10:14, 10:30, :3] = (4, 0, 3)                             # BGR: luma 0.2126*3 + 0.0722*4 = 0.93 -> 0 => scales 0
    brq[20:24, 10:30, :3] = (0, 2, 0)                             # luma 0.7152*2 = 1.43 -> 1 => scales 255
    tint = brq[40:60, 40:90, :3].astype(np.int16)
    tint[..., 0] += rng.integers(0, 14, tint.shape[:2])           # max-min up to 13 straddles the similarity threshold (12)
    brq[40:60, 40:90, :3] = np.clip(tint, 0, 255).astype(np.uint8)
    ref = o.process_frame(frame, stages=0xC, want_images=True, scales_start_y=7)
    vision.load_frame(frame)
    assert vision.crop_to_map(True) is not None
    assert np.array_equal(vision.ocr_preprocess(), ref["ocr"])
    assert int((ref["ocr"] != 255).sum()) > 100
    sc = vision.find_scales_preprocess(7)
    assert np.array_equal(sc[7:], ref["scales"][7:])
    assert (sc[10:14, 10:30] == 0).all() and (sc[20:24, 10:30] == 255).all()
    # rows above scales_start_y stay stale: a second call with a larger start leaves rows 7.. untouched
    sc2 = vision.find_scales_preprocess(qh // 2)
    assert np.array_equal(sc2, sc)
    with pytest.raises(smh.VisionError):
        vision.find_scales_preprocess(qh + 1)


def test_mask_edges_and_dilation_clipping(vision):
    """Marker pixels on all four ROI borders and corners: dilation must clip at the image edge and
    must not leak across the ROI boundary (pixels just outside the ROI are marker-coloured too)."""
    from squad_mortar_helper_amd import synth
    import squad_mortar_helper_amd as smh
    W, H = 1600, 1024
    frame, _ = synth.make_frame(W, H, 4, n_lines=0)
    x, y, rw, rh = smh.map_bounds(W, H)
    green = (0, 255, 64, 255)      # BGRA of RGB(64,255,0)
    frame[y - 1, x:x + rw] = green           # just outside (above)
    frame[y + rh, x:x + rw] = green          # just outside (below)
    frame[y:y + rh, x - 1] = green           # just outside (left)
    frame[y:y + rh, x + rw] = green          # just outside (right)
    for (px, py) in [(0, 0), (rw - 1, 0), (0, rh - 1), (rw - 1, rh - 1), (rw // 2, 0), (0, rh // 2), (rw - 1, rh // 3), (rw // 3, rh - 1),
                     (61, 61), (62, 62), (63, 63), (64, 64), (255, 61), (256, 62), (257, 123), (3, 124)]:
        frame[y + py, x + px] = green
    ref = o.process_frame(frame, stages=0x1, want_images=True)
    res = run_trait_sequence(vision, frame)
    lsd = vision.lsd_image()
    assert np.array_equal(lsd, ref["lsd"]) and ref["n_mask_px"] > 40
    assert np.array_equal(res.markers, ref["lines"])


def test_line_cap_of_32_and_many_rounds(vision):
    """A frame with far more than 32 acceptable segments stops at 32 exactly like lsd.rs:100-102."""
    from squad_mortar_helper_amd import synth
    import squad_mortar_helper_amd as smh
    W, H = 1920, 1080
    frame, _ = synth.make_frame(W, H, 9, n_lines=0)
    x, y, rw, rh = smh.map_bounds(W, H)
    for k in range(45):
        yy = y + 12 + 17 * k
        frame[yy:yy + 2, x + 30 + (k % 5) * 7: x + 30 + (k % 5) * 7 + 90 + 3 * k] = (217, 117, 192, 255)   # RGB(192,117,217)
    ref = o.process_frame(frame, stages=0x1, want_images=True)
    res = run_trait_sequence(vision, frame)
    assert ref["n_lines"] == 32 and res.markers.shape == (32, 4)
    assert np.array_equal(res.markers, ref["lines"])


def test_batch_matches_oracle_including_derived_outputs(vision):
    """Resident-batch entry point: mixed open/closed frames, per-frame anchors, every record field."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 1920, 1080, 12
    frames, infos = synth.make_batch(W, H, N, first_idx=100)
    frames[3], _ = synth.make_frame(W, H, 103, map_open=False)
    frames[7], _ = synth.make_frame(W, H, 107, n_lines=0)
    per = [(i["scales_start_y"], i["anchors"]) for i in infos]
    per[5] = (per[5][0], [])                                       # no OCR labels for frame 5 => no m/px
    per[6] = (per[6][0], per[6][1] + [(50, 20, 20)])               # a third, bogus label
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    # default (sector-culled) run first: every output but the sample count must already be the reference's
    fb.run(d.data_ptr(), N, anchors=smh.make_anchors(per), stream=torch.cuda.current_stream().cuda_stream)
    fast = smh.results_to_dicts(fb.read_results(0, N))
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=smh.make_anchors(per), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    recs = smh.results_to_dicts(fb.read_results(0, N))
    for a, b_ in zip(fast, recs):
        assert np.array_equal(a["lines"], b_["lines"]) and a["rounds"] == b_["rounds"] and a["mpx"] == b_["mpx"]
        assert a["ray_steps"] <= b_["ray_steps"] and np.array_equal(a["length_px"], b_["length_px"])
    for i in range(N):
        start_y, anc = per[i]
        ref = o.process_frame(frames[i], stages=0xF if anc else 0x7, anchors=anc, scales_start_y=start_y, want_images=True)
        r = recs[i]
        assert r["map_open"] == ref["map_open"]
        if not ref["map_open"]:
            assert r["n_lines"] == 0 and r["mpx"] is None and r["n_mask_px"] == 0
            continue
        assert r["red_pixels"] == o.button_red_pixels(frames[i])
        assert np.array_equal(r["lines"], ref["lines"]) and r["n_mask_px"] == ref["n_mask_px"]
        assert (r["rounds"], r["ray_steps"]) == (ref["rounds"], ref["steps"])       # identical ray trajectories
        assert r["mpx"] == ref["mpx"]
        assert np.array_equal(fb.read_image(smh._lib.IMAGE_UI_MAP, i), ref["ui_map"])
        assert np.array_equal(fb.read_image(smh._lib.VIEW_LSD_INPUT, i), ref["lsd"])
        assert np.array_equal(fb.read_image(smh._lib.VIEW_OCR_INPUT, i), ref["ocr"])
        if anc:
            assert np.array_equal(fb.read_image(smh._lib.VIEW_FIND_SCALES_INPUT, i)[start_y:], ref["scales"][start_y:])
        for k, ln in enumerate(ref["lines"]):
            length, meters = o.marker_new(ln, ref["mpx"] if ref["mpx"] is not None else 0.0)
            assert abs(r["length_px"][k] - length) <= TOL
            assert abs(r["meters"][k] - (meters if ref["mpx"] is not None else 0.0)) <= TOL
            assert abs(float(r["angle"][k]) - o.marker_angle(ln)) <= TOL
    fb.close()


def test_full_size_batch_properties(vision):
    """BASELINE configs[2] size (256 x 1080p): size-independent properties instead of a full oracle run:
    idempotence (a second run gives identical bytes), permutation equivariance (reversing the batch
    reverses the records), n_mask_px == popcount of the mask image, and a sample of frames against
    the oracle."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 1920, 1080, 256
    host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
    _, infos = synth.make_batch(W, H, N, out=host.numpy())
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = host.cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    s = torch.cuda.current_stream().cuda_stream
    fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
    a = bytes(fb.read_results(0, N))
    fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
    b = bytes(fb.read_results(0, N))
    assert a == b
    fast = smh.results_to_dicts(fb.read_results(0, N))
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=anchors, stream=s)
    recs = smh.results_to_dicts(fb.read_results(0, N))
    for x_, y_ in zip(fast, recs):                                # sector culling changes nothing but the sample count
        assert np.array_equal(x_["lines"], y_["lines"]) and x_["rounds"] == y_["rounds"] and x_["ray_steps"] <= y_["ray_steps"]
    assert sum(r["ray_steps"] for r in fast) < 0.6 * sum(r["ray_steps"] for r in recs)
    assert all(r["map_open"] == 1 and r["mpx"] == recs[0]["mpx"] for r in recs)
    for i in (0, 17, 101, 255):
        ref = o.process_frame(host.numpy()[i], stages=0xF, anchors=infos[i]["anchors"], scales_start_y=infos[i]["scales_start_y"], want_images=True)
        assert np.array_equal(recs[i]["lines"], ref["lines"]) and recs[i]["mpx"] == ref["mpx"]
        assert (recs[i]["rounds"], recs[i]["ray_steps"], recs[i]["n_mask_px"]) == (ref["rounds"], ref["steps"], ref["n_mask_px"])
        m = fb.read_image(smh._lib.VIEW_LSD_INPUT, i)
        assert int((m == 255).sum()) == recs[i]["n_mask_px"] and np.array_equal(m, ref["lsd"])
    rev = torch.flip(d, dims=[0]).contiguous()
    anchors_rev = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in reversed(infos)])
    fb.run(rev.data_ptr(), N, anchors=anchors_rev, stream=s)
    recs_rev = smh.results_to_dicts(fb.read_results(0, N))
    for i in range(N):
        r, q = recs[i], recs_rev[N - 1 - i]
        assert np.array_equal(r["lines"], q["lines"]) and r["rounds"] == q["rounds"] and r["n_mask_px"] == q["n_mask_px"]
    fb.close()


def test_1440p_batch_window_and_global_mask_paths(vision):
    """BASELINE configs[3] geometry: the bit-packed mask (176 KiB) does not fit LDS.  Frame 0 has
    markers spread over the whole ROI (bounding box > LDS window => global-memory mask path), the
    others use the LDS window path."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 2560, 1440, 6
    frames, infos = synth.make_batch(W, H, N, first_idx=40, n_lines=3)
    x, y, rw, rh = smh.map_bounds(W, H)
    green = (0, 255, 64, 255)
    frames[0, y + 2:y + 5, x + 2:x + 80] = green
    frames[0, y + rh - 6:y + rh - 3, x + rw - 90:x + rw - 3] = green
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    per = [(i["scales_start_y"], i["anchors"]) for i in infos]
    fb.run(d.data_ptr(), N, anchors=smh.make_anchors(per), stream=torch.cuda.current_stream().cuda_stream)
    fast = smh.results_to_dicts(fb.read_results(0, N))
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=smh.make_anchors(per), stream=torch.cuda.current_stream().cuda_stream)
    recs = smh.results_to_dicts(fb.read_results(0, N))
    for x_, y_ in zip(fast, recs):
        assert np.array_equal(x_["lines"], y_["lines"]) and x_["rounds"] == y_["rounds"]
    for i in range(N):
        ref = o.process_frame(frames[i], stages=0xF, anchors=per[i][1], scales_start_y=per[i][0], want_images=True)
        assert np.array_equal(recs[i]["lines"], ref["lines"]) and recs[i]["mpx"] == ref["mpx"]
        assert (recs[i]["rounds"], recs[i]["ray_steps"]) == (ref["rounds"], ref["steps"])
        assert np.array_equal(fb.read_image(smh._lib.VIEW_LSD_INPUT, i), ref["lsd"])
        assert np.array_equal(fb.read_image(smh._lib.IMAGE_UI_MAP, i), ref["ui_map"])
    fb.close()


def test_smoke_entry_point(vision):
    import __graft_entry__ as g
    g.smoke()


# ---------------------------------------------------------------------------------------------------
# stress cases for the restructured kernels (speculative candidate groups, batched ray walking,
# LDS hit compaction): every one compares complete outputs with the oracle
# ---------------------------------------------------------------------------------------------------
GREEN = (0, 255, 64, 255)       # BGRA of RGB(64,255,0)
PURPLE = (217, 117, 192, 255)   # BGRA of RGB(192,117,217)


def _blank(W, H, idx=0):
    from squad_mortar_helper_amd import synth
    import squad_mortar_helper_amd as smh
    frame, _ = synth.make_frame(W, H, idx, n_lines=0)
    return frame, smh.map_bounds(W, H)


def _check_markers(vision, frame, max_gap=15):
    ref = o.process_frame(frame, stages=0x1, max_gap=max_gap, want_images=True)
    res = run_trait_sequence(vision, frame, max_gap=max_gap)
    lsd = vision.lsd_image()
    assert np.array_equal(lsd, ref["lsd"])
    assert res.markers.shape == ref["lines"].shape and np.array_equal(res.markers, ref["lines"]), (res.markers, ref["lines"])
    r_fast, s_fast = vision.lsd_stats(max_gap, exact=False)
    r_exact, s_exact = vision.lsd_stats(max_gap, exact=True)
    assert r_fast == r_exact == ref["rounds"] and s_exact == ref["steps"] and s_fast <= s_exact
    return ref


def test_stress_dense_noise_many_isolated_candidates(vision):
    """Thousands of isolated marker pixels: more non-zero mask words than the LDS candidate list holds
    (segmented compaction), every candidate rejected, every ray dies in the first batch."""
    W, H = 1920, 1080
    frame, (x, y, rw, rh) = _blank(W, H, 21)
    rng = np.random.default_rng(5)
    n = 3000
    xs, ys = rng.integers(0, rw, n), rng.integers(0, rh, n)
    frame[y + ys, x + xs] = GREEN
    ref = _check_markers(vision, frame)
    assert ref["rounds"] > 2048


def test_stress_filled_areas_overflow_the_long_ray_queue(vision):
    """Large filled marker areas: most rays survive 64 samples, so the LDS queue of long rays overflows
    and rays are finished in place; also >64 pre-filter hits per wave (chunked hit compaction)."""
    W, H = 1920, 1080
    frame, (x, y, rw, rh) = _blank(W, H, 22)
    frame[y + 100:y + 420, x + 150:x + 560] = GREEN
    frame[y + 500:y + 760, x + 600:x + 980] = PURPLE          # touches the right border region
    frame[y + 600:y + 620, x + 0:x + 300] = GREEN              # starts at the left border
    _check_markers(vision, frame)


@pytest.mark.parametrize("max_gap", [0, 1, 2, 7, 22, 31, 32, 40, 49, 50, 100])
def test_stress_max_gap_values(vision, max_gap):
    """Gap thresholds around the 32-sample batch size (T <= 31: bit-trick state machine, T > 31: run loop)."""
    from squad_mortar_helper_amd import synth
    frame, info = synth.make_frame(1280, 1024, 30 + max_gap, n_lines=3)
    x, y, rw, rh = info["roi"]
    for k in range(12):                                        # dashed line: 6 px on, k px off
        x0 = x + 40 + 25 * k
        frame[y + 300:y + 303, x0:x0 + 6 + k] = GREEN
    _check_markers(vision, frame, max_gap=max_gap)


def test_stress_find_longest_line_odd_gaps_and_outside_points(vision):
    """Vision::find_longest_line with max_gap <= 0 / fractional / huge and start points outside the image."""
    frame, e, g = fx.load_fixture("point_intersect_png")
    run_trait_sequence(vision, frame)
    lsd = vision.lsd_image()
    h, w = lsd.shape
    ys, xs = np.nonzero(lsd == 255)
    inside = (float(xs[len(xs) // 2]), float(ys[len(ys) // 2]))
    pts = [inside, (-3.0, 10.0), (10.0, -2.5), (w + 5.0, 20.0), (30.0, h + 1.0), (w - 0.5, h - 0.5), (0.25, 0.75)]
    for gap in (0.0, -1.0, 0.5, 15.5, 31.0, 31.5, 64.0, 1e6):
        for p in pts:
            line, ln = vision.find_longest_line(p, gap)
            rl, rn = o.find_longest_line(lsd, p[0], p[1], gap)
            assert np.array_equal(line, rl) and ln == rn, (p, gap, line, rl, ln, rn)


def test_stress_lines_hugging_all_borders(vision):
    """Marker lines along and into all four ROI borders (rays leave the image: end-point quirk paths)."""
    W, H = 1600, 1024
    frame, (x, y, rw, rh) = _blank(W, H, 23)
    frame[y:y + 3, x + 20:x + rw - 20] = GREEN                # along the top edge
    frame[y + rh - 3:y + rh, x + 20:x + rw - 20] = PURPLE     # along the bottom edge
    frame[y + 40:y + rh - 40, x:x + 3] = GREEN                # left edge
    frame[y + 40:y + rh - 40, x + rw - 3:x + rw] = PURPLE     # right edge
    for k in range(200):                                       # diagonal into the bottom-right corner
        frame[y + rh - 1 - k, x + rw - 1 - k] = GREEN
        frame[y + rh - 1 - k, x + rw - 2 - k] = GREEN
    _check_markers(vision, frame)


def test_stress_4k_frame(vision):
    """3840x2160: 494 quads per row (8 waves per workgroup), 2160p ROI 1972x1644 does not fit LDS."""
    from squad_mortar_helper_amd import synth
    frame, info = synth.make_frame(3840, 2160, 3, n_lines=3)
    ref = o.process_frame(frame, stages=0xF, anchors=info["anchors"], scales_start_y=info["scales_start_y"], want_images=True)
    res = run_trait_sequence(vision, frame, anchors=info["anchors"])
    assert np.array_equal(res.map, ref["ui_map"]) and np.array_equal(vision.lsd_image(), ref["lsd"])
    assert np.array_equal(res.markers, ref["lines"]) and res.meters_to_px_ratio == ref["mpx"]
    assert np.array_equal(vision.ocr_preprocess(), ref["ocr"])


def test_stress_random_scenes(vision):
    """Random mixes of lines (all angles, 1-5 px thick), blobs, rings and noise at three sizes."""
    from squad_mortar_helper_amd import synth
    import squad_mortar_helper_amd as smh
    rng = np.random.default_rng(77)
    for trial, (W, H) in enumerate([(1024, 768), (1920, 1080), (1280, 1024), (1920, 1080), (2560, 1440), (1600, 1024)]):
        frame, _ = synth.make_frame(W, H, 50 + trial, n_lines=0)
        x, y, rw, rh = smh.map_bounds(W, H)
        roi = frame[y:y + rh, x:x + rw]
        for _ in range(int(rng.integers(1, 6))):
            col = GREEN if rng.random() < 0.5 else PURPLE
            p0 = rng.uniform([0, 0], [rw, rh]); ang = rng.uniform(0, 2 * np.pi); L = rng.uniform(30, 0.8 * min(rw, rh))
            t = np.linspace(0, 1, int(L * 2) + 2)
            px = np.clip(np.rint(p0[0] + np.cos(ang) * L * t), 0, rw - 1).astype(int)
            py = np.clip(np.rint(p0[1] + np.sin(ang) * L * t), 0, rh - 1).astype(int)
            th = int(rng.integers(1, 6))
            for dy in range(th):
                for dx in range(th):
                    roi[np.clip(py + dy, 0, rh - 1), np.clip(px + dx, 0, rw - 1)] = col
        for _ in range(int(rng.integers(0, 4))):               # blobs and rings
            cx, cy, r = int(rng.integers(20, rw - 20)), int(rng.integers(20, rh - 20)), int(rng.integers(4, 18))
            yy, xx = np.ogrid[-r:r + 1, -r:r + 1]
            d2 = xx * xx + yy * yy
            m = (d2 <= r * r) & ((d2 >= (r - 3) ** 2) if rng.random() < 0.5 else True)
            sub = roi[max(cy - r, 0):cy + r + 1, max(cx - r, 0):cx + r + 1]
            mm = m[max(r - cy, 0):max(r - cy, 0) + sub.shape[0], max(r - cx, 0):max(r - cx, 0) + sub.shape[1]]
            sub[mm] = GREEN
        k = int(rng.integers(0, 60))
        roi[rng.integers(0, rh, k), rng.integers(0, rw, k)] = PURPLE
        _check_markers(vision, frame, max_gap=int(rng.choice([15, 15, 22, 9])))


# ---------------------------------------------------------------------------------------------------
# find_minimap (the caller's step next to crop_to_map)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stem", fx.FULL_STEMS + ["in_mortar_png", "full_jpg"])
def test_find_minimap_matches_oracle_on_samples(vision, stem):
    frame, e, _ = fx.load_fixture(stem)
    res = run_trait_sequence(vision, frame)
    assert res.minimap_bounds == o.find_minimap(frame)


def test_find_minimap_synthetic_and_batch(vision):
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 1920, 1080, 6
    frames, infos = synth.make_batch(W, H, N, first_idx=300)
    x, y, rw, rh = smh.map_bounds(W, H)
    rng = np.random.default_rng(9)
    for i in range(1, N):                                        # flat rectangles of different extents around the centre
        l, r = int(rng.integers(5, rw // 2 - 5)), int(rng.integers(rw // 2 + 5, rw - 5))
        t, b = int(rng.integers(5, rh // 2 - 5)), int(rng.integers(rh // 2 + 5, rh - 5))
        frames[i, y + t:y + b, x + l:x + r, :3] = (40 + 30 * i, 90, 120)
        if i == 3:
            frames[i, y + t + 40:y + b - 40, x + l + 70, :3] = 255   # a bright vertical line inside: edge found earlier on one side
        if i == 4:
            frames[i, y + rh // 2 - 1, x + l:x + r, 1] = 97          # a faint (<= threshold) horizontal line through the centre row
    frames[5], _ = synth.make_frame(W, H, 305, map_open=False)
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_MINIMAP, stream=torch.cuda.current_stream().cuda_stream)
    recs = smh.results_to_dicts(fb.read_results(0, N))
    for i in range(N):
        want = o.find_minimap(frames[i]) if o.crop_to_map(frames[i]) is not None else None
        assert recs[i]["minimap"] == want, (i, recs[i]["minimap"], want)
        vision.load_frame(frames[i])
        if vision.crop_to_map(True) is not None:
            assert vision.find_minimap() == want
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL, stream=torch.cuda.current_stream().cuda_stream)
    assert all(r["minimap"] is None for r in smh.results_to_dicts(fb.read_results(0, N)))   # stage not selected
    fb.close()


def test_two_batches_in_flight_on_two_streams_do_not_interfere(vision):
    """bench.py keeps two steps in flight (two FrameBatch objects, two HIP streams): records must equal the
    ones produced one step at a time."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 1920, 1080, 48
    fa, ia = synth.make_batch(W, H, N, first_idx=500)
    fb_, ib = synth.make_batch(W, H, N, first_idx=700, n_lines=3)
    da, db = torch.from_numpy(fa).cuda(), torch.from_numpy(fb_).cuda()
    anc_a = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in ia])
    anc_b = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in ib])
    ba, bb = smh.FrameBatch(vision, W, H, N), smh.FrameBatch(vision, W, H, N)
    s0 = torch.cuda.current_stream()
    ba.run(da.data_ptr(), N, anchors=anc_a, stream=s0.cuda_stream)
    ref_a = bytes(ba.read_results(0, N))
    bb.run(db.data_ptr(), N, anchors=anc_b, stream=s0.cuda_stream)
    ref_b = bytes(bb.read_results(0, N))
    assert ref_a != ref_b
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(s0); s2.wait_stream(s0)
    for _ in range(4):                                           # A and B alternate, two steps in flight
        ba.run(da.data_ptr(), N, anchors=anc_a, stream=s1.cuda_stream)
        bb.run(db.data_ptr(), N, anchors=anc_b, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert bytes(ba.read_results(0, N)) == ref_a and bytes(bb.read_results(0, N)) == ref_b
    # the host-chosen stream assignment of bench.py: scales branches on caller streams, second step started half a
    # period after the first (smhv_batch_set_scales_stream / smhv_batch_wait_map_pass)
    s3, s4 = torch.cuda.Stream(), torch.cuda.Stream()
    ba.set_scales_stream(s3.cuda_stream); bb.set_scales_stream(s4.cuda_stream)
    for k in range(4):
        ba.run(da.data_ptr(), N, anchors=anc_a, stream=s1.cuda_stream)
        if k == 0:
            ba.wait_map_pass(s2.cuda_stream)
        bb.run(db.data_ptr(), N, anchors=anc_b, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert bytes(ba.read_results(0, N)) == ref_a and bytes(bb.read_results(0, N)) == ref_b
    ba.set_scales_stream(0); bb.set_scales_stream(0)
    ba.close(); bb.close()


def test_ocr_callback_hand_off(vision):
    """The scales branch with a pluggable OCR step: ocr_preprocess output goes to the
    callback, its hits are filtered like the reference filters Tesseract's, and the labels drive the scale scan."""
    import squad_mortar_helper_amd as smh
    frame, e, g = fx.load_fixture("point_intersect_png")
    seen = {}

    def fake_tesseract(img, w, h):
        seen["shape"] = (img.shape, w, h)
        seen["sha"] = sha(img)
        # boxes read off the screenshot; (left+right)/2 = 594, bottoms 433 / 465 (tests/golden/make_goldens.py)
        return [dict(text="300m", left=560, right=628, bottom=433), dict(text="Jensen's Training Range", left=300, right=600, bottom=500),
                dict(text="900m", left=559, right=629, bottom=465)]

    res = smh.VisionState().process(vision, frame, ocr=fake_tesseract)
    assert seen["shape"] == ((548, 657), 657, 548) and seen["sha"] == e["sha_ocr"]
    want = o.calc_meters_to_px_ratio([(300, 594, 433), (900, 594, 465)],
                                     o.find_scales_preprocess(o.crop_to_map(frame)["cropped_brq"], 433))
    assert res.meters_to_px_ratio == want and 3.7 < want < 4.0
    assert smh.VisionState().process(vision, frame, ocr=lambda img, w, h: []).meters_to_px_ratio is None


def test_stress_large_white_area_all_rays_leave_the_image(vision):
    """Marker-coloured bands along the top and bottom ROI borders: thousands of rays run to the image border (the
    reference's end-point quirk gives them length 0 there) and the long-ray queue overflows.  (A fully white ROI
    is the reference's own O(n^2) worst case -- 210,600 rounds of 3600 full-length rays, minutes even on the
    GPU -- so it is not part of the suite.)"""
    W, H = 1024, 768
    frame, (x, y, rw, rh) = _blank(W, H, 31)
    frame[y:y + 40, x:x + rw] = GREEN
    frame[y + 12:y + 28, x + 30:x + rw - 30] = (40, 40, 40, 255)         # hole: not every pixel is a candidate start
    frame[y + rh - 30:y + rh, x:x + rw] = PURPLE                          # and a solid band along the bottom border
    _check_markers(vision, frame)


def test_stress_checkerboard_of_blobs(vision):
    """A regular grid of small blobs closer together than max_gap: rays hop from blob to blob (many gap openings
    and closings per ray), lots of accepted lines up to the cap of 32."""
    W, H = 1280, 1024
    frame, (x, y, rw, rh) = _blank(W, H, 32)
    for by in range(10, rh - 10, 12):
        for bx in range(10, rw - 10, 12):
            frame[y + by:y + by + 3, x + bx:x + bx + 3] = PURPLE
    ref = _check_markers(vision, frame)
    assert ref["n_lines"] == 32


@pytest.mark.gpu
@pytest.mark.parametrize("max_gap", [3, 15, 30, 49])
def test_stress_segments_around_the_acceptance_length(vision, max_gap):
    """Sector culling (k_lsd casts only the 64-ray sectors that see a white pixel 50 - T .. 50 steps away) must
    never lose a line: a grid of short segments whose lengths straddle the len^2 > 2500 acceptance threshold, at
    all angles, solid and dashed with gaps of max_gap - 1 .. max_gap + 1 pixels."""
    W, H = 1920, 1080
    frame, (x, y, rw, rh) = _blank(W, H, 40 + max_gap)
    roi = frame[y:y + rh, x:x + rw]
    rng = np.random.default_rng(1000 + max_gap)
    cell = 130
    k = 0
    for gy in range(rh // cell):
        for gx in range(rw // cell):
            cx, cy = gx * cell + cell // 2, gy * cell + cell // 2
            L = 43.0 + (k % 16)                                  # 43 .. 58 px before dilation
            ang = rng.uniform(0, np.pi)
            gap = max(int(max_gap) - 1 + (k % 3), 0) if (k % 4) == 3 else 0   # every fourth one dashed
            t = np.arange(0.0, L, 0.5)
            on = np.ones_like(t, dtype=bool)
            if gap and L / 2 + gap < L:
                on[(t >= L / 2 - gap / 2.0) & (t < L / 2 + gap / 2.0)] = False
            px = np.rint(cx - L / 2 * np.cos(ang) + t * np.cos(ang)).astype(int)
            py = np.rint(cy - L / 2 * np.sin(ang) + t * np.sin(ang)).astype(int)
            roi[py[on], px[on]] = GREEN if k % 2 else PURPLE
            k += 1
    ref = _check_markers(vision, frame, max_gap=max_gap)
    assert 0 < len(ref["lines"]) <= 32


# ---------------------------------------------------------------------------------------------------
# ingest queue + device CRC-32 (the capture hand-off in front of load_frame)
# ---------------------------------------------------------------------------------------------------
def test_crc32_device_matches_zlib_and_oracle(vision):
    import zlib
    import torch
    import squad_mortar_helper_amd as smh
    rng = np.random.default_rng(21)
    # 4 B .. 40 MB: single group, ragged tails (scalar path), one full round of 1024 workgroups, several rounds
    for nbytes in (4, 8, 12, 16, 20, 4092, 4096, 65540, 1920 * 1080 * 4, 16 * 1024 * 1024, 16 * 1024 * 1024 + 4, 40 * 1000 * 1000):
        buf = rng.integers(0, 256, nbytes, dtype=np.uint8)
        d = torch.from_numpy(buf).cuda()
        got = smh.crc32_device(vision, d.data_ptr(), nbytes)
        assert got == zlib.crc32(buf.tobytes()), nbytes
        if nbytes <= 65540:
            assert got == o.crc32(buf)
    # misaligned device pointer (dword aligned only) takes the scalar load path
    buf = rng.integers(0, 256, 1 << 20, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    assert smh.crc32_device(vision, d.data_ptr() + 4, (1 << 20) - 8) == zlib.crc32(buf[4:-4].tobytes())
    # all-zero and all-ones messages (the init / final-xor terms are applied on the host)
    z = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    assert smh.crc32_device(vision, z.data_ptr(), 1 << 16) == zlib.crc32(bytes(1 << 16))
    with pytest.raises(smh.VisionError):
        smh.crc32_device(vision, z.data_ptr(), 7)


def test_ingest_queue_dedupes_like_the_capture_thread_and_feeds_the_batch(vision):
    import zlib
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = 1280, 1024
    A, ia = synth.make_frame(W, H, 1, n_lines=2)
    B, ib = synth.make_frame(W, H, 2, n_lines=1)
    Cc, ic = synth.make_frame(W, H, 3, n_lines=3)
    seq = [A, A, B, B, B, A, Cc, Cc]
    crcs = [zlib.crc32(f.tobytes()) for f in seq]
    keep, last = o.capture_dedupe(crcs)
    q = smh.IngestQueue(vision, W, H, slots=3, capacity=4)
    for k, f in enumerate(seq):
        if k % 2:
            q.push(f)                                      # pageable source
        else:
            buf = q.acquire()                              # capture straight into pinned staging
            buf[...] = f
            q.commit()
    ptr, n, crc = q.batch()
    assert n == int(keep.sum()) == 4 and crc == last and q.counts() == (4, 4)
    nb = W * H * 4
    fb = smh.FrameBatch(vision, W, H, n)
    fb.run(ptr, n, stages=0x3)
    recs = fb.read_results(0, n)
    want = [f for f, k in zip(seq, keep) if k]
    for r, f in zip(recs, want):
        ref = o.process_frame(f, stages=0x3)
        lines = np.array([[l.x0, l.y0, l.x1, l.y1] for l in r.lines[:r.n_lines]], np.float32).reshape(-1, 4)
        assert r.map_open == 1 and np.array_equal(lines, ref["lines"])
    for i, f in enumerate(want):
        assert smh.crc32_device(vision, ptr + i * nb, nb) == zlib.crc32(f.tobytes())
    # a slab frame feeds the per-call trait path without a host copy (capture.rs hand-off -> load_frame_device)
    vision.load_frame_device(ptr + 1 * nb, W, H)
    ui, roi = vision.crop_to_map(True)
    vision.isolate_map_markers(); vision.mask_marker_lines()
    refb = o.process_frame(want[1], stages=0x3, want_images=True)
    assert np.array_equal(ui, refb["ui_map"]) and np.array_equal(vision.find_marker_lines(15), refb["lines"])
    assert vision.get_cpu_frame() is None
    # next slab: dedupe continues against the last accepted frame (C); a frame beyond the capacity is neither an error
    # nor a drop: it stays queued and opens the next slab
    q.reset()
    for f in (Cc, A, B, A, B):
        q.push(f)
    assert q.batch()[1] == 4 and q.counts() == (8, 5)
    q.push(A)
    assert q.batch()[1] == 4 and q.counts() == (8, 5)
    q.reset()
    ptr, n, crc = q.batch()
    assert n == 1 and crc == zlib.crc32(A.tobytes()) and q.counts() == (9, 5)
    assert smh.crc32_device(vision, ptr, nb) == crc
    q.close()


def test_ingest_roi_upload_hashes_on_the_host_and_uploads_only_what_is_read(vision):
    """SMHV_INGEST_ROI_UPLOAD: the CRC-32 of the WHOLE frame is computed by the queue's host threads (so the duplicate rule
    of src/capture.rs:44-47 sees the bytes the reference hashes), only the map ROI's and the button's rows travel; the
    records, the images and the dedupe decisions equal the full-upload queue's and the oracle's."""
    import zlib
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    for (W, H) in ((1280, 1024), (1366, 768)):                   # (1366: the ROI's x is not quad aligned)
        fr = [synth.make_frame(W, H, 40 + i, n_lines=1 + i % 3)[0] for i in range(5)]
        outside = fr[1].copy()
        outside[0, 0, 0] ^= 1                                      # differs from fr[1] only OUTSIDE everything the pipeline reads
        seq = [fr[0], fr[0], fr[1], outside, outside, fr[2], fr[2], fr[3], fr[4], fr[4], fr[0]]
        crcs = [zlib.crc32(f.tobytes()) for f in seq]
        keep, last = o.capture_dedupe(crcs)
        want = [f for f, k in zip(seq, keep) if k]
        assert len(want) == 7                                      # `outside` is a new capture for the reference, and for the queue
        q = smh.IngestQueue(vision, W, H, slots=4, capacity=8, roi_upload=True)
        for k, f in enumerate(seq):
            if k % 2:
                q.push(f)
            else:
                q.acquire()[...] = f
                q.commit()
        ptr, n, crc = q.batch()
        assert n == len(want) and crc == last and q.counts() == (len(want), len(seq) - len(want))
        fb = smh.FrameBatch(vision, W, H, n)
        fb.run(ptr, n, stages=0x3)
        recs = fb.read_results(0, n)
        for r, f in zip(recs, want):
            ref = o.process_frame(f, stages=0x3)
            lines = np.array([[l.x0, l.y0, l.x1, l.y1] for l in r.lines[:r.n_lines]], np.float32).reshape(-1, 4)
            assert r.map_open == ref["map_open"] == 1 and np.array_equal(lines, ref["lines"])
        # the slab frames hold the two rectangles and zeros elsewhere
        import ctypes as C
        nb = W * H * 4
        x, y, rw, rh = o.map_bounds(W, H)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        for i, f in enumerate(want):
            fr_dev = np.empty((H, W, 4), np.uint8)
            assert hip.hipMemcpy(fr_dev.ctypes.data, ptr + i * nb, nb, 2) == 0            # hipMemcpyDeviceToHost
            assert np.array_equal(fr_dev[y:y + rh, x:x + rw], f[y:y + rh, x:x + rw])
            assert not fr_dev[:min(y, 8), :8].any() and fr_dev.sum() < f.sum()
        # a slab frame also feeds the per-call path
        vision.load_frame_device(ptr + 2 * nb, W, H)
        ui, roi = vision.crop_to_map(True)
        vision.isolate_map_markers(); vision.mask_marker_lines()
        refb = o.process_frame(want[2], stages=0x3, want_images=True)
        assert np.array_equal(ui, refb["ui_map"]) and np.array_equal(vision.find_marker_lines(15), refb["lines"])
        # full slab: the surplus stays queued, nothing is lost
        q.reset()
        more = [synth.make_frame(W, H, 70 + i, n_lines=1)[0] for i in range(8 + 4)]
        for f in more:
            q.push(f)
        with pytest.raises(smh.VisionError) as ei:
            q.push(fr[0])
        assert ei.value.code == smh._lib.E_STATE
        assert q.batch()[1] == 8
        q.reset()
        ptr, n, crc = q.batch()
        assert n == 4 and crc == zlib.crc32(more[-1].tobytes())
        q.acquire()
        with pytest.raises(smh.VisionError):                       # decoded pixel layouts need the device path
            q.commit_pixels("rgb")
        q.commit()
        q.close()


def test_stress_4k_sparse_candidates_exceed_the_row_cache(vision):
    """3840x2160: the mask does not fit LDS, k_lsd keeps a sliding cache of 434 rows.  Isolated marker pixels every
    ~90 rows make a speculative group of 8 candidates span more rows than the cache (the group is cut short and
    the rest handed back), and two long lines run the whole height (their long rays read outside the cache)."""
    W, H = 3840, 2160
    frame, (x, y, rw, rh) = _blank(W, H, 61)
    rng = np.random.default_rng(61)
    for k, yy in enumerate(range(30, rh - 30, 90)):
        frame[y + yy, x + 40 + int(rng.integers(0, rw - 80))] = GREEN if k % 2 else PURPLE
    for k in range(rh - 200):                                  # two steep lines, 3 px wide
        frame[y + 100 + k, x + 300 + k // 7:x + 303 + k // 7] = GREEN
        frame[y + 100 + k, x + rw - 300 - k // 5:x + rw - 297 - k // 5] = PURPLE
    ref = _check_markers(vision, frame)
    assert len(ref["lines"]) >= 2 and ref["rounds"] > 20
    # Vision::find_longest_line from points far apart in y (the cache is re-centred per call)
    lsd = vision.lsd_image()
    for p in [(303.0, 110.0), (float(rw - 301), 105.0), (300.0 + (rh - 300) // 7, float(rh - 120)), (50.0, 50.0)]:
        line, ln = vision.find_longest_line(p, 15.0)
        rl, rn = o.find_longest_line(lsd, p[0], p[1], 15.0)
        assert np.array_equal(line, rl) and ln == rn, (p, line, rl)


def test_c_example_runs_the_trait_sequence_without_python(vision, tmp_path):
    """examples/process_frame.c: the C ABI driven from plain C in its own process (no torch, no ctypes) on a sample
    screenshot; the printed lines are the oracle's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "process_frame")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "process_frame.c"),
                           "-L", os.path.join(root, "squad-mortar-helper_amd"), "-l:libsmh_vision_hip.so",
                           "-Wl,-rpath," + os.path.join(root, "squad-mortar-helper_amd"), "-o", exe])
    frame, e, g = fx.load_fixture("points_intersect_png")
    raw = tmp_path / "frame.bgra"
    frame.tofile(str(raw))
    out = subprocess.run([exe, str(raw), str(frame.shape[1]), str(frame.shape[0])], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ref = o.process_frame(frame, stages=0x1)
    got = [[float(v) for v in ln.replace("(", " ").replace(")", " ").replace("->", " ").replace(",", " ").split()]
           for ln in out.stdout.splitlines() if ln.startswith("  (")]
    assert len(got) == ref["n_lines"] == 7
    assert np.allclose(np.array(got), ref["lines"], atol=0.051)            # printed with one decimal


def test_decoded_images_enter_the_queue_as_the_reference_would_hash_them(vision):
    """The decode half of row f4 (src/ui/debug.rs:169): RGB / RGBA / L / LA pixels from a host decoder become the BGRA bytes of
    `into_bgra8` on the device, in front of the CRC and the duplicate rule (src/capture.rs:44-47)."""
    import zlib
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = 1282, 1023                                         # 1311486 pixels: not a multiple of the kernel's 4-pixel groups
    A, _ = synth.make_frame(W, H, 5, n_lines=2)
    rng = np.random.default_rng(11)
    A[..., 3] = rng.integers(0, 256, (H, W), dtype=np.uint8)   # a decoder's alpha is not always 255
    rgba = np.ascontiguousarray(A[..., [2, 1, 0, 3]])
    rgb = np.ascontiguousarray(A[..., [2, 1, 0]])
    lum = rng.integers(0, 256, (H, W), dtype=np.uint8)
    la = np.stack([lum, A[..., 3]], axis=-1)
    seq = [("rgba", rgba), ("bgra", A), ("rgb", rgb), ("rgb", rgb), ("l", lum), ("la", la), ("bgra", o.into_bgra8(la, "la"))]
    want = [o.into_bgra8(px, lay) for lay, px in seq]
    assert np.array_equal(want[0], A)
    keep, last = o.capture_dedupe([zlib.crc32(f.tobytes()) for f in want])
    assert keep.tolist() == [True, False, True, False, True, True, False]
    q = smh.IngestQueue(vision, W, H, slots=3, capacity=8)
    for k, (lay, px) in enumerate(seq):
        if k == 2:                                            # straight into pinned staging, then commit with the layout
            buf = q.acquire()
            buf.reshape(-1)[:px.size] = px.reshape(-1)
            q.commit_pixels(lay)
        else:
            q.push_pixels(px, lay)
    ptr, n, crc = q.batch()
    assert n == int(keep.sum()) and crc == last and q.counts() == (4, 3)
    nb = W * H * 4
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for i, f in enumerate([f for f, k in zip(want, keep) if k]):
        got = np.empty((H, W, 4), np.uint8)
        assert hip.hipMemcpy(got.ctypes.data, C.c_void_p(ptr + i * nb), nb, 2) == 0    # hipMemcpyDeviceToHost
        assert np.array_equal(got, f), i
        assert smh.crc32_device(vision, ptr + i * nb, nb) == zlib.crc32(f.tobytes())
    with pytest.raises(KeyError):
        q.push_pixels(rgb, "rgb16")
    with pytest.raises(ValueError):
        q.push_pixels(rgb, "rgba")


def test_load_frame_view_is_load_frame_of_the_repacked_rectangle(vision):
    """vision-gpu/src/lib.rs:175-179: a VisionFrame that is a rectangle of a larger image gives the results of its tight copy."""
    from squad_mortar_helper_amd import synth
    W, H = 1600, 900
    f, _ = synth.make_frame(W, H, 21, n_lines=3)
    rng = np.random.default_rng(3)
    parent = rng.integers(0, 256, (H + 37, W + 51, 4), dtype=np.uint8)
    x, y = 29, 17
    parent[y:y + H, x:x + W] = f
    ref = o.process_frame(f, stages=0xF, want_images=True)
    for view in (True, False):
        if view:
            vision.load_frame_view(parent, x, y, W, H)
        else:
            big = np.zeros((H + 5, W, 4), np.uint8); big[3:3 + H] = f
            vision.load_frame_view(big, 0, 3, W, H)            # full rows: the tight fast path
        ui, roi = vision.crop_to_map(True)
        vision.isolate_map_markers(); vision.mask_marker_lines()
        assert np.array_equal(ui, ref["ui_map"]) and np.array_equal(vision.find_marker_lines(15), ref["lines"])
        assert np.array_equal(vision.get_cpu_frame(), f)
    import squad_mortar_helper_amd as smh
    with pytest.raises(smh.VisionError):
        vision.load_frame_view(parent, W, 0, 100, 100)


def test_vision_state_gives_its_branch_threads_back(vision):
    """A VisionState that goes out of scope takes its two branch threads with it (they must not hold on to the state through
    the last job they ran)."""
    import gc
    import os
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    frame, info = synth.make_frame(1280, 1024, 5, n_lines=1)

    def threads():
        # (the interpreter's threads: the branch workers are among them; /proc/self/task also counts whatever the HIP runtime and
        # torch start lazily, which has nothing to do with a VisionState and made this test fail once the suite grew)
        import threading
        return len(threading.enumerate())
    st = smh.VisionState()
    st.process(vision, frame, ocr_labels=info["anchors"])
    base = threads() - 2                                         # (the first state's two workers are alive now)
    for _ in range(4):
        st = smh.VisionState()                                     # the previous one is dropped here
        st.process(vision, frame, ocr_labels=info["anchors"])
        gc.collect()
    assert threads() <= base + 2, (threads(), base)
    st.close()
    assert threads() <= base
