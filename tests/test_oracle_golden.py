"""CPU tests (no GPU): the oracle against the committed goldens and against known answers.

The goldens under tests/golden/ were produced from the reference's own sample screenshots by
tests/golden/make_goldens.py; the workload counts asserted in test_appendix_b_counts were
recorded independently (numpy probe) before the C oracle existed.
"""
import re
import os

import numpy as np
import pytest

import fixtures as fx
from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# SURVEY.md Appendix B (independent numpy restatement): ROI geometry per frame size
GEOMETRY = {
    (1920, 1080): ((1657, 1031, 255, 41), (914, 178, 986, 822)),
    (2560, 1440): ((2209, 1374, 340, 55), (1219, 237, 1314, 1096)),
    (1024, 768): ((837, 733, 181, 29), (650, 126, 360, 585)),
    (1280, 1024): ((1030, 977, 242, 39), (867, 169, 394, 779)),
    (1600, 1024): ((1350, 977, 242, 39), (867, 169, 714, 779)),
}

# (marker px, mask px, skipped by proximity, rounds, ray steps, lines)
APPENDIX_B = {
    "in_mortar_png": (114, 293, 292, 1, 74660, 1),
    "point_png": (501, 1351, 1273, 78, 5326764, 1),
    "point2_png": (543, 2091, 2090, 1, 75250, 1),
    "lol_png": (437, 1574, 1500, 74, 5056614, 2),
    "point_far_png": (1076, 2646, 2567, 79, 5405513, 2),
    "point_intersect_png": (954, 2306, 2227, 79, 5405676, 2),
    "point_opposite_h_png": (991, 1866, 1865, 1, 77920, 1),
    "point_opposite_v_png": (1068, 2139, 2138, 1, 78263, 1),
    "points_png": (1560, 4400, 4396, 4, 301640, 4),
    "points_intersect_png": (2722, 6817, 6810, 7, 540087, 7),
    "snowpoints_png": (1338, 4781, 4429, 352, 28069232, 24),
    "vlcsnap-2022-05-11-06h03m39s483_png": (14, 61, 0, 61, 4057200, 0),
    "full_1024x768_png": (68, 230, 0, 230, 17117542, 0),
    "full_1280x1024_png": (92, 312, 0, 312, 23370451, 0),
    "full_1600x1024_png": (303, 1098, 393, 705, 56020456, 6),
}
# Appendix B provisional lines (to ~7 significant digits)
APPENDIX_B_LINES = {
    "point_png": [[719, 558, 1092.0632, 876.6817]],
    "in_mortar_png": [[669.5, 538, 673.18024, 646.93903]],
    "point2_png": [[1077, 153, 305.15057, 618.91766]],
    "point_opposite_h_png": [[618, 497.5, 227.76489, 916.48022]],
    "point_opposite_v_png": [[1146.5, 82.5, 649.35065, 458.63406]],
    "point_intersect_png": [[719, 558, 1092.0632, 876.6817], [971, 618.5, 720.685, 699.00909]],
    "lol_png": [[721, 560, 1092.6296, 875.2829], [960, 621.5, 825.10394, 664.85156]],
}


@pytest.mark.parametrize("size", sorted(GEOMETRY))
def test_geometry_matches_appendix_b(size):
    btn, mp = GEOMETRY[size]
    assert o.button_bounds(*size) == btn
    assert o.map_bounds(*size) == mp


def test_geometry_rejects_frames_the_reference_panics_on():
    assert o.map_bounds(43, 44) is None          # convolution.png
    assert o.map_bounds(600, 1080) is None       # portrait: map width underflows


def test_hsv_known_answers():
    # exact team colours (consts.toml:33-41) and corner cases of util/src/image.rs:159-187
    assert o.hsv(64, 255, 0) == (104, 100, 100)
    assert o.hsv(192, 117, 217) == (285, 46, 85)
    assert o.hsv(93, 232, 181) == (157, 59, 90)
    assert o.hsv(0, 0, 0) == (0, 0, 0)           # s = NaN -> 0
    assert o.hsv(255, 255, 255) == (0, 0, 100)
    assert o.hsv(255, 0, 0) == (0, 100, 100)
    assert o.hsv(255, 0, 1)[0] == 359            # negative hue wraps through modulo(h, 360)
    assert o.hsv(0, 0, 255) == (240, 100, 100)
    for rgb in [(64, 255, 0), (192, 117, 217), (93, 232, 181)]:
        assert o.is_any_map_marker_color(*rgb)
    for rgb in [(0, 0, 0), (255, 255, 255), (128, 128, 128), (255, 0, 0), (200, 180, 150)]:
        assert not o.is_any_map_marker_color(*rgb)


def test_luma_matches_image_crate_formula():
    rng = np.random.default_rng(1)
    for r, g, b in rng.integers(0, 256, (2000, 3)):
        l = np.float32(0.2126) * np.float32(r) + np.float32(0.7152) * np.float32(g) + np.float32(0.0722) * np.float32(b)
        assert o.lib().orc_luma8(int(r), int(g), int(b)) == int(l)
    assert o.lib().orc_luma8(255, 255, 255) == 255 and o.lib().orc_luma8(0, 0, 0) == 0


def test_ray_table_is_glibc_and_matches_committed_include():
    dx, dy = o.ray_table()
    g = np.load(os.path.join(fx.GOLDEN, "ray_table_glibc.npz"))
    assert np.array_equal(dx.view(np.uint32), g["dx"].view(np.uint32))
    assert np.array_equal(dy.view(np.uint32), g["dy"].view(np.uint32))
    txt = open(os.path.join(ROOT, "squad-mortar-helper_amd", "csrc", "ray_table.inc")).read()
    v = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", txt)], np.uint32).reshape(3600, 2)
    assert np.array_equal(v[:, 0], dx.view(np.uint32)) and np.array_equal(v[:, 1], dy.view(np.uint32))
    assert dx[0] == 1.0 and dy[0] == 0.0
    assert dx[900] == np.float32(-4.371139e-08)    # cos(90 deg) in f32 is not 0


def test_dilation_literal_imageproc_equals_cross():
    rng = np.random.default_rng(7)
    for shape, p in [((37, 53), 0.05), ((64, 64), 0.3), ((5, 9), 0.5), ((1, 17), 0.2), ((23, 1), 0.2), ((40, 40), 0.0)]:
        img = (rng.random(shape) < p).astype(np.uint8) * 255
        assert np.array_equal(o.dilate_l1_imageproc(img, 1), o.dilate_cross(img))
    one = np.zeros((9, 9), np.uint8)
    one[4, 4] = 255
    d = o.dilate_l1_imageproc(one, 1)
    assert int((d == 255).sum()) == 5 and d[4, 4] == d[3, 4] == d[5, 4] == d[4, 3] == d[4, 5] == 255
    corner = np.zeros((4, 4), np.uint8)
    corner[0, 0] = 7                                  # any non-zero value is foreground
    assert int((o.dilate_l1_imageproc(corner, 1) == 255).sum()) == 3


def test_find_longest_line_quirks():
    # a ray leaving through the right/bottom edge while still white has len 0,
    # one leaving through the left/top edge inspects column/row 0.
    img = np.zeros((40, 200), np.uint8)
    img[20, :] = 255                                  # full-width horizontal line
    line, ln = o.find_longest_line(img, 100.0, 20.0, 15.0)
    assert ln > 0 and line[2] < 100.0                 # the leftward ray wins; the rightward one is void
    img2 = np.zeros((40, 200), np.uint8)
    img2[20, 50:150] = 255
    line2, ln2 = o.find_longest_line(img2, 100.0, 20.0, 15.0)
    # both directions end on a black sample here (~50 px each way); the winner is a ray a fraction of a
    # degree off the axis that stays inside row 20 and is therefore marginally longer than the axial one.
    assert abs(np.sqrt(ln2) - 50.0) < 0.01 and int(line2[3]) == 20
    # gap tolerance: 15 missing samples are bridged, 16 are not
    img3 = np.zeros((9, 300), np.uint8)
    img3[4, 10:100] = 255
    img3[4, 115:200] = 255                            # 15-px gap
    _, l15 = o.find_longest_line(img3, 10.0, 4.0, 15.0)
    img3[4, 115] = 0                                  # 16-px gap
    _, l16 = o.find_longest_line(img3, 10.0, 4.0, 15.0)
    assert abs(np.sqrt(l15) - 190.0) < 0.01 and abs(np.sqrt(l16) - 90.0) < 0.01   # sub-pixel steps end just inside the last white pixel


def test_get_centre_half_pixel_midpoints():
    img = np.zeros((30, 30), np.uint8)
    img[10:14, 10:15] = 255                           # 5 wide, 4 tall
    cx, cy = o.get_centre(img, 10.0, 10.0)
    assert (cx, cy) == (12.0, 11.5)
    assert o.get_centre(img, 0.0, 0.0) == (0.0, 0.0)


def test_find_scale_width_and_ladder():
    img = np.full((60, 200), 255, np.uint8)
    img[30, 40:161] = 0
    img[30:37, 40] = 0
    img[30:37, 160] = 0
    r = o.find_scale_width(300, 100, 25, img)
    assert r is not None and r[1] == (41, 30, 159, 30) and r[0] == 300 / 118
    assert o.find_scale_width(300, 100, 3, img) is None            # y < MIN_SCALE_VERTICAL_BAR_HEIGHT
    assert o.find_scale_width(300, 100, 24, img) is None           # bar row is not within round(20/640*w)=6 rows [y, y+6)
    assert o.calc_meters_to_px_ratio([], img) is None
    assert o.calc_meters_to_px_ratio([(300, 100, 25)], img) == 300 / 118
    assert o.calc_meters_to_px_ratio([(300, 100, 25), (7, 5, 5)], img) == 300 / 118
    assert o.calc_meters_to_px_ratio([(300, 100, 25), (100, 100, 26)], img) == (300 / 118 + 100 / 118) / 2.0
    # ticks that reach below the image do not count (reference: unchecked read; oracle: non-zero)
    img2 = np.full((33, 200), 255, np.uint8)
    img2[30, 40:161] = 0
    img2[30:33, 40] = 0
    img2[30:33, 160] = 0
    assert o.find_scale_width(300, 100, 25, img2) is None


def test_closed_and_invalid_samples():
    assert fx.MANIFEST["convolution_png"]["kind"] == "invalid_geometry"
    for stem in ("a_point_png", "line_angle_png"):
        frame, e, _ = fx.load_fixture(stem)
        assert e["kind"] == "closed" and o.crop_to_map(frame) is None
        assert o.button_red_pixels(frame) == e["red_pixels"]


@pytest.mark.parametrize("stem", sorted(APPENDIX_B))
def test_appendix_b_counts(stem):
    frame, e, g = fx.load_fixture(stem)
    n_marker, n_mask, skipped, rounds, steps, n_lines = APPENDIX_B[stem]
    c = o.crop_to_map(frame, True)
    assert c is not None and tuple(c["roi"]) == tuple(e["map_rect"])
    iso = o.isolate_map_markers(c["cropped_map"])
    assert int(iso.any(axis=2).sum()) == n_marker
    mask = o.mask_marker_lines(iso)
    assert int((mask == 255).sum()) == n_mask and set(np.unique(mask)) <= {0, 255}
    lines, st = o.find_lines(mask, 15)
    assert (st["skipped"], st["rounds"], st["steps"], len(lines)) == (skipped, rounds, steps, n_lines)
    if stem in APPENDIX_B_LINES:
        assert np.allclose(lines, np.array(APPENDIX_B_LINES[stem], np.float32), rtol=0, atol=2e-3)


@pytest.mark.parametrize("stem", fx.OPEN_STEMS)
def test_oracle_reproduces_committed_goldens(stem):
    frame, e, g = fx.load_fixture(stem)
    res = o.process_frame(frame, stages=0x1, want_images=True)
    assert res["map_open"] == 1 and o.button_red_pixels(frame) == e["red_pixels"]
    assert np.array_equal(np.flatnonzero(res["lsd"].reshape(-1) == 255).astype(np.uint32), g["mask_idx"])
    assert np.array_equal(res["lines"], g["lines"])
    assert (res["rounds"], res["steps"]) == (e["rounds"], e["steps"])
    lines22, _ = o.find_lines(res["lsd"], 22)
    assert np.array_equal(lines22, g["lines_gap22"])


@pytest.mark.parametrize("stem", ["point_intersect_png", "points_intersect_png"])
def test_real_scale_bars(stem):
    frame, e, g = fx.load_fixture(stem)
    res = o.process_frame(frame, stages=0xF, anchors=e["anchors"], scales_start_y=e["scales_start_y"], want_images=True)
    assert res["mpx"] == e["mpx"] and 3.7 < res["mpx"] < 4.0       # 300 m over ~78 px, 900 m over ~237 px at 1440p
    der = g["derived"]
    for i, ln in enumerate(res["lines"]):
        length, meters = o.marker_new(ln, res["mpx"])
        assert abs(length - der[i, 0]) <= 1e-4 and abs(meters - der[i, 1]) <= 1e-4
        assert abs(o.marker_angle(ln) - der[i, 2]) <= 1e-4


def test_synthetic_frames_are_deterministic_and_marker_free_terrain(built):
    from squad_mortar_helper_amd import synth
    a, ia = synth.make_frame(1024, 768, 3)
    b, ib = synth.make_frame(1024, 768, 3)
    assert np.array_equal(a, b) and ia == ib
    blank, _ = synth.make_frame(1024, 768, 5, n_lines=0)
    res = o.process_frame(blank, stages=0x1)
    assert res["map_open"] == 1 and res["n_mask_px"] == 0 and res["n_lines"] == 0
    closed, _ = synth.make_frame(1024, 768, 5, map_open=False)
    assert o.process_frame(closed)["map_open"] == 0


# find_minimap: rects produced by the oracle on the full fixtures when the row was added.
# full_1600x1024: a 689 x 690 px minimap square inside the 714 x 779 ROI, as the screenshot shows.
MINIMAP_GOLDEN = {
    "full_1024x768_png": (7, 359, 128, 493), "full_1280x1024_png": (15, 393, 214, 584), "full_1600x1024_png": (24, 713, 54, 744),
    "point_intersect_png": (141, 1208, 29, 1095), "points_intersect_png": (141, 1208, 29, 1095), "snowpoints_png": (141, 1208, 29, 1095),
    "tinyscales_png": (15, 393, 11, 778), "whiteout_png": (657, 657, 548, 548),
}


@pytest.mark.parametrize("stem", sorted(MINIMAP_GOLDEN))
def test_find_minimap_goldens(stem):
    frame, e, _ = fx.load_fixture(stem)
    assert o.find_minimap(frame) == MINIMAP_GOLDEN[stem]


def test_find_minimap_hand_made():
    """A textured (minimap-like) rectangle around the ROI centre inside flat UI: each walk stops at the first flat
    pixel that has a long flat run perpendicular to the walk, i.e. just outside the textured rectangle."""
    from squad_mortar_helper_amd import synth
    W, H = 1024, 768
    frame, _ = synth.make_frame(W, H, 1, n_lines=0)             # terrain is per-pixel noise: edginess is high everywhere
    x, y, w, h = o.map_bounds(W, H)
    assert o.find_minimap(frame) == (0, w - 1, 0, h - 1)       # never finds a flat run: walks to the ROI borders
    tex = frame[y:y + h, x:x + w].copy()
    frame[y:y + h, x:x + w, :3] = 90                             # flat UI ...
    l, r, t, b = 60, 300, 100, 500
    frame[y + t:y + b, x + l:x + r] = tex[t:b, l:r]              # ... around a textured rectangle that holds the centre
    got = o.find_minimap(frame)
    assert got == (l - 1, r, t - 1, b), got                      # the last pixel whose 3x3 neighbourhood still touches the texture
    # a flat band through the centre that is shorter (10 rows) than the perpendicular run the walk demands changes nothing
    band = frame.copy()
    band[y + h // 2 - 5:y + h // 2 + 5, x + 70:x + 290, :3] = 90
    assert o.find_minimap(band)[:2] == (l - 1, r)


def test_crc32_restatement_and_capture_dedupe_rule():
    """src/capture.rs:34,44-47: crc32fast::hash == CRC-32/IEEE (zlib's crc32 is the independent second opinion),
    and the capture loop delivers a frame iff its CRC differs from the previous capture's (initial value 0)."""
    import zlib
    assert o.crc32(b"123456789") == 0xCBF43926            # the catalogue check value of CRC-32/ISO-HDLC
    assert o.crc32(b"") == 0
    rng = np.random.default_rng(11)
    for n in (1, 3, 4, 5, 63, 64, 65, 4096, 100001):
        buf = rng.integers(0, 256, n, dtype=np.uint8)
        assert o.crc32(buf) == zlib.crc32(buf.tobytes()), n
    keep, last = o.capture_dedupe([5, 5, 7, 7, 5, 0, 0, 9])
    assert keep.tolist() == [True, False, True, False, True, True, False, True] and last == 9
    keep, last = o.capture_dedupe([0, 0, 3])                # a first frame whose CRC is 0 is dropped (initial value 0)
    assert keep.tolist() == [False, False, True] and last == 3


def test_marker_predicate_second_restatement_all_colours():
    """Independent second restatement (vectorised numpy float32, written from util/src/image.rs:159-187 and
    vision-common/src/markers/mod.rs:17-54, sharing no code with the C oracle) of hsv + is_any_map_marker_color,
    compared with the C oracle on all 2^24 colours.  The reference has no known-answer test for this; two
    independently written restatements agreeing bit for bit is the strongest pin available here."""
    table = o.marker_table()                                  # uint32[2^24 / 32], bit (r << 16 | g << 8 | b)
    f32 = np.float32
    teams = [(105, 100, 100), (285, 46, 85), (158, 60, 91)]
    chunk = 1 << 21
    for base in range(0, 1 << 24, chunk):
        idx = np.arange(base, base + chunk, dtype=np.uint32)
        r = ((idx >> 16) & 255).astype(f32) / f32(255.0)
        g = ((idx >> 8) & 255).astype(f32) / f32(255.0)
        b = (idx & 255).astype(f32) / f32(255.0)
        mx = np.maximum(r, np.maximum(g, b)); mn = np.minimum(r, np.minimum(g, b))
        delta = mx - mn
        with np.errstate(divide="ignore", invalid="ignore"):
            h = np.where(mx == mn, f32(0.0),
                         np.where(mx == r, f32(60.0) * np.fmod((g - b) / delta, f32(6.0)),
                                  np.where(mx == g, f32(60.0) * ((b - r) / delta + f32(2.0)), f32(60.0) * ((r - g) / delta + f32(4.0))))).astype(f32)
            s = (f32(100.0) * delta) / mx
        v = f32(100.0) * mx
        hm = np.fmod(h, f32(360.0))
        hm = np.where(hm < 0, hm + f32(360.0), hm)
        H = np.nan_to_num(hm, nan=0.0).astype(np.int64)        # `as u16`: truncation, NaN -> 0 (values are in range)
        S = np.nan_to_num(s, nan=0.0).astype(np.int64)
        V = v.astype(np.int64)
        hit = np.zeros(chunk, bool)
        for mh, ms, mv in teams:
            sat_ok = (np.abs(ms - S) <= 15) | (np.abs(S - (ms - 50)) <= 15)
            hit |= (np.abs(mh - H) <= 15) & sat_ok & (np.abs(mv - V) <= 15)
        hit &= S >= 35
        want = ((table[idx >> 5] >> (idx & 31)) & 1).astype(bool)
        assert np.array_equal(hit, want), "first difference at colour %06x" % int(idx[np.nonzero(hit != want)[0][0]])


def test_lsd_second_restatement_matches_the_c_oracle():
    """tests/independent_lsd.py (numpy, vectorised over rays, written separately from the reference sources) against
    the C oracle: find_longest_line on assorted points / gaps, and whole find_lines runs on a hand-made mask and on
    the dilated masks of two sample screenshots (7 and 79 rounds)."""
    import independent_lsd as ind
    rng = np.random.default_rng(12)
    img = np.zeros((180, 260), np.uint8)
    for k in range(150):                                        # a thick diagonal, a horizontal dashed line, a blob, noise
        img[20 + k // 2:23 + k // 2, 30 + k:33 + k] = 255
    for k in range(0, 200, 23):
        img[140:143, 20 + k:20 + k + 12] = 255
    img[60:80, 200:222] = 255
    img[rng.integers(0, 180, 40), rng.integers(0, 260, 40)] = 255
    pts = [(31.0, 21.0), (100.5, 55.0), (210.0, 70.0), (25.0, 141.0), (0.0, 0.0), (259.0, 179.0), (130.25, 90.75)]
    for gap in (15.0, 3.0, 0.0, 40.0):
        for (x, y) in pts:
            la, na = ind.find_longest_line(img, x, y, gap)
            lb, nb = o.find_longest_line(img, x, y, gap)
            assert np.array_equal(la, lb) and na == nb, (x, y, gap, la, lb, na, nb)
    la, ra = ind.find_lines(img, 15)
    lb, st = o.find_lines(img, 15)
    assert np.array_equal(la, lb) and ra == st["rounds"]
    for stem in ("points_intersect_png", "point_intersect_png"):
        frame, e, g = fx.load_fixture(stem)
        ref = o.process_frame(frame, stages=0x1, want_images=True)
        la, ra = ind.find_lines(ref["lsd"], 15)
        assert np.array_equal(la, ref["lines"]) and ra == ref["rounds"], stem


def test_brq_stages_second_restatement():
    """Independent numpy restatement of ocr_preprocess / find_scales_preprocess (vision-cpu/src/lib.rs:39-53,173-251,
    image 0.23.14 luma) against the C oracle: threshold-heavy random quadrants and a real sample's quadrant."""
    f32 = np.float32

    def luma(rgb):
        l = f32(0.2126) * rgb[..., 0].astype(f32) + f32(0.7152) * rgb[..., 1].astype(f32) + f32(0.0722) * rgb[..., 2].astype(f32)
        return l.astype(np.uint8)                               # NumCast f32 -> u8 truncates

    def ocr(rgb):
        h, w, _ = rgb.shape
        c = rgb.astype(np.int32)
        mono = 2 * (np.abs(c[..., 0] - c[..., 1]) + np.abs(c[..., 0] - c[..., 2]) + np.abs(c[..., 1] - c[..., 2]))   # 9 ordered pairs
        W = (mono <= 3) & (c.min(axis=2) >= 200)
        E = (mono <= 48) & (c.min(axis=2) >= 130)
        Wn = W.copy()
        Wn[:, w - 2:] = False; Wn[h - 2:, :] = False             # neighbour ranges stop at min(x + 3, w - 3) / min(y + 3, h - 3)
        pad = np.zeros((h + 6, w + 6), bool); pad[3:h + 3, 3:w + 3] = Wn
        near = np.zeros((h, w), bool)
        for dy in range(7):
            for dx in range(7):
                near |= pad[dy:dy + h, dx:dx + w]
        keep = W | (E & near)
        return np.where(keep, 255 - luma(rgb), 255).astype(np.uint8)

    rng = np.random.default_rng(8)
    for (h, w) in ((61, 83), (40, 40), (7, 9)):
        g = rng.choice([128, 129, 130, 131, 198, 199, 200, 201, 255, 0, 1, 2], (h, w))[..., None] + rng.integers(-13, 14, (h, w, 3)) * (rng.random((h, w, 1)) < 0.6)
        rgb = np.clip(g, 0, 255).astype(np.uint8)
        assert np.array_equal(ocr(rgb), o.ocr_preprocess(rgb)), (h, w)
        for start in (0, 5, h):
            want = np.where(luma(rgb) != 0, 255, 0).astype(np.uint8)
            got = o.find_scales_preprocess(rgb, start)
            assert np.array_equal(got[start:], want[start:]) and not got[:start].any()
    frame, e, g_ = fx.load_fixture("point_intersect_png")
    brq = o.crop_to_map(frame)["cropped_brq"]
    assert np.array_equal(ocr(brq), o.ocr_preprocess(brq))


def test_into_bgra8_restatement_on_hand_vectors():
    """image 0.23.14 `into_bgra8` for 8-bit decoder outputs (src/ui/debug.rs:169): channel order and the alpha rule."""
    rgb = np.array([[[1, 2, 3], [250, 128, 0]]], np.uint8)
    assert o.into_bgra8(rgb, "rgb").tolist() == [[[3, 2, 1, 255], [0, 128, 250, 255]]]
    rgba = np.array([[[1, 2, 3, 4], [9, 8, 7, 0]]], np.uint8)
    assert o.into_bgra8(rgba, "rgba").tolist() == [[[3, 2, 1, 4], [7, 8, 9, 0]]]
    assert o.into_bgra8(np.array([[7, 200]], np.uint8), "l").tolist() == [[[7, 7, 7, 255], [200, 200, 200, 255]]]
    assert o.into_bgra8(np.array([[[7, 1], [200, 77]]], np.uint8), "la").tolist() == [[[7, 7, 7, 1], [200, 200, 200, 77]]]
    bgra = np.arange(16, dtype=np.uint8).reshape(1, 4, 4)
    assert np.array_equal(o.into_bgra8(bgra, "bgra"), bgra)
    # the round trip the fake-input path relies on: a BGRA frame saved as RGBA and decoded again is the same frame
    assert np.array_equal(o.into_bgra8(bgra[..., [2, 1, 0, 3]], "rgba"), bgra)
