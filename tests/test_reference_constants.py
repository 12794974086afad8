"""The only numbers the reference itself holds for this path are its constants: the tunables of
vision-common/src/consts/consts.toml:1-63, the two screen-relative rectangles of vision-common/src/consts/mod.rs:7-19, the luma
weights its CUDA side hard-codes (vision-gpu/cuda/cuda.cu:23-25), and a handful of literals in lsd.rs / mpx_ratio.rs.  This test
parses those files where they lie (build container only: /root/reference is not on the GPU box, and nothing of it is copied
into the repo) and checks that csrc/smh_consts.h (what the HIP kernels and the host runtime are built from) and
oracle/smh_oracle.c (the checker) carry exactly those values.  It does not lift parity above "unpinned by the reference" --
the reference has no golden vectors -- but it makes the reference-held numbers machine-checked instead of eyeballed."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "vision-common")), reason="the reference tree is only present in the build container")


def _toml_consts():
    """[NAME] / type = ".." / value = <scalar | [a, b, c]> blocks (the whole grammar toml-consts accepts here)."""
    text = open(os.path.join(REF, "vision-common", "src", "consts", "consts.toml")).read()
    out = {}
    for name, body in re.findall(r"^\[([A-Z0-9_]+)\]\s*\n((?:[^\[\n].*\n?)*)", text, re.M):
        ty = re.search(r'type\s*=\s*"(\w+)"', body).group(1)
        raw = re.search(r"value\s*=\s*(.+)", body).group(1).strip()
        conv = float if ty.startswith("f") else int
        out[name] = [conv(v) for v in raw.strip("[]").split(",")] if raw.startswith("[") else conv(raw)
    return out


def _defines(path):
    """#define NAME <number> -> {NAME: literal text} (suffixes f / u kept out of the value)."""
    out = {}
    for name, val in re.findall(r"^#define\s+(\w+)\s+(-?[0-9][0-9.]*)[fu]?\b", open(path).read(), re.M):
        out[name] = val
    return out


def _num(text):
    return float(text) if "." in text else int(text)


def _rust_bounds():
    text = open(os.path.join(REF, "vision-common", "src", "consts", "mod.rs")).read()
    out = {}
    for name, body in re.findall(r"pub const (\w+): RelativeBounds2D = RelativeBounds2D \{(.*?)\};", text, re.S):
        fields = {}
        for f, corner, frac in re.findall(r"(\w):\s*(?:(Left|Right|Top|Bottom)\()?ScreenH\(([0-9.]+)\)\)?", body):
            fields[f] = (corner, frac)
        out[name] = fields
    return out


def test_consts_toml_equals_the_header_and_the_oracle():
    toml = _toml_consts()
    hdr = _defines(os.path.join(ROOT, "squad-mortar-helper_amd", "csrc", "smh_consts.h"))
    orc_text = open(os.path.join(ROOT, "oracle", "smh_oracle.c")).read()
    orc = _defines(os.path.join(ROOT, "oracle", "smh_oracle.c"))
    assert len(toml) == 17, sorted(toml)                      # every table of the file was parsed (consts.toml:1-67)

    # header name <- toml name
    scalar = {
        "SMH_BUTTON_TOLERANCE": "CLOSE_DEPLOYMENT_BUTTON_TOLERANCE",
        "SMH_BUTTON_RED_PIXEL_THRESHOLD": "CLOSE_DEPLOYMENT_BUTTON_RED_PIXEL_THRESHOLD",
        "SMH_OCR_BRIGHTNESS_THRESHOLD": "OCR_PREPROCESS_BRIGHTNESS_THRESHOLD",
        "SMH_OCR_MONOCHROMATICY_THRESHOLD": "OCR_PREPROCESS_MONOCHROMATICY_THRESHOLD",
        "SMH_OCR_BRIGHTNESS_EDGE_THRESHOLD": "OCR_PREPROCESS_BRIGHTNESS_EDGE_THRESHOLD",
        "SMH_OCR_SIMILARITY_EDGE_THRESHOLD": "OCR_PREPROCESS_SIMILARITY_EDGE_THRESHOLD",
        "SMH_OCR_DILATE_RADIUS": "OCR_PREPROCESS_DILATE_RADIUS",
        "SMH_HSV_HUE_TOLERANCE": "FIND_MARKER_HSV_HUE_TOLERANCE",
        "SMH_HSV_SAT_TOLERANCE": "FIND_MARKER_HSV_SAT_TOLERANCE",
        "SMH_HSV_VIB_TOLERANCE": "FIND_MARKER_HSV_VIB_TOLERANCE",
        "SMH_HSV_MIN_SAT": "FIND_MARKER_HSV_MIN_SAT",
        "SMH_PLAYER_DIR_ARC_SAT": "FIND_MARKER_PLAYER_DIR_ARC_SAT",
    }
    for h, t in scalar.items():
        assert _num(hdr[h]) == toml[t], (h, hdr[h], toml[t])
        assert _num(orc[t]) == toml[t], (t, orc[t], toml[t])  # the oracle keeps the reference's own names
    assert [_num(hdr["SMH_BUTTON_" + c]) for c in "RGB"] == toml["CLOSE_DEPLOYMENT_BUTTON_COLOR"]
    for team in ("ALPHA", "BRAVO", "CHARLIE"):
        assert [_num(hdr["SMH_%s_%s" % (team, c)]) for c in "HSV"] == toml[team + "_MARKER_COLOR_HSV"]
    m = re.search(r"CLOSE_DEPLOYMENT_BUTTON_COLOR\[3\]\s*=\s*\{([^}]*)\}", orc_text)
    assert [int(v) for v in m.group(1).split(",")] == toml["CLOSE_DEPLOYMENT_BUTTON_COLOR"]
    m = re.search(r"MARKER_HSV\[3\]\[3\]\s*=\s*\{\{([^;]*)\}\};", orc_text)
    teams = [[int(v) for v in t.split(",")] for t in m.group(1).split("}, {")]
    assert teams == [toml["ALPHA_MARKER_COLOR_HSV"], toml["BRAVO_MARKER_COLOR_HSV"], toml["CHARLIE_MARKER_COLOR_HSV"]]


def test_screen_relative_bounds_equal_consts_mod_rs():
    rb = _rust_bounds()
    hdr = _defines(os.path.join(ROOT, "squad-mortar-helper_amd", "csrc", "smh_consts.h"))
    orc_text = open(os.path.join(ROOT, "oracle", "smh_oracle.c")).read()
    mb, bb = rb["MAP_BOUNDS"], rb["CLOSE_DEPLOYMENT_BUTTON_BOUNDS"]
    # corners: the header's names say which edge a coordinate is measured from
    assert mb["x"][0] == "Left" and mb["y"][0] == "Bottom" and bb["x"][0] == "Right" and bb["y"][0] == "Bottom"
    pairs = {"SMH_MAP_X": mb["x"][1], "SMH_MAP_Y_BOTTOM": mb["y"][1], "SMH_MAP_W": mb["w"][1], "SMH_MAP_H": mb["h"][1],
             "SMH_BTN_X_RIGHT": bb["x"][1], "SMH_BTN_Y_BOTTOM": bb["y"][1], "SMH_BTN_W": bb["w"][1], "SMH_BTN_H": bb["h"][1]}
    for h, frac in pairs.items():
        assert hdr[h] == frac, (h, hdr[h], frac)             # the literal text: these are f64 and every digit counts
        assert re.search(r"screen_h\(%s, H\)" % re.escape(frac), orc_text), ("oracle", h, frac)


def test_luma_weights_and_lsd_literals():
    hdr = _defines(os.path.join(ROOT, "squad-mortar-helper_amd", "csrc", "smh_consts.h"))
    orc_text = open(os.path.join(ROOT, "oracle", "smh_oracle.c")).read()
    cu = open(os.path.join(REF, "vision-gpu", "cuda", "cuda.cu")).read()
    luma = dict(re.findall(r"#define LUMA_([RGB]) ([0-9.]+)f", cu))
    assert sorted(luma) == ["B", "G", "R"]
    for c in "RGB":
        assert hdr["SMH_LUMA_" + c] == luma[c]
    assert re.search(r"%sf \* \(float\)r \+ %sf \* \(float\)g \+ %sf \* \(float\)b" % (luma["R"], luma["G"], luma["B"]), orc_text)

    lsd = open(os.path.join(REF, "vision-common", "src", "lsd.rs")).read()
    cpu = open(os.path.join(REF, "vision-cpu", "src", "lib.rs")).read()
    mpx = open(os.path.join(REF, "src", "vision", "mpx_ratio.rs")).read()
    lib = open(os.path.join(REF, "vision-common", "src", "lib.rs")).read()
    accept = re.search(r"if max_length > ([0-9.]+)", lsd).group(1)               # lsd.rs:94
    prox = re.search(r"\.powi\(2\) < ([0-9.]+)", lsd).group(1)                     # lsd.rs:86
    reach = re.search(r"const MAX_DIST: f32 = ([0-9.]+);", lsd).group(1)          # lsd.rs:9
    rays = re.search(r"\(0\.\.(\d+)_u32\)", cpu[cpu.index("fn find_longest_line"):]).group(1)   # vision-cpu/src/lib.rs:434
    nmax = re.search(r"SmallVec<Line<f32>, (\d+)>", lib).group(1)
    assert float(hdr["SMH_LSD_ACCEPT_LEN_SQ"]) == float(accept) and ("max_length > %sf" % accept) in orc_text
    assert float(hdr["SMH_LSD_PROXIMITY_SQ"]) == float(prox) and ("< %sf" % prox) in orc_text
    assert float(hdr["SMH_LSD_CENTRE_REACH"]) == float(reach)
    assert int(hdr["SMH_LSD_RAYS"]) == int(rays) and ("i < %s;" % rays) in orc_text
    assert int(hdr["SMH_LSD_MAX_LINES"]) == int(nmax)
    assert int(hdr["SMH_MIN_SCALE_WIDTH"]) == int(re.search(r"MIN_SCALE_WIDTH: u32 = (\d+)", mpx).group(1))
    assert int(hdr["SMH_MIN_SCALE_VERTICAL_BAR_HEIGHT"]) == int(re.search(r"MIN_SCALE_VERTICAL_BAR_HEIGHT: u32 = (\d+)", mpx).group(1))
