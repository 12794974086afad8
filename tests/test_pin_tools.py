"""oracle/pin/diff_pin.py (the comparison of the real vision-cpu's dump with the committed goldens) must accept a dump that
IS the goldens and flag a corrupted one; rust/ holds the crate sources it refers to."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _pin_from_goldens():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    ref = {}
    for stem, e in man.items():
        src = e.get("source")
        if not src or not src.endswith(".png"):
            continue
        if not e.get("map_open"):
            ref[src] = {"W": e["W"], "H": e["H"], "map_open": 0}
            continue
        g = np.load(os.path.join(GOLDEN, stem + ".golden.npz"))
        r = {"W": e["W"], "H": e["H"], "map_open": 1, "map_rect": e["map_rect"], "sha_lsd": e["sha_lsd"], "n_mask_px": e["n_mask_px"],
             "sha_mask_idx": hashlib.sha256(np.ascontiguousarray(g["mask_idx"], "<u4").tobytes()).hexdigest()}
        for k in ("lines", "lines_gap22"):
            r[k] = np.ascontiguousarray(g[k], np.float32).view(np.uint32).reshape(-1, 4).tolist()
        for k in ("sha_ui_gray", "sha_ui_color", "sha_ocr", "sha_scales0", "sha_isolated", "sha_brq"):
            r[k] = e[k]
        ref[src] = r
    return ref


def test_diff_pin_accepts_the_goldens_and_flags_a_changed_endpoint(tmp_path):
    ref = _pin_from_goldens()
    good = tmp_path / "pin_good.json"
    good.write_text(json.dumps(ref))
    tool = os.path.join(ROOT, "oracle", "pin", "diff_pin.py")
    p = subprocess.run([sys.executable, tool, str(good)], capture_output=True, text=True)
    assert p.returncode == 0 and "0 mismatching" in p.stdout, p.stdout + p.stderr
    name = next(k for k, v in ref.items() if v.get("map_open") and len(v["lines"]) > 0)
    ref[name]["lines"][0][2] ^= 1                                   # one ulp in one end point
    bad = tmp_path / "pin_bad.json"
    bad.write_text(json.dumps(ref))
    p = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert p.returncode == 1 and "MISMATCH" in p.stdout and name in p.stdout


def test_rust_sources_are_present_and_name_the_c_abi():
    lib = open(os.path.join(ROOT, "rust", "smh-vision-hip", "src", "lib.rs")).read()
    hdr = open(os.path.join(ROOT, "include", "smh_vision_hip.h")).read()
    import re
    for fn in re.findall(r"fn (smhv_[a-z_]+)\(", lib):
        assert ("SMHV_API int %s(" % fn) in hdr or ("SMHV_API void %s(" % fn) in hdr or ("SMHV_API const char *%s(" % fn) in hdr, fn
    assert "export_dylib_wrapper!" in lib and "impl Vision for HipInstance" in lib
    pin = open(os.path.join(ROOT, "rust", "vision-cpu-pin", "pin_goldens.rs")).read()
    assert "CPUFallback" in pin and "find_marker_lines(22)" in pin
