#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the reference's own sample
screenshots (/root/reference/vision-common/samples) and the CPU oracle (oracle/smh_oracle.c).

Runs only in the build container (the reference tree does not exist on the GPU box); the
fixtures it writes are plain data: input pixels + the oracle's expected outputs.

Two kinds of fixture (size budget: the 26 decodable frames would be ~28 MB even as lossless WebP):
  * "full"   -- the complete map ROI + button ROI pixels of the sample.  Every stage output is
                covered (ui_map, isolated map, mask, OCR-preprocess, scales, lines, m/px).
  * "sparse" -- only the 16x16 tiles of the map ROI that contain at least one marker-coloured
                pixel are kept (all other ROI pixels become (0,0,0), which is not a marker colour),
                plus the button ROI.  The script ASSERTS that the oracle's mask and line list on
                the sparse frame equal those on the original frame, so the expected marker/segment
                outputs are the real sample's outputs.

A frame is rebuilt from a fixture by `tests/fixtures.py:load_fixture` (background (32,32,32),
alpha 255, button ROI and map ROI pasted at the reference's bounds).

JPEG samples are decoded here with PIL; the fixture stores the decoded pixels losslessly, so the
goldens do not depend on any JPEG decoder afterwards (they are *not* cross-decoder goldens).
"""
import glob
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import oracle as o  # noqa: E402

SAMPLES = "/root/reference/vision-common/samples"
FULL = {"point_intersect.png", "points_intersect.png", "snowpoints.png", "full_1024x768.png",
        "full_1280x1024.png", "full_1600x1024.png", "tinyscales.png", "whiteout.png"}
TILE = 16
BG = (32, 32, 32)
# OCR label anchors (meters, (left+right)/2, bottom) in BRQ coordinates, read off the screenshots by
# hand (the "300m"/"900m" labels above the two scale bars); OCR itself is out of scope.  The third
# anchor is deliberately bogus (no bar below it) to exercise the Some/None averaging ladder.
REAL_ANCHORS = {
    "point_intersect.png": [(300, 594, 433), (900, 594, 465), (100, 50, 50)],
    "points_intersect.png": [(300, 594, 433), (900, 594, 465), (100, 50, 50)],
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def webp_bytes(rgb):
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "WEBP", lossless=True, quality=100, method=6, exact=True)
    back = np.array(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
    assert (back == rgb).all(), "webp roundtrip not lossless"
    return b.getvalue()


def rebuild(W, H, brect, btn_rgb, mrect, roi_rgb):
    f = np.empty((H, W, 4), np.uint8)
    f[:, :, 0], f[:, :, 1], f[:, :, 2], f[:, :, 3] = BG[2], BG[1], BG[0], 255
    bx, by, bw, bh = brect
    f[by:by + bh, bx:bx + bw, :3] = btn_rgb[:, :, ::-1]
    x, y, w, h = mrect
    f[y:y + h, x:x + w, :3] = roi_rgb[:, :, ::-1]
    return f


def find_anchors(scales_img, max_found=3):
    """Brute-force plausible OCR label anchors (OCR itself is out of scope): positions from which
    find_scale_width succeeds with distinct, reasonably wide bars; plus two failing anchors."""
    h, w = scales_img.shape
    found, widths = [], set()
    meters = [100, 300, 900]
    for y in range(8, h - 8, 3):
        for x in range(4, w - 4, 5):
            r = o.find_scale_width(100, x, y, scales_img)
            if r is None:
                continue
            _, (l, yy, rr, _) = r
            width = rr - l
            if width < 30 or width > w or (l, yy) in widths:
                continue
            widths.add((l, yy))
            found.append((meters[len(found)], x, y))
            if len(found) == max_found:
                return found
    return found


def main():
    os.makedirs(HERE, exist_ok=True)
    manifest = {}
    paths = sorted(glob.glob(SAMPLES + "/*.png")) + sorted(glob.glob(SAMPLES + "/*.jpg"))
    for p in paths:
        name = os.path.basename(p)
        stem = name.replace(".", "_")
        rgb = np.array(Image.open(p).convert("RGB"))
        H, W, _ = rgb.shape
        mrect, brect = o.map_bounds(W, H), o.button_bounds(W, H)
        entry = dict(source=name, W=W, H=H)
        if mrect is None or brect is None:
            entry["kind"] = "invalid_geometry"
            manifest[stem] = entry
            continue
        bgra_orig = np.ascontiguousarray(np.dstack([rgb[:, :, 2], rgb[:, :, 1], rgb[:, :, 0], np.full((H, W), 255, np.uint8)]))
        x, y, w, h = mrect
        bx, by, bw, bh = brect
        btn = rgb[by:by + bh, bx:bx + bw].copy()
        roi = rgb[y:y + h, x:x + w].copy()
        red = o.button_red_pixels(bgra_orig)
        ref = o.process_frame(bgra_orig, grayscale=True, max_gap=15, stages=0x1, want_images=True)
        entry.update(map_rect=list(mrect), button_rect=list(brect), red_pixels=red, map_open=ref["map_open"])
        if not ref["map_open"]:
            # keep closed-map samples as tiny fixtures: button ROI only, ROI blanked
            entry["kind"] = "closed"
            roi_keep = np.zeros_like(roi)
        elif name in FULL:
            entry["kind"] = "full"
            roi_keep = roi
        else:
            entry["kind"] = "sparse"
            iso = o.isolate_map_markers(roi)
            on = iso.any(axis=2)
            th, tw = -(-h // TILE), -(-w // TILE)
            pad = np.zeros((th * TILE, tw * TILE), bool)
            pad[:h, :w] = on
            tiles = pad.reshape(th, TILE, tw, TILE).any(axis=(1, 3))
            keep = np.repeat(np.repeat(tiles, TILE, 0), TILE, 1)[:h, :w]
            roi_keep = np.where(keep[:, :, None], roi, 0).astype(np.uint8)
        frame = rebuild(W, H, brect, btn, mrect, roi_keep)
        assert o.button_red_pixels(frame) == red
        with open(os.path.join(HERE, stem + ".roi.webp"), "wb") as f:
            f.write(webp_bytes(roi_keep))
        with open(os.path.join(HERE, stem + ".btn.webp"), "wb") as f:
            f.write(webp_bytes(btn))

        gold = {}
        if ref["map_open"]:
            res = o.process_frame(frame, grayscale=True, max_gap=15, stages=0x1, want_images=True)
            # the fixture frame must reproduce the ORIGINAL sample's marker/segment outputs
            assert (res["lsd"] == ref["lsd"]).all(), name
            assert res["n_lines"] == ref["n_lines"] and (res["lines"] == ref["lines"]).all(), name
            assert res["rounds"] == ref["rounds"] and res["steps"] == ref["steps"], name
            lsd = res["lsd"]
            gold["mask_idx"] = np.flatnonzero(lsd.reshape(-1) == 255).astype(np.uint32)
            gold["lines"] = res["lines"].astype(np.float32)
            lines22, st22 = o.find_lines(lsd, 22)   # the reference's own GPU test uses max_gap 22
            gold["lines_gap22"] = lines22
            entry.update(n_mask_px=int(res["n_mask_px"]), n_lines=int(res["n_lines"]), rounds=int(res["rounds"]),
                         steps=int(res["steps"]), rounds_gap22=st22["rounds"], sha_lsd=sha(lsd))
            c = o.crop_to_map(frame, True)
            iso = o.isolate_map_markers(c["cropped_map"])
            entry["n_marker_px"] = int(iso.any(axis=2).sum())
            entry["sha_isolated"] = sha(iso)
            entry["sha_ui_gray"] = sha(c["ui_map"])
            entry["sha_ui_color"] = sha(o.crop_to_map(frame, False)["ui_map"])
            entry["sha_brq"] = sha(c["cropped_brq"])
            ocr = o.ocr_preprocess(c["cropped_brq"])
            entry["sha_ocr"] = sha(ocr)
            sc0 = o.find_scales_preprocess(c["cropped_brq"], 0)
            entry["sha_scales0"] = sha(sc0)
            entry["n_ocr_kept"] = int((ocr != 255).sum())
            entry["n_scales_zero"] = int((sc0 == 0).sum())
            if entry["kind"] == "full":
                anchors = REAL_ANCHORS.get(name) or find_anchors(sc0)
                entry["anchors"] = [list(map(int, a)) for a in anchors]
                if anchors:
                    start_y = min(a[2] for a in anchors)
                    sc = o.find_scales_preprocess(c["cropped_brq"], start_y)
                    entry["scales_start_y"] = int(start_y)
                    entry["per_anchor"] = []
                    for a in anchors:
                        r = o.find_scale_width(a[0], a[1], a[2], sc)
                        entry["per_anchor"].append(None if r is None else dict(ratio=r[0], bar=list(r[1])))
                    mpx = o.calc_meters_to_px_ratio(anchors, sc)
                    entry["mpx"] = mpx
                    # derived marker outputs (src/ui/mod.rs:131-140, src/ui/markers.rs:98)
                    der = []
                    for ln in res["lines"]:
                        length, meters = o.marker_new(ln, mpx if mpx is not None else 0.0)
                        der.append([length, meters, o.marker_angle(ln)])
                    gold["derived"] = np.array(der, np.float64).reshape(-1, 3)
        np.savez_compressed(os.path.join(HERE, stem + ".golden.npz"), **gold)
        manifest[stem] = entry
        print(stem, entry["kind"], entry.get("n_mask_px"), entry.get("n_lines"), entry.get("rounds"), entry.get("anchors"), entry.get("mpx"))

    # the 3600-direction table as produced by glibc cosf/sinf in this image (pins csrc/ray_table.inc)
    dx, dy = o.ray_table()
    np.savez_compressed(os.path.join(HERE, "ray_table_glibc.npz"), dx=dx, dy=dy)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
