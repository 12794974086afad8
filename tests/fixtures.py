"""Loads the committed golden fixtures (tests/golden/, written by tests/golden/make_goldens.py)."""
import json
import os

import numpy as np
from PIL import Image

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BG = (32, 32, 32)

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)

FRAME_STEMS = sorted(k for k, v in MANIFEST.items() if v["kind"] in ("full", "sparse", "closed"))
OPEN_STEMS = sorted(k for k, v in MANIFEST.items() if v["kind"] in ("full", "sparse"))
FULL_STEMS = sorted(k for k, v in MANIFEST.items() if v["kind"] == "full")


def load_fixture(stem):
    """-> (frame uint8[H,W,4] BGRA, manifest entry, golden dict of arrays)."""
    e = MANIFEST[stem]
    W, H = e["W"], e["H"]
    f = np.empty((H, W, 4), np.uint8)
    f[..., 0], f[..., 1], f[..., 2], f[..., 3] = BG[2], BG[1], BG[0], 255
    btn = np.array(Image.open(os.path.join(GOLDEN, stem + ".btn.webp")).convert("RGB"))
    roi = np.array(Image.open(os.path.join(GOLDEN, stem + ".roi.webp")).convert("RGB"))
    bx, by, bw, bh = e["button_rect"]
    x, y, w, h = e["map_rect"]
    f[by:by + bh, bx:bx + bw, :3] = btn[..., ::-1]
    f[y:y + h, x:x + w, :3] = roi[..., ::-1]
    g = dict(np.load(os.path.join(GOLDEN, stem + ".golden.npz")))
    return f, e, g
