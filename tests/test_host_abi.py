"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/*.h declares,
its host-side logic (bounds arithmetic, error reporting, record layout) is right, and the
multi-process gather path works (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    # the boundary (smh_vision_hip.h) and the diagnostics beside it (smh_vision_hip_debug.h): everything include/*.h declares
    txt = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h"))
    return sorted(set(re.findall(r"SMHV_API\s+[\w\s\*]+?\b(smhv_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    from squad_mortar_helper_amd import _lib
    names = header_symbols()
    assert len(names) >= 31 and set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (\w+)", out))
    assert set(names) <= exported
    # nothing but the C ABI leaks out of the library (visibility=hidden), and no oracle symbol is linked in
    assert not [s for s in exported if s.startswith("orc_")]
    assert "libsmh_oracle" not in subprocess.check_output(["ldd", _lib.LIB_PATH], text=True)


def test_product_package_does_not_import_the_oracle(built):
    pkg = os.path.join(ROOT, "squad-mortar-helper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "smh_oracle" not in src, f


def test_bounds_match_oracle_for_many_sizes(built):
    import squad_mortar_helper_amd as smh
    from oracle import oracle as o
    sizes = [(1920, 1080), (2560, 1440), (1024, 768), (1280, 1024), (1600, 1024), (3840, 2160), (1366, 768), (1280, 720),
             (3440, 1440), (5120, 1440), (7680, 4320), (800, 600), (641, 479)]
    for w, h in sizes:
        om, ob = o.map_bounds(w, h), o.button_bounds(w, h)
        if om is None or ob is None:
            with pytest.raises(smh.VisionError) as ei:
                smh.map_bounds(w, h)
                smh.button_bounds(w, h)
            assert ei.value.code == -2
        else:
            assert smh.map_bounds(w, h) == om and smh.button_bounds(w, h) == ob
    for w, h in [(43, 44), (600, 1080), (1, 1)]:
        with pytest.raises(smh.VisionError) as ei:
            smh.map_bounds(w, h)
        assert ei.value.code == -2 and "frame" in str(ei.value)


def test_init_without_gpu_fails_loudly_not_silently(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import squad_mortar_helper_amd as smh
    with pytest.raises(smh.VisionError) as ei:
        smh.HipVision.init(0)
    assert ei.value.code == -4          # SMHV_E_NO_DEVICE: the caller falls back, the library never does


def test_record_layout_matches_header(built):
    from squad_mortar_helper_amd import _lib
    # smhv_frame_result: 8 + 32*16 + 8 + 4*4 + 8 + 32*8*2 + 32*4 + 6*4 = 1216
    assert C.sizeof(_lib.FrameResult) == 1216 and _lib.FrameResult.minimap.offset == 1192
    assert _lib.FrameResult.lines.offset == 8 and _lib.FrameResult.mpx.offset == 520
    assert _lib.FrameResult.ray_steps.offset == 544 and _lib.FrameResult.length_px.offset == 552
    assert C.sizeof(_lib.Anchors) == 44 and C.sizeof(_lib.Line) == 16
    # the per-frame status word (SMHV_FRAME_*) is the record's last field; the header's values are the binding's
    assert _lib.FrameResult.status.offset == 1212 and (_lib.FRAME_OK, _lib.FRAME_LSD_STUCK) == (0, 1)
    hdr = open(os.path.join(ROOT, "include", "smh_vision_hip.h")).read()
    assert "#define SMHV_FRAME_OK 0u" in hdr and "#define SMHV_FRAME_LSD_STUCK 1u" in hdr and "uint32_t status;" in hdr


def test_pipeline_options_and_flags_match_the_header(built):
    """smhv_pipeline_options as the binding declares it == the struct in the header, field for field (all uint32_t, in order),
    and the SMHV_SEARCH_* / SMHV_PIPE_* / SMHV_INGEST_* values the binding uses are the header's.  A C program compiled against
    the header agrees on the size."""
    import re
    import subprocess
    from squad_mortar_helper_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "smh_vision_hip.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} smhv_pipeline_options;", hdr).group(1)
    fields = re.findall(r"uint32_t\s+(\w+);", body)
    assert fields == [f[0] for f in _lib.PipelineOptions._fields_] and all(f[1] is C.c_uint32 for f in _lib.PipelineOptions._fields_)
    defs = dict(re.findall(r"#define (SMHV_(?:SEARCH|PIPE|INGEST)_\w+) (\d+)u", hdr))
    assert (int(defs["SMHV_SEARCH_AUTO"]), int(defs["SMHV_SEARCH_BATCH"]), int(defs["SMHV_SEARCH_FRAME"])) == (_lib.SEARCH_AUTO, _lib.SEARCH_BATCH, _lib.SEARCH_FRAME)
    assert (int(defs["SMHV_PIPE_NO_TEAM_HELP"]), int(defs["SMHV_PIPE_NO_STREAM_PRIORITY"]), int(defs["SMHV_PIPE_NO_PROLOGUE"])) == \
        (_lib.PIPE_NO_TEAM_HELP, _lib.PIPE_NO_STREAM_PRIORITY, _lib.PIPE_NO_PROLOGUE)
    assert (int(defs["SMHV_PIPE_NO_REMOTE_HELP"]), int(defs["SMHV_PIPE_HELP_FIRST"])) == (_lib.PIPE_NO_REMOTE_HELP, _lib.PIPE_HELP_FIRST)
    assert int(defs["SMHV_INGEST_ROI_UPLOAD"]) == 1 and int(defs["SMHV_INGEST_NO_AFFINITY"]) == 2 and sorted(k for k in defs if k.startswith("SMHV_PIPE_")) == \
        ["SMHV_PIPE_HELP_FIRST", "SMHV_PIPE_NO_PROLOGUE", "SMHV_PIPE_NO_REMOTE_HELP", "SMHV_PIPE_NO_STREAM_PRIORITY", "SMHV_PIPE_NO_TEAM_HELP", "SMHV_PIPE_THREE_LOAD_SETS",
         "SMHV_PIPE_WALK_BIT_ROWS"]
    assert (int(defs["SMHV_PIPE_WALK_BIT_ROWS"]), int(defs["SMHV_PIPE_THREE_LOAD_SETS"])) == (_lib.PIPE_WALK_BIT_ROWS, _lib.PIPE_THREE_LOAD_SETS)
    src = os.path.join(os.environ.get("TMPDIR", "/tmp"), "smhv_opt_size.c")
    exe = src[:-2]
    with open(src, "w") as f:
        f.write('#include <stdio.h>\n#include "smh_vision_hip.h"\nint main(void) { printf("%zu\\n", sizeof(smhv_pipeline_options)); return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    assert int(subprocess.check_output([exe]).decode()) == C.sizeof(_lib.PipelineOptions) == 48


def test_host_crc32_equals_zlib_for_ragged_lengths_and_alignments(built):
    """smhv_crc32_host (smh_crc_host.cpp: carry-less-multiply folding with constants derived from the polynomial, table loop
    for the head / tail and short messages) == zlib.crc32 == crc32fast::hash (src/capture.rs:44): the ingest queue's
    region-of-interest mode decides duplicates with it.  Needs no device."""
    import zlib
    from squad_mortar_helper_amd import ingest
    rng = np.random.default_rng(7)
    buf = rng.integers(0, 256, size=(1 << 20) + 64, dtype=np.uint8)
    for n in list(range(0, 200)) + [255, 256, 257, 1023, 4096, 65537, (1 << 20) - 3, 1 << 20]:
        for off in (0, 1, 5, 8, 15):
            a = buf[off:off + n]
            assert ingest.crc32_host(a) == zlib.crc32(a.tobytes()), (n, off)
    assert ingest.crc32_host(b"123456789") == 0xCBF43926           # the CRC-32/IEEE check value
    assert ingest.crc32_host(bytes(1 << 16)) == zlib.crc32(bytes(1 << 16))
    assert ingest.crc32_host(b"\xff" * 100000) == zlib.crc32(b"\xff" * 100000)


def test_every_host_crc32_loop_equals_zlib(built):
    """The library has three host loops for the CRC -- 512-bit VPCLMULQDQ folding (2048 bits per step), 128-bit PCLMULQDQ lanes,
    slicing-by-8 tables -- and picks by CPUID.  smhv_debug_crc32_host_level runs ONE of them (a loop the machine lacks falls
    back to the next lower one): every loop this machine has, against zlib, on lengths around every block size of the three
    (16, 64, 256, 512 bytes) and a frame-sized message, at several alignments."""
    import ctypes as C
    import zlib
    from squad_mortar_helper_amd import _lib
    lib = _lib.load()
    have = lib.smhv_debug_crc32_host_level(None, 0, -1, None)
    assert have in (0, 1, 2)
    rng = np.random.default_rng(11)
    buf = rng.integers(0, 256, size=(8 << 20) + 128, dtype=np.uint8)
    lengths = sorted(set(list(range(0, 80)) + [k + d for k in (128, 256, 512, 768, 1024, 2048, 61440, 65536) for d in (-17, -1, 0, 1, 15, 16, 63, 64, 65)] +
                         [1920 * 1080 * 4, (8 << 20) - 5]))
    c = C.c_uint32()
    for n in lengths:
        for off in (0, 3, 16, 33):
            ref = zlib.crc32(buf[off:off + n].tobytes())
            for level in range(have + 1):
                assert lib.smhv_debug_crc32_host_level(C.c_void_p(buf.ctypes.data + off), n, level, C.byref(c)) == have
                assert c.value == ref, (n, off, level)


def test_shard_range_partitions_exactly(built):
    from squad_mortar_helper_amd.dist import shard_range
    for n in (1, 7, 8, 255, 256, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert [shard_range(8192, r, 8) for r in range(8)] == [(1024 * r, 1024 * (r + 1)) for r in range(8)]


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import ctypes as C
from squad_mortar_helper_amd import _lib
from squad_mortar_helper_amd.dist import gather_records, records_from_bytes, shard_range, RECORD_BYTES
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N = 7                                     # uneven shards: 3 + 4
lo, hi = shard_range(N, rank, world)
recs = (_lib.FrameResult * (hi - lo))()
for i, f in enumerate(range(lo, hi)):
    recs[i].map_open = 1; recs[i].n_lines = f % 5; recs[i].rounds = 100 + f; recs[i].mpx = 0.5 * f; recs[i].has_mpx = 1
    recs[i].lines[0].x0 = float(f)
buf = torch.from_numpy(np.frombuffer(bytes(recs), np.uint8).copy())
out = gather_records(buf, dist)
if rank == 0:
    allrecs = []
    for t in out:
        allrecs += list(records_from_bytes(t.numpy()))
    assert len(allrecs) == N
    for f, r in enumerate(allrecs):
        assert r.rounds == 100 + f and r.n_lines == f % 5 and r.mpx == 0.5 * f and r.lines[0].x0 == float(f)
    # equal shards with known sizes: single collective
out2 = gather_records(buf[:3 * RECORD_BYTES].contiguous(), dist, sizes=[3 * RECORD_BYTES] * world)
if rank == 0:
    assert [t.numel() for t in out2] == [3 * RECORD_BYTES] * world
# the pre-allocated gather bench.py uses per pass (equal shards, one receive buffer set per pipeline slot)
from squad_mortar_helper_amd.dist import RecordGather
g = RecordGather(dist, 3, world, rank, device="cpu", slots=2)
for slot in (0, 1, 0):
    mine = buf[:3 * RECORD_BYTES].clone()
    mine[0] = 10 * slot + rank + 1                       # first byte of map_open: marks (slot, rank)
    g.run(mine, slot)
    if rank == 0:
        got = g.records(slot)
        assert len(got) == 3 * world and [got[3 * r].map_open & 0xFF for r in range(world)] == [10 * slot + r + 1 for r in range(world)]
if rank == 0:
    print("GATHER_OK")
dist.destroy_process_group()
'''


def test_gather_of_result_records_gloo_world2(built, tmp_path):
    """The N>1 path of bench.py (block shard + gather of per-frame records) on CPU with gloo."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script), ROOT]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "GATHER_OK" in p.stdout


def test_bench_launches_its_own_ranks(built):
    """`python bench.py --gpus 2` with no launcher in front (how the driver starts it) must start its two ranks itself -- as a
    child process, before anything touches the GPU -- and relay rank 0's JSON line and exit code.  --rendezvous-only stops
    after the ranks have met over gloo (no GPU here); the GPU suite runs the same launch with the real workload.  With
    --gpus > 1 the default workload is BASELINE configs[4]: 1024 frames per GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0 and out["baseline_config"] == 4 and out["global_batch"] == 2048
    # a rank that fails makes the launcher's exit code non-zero (an unknown flag here)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--no-such-flag"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # under a launcher whose world does not match --gpus: refuse
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rendezvous-only"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 2


def test_parse_ocr_labels_follows_the_reference_filter(built):
    """src/vision/mod.rs:150-196: ascii only, text up to the LAST 'm' must parse as a non-zero u32, anchor =
    ((left+right)/2, bottom), duplicates dropped (but still lower scales_start_y), at most 3."""
    from squad_mortar_helper_amd import parse_ocr_labels
    hits = [
        dict(text="300m", left=560, right=630, bottom=433),
        dict(text="Jensen's Training Range", left=300, right=620, bottom=500),   # 'm' absent -> skipped... has no digits
        dict(text="900m", left=561, right=628, bottom=465),
        dict(text="300m", left=10, right=20, bottom=400),                        # duplicate value: only lowers start_y
        dict(text="0m", left=1, right=2, bottom=3),                              # zero
        dict(text="12 m", left=1, right=2, bottom=3),                            # space does not parse
        dict(text="٣٠٠m", left=1, right=2, bottom=3),                            # not ascii
        dict(text="100mm", left=100, right=141, bottom=480),                     # "100m" before the last 'm' does not parse
        dict(text="+50m", left=40, right=60, bottom=470),                        # Rust u32 parse accepts a leading '+'
        dict(text="70m", left=1, right=3, bottom=490),                           # 4th distinct value: never reached
    ]
    scales, start_y = parse_ocr_labels(hits)
    assert scales == [(300, 595, 433), (900, 594, 465), (50, 50, 470)] and start_y == 400
    assert parse_ocr_labels([dict(text="map", left=0, right=1, bottom=2)]) == ([], None)
    assert parse_ocr_labels([]) == ([], None)
    assert parse_ocr_labels([dict(text="4294967296m", left=0, right=2, bottom=9)]) == ([], None)   # overflows u32


def test_header_is_plain_c_and_the_c_example_links(built, tmp_path):
    """include/smh_vision_hip.h is a C ABI: it must compile as pedantic C99, and a plain C program must link against
    the library with nothing but the header (examples/process_frame.c; it is run by a GPU test)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "examples", "process_frame.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), src])
    exe = str(tmp_path / "process_frame")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), src, "-L", os.path.join(root, "squad-mortar-helper_amd"),
                           "-l:libsmh_vision_hip.so", "-Wl,-rpath," + os.path.join(root, "squad-mortar-helper_amd"), "-o", exe])
    assert os.path.exists(exe)


class _RecordingVision:
    """Stands in for a Vision back-end: records the trait calls VisionState.process issues (no GPU)."""

    def __init__(self, fail_ui_map=False):
        import threading
        self.calls, self.threads, self.fail_ui_map, self._lock = [], {}, fail_ui_map, threading.Lock()

    def _rec(self, name):
        import threading
        with self._lock:
            self.calls.append(name)
            self.threads[name] = threading.get_ident()

    def load_frame(self, f): self._rec("load_frame")
    def crop_to_map(self, gray, lazy=False):
        self._rec("crop_to_map_lazy" if lazy else "crop_to_map_eager")
        return (None if lazy else np.full((2, 2, 4), 7, np.uint8)), [1, 2, 3, 4]
    def ui_map(self, copy=False):
        self._rec("ui_map")
        if self.fail_ui_map:
            raise RuntimeError("ui_map failed")
        return np.full((2, 2, 4), 9, np.uint8)
    def find_minimap(self): self._rec("find_minimap"); return None
    def thread_ctx(self): self._rec("thread_ctx")
    def isolate_map_markers(self): self._rec("isolate_map_markers")
    def mask_marker_lines(self): self._rec("mask_marker_lines")
    def find_marker_lines(self, gap): self._rec("find_marker_lines"); return np.ones((1, 4), np.float32)
    def ocr_preprocess(self): self._rec("ocr_preprocess"); return np.zeros((4, 4), np.uint8)
    def find_scales_preprocess(self, y): self._rec("find_scales_preprocess")
    def calc_meters_to_px_ratio(self, labels): self._rec("calc_meters_to_px_ratio"); return 0.5
    def get_debug_view(self, c): self._rec("get_debug_view"); return None


def test_vision_state_follows_the_reference_call_contract(built):
    """src/vision/mod.rs:121-124, 219-223: `heightmaps::is_set()` => no scales closure, `(markers(), Ok(None))` on the calling
    thread; otherwise `threads.join(markers, scales)`.  Also: a failing ui_map leaves no completion token behind."""
    import threading
    import squad_mortar_helper_amd as smh
    me = threading.get_ident()
    frame = np.zeros((4, 4, 4), np.uint8)
    labels = [(100, 1, 1)]
    # no heightmap (default): both branches, on two threads other than the caller's
    v, st = _RecordingVision(), smh.VisionState()
    res = st.process(v, frame, ocr_labels=labels)
    assert res.meters_to_px_ratio == 0.5 and len(res.markers) == 1 and res.map[0, 0, 0] == 9
    assert {"ocr_preprocess", "find_scales_preprocess", "calc_meters_to_px_ratio", "find_marker_lines"} <= set(v.calls)
    assert v.threads["find_marker_lines"] != me and v.threads["ocr_preprocess"] != me and v.threads["find_marker_lines"] != v.threads["ocr_preprocess"]
    assert v.calls[:3] == ["load_frame", "crop_to_map_lazy", "find_minimap"] and v.calls[-1] == "get_debug_view"
    # heightmap selected: the scales branch is not there at all
    v2, st2 = _RecordingVision(), smh.VisionState(heightmap_is_set=True)
    res2 = st2.process(v2, frame, ocr_labels=labels)
    assert res2.meters_to_px_ratio is None and len(res2.markers) == 1
    assert not {"ocr_preprocess", "find_scales_preprocess", "calc_meters_to_px_ratio"} & set(v2.calls)
    assert v2.threads["find_marker_lines"] == me
    # the trait-shaped sequence: the image comes back from crop_to_map, ui_map is never asked for
    v3, st3 = _RecordingVision(), smh.VisionState(lazy_map=False)
    res3 = st3.process(v3, frame, ocr_labels=labels)
    assert "crop_to_map_eager" in v3.calls and "ui_map" not in v3.calls and res3.map[0, 0, 0] == 7 and res3.meters_to_px_ratio == 0.5
    # detect_markers off: Default::default()
    v4 = _RecordingVision()
    res4 = smh.VisionState(detect_markers=False).process(v4, frame, ocr_labels=labels)
    assert "find_marker_lines" not in v4.calls and res4.markers.shape == (0, 4) and res4.meters_to_px_ratio == 0.5
    # ui_map raising: the error surfaces, both branches are waited for, and the NEXT frame waits for its own branches
    bad = _RecordingVision(fail_ui_map=True)
    with pytest.raises(RuntimeError):
        st.process(bad, frame, ocr_labels=labels)
    assert "find_marker_lines" in bad.calls and "calc_meters_to_px_ratio" in bad.calls
    for _, _, q_out in st._workers:
        assert q_out.empty()
    v5 = _RecordingVision()
    res5 = st.process(v5, frame, ocr_labels=labels)
    assert res5.meters_to_px_ratio == 0.5 and len(res5.markers) == 1
    for s in (st, st2, st3):
        s.close()


def test_band_heights_of_the_streaming_pass(built):
    """band_rows_for (smh_stream.hip) through smhv_debug_band_rows, no device: whole tile rows -- and with them the tile-major mask --
    where the ROI is at most 900 rows tall (frames up to 1080p: 24-row bands for a launch alone, 56 beside the search service) or
    they cost the launch no band (56), the kernel's own
    58 / 62 rows elsewhere; fewer frames than fill the chip: shorter bands, whole tile rows again at the end of the halving;
    smhv_debug_map_band_rows overrides the height of the launches that write the tile-major mask."""
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import _lib
    lib = _lib.load()

    def q(w, h, n, fused):
        rows, bands, tiles = C.c_uint32(), C.c_uint32(), C.c_int()
        _lib.check(lib.smhv_debug_band_rows(w, h, n, fused, C.byref(rows), C.byref(bands), C.byref(tiles)))
        return rows.value, bands.value, bool(tiles.value)
    assert q(1920, 1080, 256, 1) == (24, 35, True)          # 822 rows
    assert q(1920, 1080, 256, 0) == (24, 35, True)
    assert q(1920, 1080, 16, 1) == (16, 52, True)           # 35 bands x 16 frames do not fill the chip's 768 workgroup slots
    assert q(1920, 1080, 256, 2) == (56, 15, True)          # beside the search service (frame-granular pipelines): 56
    assert q(2560, 1440, 128, 2) == (58, 19, False)
    try:
        _lib.check(lib.smhv_debug_map_band_rows(56))
        assert q(1920, 1080, 256, 1) == (56, 15, True) and q(2560, 1440, 128, 1) == (56, 20, True)
        _lib.check(lib.smhv_debug_map_band_rows(32))
        assert q(1920, 1080, 256, 0) == (32, 26, True)
        assert lib.smhv_debug_map_band_rows(12) != 0         # whole tile rows only
    finally:
        _lib.check(lib.smhv_debug_map_band_rows(0))
    assert q(2560, 1440, 128, 1) == (58, 19, False)         # 1096 rows: 56 would cost a twentieth band
    assert q(2560, 1440, 128, 0) == (62, 18, False)
    assert q(3840, 2160, 64, 1) == (58, 29, False)
    assert q(1024, 768, 256, 1)[2] and q(1280, 1024, 256, 1)[2] and q(800, 600, 256, 1)[2]
    for (w, h) in [(1920, 1080), (2560, 1440), (3840, 2160), (1024, 768), (5120, 1440)]:
        for fused in (0, 1):
            rows, bands, tiles = q(w, h, 1, fused)           # one frame: bands of 8 rows
            assert rows == 8 and tiles and bands == -(-smh.map_bounds(w, h)[3] // 8)
            for n in (2, 5, 17, 40, 300):
                rows, bands, tiles = q(w, h, n, fused)
                assert 8 <= rows <= (58 if fused else 62) and tiles == (rows % 8 == 0) and bands == -(-smh.map_bounds(w, h)[3] // rows)
