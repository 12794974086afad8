"""GPU parity tests (-m gpu) for the BASELINE.json configurations at their FULL sizes, through the C ABI:

  configs[1]  1 x 1920x1080, marker threshold + LSD only (smhv_batch_run with n = 1, stages = SMHV_STAGE_MARKERS)
  configs[3]  128 x 2560x1440 (the mask does not fit LDS: all three mask residency paths of k_lsd in one launch)
  configs[4]  the per-GPU shard of the 8-GPU run: 1024 x 1920x1080 resident frames (8.5 GB, byte offsets beyond 2^32)

Sizes the oracle cannot cover in seconds are checked through size-independent properties (idempotence, permutation
equivariance, periodicity of a tiled batch, popcounts) plus a sample of frames against the oracle; then bounded slices of
the two fuzzers (tools/fuzz_lsd.py, tools/fuzz_stream.py) so that the driver's own run sees random scenes too."""
import os

import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-4
GREEN = (0, 255, 64, 255)       # BGRA of RGB(64,255,0)
PURPLE = (217, 117, 192, 255)   # BGRA of RGB(192,117,217)


def _lines(r):
    return np.array([[r.lines[a][b] for b in range(4)] for a in range(r.n_lines)], np.float32).reshape(-1, 4)


def test_config1_batch_of_one_markers_only(vision):
    """BASELINE configs[1]: batch = 1, 1920x1080 synthetic map, marker threshold + LSD only -- exactly the call the
    bench line `--config 1` times.  Every record field and the mask against the oracle; ui_map / ocr / scales stages are
    not run (their record fields stay empty)."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = 1920, 1080
    fb = smh.FrameBatch(vision, W, H, 1)
    s = torch.cuda.current_stream().cuda_stream
    for idx, n_lines in ((0, 2), (1, 1), (2, 4), (3, 0)):
        frame, info = synth.make_frame(W, H, idx, n_lines=n_lines)
        d = torch.from_numpy(frame).cuda()
        fb.run(d.data_ptr(), 1, stages=smh.STAGE_MARKERS, max_gap=15, stream=s)
        fast = smh.results_to_dicts(fb.read_results(0, 1))[0]
        fb.run(d.data_ptr(), 1, stages=smh.STAGE_MARKERS | smh.STAGE_EXACT_STATS, max_gap=15, stream=s)
        r = smh.results_to_dicts(fb.read_results(0, 1))[0]
        ref = o.process_frame(frame, stages=0x1, max_gap=15, want_images=True)
        assert r["map_open"] == ref["map_open"] == 1 and r["red_pixels"] == o.button_red_pixels(frame)
        assert np.array_equal(r["lines"], ref["lines"]) and np.array_equal(fast["lines"], ref["lines"])
        assert (r["rounds"], r["ray_steps"], r["n_mask_px"]) == (ref["rounds"], ref["steps"], ref["n_mask_px"])
        assert fast["rounds"] == ref["rounds"] and fast["ray_steps"] <= ref["steps"]
        assert r["mpx"] is None and r["minimap"] is None
        m = fb.read_image(smh._lib.VIEW_LSD_INPUT, 0)
        assert np.array_equal(m, ref["lsd"])
        assert np.array_equal(np.flatnonzero(m.reshape(-1) == 255), np.flatnonzero(ref["lsd"].reshape(-1) == 255))   # marker pixel coords
        for k, ln in enumerate(ref["lines"]):
            length, _ = o.marker_new(ln, 0.0)
            assert abs(r["length_px"][k] - length) <= TOL and r["meters"][k] == 0.0
            assert abs(float(r["angle"][k]) - o.marker_angle(ln)) <= TOL
    # a closed map through the same entry point
    frame, _ = synth.make_frame(W, H, 9, map_open=False)
    d = torch.from_numpy(frame).cuda()
    fb.run(d.data_ptr(), 1, stages=smh.STAGE_MARKERS, stream=s)
    r = smh.results_to_dicts(fb.read_results(0, 1))[0]
    assert r["map_open"] == 0 and r["n_lines"] == 0 and r["n_mask_px"] == 0 and r["rounds"] == 0
    fb.close()


def test_config3_full_size_1440p_batch_properties(vision):
    """BASELINE configs[3] at its full size, 128 x 2560x1440: idempotence, permutation equivariance, mask popcounts, the
    culled run against the exact-statistics run, and frames against the oracle -- among them three frames built to take
    the three mask residency paths of k_lsd (whole rows in LDS, bounding-box window in LDS, mask in global memory with
    an LDS row cache), which run as concurrent kernels over the same record array."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 2560, 1440, 128
    host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
    frames = host.numpy()
    _, infos = synth.make_batch(W, H, N, first_idx=2000, n_lines=2, out=frames)
    x, y, rw, rh = smh.map_bounds(W, H)
    # frame 5: marker pixels in opposite corners -> bounding box = whole ROI (does not fit LDS)
    frames[5, y + 2:y + 5, x + 2:x + 90] = GREEN
    frames[5, y + rh - 6:y + rh - 3, x + rw - 100:x + rw - 3] = GREEN
    # frame 6: no synthetic lines; one tall, narrow, steep line -> more rows than the whole-row window holds, few columns
    frames[6], info6 = synth.make_frame(W, H, 2006, n_lines=0)
    for k in range(rh - 120):
        frames[6, y + 60 + k, x + 500 + k // 9:x + 503 + k // 9] = PURPLE
    infos[6] = info6
    # frame 7: one short line -> small bounding box (whole rows in LDS)
    frames[7], info7 = synth.make_frame(W, H, 2007, n_lines=0)
    frames[7, y + 300:y + 303, x + 200:x + 420] = GREEN
    infos[7] = info7
    # frame 8: closed map; frame 9: open, empty mask
    frames[8], infos[8] = synth.make_frame(W, H, 2008, map_open=False)
    frames[9], infos[9] = synth.make_frame(W, H, 2009, n_lines=0)
    per = [(i["scales_start_y"], i["anchors"]) for i in infos]
    anchors = smh.make_anchors(per)
    d = host.cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    s = torch.cuda.current_stream().cuda_stream
    fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
    a = bytes(fb.read_results(0, N))
    for _ in range(3):                                            # repeated: the concurrent mode kernels must not race on the records
        fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
        assert bytes(fb.read_results(0, N)) == a
    fast = smh.results_to_dicts(fb.read_results(0, N))
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=anchors, stream=s)
    recs = smh.results_to_dicts(fb.read_results(0, N))
    for i, (p, q) in enumerate(zip(fast, recs)):
        assert np.array_equal(p["lines"], q["lines"]) and p["rounds"] == q["rounds"] and p["ray_steps"] <= q["ray_steps"], i
        assert p["n_mask_px"] == q["n_mask_px"] and p["mpx"] == q["mpx"]
        if q["map_open"] and q["n_mask_px"]:
            assert q["rounds"] > 0, "frame %d has marker pixels but was never searched" % i
    assert recs[8]["map_open"] == 0 and recs[8]["n_lines"] == 0 and recs[9]["map_open"] == 1 and recs[9]["n_mask_px"] == 0 and recs[9]["rounds"] == 0
    for i in (0, 5, 6, 7, 64, 127):
        ref = o.process_frame(frames[i], stages=0xF, anchors=per[i][1], scales_start_y=per[i][0], want_images=True)
        r = recs[i]
        assert np.array_equal(r["lines"], ref["lines"]) and r["mpx"] == ref["mpx"], i
        assert (r["rounds"], r["ray_steps"], r["n_mask_px"]) == (ref["rounds"], ref["steps"], ref["n_mask_px"]), i
        m = fb.read_image(smh._lib.VIEW_LSD_INPUT, i)
        assert int((m == 255).sum()) == r["n_mask_px"] and np.array_equal(m, ref["lsd"])
        assert np.array_equal(fb.read_image(smh._lib.IMAGE_UI_MAP, i), ref["ui_map"])
        assert np.array_equal(fb.read_image(smh._lib.VIEW_OCR_INPUT, i), ref["ocr"])
        assert np.array_equal(fb.read_image(smh._lib.VIEW_FIND_SCALES_INPUT, i)[per[i][0]:], ref["scales"][per[i][0]:])
        for k, ln in enumerate(ref["lines"]):
            length, meters = o.marker_new(ln, ref["mpx"] if ref["mpx"] is not None else 0.0)
            assert abs(r["length_px"][k] - length) <= TOL and abs(r["meters"][k] - (meters if ref["mpx"] is not None else 0.0)) <= TOL
            assert abs(float(r["angle"][k]) - o.marker_angle(ln)) <= TOL
    assert len(recs[6]["lines"]) >= 1 and len(recs[7]["lines"]) >= 1
    rev = torch.flip(d, dims=[0]).contiguous()
    anchors_rev = smh.make_anchors(list(reversed(per)))
    fb.run(rev.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=anchors_rev, stream=s)
    recs_rev = smh.results_to_dicts(fb.read_results(0, N))
    for i in range(N):
        r, q = recs[i], recs_rev[N - 1 - i]
        assert np.array_equal(r["lines"], q["lines"]) and (r["rounds"], r["ray_steps"], r["n_mask_px"], r["mpx"]) == (q["rounds"], q["ray_steps"], q["n_mask_px"], q["mpx"]), i
    fb.close()


def test_config4_shard_of_1024_resident_frames(vision):
    """The per-GPU shard of BASELINE configs[4]: 1024 x 1920x1080 frames resident in HBM (8.5 GB of input: frame byte
    offsets pass 2^32 after frame 517).  64 distinct frames tiled 16 times on the device: the records must be periodic,
    a second run identical, two frames planted in the last 4 GB must be found where they were put, and the distinct
    frames equal the oracle."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, K, REP = 1920, 1080, 64, 16
    N = K * REP
    base, infos = synth.make_batch(W, H, K, first_idx=4096)
    d = torch.from_numpy(base).cuda().repeat(REP, 1, 1, 1)
    assert d.shape[0] == N and d.numel() > (1 << 32)
    x, y, rw, rh = smh.map_bounds(W, H)
    sp_a, info_a = synth.make_frame(W, H, 5001, n_lines=3)         # planted beyond the 4 GB boundary
    sp_b, info_b = synth.make_frame(W, H, 5002, map_open=False)
    ia, ib = 1000, 1023
    d[ia] = torch.from_numpy(sp_a).cuda()
    d[ib] = torch.from_numpy(sp_b).cuda()
    per = [(infos[i % K]["scales_start_y"], infos[i % K]["anchors"]) for i in range(N)]
    per[ia] = (info_a["scales_start_y"], info_a["anchors"])
    anchors = smh.make_anchors(per)
    fb = smh.FrameBatch(vision, W, H, N)
    s = torch.cuda.current_stream().cuda_stream
    fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
    a = bytes(fb.read_results(0, N))
    fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
    assert bytes(fb.read_results(0, N)) == a
    fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=anchors, stream=s)
    raw = fb.read_results(0, N)
    recs = smh.results_to_dicts(raw)
    rb = bytes(raw)
    import ctypes
    sz = ctypes.sizeof(smh._lib.FrameResult)
    for i in range(K, N):
        if i in (ia, ib):
            continue
        assert rb[i * sz:(i + 1) * sz] == rb[(i % K) * sz:(i % K + 1) * sz], "record %d differs from record %d of the same frame" % (i, i % K)
    threads = min(os.cpu_count() or 1, K)
    a9 = np.zeros((K, 3, 3), np.uint32)
    for i in range(K):
        for j, sc in enumerate(infos[i]["anchors"][:3]):
            a9[i, j] = sc
    ref = o.process_batch(base, threads, stages=0xF, anchors=a9, n_anchors=len(infos[0]["anchors"]), scales_start_y=infos[0]["scales_start_y"])
    for i in range(K):
        assert np.array_equal(recs[i]["lines"], _lines(ref[i])) and recs[i]["rounds"] == ref[i].rounds and recs[i]["ray_steps"] == ref[i].steps, i
        assert recs[i]["n_mask_px"] == ref[i].n_mask_px and recs[i]["mpx"] == (ref[i].mpx if ref[i].has_mpx else None)
    ra = o.process_frame(sp_a, stages=0xF, anchors=info_a["anchors"], scales_start_y=info_a["scales_start_y"], want_images=True)
    assert np.array_equal(recs[ia]["lines"], ra["lines"]) and (recs[ia]["rounds"], recs[ia]["ray_steps"]) == (ra["rounds"], ra["steps"]) and recs[ia]["mpx"] == ra["mpx"]
    assert np.array_equal(fb.read_image(smh._lib.VIEW_LSD_INPUT, ia), ra["lsd"]) and np.array_equal(fb.read_image(smh._lib.IMAGE_UI_MAP, ia), ra["ui_map"])
    assert np.array_equal(fb.read_image(smh._lib.VIEW_OCR_INPUT, ia), ra["ocr"])
    assert recs[ib]["map_open"] == 0 and recs[ib]["n_lines"] == 0
    # images of a frame behind the boundary equal those of its twin in front of it
    for which in (smh._lib.VIEW_LSD_INPUT, smh._lib.IMAGE_UI_MAP, smh._lib.VIEW_OCR_INPUT, smh._lib.VIEW_FIND_SCALES_INPUT):
        assert np.array_equal(fb.read_image(which, 3), fb.read_image(which, 3 + 15 * K))
    fb.close()


@pytest.mark.parametrize("seed,size,max_gap", [(11, (1920, 1080), 15), (12, (2560, 1440), 15), (13, (1280, 1024), 22), (14, (1024, 768), 3),
                                               (15, (1600, 1024), 49), (16, (1920, 1080), 9)])
def test_fuzz_lsd_slice(vision, seed, size, max_gap):
    """Bounded slice of tools/fuzz_lsd.py (36 random scenes per case, ~1000 rounds each): lines of all angles and widths,
    dashed lines with gaps around max_gap, blobs, rings, noise, shapes crossing the borders -- line lists and round counts
    in the culled run, plus the sample counts in the exact-statistics run, against the oracle."""
    import torch
    import squad_mortar_helper_amd as smh
    from fuzz_scenes import scene
    W, H = size
    n = 36
    rng = np.random.default_rng(seed)
    frames = np.stack([scene(rng, W, H, 100 * seed + i, max_gap) for i in range(n)])
    ref = o.process_batch(frames, min(os.cpu_count() or 1, n), stages=0x1, max_gap=max_gap)
    fb = smh.FrameBatch(vision, W, H, n)
    d = torch.from_numpy(frames).cuda()
    for exact in (0, smh.STAGE_EXACT_STATS):
        fb.run(d.data_ptr(), n, stages=smh.STAGE_MARKERS | exact, max_gap=max_gap, stream=torch.cuda.current_stream().cuda_stream)
        got = smh.results_to_dicts(fb.read_results(0, n))
        for i in range(n):
            assert got[i]["n_lines"] == ref[i].n_lines and np.array_equal(got[i]["lines"], _lines(ref[i])), (seed, i, bool(exact))
            assert got[i]["rounds"] == ref[i].rounds and got[i]["n_mask_px"] == ref[i].n_mask_px, (seed, i, bool(exact))
            if exact:
                assert got[i]["ray_steps"] == ref[i].steps, (seed, i)
    assert sum(r.rounds for r in ref) > 100
    fb.close()


@pytest.mark.parametrize("size,seed", [((1920, 1080), 31), ((2440, 1376), 32), ((2344, 1320), 33), ((800, 600), 34), ((2560, 1440), 35), ((3440, 1440), 36), ((5120, 1440), 37)])
def test_search_service_builds_its_tile_store_from_the_tile_major_mask(vision, size, seed):
    """Random scenes through a frame-granular pipeline with the tile store built from the pass's tile-major mask (the default where
    the launch writes it) and by the walk over the bit rows (SMHV_PIPE_WALK_BIT_ROWS): byte-identical records, equal to the oracle.
    Sizes: 1080p (16-bit tile table, bit rows 2 bits left of pixel 0); 2440 x 1376 and 2344 x 1320 (above 1080p, 56-row bands cost no
    band there: the compact index built from the occupancy bytes, bit rows 1 / 2 bits left of pixel 0); 800 x 600 (bit rows start on
    pixel 0); 1440p (58-row bands: the launch writes no tile-major mask, FrameAux::tiles sends the search to the walk); the ultrawide
    3440 x 1440 and 5120 x 1440 (ROIs 69 and 122 tile columns wide: the walk behind the compact index takes a tile row in chunks of
    64 columns -- round 5's looked at the first 64 only and lost the lines to the right of pixel 2048: found by this round's fuzz)."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import _lib
    from fuzz_scenes import scene
    W, H = size
    n, max_gap = 24, 15
    rng = np.random.default_rng(seed)
    frames = np.stack([scene(rng, W, H, 100 * seed + i, max_gap) for i in range(n)])
    ref = o.process_batch(frames, min(os.cpu_count() or 1, n), stages=0x1, max_gap=max_gap)
    d = torch.from_numpy(frames).cuda()
    recs = {}
    for name, flags in (("tiles", 0), ("walk", _lib.PIPE_WALK_BIT_ROWS)):
        pipe = smh.Pipeline(vision, W, H, n, 4, search="frame", flags=flags)
        # (markers + ui_map + ocr_preprocess: the fused pass, whose column masks hold 58 rows)
        slots = [pipe.submit(d.data_ptr(), n, stages=0x7, max_gap=max_gap) for _ in range(6)]
        pipe.wait()
        got = [bytes(pipe.slots[s_].read_results(0, n)) for s_ in sorted(set(slots))]
        assert all(g == got[0] for g in got), (size, name)
        recs[name] = got[0]
        dicts = smh.results_to_dicts(pipe.slots[slots[-1]].read_results(0, n))
        for i in range(n):
            assert dicts[i]["n_lines"] == ref[i].n_lines and np.array_equal(dicts[i]["lines"], _lines(ref[i])) and dicts[i]["rounds"] == ref[i].rounds, (size, name, i)
        # which builder ran is what the launch wrote: the tile-major mask where bands are whole tile rows
        _, occ, _, xoff = pipe.slots[slots[-1]].tile_mask(0)
        x, y, rw, rh = smh.map_bounds(W, H)
        assert (occ is not None) == (rh <= 900 or -(-rh // 56) == -(-rh // 58)), (size, rh)
        pipe.close()
    assert recs["tiles"] == recs["walk"], size
    assert sum(r.rounds for r in ref) > 50


def test_both_line_segment_kernels_agree_with_the_oracle(vision):
    """find_lines has three implementations: the task-based k_lsd_tile (sparse tile store of the mask, reorder buffer, waves
    claim 64-ray units), the workgroup-synchronous k_lsd (smhv_debug_lsd_classic) and the one-wave-per-frame sequential scan
    that the frame-granular search service of a pipeline of depth >= 3 runs (k_lsd_service).  Random scenes and synthetic
    frames through all of them, culled and exact; the tile-store searches also with the store capped so low that some
    (cap 48) or all (cap 4) frames overflow it and are searched on the mask in global memory.  (A pipeline whose stage set
    switches between culled and exact statistics also exercises the service's drain-and-relaunch with another sector table.)"""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    from fuzz_scenes import scene
    lib = smh._lib.load()

    def check(got, ref, n, tag, exact):
        for i in range(n):
            assert got[i]["n_lines"] == ref[i].n_lines and np.array_equal(got[i]["lines"], _lines(ref[i])), (i,) + tag
            assert got[i]["rounds"] == ref[i].rounds, (i,) + tag
            if exact:
                assert got[i]["ray_steps"] == ref[i].steps, (i,) + tag

    try:
        for seed, (W, H), max_gap in ((31, (1920, 1080), 15), (32, (2560, 1440), 15), (33, (1280, 1024), 30), (34, (1024, 768), 0), (35, (1600, 1024), 50)):
            n = 16
            rng = np.random.default_rng(seed)
            frames = np.stack([scene(rng, W, H, 100 * seed + i, max_gap) if i % 3 else synth.make_frame(W, H, 100 * seed + i, n_lines=3)[0] for i in range(n)])
            ref = o.process_batch(frames, min(os.cpu_count() or 1, n), stages=0x1, max_gap=max_gap)
            fb = smh.FrameBatch(vision, W, H, n)
            d = torch.from_numpy(frames).cuda()
            for classic, cap, threads in ((0, 0, 0), (1, 0, 0), (0, 48, 0), (0, 4, 0), (0, 0, 256)):
                lib.smhv_debug_lsd_classic(classic)
                lib.smhv_debug_lsd_tile_cap(cap)
                lib.smhv_debug_lsd_threads(threads)
                for exact in (0, smh.STAGE_EXACT_STATS):
                    for rep in range(2):
                        fb.run(d.data_ptr(), n, stages=smh.STAGE_MARKERS | exact, max_gap=max_gap, stream=torch.cuda.current_stream().cuda_stream)
                        check(smh.results_to_dicts(fb.read_results(0, n)), ref, n, (seed, classic, cap, threads, bool(exact)), exact)
            fb.close()
            lib.smhv_debug_lsd_classic(0)
            lib.smhv_debug_lsd_threads(0)
            # the one-wave-per-frame scan: through the search service of a depth-3 pipeline (the tile cap is read when the
            # pipeline is created)
            for cap in (0, 48, 4):
                lib.smhv_debug_lsd_tile_cap(cap)
                pipe = smh.Pipeline(vision, W, H, n, 3, search="frame")
                lib.smhv_debug_lsd_tile_cap(0)
                for exact in (0, smh.STAGE_EXACT_STATS, 0):
                    slots = [pipe.submit(d.data_ptr(), n, stages=smh.STAGE_MARKERS | exact, max_gap=max_gap) for _ in range(4)]
                    pipe.wait()
                    for s_ in sorted(set(slots)):
                        check(smh.results_to_dicts(pipe.slots[s_].read_results(0, n)), ref, n, (seed, "service", cap, s_, bool(exact)), exact)
                pipe.close()
    finally:
        lib.smhv_debug_lsd_classic(0)
        lib.smhv_debug_lsd_tile_cap(0)
        lib.smhv_debug_lsd_threads(0)


def test_sample_screenshots_through_the_batch_path(vision):
    """The reference's own sample screenshots (tests/golden fixtures), grouped by size and run as batches through
    smhv_batch_run -- i.e. through k_lsd_tile, which the per-call trait path of test_gpu_parity does not use: lines, round
    counts and (exact mode) sample counts against the goldens, and once more with the tile store capped at 64 tiles (the
    larger scenes then take the global-memory path)."""
    import torch
    import squad_mortar_helper_amd as smh
    import fixtures as fx
    lib = smh._lib.load()
    by_size = {}
    for stem in fx.OPEN_STEMS:
        frame, e, g = fx.load_fixture(stem)
        by_size.setdefault(frame.shape[:2], []).append((stem, frame, e, g))
    try:
        for (H, W), items in sorted(by_size.items()):
            frames = np.stack([it[1] for it in items])
            n = len(items)
            fb = smh.FrameBatch(vision, W, H, n)
            d = torch.from_numpy(frames).cuda()
            for cap in (0, 64):
                lib.smhv_debug_lsd_tile_cap(cap)
                for exact in (0, smh.STAGE_EXACT_STATS):
                    fb.run(d.data_ptr(), n, stages=smh.STAGE_MARKERS | exact, max_gap=15, stream=torch.cuda.current_stream().cuda_stream)
                    got = smh.results_to_dicts(fb.read_results(0, n))
                    for i, (stem, _, e, g) in enumerate(items):
                        assert got[i]["n_lines"] == len(g["lines"]) and np.array_equal(got[i]["lines"], g["lines"]), (stem, cap, bool(exact))
                        assert got[i]["rounds"] == e["rounds"], (stem, cap, bool(exact))
                        if exact:
                            assert got[i]["ray_steps"] == e["steps"], (stem, cap)
            fb.close()
    finally:
        lib.smhv_debug_lsd_tile_cap(0)


def test_sample_screenshots_through_the_search_service(vision):
    """The same screenshots through the pipelined line search, k_lsd_service (one wave per frame, help desk, helpers of other
    workgroups): grouped by size, at depth 3 and 12, culled and exact statistics, the tile store at its own cap and capped at
    64 tiles (the larger scenes then go through the global-memory scan inside the service).  More submissions than slots, so
    that slots are reused while waves are still helping; lines, rounds and (exact) sample counts against the goldens."""
    import torch
    import squad_mortar_helper_amd as smh
    import fixtures as fx
    lib = smh._lib.load()
    by_size = {}
    for stem in fx.OPEN_STEMS:
        frame, e, g = fx.load_fixture(stem)
        by_size.setdefault(frame.shape[:2], []).append((stem, frame, e, g))
    try:
        for (H, W), items in sorted(by_size.items()):
            frames = np.stack([it[1] for it in items])
            n = len(items)
            d = torch.from_numpy(frames).cuda()
            for depth in (3, 12):
                for cap in (0, 64):
                    lib.smhv_debug_lsd_tile_cap(cap)          # (read when the pipeline is created)
                    pipe = smh.Pipeline(vision, W, H, n, depth, search="frame")
                    lib.smhv_debug_lsd_tile_cap(0)
                    for exact in (0, smh.STAGE_EXACT_STATS, 0):
                        slots = [pipe.submit(d.data_ptr(), n, stages=smh.STAGE_MARKERS | exact, max_gap=15) for _ in range(depth + 2)]
                        pipe.wait()
                        for s_ in sorted(set(slots)):
                            got = smh.results_to_dicts(pipe.slots[s_].read_results(0, n))
                            for i, (stem, _, e, g) in enumerate(items):
                                assert got[i]["n_lines"] == len(g["lines"]) and np.array_equal(got[i]["lines"], g["lines"]), (stem, depth, cap, s_, bool(exact))
                                assert got[i]["rounds"] == e["rounds"], (stem, depth, cap, s_, bool(exact))
                                if exact:
                                    assert got[i]["ray_steps"] == e["steps"], (stem, depth, cap, s_)
                    pipe.close()
    finally:
        lib.smhv_debug_lsd_tile_cap(0)


def test_search_service_1440p_geometry_with_frames_beyond_its_tile_store(vision):
    """Above 1080p the service's waves keep their tile stores (272 tiles at most: smh_runtime.cpp, SMH_SVC_TILE_LIMIT) behind the
    compact index (LSD_MODE_TILEC: an occupancy mask and a running count per tile row instead of a 16-bit table), which is what
    lets four of them share a workgroup.  The reference's 1440p screenshots stay below that (<= 261 tiles); here they are run in a batch
    together with copies that carry a few hundred extra marker specks each -- more non-empty tiles than the store holds, so
    those frames take the service's scan on the mask in global memory while their neighbours take the tile store and ask
    for help.  Every record against the oracle, culled and exact."""
    import torch
    import squad_mortar_helper_amd as smh
    import fixtures as fx
    W, H = 2560, 1440
    base = [fx.load_fixture(stem)[0] for stem in fx.OPEN_STEMS]
    base = [f for f in base if f.shape[:2] == (H, W)]
    assert len(base) >= 4
    rng = np.random.default_rng(272)
    x, y, rw, rh = smh.map_bounds(W, H)
    frames = []
    for k, f in enumerate(base):
        frames.append(f)
        if k % 7 == 0:
            g = f.copy()
            m = 300 + 10 * k
            g[y + rng.integers(0, rh, m), x + rng.integers(0, rw, m)] = PURPLE
            frames.append(g)
    frames = np.stack(frames)
    n = len(frames)
    ref = o.process_batch(frames, min(os.cpu_count() or 1, n), stages=0x1, max_gap=15)
    # (32 x 8 px tiles of the dilated mask, as frame_setup_tile counts them)
    tiles = []
    for f in frames[:]:
        mask = o.mask_marker_lines(np.ascontiguousarray(f[y:y + rh, x:x + rw, 2::-1])) != 0
        mp = np.zeros(((rh + 7) // 8 * 8, (rw + 31) // 32 * 32), bool)
        mp[:rh, :rw] = mask
        tiles.append(int(mp.reshape(mp.shape[0] // 8, 8, mp.shape[1] // 32, 32).any(axis=(1, 3)).sum()))
    assert max(tiles) > 272 and min(tiles) <= 261, tiles
    d = torch.from_numpy(frames).cuda()
    pipe = smh.Pipeline(vision, W, H, n, 12, search="frame")
    geo = pipe.peek()
    assert geo["waves_per_workgroup"] == 4 and geo["service_workgroups"] < 256, geo   # (the compact tile index: four waves of 24 KB instead of three of 36.6; on five eighths of the CUs)
    for exact, subs in ((0, 14), (smh.STAGE_EXACT_STATS, 3), (0, 3)):
        slots = [pipe.submit(d.data_ptr(), n, stages=smh.STAGE_MARKERS | exact, max_gap=15) for _ in range(subs)]
        pipe.wait()
        for s_ in sorted(set(slots)):
            got = smh.results_to_dicts(pipe.slots[s_].read_results(0, n))
            for i in range(n):
                assert got[i]["n_lines"] == ref[i].n_lines and np.array_equal(got[i]["lines"], _lines(ref[i])), (i, s_, bool(exact), tiles[i])
                assert got[i]["rounds"] == ref[i].rounds, (i, s_, bool(exact), tiles[i])
                if exact:
                    assert got[i]["ray_steps"] == ref[i].steps, (i, s_, tiles[i])
    pipe.close()


@pytest.mark.parametrize("size", [(1920, 1080), (2560, 1440), (1024, 768), (1280, 1024), (1600, 1024), (3840, 2160)])
def test_tile_major_mask_of_the_streaming_passes(vision, size):
    """What the streaming passes leave for the line search (include/smh_vision_hip.h, smhv_batch_tile_mask): the occupancy bytes name
    exactly the non-empty 32 x 8 tiles of the bit-packed rows, and every such tile holds those rows' words -- for the fused pass
    (all stages) and the plain one (markers only), with few frames (bands of 8 rows) and with enough of them for full-height bands;
    and the bit rows themselves are the oracle's dilated mask."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = size
    x, y, rw, rh = smh.map_bounds(W, H)
    for n in (1, 40):
        host = torch.empty((n, H, W, 4), dtype=torch.uint8, pin_memory=True)
        frames = host.numpy()
        _, infos = synth.make_batch(W, H, n, first_idx=7000 + n, n_lines=3, out=frames)
        # marker pixels on the ROI's first and last columns and rows, and across a tile-row / band boundary
        frames[0, y:y + 3, x:x + 70] = GREEN
        frames[0, y + rh - 2:y + rh, x + rw - 40:x + rw] = GREEN
        frames[0, y + 50:y + 62, x:x + 2] = PURPLE
        frames[0, y + 100:y + 130, x + rw - 1:x + rw] = PURPLE
        if n > 1:
            frames[n - 1], infos[n - 1] = synth.make_frame(W, H, 7999, n_lines=0)       # an open frame without a marker pixel
        anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
        d = host.cuda()
        fb = smh.FrameBatch(vision, W, H, n)
        s = torch.cuda.current_stream().cuda_stream
        want_mask = {f: o.process_frame(frames[f], stages=0x1, want_images=True)["lsd"] for f in sorted({0, n // 2, n - 1})}
        for stages, anc in ((smh.STAGE_ALL, anchors), (smh.STAGE_MARKERS, None)):
            fb.run(d.data_ptr(), n, stages=stages, anchors=anc, stream=s)
            for f in sorted(want_mask):
                tiled, occ, bits, xoff = fb.tile_mask(f)
                wcols = bits.shape[1]
                trows = (rh + 7) // 8
                # whole tile rows per band where that is free or pays (band_rows_for, smh_stream.hip): few frames always (bands of 8 / 16 / 32
                # rows), full-height bands when the ROI is at most 900 rows tall or 56-row bands need no more bands than the kernel's own
                cap = 58 if stages == smh.STAGE_ALL else 62
                expect_tiles = n == 1 or rh <= 900 or -(-rh // 56) == -(-rh // cap)
                assert (tiled is not None) == expect_tiles, (size, n, stages, f)
                assert bits.shape == (rh, wcols)
                # the bit rows are the oracle's mask (bit x + xoff of row y = pixel x)
                px = np.zeros((rh, wcols * 32), np.uint8)
                px[:, xoff:xoff + rw] = want_mask[f] != 0
                want_bits = np.packbits(px.reshape(rh, wcols, 32), axis=2, bitorder="little").view(np.uint32).reshape(rh, wcols)
                assert np.array_equal(bits, want_bits), (size, n, stages, f)
                if tiled is None:
                    continue
                assert tiled.shape == (trows, wcols, 8)
                # occupancy == non-empty tiles of those rows; the tiles' words == the rows' words
                padded = np.zeros((trows * 8, wcols), np.uint32)
                padded[:rh] = bits
                by_tile = padded.reshape(trows, 8, wcols).transpose(0, 2, 1)            # [ty, wx, r]
                nonempty = by_tile.any(axis=2)
                got_occ = np.unpackbits(occ, axis=1, bitorder="little")[:, :wcols].astype(bool)
                assert np.array_equal(got_occ, nonempty), (size, n, stages, f, int(nonempty.sum()), int(got_occ.sum()))
                tail = rh - (trows - 1) * 8                                               # rows of the last tile row inside the image
                a, b_ = tiled[nonempty], by_tile[nonempty]
                last = np.repeat(np.arange(trows)[:, None], wcols, axis=1)[nonempty] == trows - 1
                assert np.array_equal(a[~last], b_[~last]) and np.array_equal(a[last][:, :tail], b_[last][:, :tail]), (size, n, stages, f)
            if n > 1 and fb.tile_mask(n - 1)[1] is not None:
                assert not fb.tile_mask(n - 1)[1].any()                                   # no marker pixel: every occupancy byte written, all zero
        fb.close()


def test_8k_frames_take_the_large_tile_index(vision):
    """7680x4320: the 16-bit tile index of the map ROI alone is 106 KB, so k_lsd_tile runs one workgroup per CU with 140 KB of
    dynamic LDS instead of two with 60 KB each.  Two synthetic frames, both line-search kernels, against the oracle."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, n = 7680, 4320, 2
    frames = np.stack([synth.make_frame(W, H, 900 + i, n_lines=3)[0] for i in range(n)])
    ref = o.process_batch(frames, n, stages=0x1, max_gap=15)
    lib = smh._lib.load()
    fb = smh.FrameBatch(vision, W, H, n)
    d = torch.from_numpy(frames).cuda()
    try:
        for classic in (0, 1):
            lib.smhv_debug_lsd_classic(classic)
            fb.run(d.data_ptr(), n, stages=smh.STAGE_MARKERS, max_gap=15, stream=torch.cuda.current_stream().cuda_stream)
            got = smh.results_to_dicts(fb.read_results(0, n))
            for i in range(n):
                assert got[i]["n_lines"] == ref[i].n_lines and np.array_equal(got[i]["lines"], _lines(ref[i])), (i, classic)
                assert got[i]["rounds"] == ref[i].rounds, (i, classic)
        assert sum(r.n_lines for r in ref) >= 4
    finally:
        lib.smhv_debug_lsd_classic(0)
        fb.close()


def test_fuzz_stream_slice(vision):
    """Bounded slice of tools/fuzz_stream.py: pixels drawn around every decision threshold of the streaming stages."""
    import squad_mortar_helper_amd as smh
    from fuzz_scenes import random_frame
    rng = np.random.default_rng(5)
    for it, (W, H) in enumerate([(1920, 1080), (2560, 1440), (1024, 768), (1366, 768), (1680, 1050), (1280, 1024)]):
        frame = random_frame(rng, W, H)
        bx, by, bw, bh = smh.button_bounds(W, H)
        frac = [0.64, 0.66, 1.0, 0.65, 0.7, 0.60][it]
        red = rng.random((bh, bw)) < frac
        frame[by:by + bh, bx:bx + bw, :3] = np.where(red[..., None], np.array([49, 67, 217]) + rng.integers(-25, 26, (bh, bw, 3)), 0).astype(np.uint8)
        start_y = int(rng.integers(0, 50))
        ref = o.process_frame(frame, stages=0x0E, scales_start_y=start_y, anchors=[(100, 10, start_y)], want_images=True)
        vision.load_frame(frame)
        crop = vision.crop_to_map(True)
        assert (crop is not None) == bool(ref["map_open"]) and vision.red_pixels() == o.button_red_pixels(frame)
        if crop is None:
            continue
        vision.isolate_map_markers(); vision.mask_marker_lines()
        x, y, rw, rh = smh.map_bounds(W, H)
        mask_ref = o.mask_marker_lines(np.ascontiguousarray(frame[y:y + rh, x:x + rw, 2::-1]))
        assert np.array_equal(crop[0], ref["ui_map"]) and np.array_equal(vision.lsd_image(), mask_ref)
        assert np.array_equal(vision.ocr_preprocess(), ref["ocr"])
        assert np.array_equal(vision.find_scales_preprocess(start_y)[start_y:], ref["scales"][start_y:])
        colour = vision.crop_to_map(False)
        assert np.array_equal(colour[0][..., :3], frame[y:y + rh, x:x + rw, 2::-1])


def test_fused_streaming_pass_on_threshold_frames(vision):
    """The batched pipeline computes ui_map, the marker mask and the two bottom-right-quadrant images in ONE pass over
    the ROI (k_map_brq_pass).  Frames whose pixels sit around every decision threshold (tools/fuzz_stream.py), at sizes
    whose quadrant starts at different phases of the 4-pixel quad grid and of the 58-row bands; every image against the
    oracle, for all stage subsets that select the fused kernel, grey and colour ui_map."""
    import torch
    import squad_mortar_helper_amd as smh
    from fuzz_scenes import random_frame
    rng = np.random.default_rng(17)
    for (W, H) in [(1920, 1080), (2560, 1440), (1024, 768), (1366, 768), (1680, 1050), (1280, 1024), (1600, 1024), (3840, 2160)]:
        N = 3
        frames = np.stack([random_frame(rng, W, H) for _ in range(N)])
        bx, by, bw, bh = smh.button_bounds(W, H)
        frames[:, by:by + bh, bx:bx + bw, :3] = (49, 67, 217)                       # map open
        start = [int(rng.integers(0, 60)) for _ in range(N)]
        per = [(start[i], [(100, 10, start[i])]) for i in range(N)]
        per[1] = (per[1][0], [])                                                     # no labels: scales image must stay untouched
        d = torch.from_numpy(frames).cuda()
        fb = smh.FrameBatch(vision, W, H, N)
        s = torch.cuda.current_stream().cuda_stream
        x, y, rw, rh = smh.map_bounds(W, H)
        refs = {}                                                                    # the oracle's images of (frame, grey): once
        for stages, gray in ((0xF, True), (0xF, False), (0x7, True), (0xD, True), (0xE, True), (0x6, False)):
            fb.run(d.data_ptr(), N, stages=stages, grayscale=gray, anchors=smh.make_anchors(per), stream=s)
            torch.cuda.synchronize()
            for i in range(N):
                if (i, gray) not in refs:
                    refs[(i, gray)] = o.process_frame(frames[i], grayscale=gray, stages=0xF, anchors=per[i][1] or None, scales_start_y=per[i][0], want_images=True)
                ref = refs[(i, gray)]
                if stages & 0x2:
                    assert np.array_equal(fb.read_image(smh._lib.IMAGE_UI_MAP, i), ref["ui_map"]), (W, H, stages, i)
                if stages & 0x1:
                    assert np.array_equal(fb.read_image(smh._lib.VIEW_LSD_INPUT, i), ref["lsd"]), (W, H, stages, i)
                if stages & 0x4:
                    assert np.array_equal(fb.read_image(smh._lib.VIEW_OCR_INPUT, i), ref["ocr"]), (W, H, stages, i)
                if (stages & 0x8) and per[i][1]:
                    assert np.array_equal(fb.read_image(smh._lib.VIEW_FIND_SCALES_INPUT, i)[per[i][0]:], ref["scales"][per[i][0]:]), (W, H, stages, i)
        fb.close()


@pytest.mark.parametrize("size,n,forced", [((1920, 1080), 36, 0), ((1920, 1080), 36, 56), ((2560, 1440), 28, 0), ((2560, 1440), 28, 48), ((1280, 1024), 40, 0), ((1280, 1024), 40, 40)])
def test_fused_pass_full_height_bands_on_threshold_frames(vision, size, n, forced):
    """The same threshold frames with enough of them for FULL-HEIGHT bands (the test above runs three frames: bands of 8 rows): 24 rows
    and the tile-major mask at 1080p and 1280 x 1024, 58 rows without it at 1440p (smhv_debug_band_rows says which), and the heights
    the rule does not take by itself through smhv_debug_map_band_rows (56: seven tile rows per band) -- every image of every frame
    against the oracle, and where the launch wrote the tile-major mask, its occupancy bytes and tiles against the bit rows."""
    from squad_mortar_helper_amd import _lib
    _lib.check(_lib.load().smhv_debug_map_band_rows(forced))
    try:
        _full_height_bands(vision, size[0], size[1], n, forced)
    finally:
        _lib.check(_lib.load().smhv_debug_map_band_rows(0))


def _full_height_bands(vision, W, H, n, forced):
    import ctypes as C
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import _lib
    from fuzz_scenes import random_frame
    size = (W, H)
    rows, bands, tiles = C.c_uint32(), C.c_uint32(), C.c_int()
    _lib.check(_lib.load().smhv_debug_band_rows(W, H, n, 1, C.byref(rows), C.byref(bands), C.byref(tiles)))
    assert rows.value == (forced or (58 if H == 1440 else 24)) and bool(tiles.value) == (rows.value % 8 == 0)
    rng = np.random.default_rng(W + n)
    frames = np.stack([random_frame(rng, W, H) for _ in range(n)])
    bx, by, bw, bh = smh.button_bounds(W, H)
    frames[:, by:by + bh, bx:bx + bw, :3] = (49, 67, 217)                       # map open
    start = [int(rng.integers(0, 60)) for _ in range(n)]
    per = [(start[i], [(100, 10, start[i])]) for i in range(n)]
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, n)
    fb.run(d.data_ptr(), n, stages=0xF, grayscale=True, anchors=smh.make_anchors(per), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    x, y, rw, rh = smh.map_bounds(W, H)
    for i in range(n):
        ref = o.process_frame(frames[i], grayscale=True, stages=0xE, anchors=per[i][1], scales_start_y=per[i][0], want_images=True)
        mask_ref = o.mask_marker_lines(np.ascontiguousarray(frames[i][y:y + rh, x:x + rw, 2::-1]))
        assert np.array_equal(fb.read_image(smh._lib.IMAGE_UI_MAP, i), ref["ui_map"]), (size, i)
        assert np.array_equal(fb.read_image(smh._lib.VIEW_LSD_INPUT, i), mask_ref), (size, i)
        assert np.array_equal(fb.read_image(smh._lib.VIEW_OCR_INPUT, i), ref["ocr"]), (size, i)
        assert np.array_equal(fb.read_image(smh._lib.VIEW_FIND_SCALES_INPUT, i)[per[i][0]:], ref["scales"][per[i][0]:]), (size, i)
        tiled, occ, bits, xoff = fb.tile_mask(i)
        assert (tiled is not None) == bool(tiles.value), (size, i)
        wcols = bits.shape[1]
        px = np.zeros((rh, wcols * 32), np.uint8)
        px[:, xoff:xoff + rw] = mask_ref != 0
        assert np.array_equal(bits, np.packbits(px.reshape(rh, wcols, 32), axis=2, bitorder="little").view(np.uint32).reshape(rh, wcols)), (size, i)
        if tiled is not None:
            trows = (rh + 7) // 8
            padded = np.zeros((trows * 8, wcols), np.uint32)
            padded[:rh] = bits
            by_tile = padded.reshape(trows, 8, wcols).transpose(0, 2, 1)
            nonempty = by_tile.any(axis=2)
            assert np.array_equal(np.unpackbits(occ, axis=1, bitorder="little")[:, :wcols].astype(bool), nonempty), (size, i)
            tail = rh - (trows - 1) * 8
            a, b_ = tiled[nonempty], by_tile[nonempty]
            last = np.repeat(np.arange(trows)[:, None], wcols, axis=1)[nonempty] == trows - 1
            assert np.array_equal(a[~last], b_[~last]) and np.array_equal(a[last][:, :tail], b_[last][:, :tail]), (size, i)
    fb.close()


def test_pipeline_without_host_atomics_keeps_the_batch_granular_search(vision):
    """The search service's life cycle needs device-side 64-bit atomics on mapped host memory; smhv_pipeline_create probes for
    them.  With the probe's answer forced to "no" (smhv_debug_no_host_atomics): the default search of a deep pipeline is the
    batch-granular one (no service: search_stats() is None), an explicit "frame" is E_INVALID, and the records are those of a
    plain run."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    lib = smh._lib.load()
    W, H, N = 1920, 1080, 12
    fr, inf = synth.make_batch(W, H, N, first_idx=4100, n_lines=3)
    d = torch.from_numpy(fr).cuda()
    a = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in inf])
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=a, stream=torch.cuda.current_stream().cuda_stream)
    want = bytes(fb.read_results(0, N))
    fb.close()
    lib.smhv_debug_no_host_atomics(1)
    try:
        with pytest.raises(smh.VisionError) as ei:
            smh.Pipeline(vision, W, H, N, 8, search="frame")
        assert ei.value.code == smh._lib.E_INVALID and "atomics" in str(ei.value)
        pipe = smh.Pipeline(vision, W, H, N, 8)
        slots = [pipe.submit(d.data_ptr(), N, anchors=a) for _ in range(10)]
        pipe.wait()
        assert pipe.search_stats() is None
        for s_ in sorted(set(slots)):
            assert bytes(pipe.slots[s_].read_results(0, N)) == want, s_
        pipe.close()
    finally:
        lib.smhv_debug_no_host_atomics(0)
    pipe = smh.Pipeline(vision, W, H, N, 8, search="frame")             # (the probe itself passes on this machine)
    s_ = pipe.submit(d.data_ptr(), N, anchors=a)
    pipe.wait()
    assert bytes(pipe.slots[s_].read_results(0, N)) == want
    pipe.close()


def test_foreign_kernel_runs_beside_the_search_service(vision):
    """A search workgroup holds most of its CU's LDS for as long as the pipeline is busy, so a kernel of another owner with a
    large footprint -- RCCL's kernels on gfx950 take 19.7-21.2 KB of LDS and 261-280 VGPRs per workgroup -- fits no CU that
    has one.  A pipeline created with room_for_others (what smhv_node and bench.py --gpus N > 1 use) leaves an eighth of the
    CUs without: a probe kernel with that footprint (smhv_debug_side_kernel, 8 workgroups), launched once per pass on a
    stream of its own beside a SATURATED frame-granular depth-12 pipeline at 1080p, gets onto the chip within a fraction of a
    millisecond (median; 99th percentile about one; never the seconds a pipeline without room shows) and costs the pipeline a few percent (0-8 % measured; 60-75 % without room); the records stay those
    of a plain run.  (A probe of 32 workgroups -- one for every CU left free -- also gets there within 1-2 ms but costs the
    pipeline ~10 %: bench.py's co_residency leg reports both.)"""
    import time
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    lib = smh._lib.load()
    W, H, N, depth, passes = 1920, 1080, 256, 12, 240
    fr, inf = synth.make_batch(W, H, 64, first_idx=700, n_lines=2)
    fr = np.concatenate([fr] * 4)
    inf = [inf[i % 64] for i in range(N)]
    d = torch.from_numpy(fr).cuda()
    a = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in inf])
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=a, stream=torch.cuda.current_stream().cuda_stream)
    want = bytes(fb.read_results(0, N))
    fb.close()
    pipe = smh.Pipeline(vision, W, H, N, depth, search="frame", room_for_others=1)
    geo = pipe.peek()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert geo["service_workgroups"] == cus - max(cus // 8, 1), geo
    side = torch.cuda.Stream()

    def run(with_probe):
        for _ in range(2 * depth):
            pipe.submit(d.data_ptr(), N, anchors=a)
        pipe.wait()
        evs = []
        t0 = time.perf_counter()
        for _ in range(passes):
            pipe.submit(d.data_ptr(), N, anchors=a)
            if with_probe:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                smh._lib.check(lib.smhv_debug_side_kernel(vision._ctx, 8, side.cuda_stream))
                e1.record(side)
                evs.append((e0, e1))
        pipe.wait()
        dt = time.perf_counter() - t0
        side.synchronize()
        return N * passes / dt, np.array([x.elapsed_time(y) for x, y in evs]) if evs else None

    best = None
    for attempt in range(3):                                     # (a shared box: the best of three, each measured between two runs without probes)
        r0, _ = run(False)
        r1, lat = run(True)
        r2, _ = run(False)
        cost = 1.0 - r1 / (0.5 * (r0 + r2))
        if best is None or cost < best[0]:
            best = (cost, lat)
        assert lat.max() < 20.0, (attempt, float(lat.max()))     # never the seconds a pipeline without room shows
        if cost < 0.05 and np.percentile(lat, 99) < 2.0:
            break                                                    # (usually the first attempt: 0-5 %; a shared box now and then shows 8 %)
    cost, lat = best
    # (measured over the round's boxes: median 0.2-0.3 ms, 99th percentile 0.5-1.1 ms, maximum 0.8-1.6 ms)
    assert np.median(lat) < 0.6 and np.percentile(lat, 99) < 2.0, (float(np.median(lat)), float(np.percentile(lat, 99)), float(lat.max()))
    assert cost < 0.12, cost                                     # (measured 0.1-8 % over the round's boxes; without room: 60-75 %)
    for s_ in range(depth):
        assert bytes(pipe.slots[s_].read_results(0, N)) == want, s_
    pipe.close()


def test_pipeline_object_gives_the_records_of_plain_runs(vision):
    """smhv_pipeline_*: depth 1..4 with the batch-granular search, depth 3 / 4 / 8 with the frame-granular search service
    (with and without the workgroup help desk, with one and three streaming streams), with idle streams the host created
    beforehand; every submission's records equal those of a plain smhv_batch_run of the same frames, the slot hand-back is
    round robin, submit never loses a batch when more than `depth` are pushed back to back."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N = 1920, 1080, 24
    sets = []
    for k in range(3):
        fr, inf = synth.make_batch(W, H, N, first_idx=9000 + 100 * k, n_lines=2 + k)
        sets.append((torch.from_numpy(fr).cuda(), smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in inf])))
    fb = smh.FrameBatch(vision, W, H, N)
    want = []
    for d, a in sets:
        fb.run(d.data_ptr(), N, anchors=a, stream=torch.cuda.current_stream().cuda_stream)
        want.append(bytes(fb.read_results(0, N)))
    fb.close()
    assert len(set(want)) == 3
    idle = [torch.cuda.Stream() for _ in range(3)]                  # streams created before the pipeline: must not matter
    for depth, cus in ((1, {}), (2, {}), (3, {}), (4, {}), (3, dict(search="frame")), (4, dict(search="frame", flags=smh._lib.PIPE_NO_TEAM_HELP)),
                       (8, {}), (8, dict(streams=3, flags=smh._lib.PIPE_NO_PROLOGUE)), (5, dict(search="frame", streams=1, idle_close_us=2000))):
        pipe = smh.Pipeline(vision, W, H, N, depth, **cus)
        order = [0, 1, 2, 2, 1, 0, 1, 1, 0, 2, 0, 1]
        slots = []
        for j, k in enumerate(order):
            slot = pipe.submit(sets[k][0].data_ptr(), N, anchors=sets[k][1])
            assert slot == j % depth
            slots.append((slot, k, j))
            if j >= depth - 1:                                          # the oldest submission still in flight
                s_old, k_old, j_old = slots[j - (depth - 1)]
                pipe.wait(s_old)
                assert bytes(pipe.slots[s_old].read_results(0, N)) == want[k_old], (depth, cus, j_old)
        pipe.wait()
        # a producer stream: the frames are written on another stream right before the submission
        prod = torch.cuda.Stream()
        with torch.cuda.stream(prod):
            tmp = sets[2][0].clone()
        slot = pipe.submit(tmp.data_ptr(), N, anchors=sets[2][1], after_stream=prod.cuda_stream)
        pipe.wait(slot)
        assert bytes(pipe.slots[slot].read_results(0, N)) == want[2]
        pipe.close()
    del idle


def test_adaptive_pipeline_measures_both_searches_and_never_changes_a_record(vision):
    """SMHV_SEARCH_AUTO at depth 8: the pipeline runs its submissions through the frame-granular search service first, then
    through the batch-granular search, times a window of each and keeps the faster; submissions of both kinds are in flight
    together around every switch.  260 submissions of three alternating frame sets: every one's records equal those of a
    plain smhv_batch_run, whatever search it went to; afterwards the pipeline has settled and has a rate for both.  A change
    of the submissions' shape (stages) starts the measurement again."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, N, depth = 1920, 1080, 16, 8
    sets = []
    for k in range(3):
        fr, inf = synth.make_batch(W, H, N, first_idx=9500 + 100 * k, n_lines=1 + 2 * k)
        sets.append((torch.from_numpy(fr).cuda(), smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in inf])))
    fb = smh.FrameBatch(vision, W, H, N)
    want, want3 = [], []
    for d, a in sets:
        fb.run(d.data_ptr(), N, anchors=a, stream=torch.cuda.current_stream().cuda_stream)
        want.append(bytes(fb.read_results(0, N)))
        fb.run(d.data_ptr(), N, stages=0x3, stream=torch.cuda.current_stream().cuda_stream)
        want3.append(bytes(fb.read_results(0, N)))
    fb.close()
    pipe = smh.Pipeline(vision, W, H, N, depth)
    st0 = pipe.search_stats()
    assert st0["adaptive"] and not st0["settled"] and st0["mode"] == "frame-granular"
    inflight = []

    def push(k, stages, expect):
        slot = pipe.submit(sets[k][0].data_ptr(), N, stages=stages, anchors=sets[k][1] if stages & 8 else None)
        inflight.append((slot, expect[k]))
        if len(inflight) == depth:                                  # the oldest submission still in flight: check it before its slot is reused
            s_old, w_old = inflight.pop(0)
            pipe.wait(s_old)
            assert bytes(pipe.slots[s_old].read_results(0, N)) == w_old

    for j in range(26 * depth + 52):
        push((j * 7 + j // 5) % 3, 0xF, want)
    st = pipe.search_stats()
    assert st["settled"] and st["launches"] >= 1 and min(st["measured_frames_per_s"].values()) > 0, st
    for j in range(3 * depth):                                      # another shape: the measurement starts again
        push(j % 3, 0x3, want3)
    assert not pipe.search_stats()["settled"]
    while inflight:
        s_old, w_old = inflight.pop(0)
        pipe.wait(s_old)
        assert bytes(pipe.slots[s_old].read_results(0, N)) == w_old
    pipe.close()
    # a pinned schedule does not measure anything
    for mode in ("batch", "frame"):
        pipe = smh.Pipeline(vision, W, H, N, depth, search=mode)
        for j in range(2 * depth):
            pipe.submit(sets[0][0].data_ptr(), N, anchors=sets[0][1])
        pipe.wait()
        st = pipe.search_stats()
        assert (st is None) if mode == "batch" else (not st["adaptive"] and st["mode"] == "frame-granular")
        pipe.close()


def _oracle_batch(frames, infos, stages=0xF):
    k = len(frames)
    a9 = np.zeros((k, 3, 3), np.uint32)
    for i in range(k):
        for j, sc in enumerate(infos[i]["anchors"][:3]):
            a9[i, j] = sc
    return o.process_batch(frames, min(os.cpu_count() or 1, k), stages=stages, anchors=a9, n_anchors=len(infos[0]["anchors"]),
                           scales_start_y=infos[0]["scales_start_y"])


@pytest.mark.parametrize("W,H,N", [(1920, 1080, 256), (2560, 1440, 128)])
def test_headline_configuration_pipelined_depth4(vision, W, H, N):
    """The EXACT configuration bench.py's `value` is measured on (BASELINE configs[2] / configs[3]): N resident frames,
    smhv_pipeline with twelve batches in flight, the submissions of its first phase -- the frame-granular search service, one
    wave per frame, help desk -- back to back -- and the configuration it was measured on until round 3: four batches in flight, batch-granular
    search (k_lsd_tile, 512-thread workgroups, occupancy policy), twelve submissions.  Every slot's N records must equal, byte
    for byte, a plain smhv_batch_run of the same frames (k_lsd_tile with 1024-thread workgroups, nothing beside it), the
    depth-1 pipeline (k_lsd with helper workgroups at 1080p) must give them too, and eight frames spread over the batch are
    checked against the oracle field by field."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    frames, infos = synth.make_batch(W, H, N, first_idx=0)            # the bench's own frames (rank 0)
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
    want_raw = fb.read_results(0, N)
    want = bytes(want_raw)
    recs = smh.results_to_dicts(want_raw)
    fb.close()
    for depth, subs in ((12, 30), (4, 12)):                          # bench.py's default; round 3's
        pipe = smh.Pipeline(vision, W, H, N, depth)
        for j in range(subs):
            assert pipe.submit(d.data_ptr(), N, anchors=anchors) == j % depth
        pipe.wait()
        for s_ in range(depth):
            assert bytes(pipe.slots[s_].read_results(0, N)) == want, "slot %d of the depth-%d pipeline differs from the plain run" % (s_, depth)
        st = pipe.search_stats()
        assert (st is not None) == (depth >= 6)                     # which schedule the library picked
        if st:
            assert st["frames"] == subs * N and st["submissions"] == subs
        pipe.close()
    pipe1 = smh.Pipeline(vision, W, H, N, 1)
    for _ in range(2):
        pipe1.submit(d.data_ptr(), N, anchors=anchors)
    pipe1.wait()
    assert bytes(pipe1.slots[0].read_results(0, N)) == want, "the depth-1 pipeline differs from the plain run"
    pipe1.close()
    pick = [0, 1, N // 5, N // 3, N // 2, N // 2 + 7, N - 2, N - 1]
    ref = _oracle_batch(np.ascontiguousarray(frames[pick]), [infos[i] for i in pick])
    for r, i in zip(ref, pick):
        g = recs[i]
        assert g["map_open"] == 1 and g["status"] == 0
        assert np.array_equal(g["lines"], _lines(r)) and g["rounds"] == r.rounds and g["n_mask_px"] == r.n_mask_px, i
        assert g["mpx"] == (r.mpx if r.has_mpx else None), i
        for k in range(r.n_lines):                                      # derived f64 / f32 outputs: the north star's 1e-4
            x0, y0, x1, y1 = [float(v) for v in _lines(r)[k]]
            assert abs(g["length_px"][k] - np.hypot(x0 - x1, y0 - y1)) <= TOL and abs(g["angle"][k] - np.arctan2(np.float32(y0 - y1), np.float32(x0 - x1))) <= TOL


def test_pipeline_occupancy_policy_does_not_change_any_output(vision):
    """Whatever schedule a pipeline runs -- batch-granular search with the occupancy policy of depth >= 3 (LDS reservation and
    capped, grid-stride grid of the streaming pass; tile limit of the line search) on, off or adaptive, late helpers always /
    never, the frame-granular search service with and without its help desk, wave priority and prologue stream -- no output
    byte may change: records AND images (ui_map, mask, ocr, scales) equal a plain smhv_batch_run's, on synthetic frames and on
    frames with MORE marker tiles than the pipelines' tile limit (dense scenes: those are searched on the mask in global
    memory)."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    from fuzz_scenes import scene
    L = smh._lib
    W, H, N = 1920, 1080, 40
    frames, infos = synth.make_batch(W, H, N, first_idx=2100, n_lines=3)
    rng = np.random.default_rng(77)
    for i in range(0, N, 5):                                           # dense random scenes: hundreds of mask tiles
        frames[i] = scene(rng, W, H, 9000 + i, 15)
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
    want = bytes(fb.read_results(0, N))
    which = (L.IMAGE_UI_MAP, L.VIEW_LSD_INPUT, L.VIEW_OCR_INPUT, L.VIEW_FIND_SCALES_INPUT)
    pick = (0, 5, 17, N - 1)
    imgs = {(w, f): fb.read_image(w, f).copy() for w in which for f in pick}
    fb.close()
    for depth, opts in ((4, {}), (4, dict(occupancy_policy=1)), (4, dict(occupancy_policy=2)), (4, dict(occupancy_policy=1, late_helpers=1)), (3, dict(late_helpers=2)),
                        (8, {}), (4, dict(search="frame")), (8, dict(flags=L.PIPE_NO_TEAM_HELP)), (8, dict(flags=L.PIPE_NO_STREAM_PRIORITY | L.PIPE_NO_PROLOGUE, streams=3)),
                        (16, dict(idle_close_us=500))):
        pipe = smh.Pipeline(vision, W, H, N, depth, **opts)
        for j in range(depth + 2):
            slot = pipe.submit(d.data_ptr(), N, anchors=anchors)
        pipe.wait()
        for s_ in range(depth):
            assert bytes(pipe.slots[s_].read_results(0, N)) == want, (depth, opts, s_)
        for (w, f), img in imgs.items():
            assert np.array_equal(pipe.slots[slot].read_image(w, f), img), (depth, opts, w, f)
        pipe.close()


def test_line_search_watchdog_becomes_an_error(vision):
    """A frame the line search gives up (its waves made no progress for the spin budget) must not pass as "no marker
    lines": the record carries status = SMHV_FRAME_LSD_STUCK (n_lines 0, rounds 0xFFFFFFFF), and read_results /
    pipeline_wait return SMHV_E_STATE naming the frame -- once; the reference logs and drops such a frame
    (src/vision/mod.rs:272-276).  Forced here by a spin budget of one poll; with the default budget the same batch is clean."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    lib = smh._lib.load()
    W, H, N = 1920, 1080, 32
    frames, infos = synth.make_batch(W, H, N, first_idx=300, n_lines=3)
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    st = torch.cuda.current_stream().cuda_stream
    fb.run(d.data_ptr(), N, anchors=anchors, stream=st)
    clean = fb.read_results(0, N)
    clean_b = bytes(clean)
    assert all(r.status == 0 for r in clean)
    lib.smhv_debug_lsd_spin_limit(1)
    try:
        fb.run(d.data_ptr(), N, anchors=anchors, stream=st)
        with pytest.raises(smh.VisionError) as ei:
            fb.read_results(0, N)
        assert ei.value.code == smh._lib.E_STATE and "SMHV_FRAME_LSD_STUCK" in str(ei.value) and "frame" in str(ei.value)
        recs = fb.read_results(0, N, check=False)                       # reported once; the records are there either way
        stuck = [i for i in range(N) if recs[i].status == smh._lib.FRAME_LSD_STUCK]
        assert stuck, "a spin budget of one poll did not trip the watchdog on any of %d frames" % N
        dd = smh.results_to_dicts(recs)
        for i in range(N):
            if i in stuck:
                assert recs[i].n_lines == 0 and recs[i].rounds == 0xFFFFFFFF and dd[i]["error"]
                # everything but the line search is valid for such a frame
                assert recs[i].map_open == 1 and recs[i].n_mask_px == clean[i].n_mask_px and recs[i].mpx == clean[i].mpx
            else:
                assert dd[i]["error"] is None and bytes(recs[i]) == bytes(clean[i])
        # the pipeline reports it from wait() (the watchdog belongs to k_lsd_tile: the batch-granular search)
        pipe = smh.Pipeline(vision, W, H, N, 3, search="batch")
        slot = pipe.submit(d.data_ptr(), N, anchors=anchors)
        with pytest.raises(smh.VisionError) as ei:
            pipe.wait(slot)
        assert ei.value.code == smh._lib.E_STATE
        pipe.wait(slot)                                                 # reported once
        lib.smhv_debug_lsd_spin_limit(0)
        slot = pipe.submit(d.data_ptr(), N, anchors=anchors)
        pipe.wait(slot)
        assert bytes(pipe.slots[slot].read_results(0, N)) == clean_b
        pipe.close()
    finally:
        lib.smhv_debug_lsd_spin_limit(0)
    fb.run(d.data_ptr(), N, anchors=anchors, stream=st)
    assert bytes(fb.read_results(0, N)) == clean_b
    fb.close()


def test_bench_two_ranks_on_one_gpu_over_gloo(vision):
    """bench.py's N > 1 path started the way the driver starts it -- `python bench.py --gpus 2`, no launcher in front: it
    launches its own two ranks (block shard, per-pass gather of the records on the slot's stream, MAX-over-ranks timing of
    every sub-region), here sharing this box's one GPU over gloo.  The JSON line must report both ranks' frames, and the
    gathered records of rank 0's block must be its own."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--frames-per-gpu", "16", "--rounds-per-step", "2",
           "--dist-backend", "gloo", "--force-device", "0", "--cpu-sample", "0", "--ingest-frames", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["value"] > 0 and out["value_depth1"] > 0
    assert out["config"]["frames_per_step"] == 64 and out["scaling"] == "weak" and out["config"]["baseline_config"] == 4
    assert out["slots_identical"] and out["gather_matches_rank0_records"] and out["value_min"] <= out["value"] <= out["value_max"]


def test_bench_node_leg_and_config0(vision):
    """bench.py --node (one process, smhv_node_run + smhv_node_gather per pass; a world of one on this box) and
    bench.py --config 0 (the reference's sample screenshot through the C oracle beside the GPU trait path)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--node", "--gpus", "1", "--config", "4", "--frames-per-gpu", "24", "--steps", "5",
                        "--warmup", "1", "--rounds-per-step", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["all_map_open"] and "smhv_node" in out["config"]["schedule"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["cases"][0]["size"] == [2560, 1440] and out["cases"][0]["gpu_lines_equal_cpu"] and out["cases"][1]["gpu_lines_equal_cpu"]
    assert set(out["cases"][0]["cpu_single_thread_stage_ms"]) == {"crop_to_map", "threshold_dilate", "lsd", "ocr_preprocess", "find_scales_preprocess"}


def test_node_entry_points_on_one_gpu(vision):
    """smhv_node_*: the single-process multi-GPU driver (ncclCommInitAll + one ncclGather of the records to the root) on the
    one GPU this box has: world = 1 communicator, gathered records == the pipeline's own, several runs, a short shard."""
    import ctypes as C
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    lib = smh._lib.load()
    W, H, N = 1920, 1080, 20
    frames, infos = synth.make_batch(W, H, N, first_idx=12000)
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
    want = bytes(fb.read_results(0, N))
    fb.close()
    node = C.c_void_p()
    devs = (C.c_int * 1)(0)
    smh._lib.check(lib.smhv_node_create(devs, 1, W, H, N, 2, smh._lib.LOG_FN(), C.byref(node)))
    try:
        sz = C.sizeof(smh._lib.FrameResult)
        for n_run in (N, N, 7, N):
            ptrs = (C.c_void_p * 1)(d.data_ptr())
            ns = (C.c_uint32 * 1)(n_run)
            anc = (C.c_void_p * 1)(C.cast(anchors, C.c_void_p))
            smh._lib.check(lib.smhv_node_run(node, ptrs, ns, smh.STAGE_ALL, 1, 15, anc))
            out = (smh._lib.FrameResult * N)()
            tot = C.c_uint32(0)
            smh._lib.check(lib.smhv_node_gather(node, out, C.byref(tot)))
            assert tot.value == n_run and bytes(out)[:n_run * sz] == want[:n_run * sz]
    finally:
        lib.smhv_node_destroy(node)
    lo, hi = C.c_uint64(), C.c_uint64()
    covered = []
    for r in range(8):
        lib.smhv_shard_range(8192 + 5, r, 8, C.byref(lo), C.byref(hi))
        covered.append((lo.value, hi.value))
    assert covered[0][0] == 0 and covered[-1][1] == 8197 and all(covered[i][1] == covered[i + 1][0] for i in range(7))
    assert max(h - l for l, h in covered) - min(h - l for l, h in covered) <= 1


def test_lsd_helpers_do_not_change_any_record(vision):
    """SMHV_STAGE_LSD_HELPERS: workgroups that have finished their frame ray-cast candidates for the frames still being
    searched.  Ray casting is a pure function of (mask, pixel), so every record must stay byte-identical -- on a batch
    with a few very heavy frames (where helpers do attach), at 1080p (one kernel) and 1440p (three residency kernels)."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    for (W, H, N) in ((1920, 1080, 96), (2560, 1440, 48)):
        frames, infos = synth.make_batch(W, H, N, first_idx=7000, n_lines=2)
        for i in range(0, N, 12):                                   # heavy frames: more lines, more blobs
            frames[i], infos[i] = synth.make_frame(W, H, 7500 + i, n_lines=6)
        anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
        d = torch.from_numpy(frames).cuda()
        fb = smh.FrameBatch(vision, W, H, N)
        s = torch.cuda.current_stream().cuda_stream
        fb.run(d.data_ptr(), N, anchors=anchors, stream=s)
        plain = bytes(fb.read_results(0, N))
        for stages in (smh.STAGE_ALL | smh.STAGE_LSD_HELPERS, smh.STAGE_ALL | smh.STAGE_LSD_HELPERS, smh.STAGE_ALL):
            fb.run(d.data_ptr(), N, stages=stages, anchors=anchors, stream=s)
            assert bytes(fb.read_results(0, N)) == plain
        fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_LSD_HELPERS | smh.STAGE_EXACT_STATS, anchors=anchors, stream=s)
        exact_h = bytes(fb.read_results(0, N))
        fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | smh.STAGE_EXACT_STATS, anchors=anchors, stream=s)
        assert bytes(fb.read_results(0, N)) == exact_h
        fb.close()


def test_ingest_queue_places_its_threads_next_to_the_gpu(vision):
    """On a multi-socket host the queue reports the CPUs of the GPU's NUMA node (sysfs), runs its hashing threads there,
    binds a producer thread to them on request, and accepts the same frames with and without the placement."""
    import os
    import threading
    import zlib
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = 1024, 768
    fr = [synth.make_frame(W, H, 950 + i, n_lines=1)[0] for i in range(5)]
    p = torch.cuda.get_device_properties(0)
    sysfs = "/sys/bus/pci/devices/%04x:%02x:%02x.0/" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    expect = set()
    try:
        if int(open(sysfs + "numa_node").read()) >= 0 and (os.path.exists("/sys/devices/system/node/node1") or int(open(sysfs + "numa_node").read()) > 0):
            for part in open(sysfs + "local_cpulist").read().strip().split(","):
                lo, _, hi = part.partition("-")
                expect |= set(range(int(lo), int(hi or lo) + 1))
    except OSError:
        pass
    crcs = []
    for affinity in (True, False):
        q = smh.IngestQueue(vision, W, H, slots=4, capacity=8, roi_upload=True, affinity=affinity)
        assert q.local_cpus() == expect
        seen = {}
        def producer():
            q.bind_thread()
            seen["cpus"] = os.sched_getaffinity(0)
            for f in fr:
                q.push(f)
        t = threading.Thread(target=producer)                      # (a thread of its own: the test process keeps its affinity)
        t.start(); t.join()
        assert (seen["cpus"] and seen["cpus"] <= expect) if expect else seen["cpus"] == os.sched_getaffinity(0)
        ptr, n, crc = q.batch()
        assert n == len(fr) and crc == zlib.crc32(fr[-1].tobytes())
        crcs.append(crc)
        q.close()
    assert crcs[0] == crcs[1]


# ---------------------------------------------------------------------------------------------------
# robustness of the boundary (round-1 advisor findings)
# ---------------------------------------------------------------------------------------------------
def test_ingest_queue_survives_more_frames_than_the_slab_holds(vision):
    """capacity + slots distinct frames: the slab fills up, the surplus stays queued in the staging slots, and after
    batch + reset it lands in the next slab -- nothing is lost and the queue never wedges."""
    import zlib
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H, SLOTS, CAP = 1024, 768, 3, 4
    fr = [synth.make_frame(W, H, 900 + i, n_lines=1)[0] for i in range(CAP + SLOTS + 2)]
    nb = W * H * 4
    q = smh.IngestQueue(vision, W, H, slots=SLOTS, capacity=CAP)
    for f in fr[:CAP + SLOTS]:
        q.push(f)                                                  # the last SLOTS frames cannot be appended: they stay queued
    with pytest.raises(smh.VisionError) as ei:                     # no free staging slot and a full slab: reported, recoverable
        q.push(fr[CAP + SLOTS])
    assert ei.value.code == smh._lib.E_STATE
    ptr, n, crc = q.batch()
    assert n == CAP and crc == zlib.crc32(fr[CAP - 1].tobytes())
    for i in range(CAP):
        assert smh.crc32_device(vision, ptr + i * nb, nb) == zlib.crc32(fr[i].tobytes())
    assert q.batch()[1] == CAP                                     # asking again changes nothing
    q.reset()
    q.push(fr[CAP + SLOTS])                                        # works again; the queued frames come first
    q.push(fr[CAP + SLOTS])                                        # duplicate of the previous capture: dropped
    ptr, n, crc = q.batch()
    assert n == SLOTS + 1 and crc == zlib.crc32(fr[CAP + SLOTS].tobytes())
    for i in range(SLOTS + 1):
        assert smh.crc32_device(vision, ptr + i * nb, nb) == zlib.crc32(fr[CAP + i].tobytes())
    assert q.counts() == (CAP + SLOTS + 1, 1)
    q.close()


def test_context_shutdown_before_its_children(built):
    """HipVision.shutdown() with a FrameBatch and an IngestQueue still alive: their calls fail cleanly and closing them
    afterwards is safe (the context object goes with its last child)."""
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    v = smh.HipVision.init(0)
    W, H = 1024, 768
    frame, _ = synth.make_frame(W, H, 1)
    d = torch.from_numpy(frame).cuda()
    fb = smh.FrameBatch(v, W, H, 2)
    q = smh.IngestQueue(v, W, H, slots=2, capacity=2)
    fb.run(d.data_ptr(), 1, stages=smh.STAGE_MARKERS)
    assert fb.read_results(0, 1)[0].map_open == 1
    v.shutdown()
    with pytest.raises(smh.VisionError):
        fb.run(d.data_ptr(), 1, stages=smh.STAGE_MARKERS)
    with pytest.raises(smh.VisionError):
        smh.FrameBatch(v, W, H, 1)
    fb.close(); q.close()
    v2 = smh.HipVision.init(0)                                      # the device is still usable
    fb2 = smh.FrameBatch(v2, W, H, 1)
    fb2.run(d.data_ptr(), 1, stages=smh.STAGE_MARKERS)
    assert fb2.read_results(0, 1)[0].map_open == 1
    fb2.close(); v2.shutdown()


def test_abi_bounds_checks(vision):
    import ctypes as C
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    W, H = 1024, 768
    fb = smh.FrameBatch(vision, W, H, 4)
    lib = smh._lib.load()
    recs = (smh._lib.FrameResult * 2)()
    assert lib.smhv_batch_read_results(fb._b, 0xFFFFFFFF, 2, recs) == smh._lib.E_INVALID     # first + n must not wrap
    assert lib.smhv_batch_read_results(fb._b, 3, 2, recs) == smh._lib.E_INVALID
    frames, infos = synth.make_batch(W, H, 4)
    d = torch.from_numpy(frames).cuda()
    short = smh.make_anchors([(infos[0]["scales_start_y"], infos[0]["anchors"])])
    with pytest.raises(ValueError):
        fb.run(d.data_ptr(), 4, anchors=short)
    many = smh.make_anchors([(10, [(100, 5, 10), (200, 6, 11), (300, 7, 12), (400, 8, 13)])])
    assert many[0].n == 3
    fb.close()


def test_host_supplied_ray_table(vision):
    """smhv_set_ray_table: the glibc table gives the committed results; a table that is not (cos, sin) of 0.1-degree
    steps is rejected; a table perturbed by one ulp in a few entries is accepted and restoring the original restores the
    results."""
    import fixtures as fx
    import squad_mortar_helper_amd as smh
    dx, dy = o.ray_table()
    frame, e, g = fx.load_fixture("points_intersect_png")
    st = smh.VisionState()
    try:
        vision.set_ray_table(dx, dy)
        assert np.array_equal(st.process(vision, frame).markers, g["lines"])
        with pytest.raises(smh.VisionError):
            vision.set_ray_table(dy, dx)
        dx2 = dx.copy()
        dx2[[7, 900, 1801, 3599]] = np.nextafter(dx2[[7, 900, 1801, 3599]], np.float32(2.0))
        vision.set_ray_table(dx2, dy)
        assert st.process(vision, frame).markers.shape[1] == 4
    finally:
        vision.set_ray_table(dx, dy)
    assert np.array_equal(st.process(vision, frame).markers, g["lines"])


def test_tracked_load_fallback_of_the_streaming_pass_gives_the_same_outputs():
    """`make tracked` (what the build falls back to when tools/check_untracked_loads.py rejects the compiled streaming pass): the
    same records and the same output images as the library with the hand-placed waits.  Own processes: one library each."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tracked = os.path.join(root, "squad-mortar-helper_amd", "libsmh_vision_hip_tracked.so")
    if not os.path.exists(tracked):
        pytest.skip("libsmh_vision_hip_tracked.so not built (__graft_entry__.build() builds it)")
    code = ("import sys; sys.path.insert(0, %r); import hashlib, numpy as np, torch, squad_mortar_helper_amd as smh\n"
            "from squad_mortar_helper_amd import synth\n"
            "h = hashlib.sha256()\n"
            "for (W, H, n) in ((1920, 1080, 12), (2560, 1440, 6), (1282, 1023, 5)):\n"
            "    fr, inf = synth.make_batch(W, H, n, first_idx=4100, n_lines=3)\n"
            "    a = smh.make_anchors([(i['scales_start_y'], i['anchors']) for i in inf]); d = torch.from_numpy(fr).cuda(); v = smh.HipVision.init(0)\n"
            "    p = smh.Pipeline(v, W, H, n, 4); [p.submit(d.data_ptr(), n, anchors=a) for _ in range(5)]; p.wait()\n"
            "    fb = smh.FrameBatch(v, W, H, n); fb.run(d.data_ptr(), n, anchors=a); torch.cuda.synchronize()\n"
            "    for b in (p.slots[0], fb):\n"
            "        h.update(bytes(b.read_results(0, n)))\n"
            "        for f in range(n):\n"
            "            for w in (smh._lib.IMAGE_UI_MAP, smh._lib.VIEW_LSD_INPUT, smh._lib.VIEW_OCR_INPUT, smh._lib.VIEW_FIND_SCALES_INPUT):\n"
            "                h.update(b.read_image(w, f).tobytes())\n"
            "print('SHA', h.hexdigest(), 'maps', open('/proc/self/maps').read().count('_tracked.so') > 0)\n") % root
    outs = []
    for env in ({}, dict(SMH_VISION_HIP_LIB=tracked)):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("SHA")][-1].split())
    assert outs[0][1] == outs[1][1], outs
    assert outs[0][3] == "False" and outs[1][3] == "True", outs     # each process really had the library it was meant to have
