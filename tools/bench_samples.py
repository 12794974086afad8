"""Throughput on the reference's own sample screenshots (tests/golden fixtures rebuilt into 2560x1440 frames), next
to the synthetic workload of bench.py: batches of 256 frames cycling the open-map 1440p fixtures through smhv_pipeline (depth 8; usage: [frames per batch] [depth]),
full marker pipeline
(button, ui_map, mask + dilation, LSD); GPU results are checked against the C oracle on every distinct frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import fixtures as fx
import squad_mortar_helper_amd as smh
from oracle import oracle as orc   # checker + CPU timing only


STAGES = int(os.environ.get("SMH_BENCH_STAGES", "0x3"), 0)   # (0x43: with helper workgroups, k_lsd)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    frames, stems = [], []
    for stem in fx.OPEN_STEMS:
        f, e, g = fx.load_fixture(stem)
        if f.shape[:2] == (1440, 2560):
            frames.append(f); stems.append(stem)
    k = len(frames)
    batch = np.stack([frames[i % k] for i in range(n)])
    vision = smh.HipVision.init(0)
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    search = os.environ.get("SAMPLES_SEARCH", "auto")
    if os.environ.get("SAMPLES_TILE_CAP"):
        smh._lib.load().smhv_debug_lsd_tile_cap(int(os.environ["SAMPLES_TILE_CAP"]))
    if os.environ.get("SAMPLES_TOUCH_FIRST"):
        torch.zeros(int(os.environ["SAMPLES_TOUCH_FIRST"]), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
    if os.environ.get("SAMPLES_FRAMES_FIRST"):
        d = torch.from_numpy(batch).cuda()
    pipe = smh.Pipeline(vision, 2560, 1440, n, depth=depth, search=search, service_workgroups=int(os.environ.get("SAMPLES_WGS", "0")),
                        flags=int(os.environ.get("SAMPLES_FLAGS", "0")), remote_after=int(os.environ.get("SAMPLES_AFTER", "0")), remote_tickets=int(os.environ.get("SAMPLES_TICKETS", "0")), remote_last=int(os.environ.get("SAMPLES_LAST", "0")))
    if not os.environ.get("SAMPLES_FRAMES_FIRST"):
        d = torch.from_numpy(batch).cuda()
    torch.cuda.synchronize()
    # warm: a GPU that comes out of idle needs a few hundred milliseconds to reach its clocks (40 cold steps measured 107 k
    # frames/s where the same pipeline runs at 190 k once warm)
    t_w = time.perf_counter()
    # (no drain inside the warm-up: a pipeline with SMHV_SEARCH_AUTO measures its two searches over windows of submissions here)
    while time.perf_counter() - t_w < float(os.environ.get("SAMPLES_WARM_S", "1.0")):
        for i in range(2 * depth):
            pipe.submit(d.data_ptr(), n, stages=STAGES, max_gap=15)
    pipe.wait()
    steps = int(os.environ.get("SAMPLES_STEPS", "400"))
    t0 = time.perf_counter()
    stamps = []
    for i in range(steps):
        slot = pipe.submit(d.data_ptr(), n, stages=STAGES, max_gap=15)
        stamps.append(time.perf_counter())
    pipe.wait()
    dt = time.perf_counter() - t0
    if os.environ.get("RATE_TIMELINE"):
        w = depth
        print("[timeline] k frames/s per %d submissions: %s" % (w, " ".join("%d" % (n * w / 1e3 / (stamps[i + w] - stamps[i])) for i in range(0, len(stamps) - w, w))), file=sys.stderr, flush=True)
    got = smh.results_to_dicts(pipe.slots[slot].read_results(0, n))
    pipe_stats = pipe.search_stats()
    if pipe_stats:
        pipe_stats = {k: pipe_stats[k] for k in ("launches", "frames", "busy_fraction", "cycles_per_frame", "help_cycles_per_frame", "remote_help", "timeline_ms", "adaptive", "mode", "measured_frames_per_s") if k in pipe_stats}
    t1 = time.perf_counter()
    ref = orc.process_batch(np.stack(frames), min(os.cpu_count() or 1, k), stages=STAGES, max_gap=15)
    cdt = time.perf_counter() - t1
    ok = True
    for i in range(n):
        r = ref[i % k]
        rl = np.array([[r.lines[a][b] for b in range(4)] for a in range(r.n_lines)], np.float32).reshape(-1, 4)
        ok = ok and got[i]["n_lines"] == r.n_lines and np.array_equal(got[i]["lines"], rl) and got[i]["rounds"] == r.rounds
    print("%d distinct 2560x1440 sample frames (%s ...), batch %d, stages ui_map+markers" % (k, ", ".join(stems[:3]), n))
    print("rounds per frame: %s" % [int(r.rounds) for r in ref])
    print("GPU: %.0f frames/s (%.3f ms per %d-frame step, smhv_pipeline depth %d); lines + rounds equal to the oracle: %s" % (n * steps / dt, dt / steps * 1e3, n, depth, ok))
    print("CPU oracle: %.1f frames/s on %d threads (%.2f s for %d frames)" % (k / cdt, min(os.cpu_count() or 1, k), cdt, k))
    import json
    print(json.dumps({"metric": "map frames/sec (2560x1440 sample screenshots, stages 0x%x), one GPU" % STAGES, "value": n * steps / dt, "unit": "frames/s",
                      "ms_per_step": dt / steps * 1e3, "config": {"workload": "%d distinct 2560x1440 frames rebuilt from tests/golden, cycled through a batch of %d" % (k, n),
                                                                "batch": n, "pipeline_depth": depth, "search": search, "search_service": pipe_stats, "stages": STAGES},
                      "rounds_per_frame": [int(r.rounds) for r in ref], "records_equal_oracle": bool(ok),
                      "cpu_oracle_frames_per_s": k / cdt, "cpu_threads": min(os.cpu_count() or 1, k), "data": "the reference's sample screenshots"}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
