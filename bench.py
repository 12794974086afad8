#!/usr/bin/env python3
"""bench.py -- map frames/s through the vision hot path on N MI355X GPUs (one process per GPU).

Workloads (BASELINE.json `configs`; --config picks one; default 2 = the configuration the metric is quoted on, 4 when
--gpus > 1):
  0  the reference's own CPU-runnable case: one real sample screenshot (the committed `point_intersect` fixture, 2560x1440,
     vision-gpu/src/lib.rs:571) and its 1080p synthetic analogue through the C oracle -- single-thread ms/frame + stage
     split + all-core frames/s -- beside the same frames' latency on the GPU through the per-call trait path
  1  batch = 1, 1920x1080 synthetic map, marker threshold + dilation + LSD only
  2  256 x 1920x1080 frames RESIDENT IN HBM per GPU, full pipeline: close-deployment button test, ui_map, marker
     threshold + dilation, ocr_preprocess, find_scales_preprocess + m/px, ray-cast line-segment detection, derived
     marker lengths / angles
  3  128 x 2560x1440 frames, full pipeline
  4  8192-frame batch block-sharded 1024 x 1920x1080 per GPU, full pipeline, the per-frame result records of every pass
     gathered to rank 0 over RCCL inside the timed step (N = 1: the 1024-frame shard alone)
Launch: `python bench.py --gpus N` starts its own N ranks (a child `python -m torch.distributed.run`, before anything touches
the GPU) and relays rank 0's JSON line; under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  Every rank
runs the same workload on its own block of the global batch (frames are independent: weak scaling).  `--node` drives the
same workload from ONE process through the C ABI's smhv_node_run / smhv_node_gather (ncclCommInitAll + ncclGather).

One "step" = --rounds-per-step passes of the hot path over the resident batch (sized so that the default 20 steps last
about a second); `value` counts every frame of every pass.  The K timed steps run as up to five sub-regions, each
bracketed by barrier + synchronize, and `value` is the MEDIAN sub-region's rate (min / max / whole-region mean beside it).
The passes go through smhv_pipeline_submit: the library owns the streams and the schedule (--pipeline-depth batches in
flight, default 4); `value_depth1` is the same workload with ONE batch in flight, timed right after.

Prints ONE JSON line on rank 0 (contract in the task statement).  Extra objects:
  roofline     -- the dominant HBM streaming kernel: algorithmic bytes / hipEvent launch duration on the run's stream
  stages_ms    -- average per-stage device time over the timed passes
  lsd          -- workload statistics of the (non-HBM-bound) ray-cast stage
  cpu_baseline -- the C oracle (a port of the reference's vision-cpu; the Rust original cannot be built here) timed on this
                  box's host cores on a bounded sample of the same frames, plus its single-thread per-stage split
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

CONFIGS = {
    0: dict(frames=1, width=2560, height=1440, stages=0xF, rounds=1,
            name="BASELINE configs[0]: single vision-common/samples screenshot (point_intersect, 2560x1440) through the CPU path"),
    1: dict(frames=1, width=1920, height=1080, stages=0x1, rounds=512,
            name="BASELINE configs[1]: batch=1 1920x1080 synthetic map, marker threshold + dilation + LSD only"),
    2: dict(frames=256, width=1920, height=1080, stages=0xF, rounds=96,
            name="BASELINE configs[2]: 256 x 1920x1080 BGRA frames resident in HBM per GPU, full pipeline (button, ui_map, "
                 "marker mask+dilate, LSD, ocr_preprocess, scales+m/px)"),
    3: dict(frames=128, width=2560, height=1440, stages=0xF, rounds=96,
            name="BASELINE configs[3]: 128 x 2560x1440 BGRA frames resident in HBM per GPU, full pipeline"),
    4: dict(frames=1024, width=1920, height=1080, stages=0xF, rounds=24,
            name="BASELINE configs[4]: 8192-frame batch block-sharded 1024 x 1920x1080 per GPU (8.5 GB resident per GPU), full "
                 "pipeline, RCCL gather of the per-frame result records (segment lists, m/px) to rank 0 inside every pass"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[] index of the workload (default: 2, or 4 when --gpus > 1)")
    ap.add_argument("--frames-per-gpu", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--stages", type=lambda s: int(s, 0), default=None)
    ap.add_argument("--rounds-per-step", type=int, default=None, help="passes over the resident batch per step")
    ap.add_argument("--lines", type=int, default=2, help="marker lines per synthetic frame")
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="batches in flight (smhv_pipeline_create depth; 0 = default: 12, or 8 for 1024-frame shards; 12 x 256 frames resident outputs = 16 GB of the 288: what the frame-granular search needs to hide its one-wave-per-frame latency)")
    ap.add_argument("--distinct", type=int, default=None,
                    help="distinct synthetic frames per GPU (0 = every frame distinct); fewer are tiled on the device.  Default: every "
                         "frame distinct, except config 4 (1024 frames per GPU): 256 distinct frames per GPU, tiled four times -- "
                         "generating 1024 frames per rank on the host cores is minutes of set-up before the first GPU call")
    ap.add_argument("--node", action="store_true",
                    help="ONE process drives all --gpus devices through the C ABI (smhv_node_run + smhv_node_gather) instead of one rank per GPU")
    ap.add_argument("--tile-cap", type=int, default=0,
                    help="diagnostic: cap the tile store of k_lsd_tile (smhv_debug_lsd_tile_cap): fewer tiles = less LDS = more workgroups per CU")
    ap.add_argument("--idle-streams", type=int, default=0,
                    help="diagnostic: create this many HIP streams before the pipeline (the schedule must not depend on them)")
    ap.add_argument("--search", default="auto", choices=("auto", "batch", "frame"),
                    help="line-search schedule of the pipeline (smhv_pipeline_options::search); auto: batch-granular below depth 6, from there on the pipeline measures both")
    ap.add_argument("--room", default="auto", choices=("auto", "on", "off"),
                    help="smhv_pipeline_options::room_for_others: the search kernel leaves an eighth of the CUs without a workgroup of its own so that "
                         "other owners' kernels (RCCL's gather) find a CU; auto: on when --gpus > 1 (every pass gathers over RCCL), off for one GPU")
    ap.add_argument("--side-probe", type=int, default=32,
                    help="workgroups of the co-residency probe (a kernel with RCCL's footprint: 21 KB LDS, 280 VGPRs) launched once per pass beside "
                         "a pipeline with room_for_others, after the timed region (N=1, config 2; 0 = skip): `co_residency` in the JSON line")
    ap.add_argument("--gather-probe", action="store_true", help="N = 1: also run the co-residency leg with RCCL itself (a communicator of one rank, one all_gather per pass on a side stream; RCCL prints its banner to stdout in front of the JSON line)")
    ap.add_argument("--cpu-sample", type=int, default=128, help="frames for the CPU baseline (0 = skip)")
    ap.add_argument("--timed-regions", type=int, default=1, help="cut the K timed steps into this many regions, each bracketed by barrier + synchronize (value = median; default 1 = the contract's one region)")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--no-traffic-probe", action="store_true", help="skip the live HBM-traffic measurement of the streaming kernel (two short rocprofv3 --pmc child runs after the timed region); roofline.traffic then comes from the committed profiles/traffic.json")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-real-samples", action="store_true", help="skip the leg on the reference's own 1440p screenshots (real_samples)")
    ap.add_argument("--no-depth1", action="store_true", help="skip the one-batch-in-flight region (value_depth1)")
    ap.add_argument("--no-back-to-back", action="store_true",
                    help="skip the back-to-back leg of the streaming pass (profiled runs: its overlapping launches would be averaged "
                         "into the kernel's isolated duration)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo (single-box testing of the N>1 code path)")
    ap.add_argument("--force-device", type=int, default=None, help="testing only: put every rank on this device")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="testing only (no GPU needed): the ranks meet, reduce one number over the process group, rank 0 prints a JSON line")
    ap.add_argument("--ingest-affinity", default="on", choices=("on", "off"), help="A/B: the ingest queue's hashing threads and the producer on the CPUs next to the GPU (on) or left to the scheduler")
    ap.add_argument("--ingest-frames", type=int, default=8192,
                    help="frames streamed through the ingest queue for the PCIe-inclusive figure (0 = skip; rank 0, N=1, config 2 only)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher in front: start the N ranks as a CHILD process (never exec: this
    process may not have touched the GPU, but a replaced process image is not worth the risk on this pool), relay their
    output and leave with their exit code.  Nothing here imports torch or initialises HIP."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def emit(obj):
    """The JSON line, LAST on stdout: RCCL prints its version banner through C stdio, whose buffer (when stdout is a pipe)
    would otherwise be flushed after Python's at exit and end up behind the line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass
    sys.stdout.write(json.dumps(obj) + "\n")
    sys.stdout.flush()


def algorithmic_bytes(roi_w, roi_h, btn_w, btn_h, stages):
    """each input ROI pixel read once as BGRA, each API-visible output written once.
    -> (bytes of the streaming kernel per frame, bytes of the whole step per frame)"""
    qw, qh = roi_w // 2, roi_h // 2
    px = roi_w * roi_h
    kernel = px * 4                                            # read the ROI as BGRA
    if stages & 0x2:
        kernel += px * 4                                       # ui_map RGBA
    if stages & 0x1:
        kernel += px                                           # u8 marker mask
    if stages & 0x4:
        kernel += qw * qh                                      # ocr_out
    if stages & 0x8:
        kernel += qw * qh                                      # scales
    full = btn_w * btn_h * 4 + kernel + (516 if stages & 0x1 else 0)   # + button pixels, <= 32 lines
    return kernel, full


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_threads(k):
    quota = cpu_quota_cores()
    cores = min(os.cpu_count() or 1, k)              # threads actually used: one frame per thread at a time
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))  # more threads than the container's CPU quota only time-slice
    return cores


def omp_threads(n):
    """The C oracle's per-pixel loops are OpenMP loops (its stand-alone stage functions would otherwise use every core):
    pin the team size for the single-thread figures.  -> previous maximum."""
    import ctypes
    try:
        g = ctypes.CDLL("libgomp.so.1")
        prev = g.omp_get_max_threads()
        g.omp_set_num_threads(int(n))
        return prev
    except OSError:
        return None


def cpu_stage_split(orc, frames, infos, max_gap=15):
    """Single-thread per-stage milliseconds of the C oracle: median over the given frames, after one
    untimed pass over the same frames (first-touch page faults in freshly mapped output arrays moved whole stages by 10-40x
    between runs when it was a plain mean over cold calls)."""
    names = ("crop_to_map", "threshold_dilate", "lsd", "ocr_preprocess", "find_scales_preprocess")
    samples = {k: [] for k in names}

    def one(fr, info, record):
        t0 = time.perf_counter(); crop = orc.crop_to_map(fr, True)
        t1 = time.perf_counter(); mask = orc.mask_marker_lines(crop["cropped_map"])
        t2 = time.perf_counter(); orc.find_lines(mask, max_gap)
        t3 = time.perf_counter(); orc.ocr_preprocess(crop["cropped_brq"])
        t4 = time.perf_counter(); orc.find_scales_preprocess(crop["cropped_brq"], info["scales_start_y"])
        t5 = time.perf_counter()
        if record:
            for k, dt in zip(names, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                samples[k].append(dt)

    prev = omp_threads(1)
    for fr, info in zip(frames, infos):                      # untimed pass: the allocator and the page cache settle
        one(fr, info, False)
    for fr, info in zip(frames, infos):
        one(fr, info, True)
    if prev:
        omp_threads(prev)
    return {k: sorted(v)[len(v) // 2] * 1e3 for k, v in samples.items()}


def records_equal_oracle(np, rec, ref):
    """One GPU record (dict) against one oracle record: integer fields and line end points bit for bit."""
    lines = np.array([[ref.lines[i][j] for j in range(4)] for i in range(ref.n_lines)], np.float32).reshape(-1, 4)
    return (rec["n_lines"] == ref.n_lines and rec["rounds"] == ref.rounds and rec["n_mask_px"] == ref.n_mask_px
            and np.array_equal(rec["lines"], lines) and rec["mpx"] == (ref.mpx if ref.has_mpx else None))


class DeviceWatch:
    """Shader clock, temperature and power of the device while a region runs (sysfs hwmon of the amdgpu cards, sampled by a thread
    every 50 ms; best effort: None where the box does not show them).  A box shows every GPU of its host, whichever one the
    process was given: all of them are sampled and the one that drew the most power in the region -- the one under load -- is
    reported.  A 513 k vs 551 k spread between boxes is attributable only with these beside the value."""

    def __init__(self, pci=None):
        import glob
        self.pci = pci.lower() if pci else None                 # "dddd:bb:dd" of the process's device (torch: pci_domain_id / pci_bus_id / pci_device_id)
        self.cards = []
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(card, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            files = {}
            for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                for key, names in (("sclk_mhz", ("freq1_input",)), ("temp_c", ("temp2_input", "temp1_input")), ("power_w", ("power1_average", "power1_input"))):
                    for nm in names:
                        fpath = os.path.join(hw, nm)
                        if key not in files and os.path.exists(fpath):
                            files[key] = fpath
            if files:
                self.cards.append((os.path.basename(os.path.realpath(card)), files, {k: [] for k in files}))
        self.files = bool(self.cards)
        self._stop = None
        self._thr = None

    def _read(self):
        scale = {"sclk_mhz": 1e-6, "temp_c": 1e-3, "power_w": 1e-6}
        for _, files, samples in self.cards:
            for k, f in files.items():
                try:
                    samples[k].append(float(open(f).read().strip()) * scale[k])
                except (OSError, ValueError):
                    pass

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self._read()
                self._stop.wait(0.05)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thr.join(timeout=1.0)

    def summary(self):
        best = None
        for name, _, samples in self.cards:
            if self.pci and name.lower().startswith(self.pci):
                best = (0.0, name, samples)
                break
        for name, _, samples in ([] if best else self.cards):
            load = samples.get("power_w") or samples.get("sclk_mhz") or []
            if load and (best is None or sum(load) / len(load) > best[0]):
                best = (sum(load) / len(load), name, samples)
        if best is None:
            return None
        out = {"card": best[1], "cards_sampled": len(self.cards), "card_is": "the process's device (PCI address)" if self.pci and best[1].lower().startswith(self.pci) else "the card that drew the most power"}
        for k, v in best[2].items():
            if v:
                out[k] = {"min": min(v), "max": max(v), "mean": sum(v) / len(v), "samples": len(v)}
        return out


def co_residency_leg(smh, torch, vision, W, H, n, depth, fptr, anchors, stages, probe_wgs, passes=300, rccl_probe=False):
    """Can a kernel of ANOTHER owner run beside the pipeline (never `value`)?  A pipeline created with room_for_others (what every
    N > 1 run uses: its gather is an RCCL kernel) is saturated; once per pass a probe kernel with RCCL's footprint on gfx950 (21 KB
    of LDS, 280 VGPRs per 256-thread workgroup; smhv_debug_side_kernel) goes onto a stream of its own.  -> the pipeline's rate
    without and with the probes, and the probes' enqueue-to-completion times (hipEvents on the probe stream)."""
    import numpy as np
    lib = smh._lib.load()
    pipe = smh.Pipeline(vision, W, H, n, depth, search="frame", room_for_others=1)
    side = torch.cuda.Stream()

    def run(with_probe, wgs=8):
        for _ in range(2 * depth):
            pipe.submit(fptr, n, stages=stages, anchors=anchors)
        pipe.wait()
        evs = []
        t0 = time.perf_counter()
        for _ in range(passes):
            pipe.submit(fptr, n, stages=stages, anchors=anchors)
            if with_probe:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                smh._lib.check(lib.smhv_debug_side_kernel(vision._ctx, wgs, side.cuda_stream))
                e1.record(side)
                evs.append((e0, e1))
        pipe.wait()
        dt = time.perf_counter() - t0
        side.synchronize()
        return n * passes / dt, (np.array([a.elapsed_time(b) for a, b in evs]) if evs else None)

    r0, _ = run(False)
    by_size = {}
    for wg in sorted({8, probe_wgs}):
        r1, lat = run(True, wg)
        r2, _ = run(False)
        base = 0.5 * (r0 + r2)
        by_size[str(wg)] = {"frames_per_s_without_probe": base, "frames_per_s_with_probe": r1, "cost": 1.0 - r1 / base,
                            "probe_ms": {"median": float(np.median(lat)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())}}
        r0 = r2
    rccl = None
    if rccl_probe:
        # --gather-probe: the same with RCCL itself -- a communicator of one rank, one all_gather of a batch's records per pass on
        # the side stream.  (With one rank RCCL has nobody to talk to: whether it still launches its kernel or copies is its
        # business; the probe kernel above has the footprint of the kernels it launches between ranks.)
        import socket
        import torch.distributed as dist
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        try:
            rec = torch.zeros(n * 1024, dtype=torch.uint8, device="cuda")     # (a batch's records: n x sizeof(smhv_frame_result) is of this order)
            out_t = torch.empty_like(rec)

            def run_rccl():
                for _ in range(2 * depth):
                    pipe.submit(fptr, n, stages=stages, anchors=anchors)
                pipe.wait()
                evs = []
                t0 = time.perf_counter()
                for _ in range(passes):
                    pipe.submit(fptr, n, stages=stages, anchors=anchors)
                    with torch.cuda.stream(side):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(side)
                        dist.all_gather_into_tensor(out_t, rec)
                        e1.record(side)
                        evs.append((e0, e1))
                pipe.wait()
                dt = time.perf_counter() - t0
                side.synchronize()
                return n * passes / dt, np.array([a_.elapsed_time(b_) for a_, b_ in evs])
            run_rccl()
            r1, lat = run_rccl()
            r2, _ = run(False)
            rccl = {"frames_per_s_with_all_gather": r1, "frames_per_s_without": r2, "cost": 1.0 - r1 / r2,
                    "all_gather_ms": {"median": float(np.median(lat)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
                    "what": "torch.distributed (backend nccl = RCCL), world size 1: one all_gather_into_tensor of %d bytes per pass on a side stream" % rec.numel()}
        finally:
            dist.destroy_process_group()
    geo = pipe.peek()
    pipe.close()
    first = by_size[str(min(int(k) for k in by_size))]
    return {"rccl_world_of_one": rccl, "frames_per_s_without_probe": first["frames_per_s_without_probe"], "frames_per_s_with_probe": first["frames_per_s_with_probe"], "cost": first["cost"],
            "probe_ms": first["probe_ms"], "by_probe_workgroups": by_size,
            "probe": "8 (headline figures) and %d workgroups x 256 threads, 21 KB LDS, 280 VGPRs (RCCL's kernels on gfx950: 19.7-21.2 KB, 261-280), one launch per pass on its own stream" % probe_wgs,
            "service_workgroups": geo["service_workgroups"], "waves_per_workgroup": geo["waves_per_workgroup"], "pipeline_depth": depth, "passes": passes,
            "note": "room_for_others = 1 (an eighth of the CUs without a search workgroup): what bench.py --gpus N > 1 and smhv_node run with"}


def trait_path_leg(smh, vision, frame, labels, reps=40):
    """The per-call (drop-in) path on one frame of the workload (never `value`): VisionState.process -- load_frame, crop_to_map,
    find_minimap, then the markers branch and the scales branch on two threads -- `reps` times; wall time per frame and the
    library's own per-call table (smhv_trait_times: the reference wraps every trait call in a Timeshares entry,
    vision-common/src/debug.rs:3-30, src/vision/mod.rs:54-66).  Two sequences: the one the trait allows (the drop-in latency) and the
    one with the lazy ui_map."""
    import numpy as np

    def run(state):
        for _ in range(5):
            state.process(vision, frame, ocr_labels=labels)
        vision.trait_times(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            res = state.process(vision, frame, ocr_labels=labels)
        ms = (time.perf_counter() - t0) / reps * 1e3
        tt = vision.trait_times(reset=True)
        state.close()
        per = {k: v[0] / max(v[1], 1) for k, v in tt.items() if v[1]}
        crit = per.get("load_frame", 0.0) + per.get("crop_to_map", 0.0) + per.get("find_minimap", 0.0) + max(
            per.get("isolate_map_markers", 0.0) + per.get("mask_marker_lines", 0.0) + per.get("find_marker_lines", 0.0),
            per.get("ocr_preprocess", 0.0) + per.get("find_scales_preprocess", 0.0) + per.get("calc_meters_to_px_ratio", 0.0))
        return ms, per, crit, res
    # the sequence the trait allows and rust/smh-vision-hip issues: crop_to_map returns the ui_map BY VALUE
    # (vision-common/src/lib.rs:47; the CUDA back-end copies and synchronises inside the call, vision-gpu/src/lib.rs:291-297)
    e_ms, e_per, e_crit, e_res = run(smh.VisionState(lazy_map=False))
    # a host written for this library: crop_to_map returns at the button test, smhv_ui_map hands the image out of pinned memory
    ms, per, crit, res = run(smh.VisionState(lazy_map=True, copy_map=False))
    return {"eager_ms_per_frame": e_ms, "eager_per_call_ms": e_per, "eager_critical_path_ms": e_crit,
            "ms_per_frame": ms, "per_call_ms": per, "critical_path_ms": crit, "frames": reps, "lines": int(len(res.markers)),
            "same_results": bool(np.array_equal(res.markers, e_res.markers) and np.array_equal(res.map, e_res.map) and res.meters_to_px_ratio == e_res.meters_to_px_ratio),
            "what": "VisionState.process through the C ABI's trait functions (host frame in pageable memory -> results on the host), one frame at a time. "
                    "eager_* = THE DROP-IN LATENCY: the sequence the reference's host issues through the Rust shim -- smhv_crop_to_map(ctx, gray, &open, roi, ui) "
                    "returns the ui_map by value as the trait demands. ms_per_frame = the same with crop_to_map(ui = NULL) + smhv_ui_map (the image travels to "
                    "pinned memory while the branches run): needs a host that asks for the image later, i.e. a change to src/vision/mod.rs. "
                    "per_call_ms = the library's wall-clock table per trait call; critical_path_ms = load_frame + crop_to_map + find_minimap + the longer branch; "
                    "the rest is the Python caller (thread hand-over, ctypes)"}


def real_samples_leg(smh, torch, vision, depth, batch=128, steps=200):
    """The reference's own screenshots beside the synthetic scene (never `value`): the 2560x1440 open-map fixtures of
    tests/golden (crops of vision-common/samples/*, the images the reference's one GPU test runs on, vision-gpu/src/lib.rs:571)
    rebuilt into full frames and cycled through a batch of `batch` frames, ui_map + markers, through the same smhv_pipeline.
    A real scene's cost per frame is anything from 0 to 372 find_longest_line rounds; the synthetic generator's is 16-157."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures as fx
    from oracle import oracle as orc                   # checker only, outside every timed region
    frames, stems = [], []
    for stem in fx.OPEN_STEMS:
        f, _e, _g = fx.load_fixture(stem)
        if f.shape[:2] == (1440, 2560):
            frames.append(f)
            stems.append(stem)
    k = len(frames)
    d = torch.from_numpy(np.stack([frames[i % k] for i in range(batch)])).cuda()
    stages = 0x3

    def rate(dep):
        pipe = smh.Pipeline(vision, 2560, 1440, batch, depth=dep)
        for _ in range(26 * dep if dep >= 6 else 2 * dep):     # (from depth 6 on the pipeline first measures both of its searches: 24 x depth submissions)
            pipe.submit(d.data_ptr(), batch, stages=stages, max_gap=15)
        pipe.wait()
        modes[str(dep)] = (pipe.search_stats() or {}).get("mode", "batch-granular")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            slot = pipe.submit(d.data_ptr(), batch, stages=stages, max_gap=15)
        pipe.wait()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        got_ = smh.results_to_dicts(pipe.slots[slot].read_results(0, batch))
        pipe.close()
        return dt_, got_

    modes = {}
    dt, got = rate(depth)
    by_depth = {str(depth): batch * steps / dt}
    for dep in (4, 8, 20):                                 # (depth 4: the batch-granular search; 20: frames are cheap to keep in flight -- 24 GB of the 288)
        if dep != depth:
            dt2, got2 = rate(dep)
            by_depth[str(dep)] = batch * steps / dt2
            same = all(a["n_lines"] == b["n_lines"] and np.array_equal(a["lines"], b["lines"]) and a["rounds"] == b["rounds"] for a, b in zip(got, got2))
            if not same:
                raise SystemExit("bench.py: the sample screenshots' records differ between pipeline depths %d and %d" % (depth, dep))
    ref = orc.process_batch(np.stack(frames), cpu_threads(k), stages=stages, max_gap=15)
    ok = True
    for i in range(batch):
        r = ref[i % k]
        rl = np.array([[r.lines[a][b] for b in range(4)] for a in range(r.n_lines)], np.float32).reshape(-1, 4)
        ok = ok and got[i]["n_lines"] == r.n_lines and np.array_equal(got[i]["lines"], rl) and got[i]["rounds"] == r.rounds and got[i]["n_mask_px"] == r.n_mask_px
    rounds = [int(r.rounds) for r in ref]
    return {"frames_per_s": batch * steps / dt, "ms_per_pass": dt / steps * 1e3,
            "workload": "%d distinct 2560x1440 open-map screenshots of vision-common/samples (committed fixtures), cycled through a batch of %d; ui_map + markers" % (k, batch),
            "batch": batch, "pipeline_depth": depth, "stages": stages, "passes": steps, "frames_per_s_by_depth": by_depth, "search_by_depth": modes,
            "rounds_per_frame": {"min": min(rounds), "max": max(rounds), "mean": float(np.mean(rounds)), "all": rounds},
            "lines_per_frame_mean": float(np.mean([int(r.n_lines) for r in ref])),
            "records_equal_oracle": bool(ok), "note": "outside the timed region of `value`; the headline stays the synthetic configs[2] scene"}


def cpu_throttled():
    """(periods, microseconds) this container's cgroup has been throttled by its CPU quota so far (cgroup v2 cpu.stat; (0, 0) when unknown)."""
    try:
        kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except (OSError, ValueError):
        return 0, 0


def ingest_leg(smh, torch, vision, pipe, src, anchors, stages, frames_total, W, H, n, affinity=True):
    """PCIe-inclusive rate (never `value`): frames travel pinned host memory -> HBM through the ingest queue and every full
    slab goes through the same pipeline; two queues alternate so the uploads of one slab overlap the compute of the other.
    The staging buffers are filled once; each frame then gets a fresh counter in pixel (0,0) (outside every ROI) so that no
    CRC repeats.  `frames_per_s`: region-of-interest upload (whole-frame CRC-32 on the host cores, one worker per staging
    slot; only the map ROI's and the button's rows cross PCIe); `full_upload_frames_per_s`: the whole frame crosses and the
    device computes the CRC (the mode a device-side producer or a decoded RGB image uses), on a quarter of the frames."""
    def run(roi, slots, total, native=True):
        qs = [smh.IngestQueue(vision, W, H, slots=slots, capacity=n, roi_upload=roi, affinity=affinity) for _ in range(2)]
        for q in qs:                                           # prime the staging buffers (their content persists)
            for i in range(slots):
                q.acquire()[...] = src[i % len(src)]
                q.commit()
            q.batch()
            q.reset()
        slabs = max(2, (total + n - 1) // n)
        counter = 1
        pending = [None, None]
        if affinity:
            try:
                qs[0].bind_thread()                            # the producer fills staging buffers that are pinned next to the GPU: run on that socket
            except smh.VisionError as e:                       # (a cpuset that excludes those CPUs: the leg still runs, wherever the scheduler puts it)
                print("bench.py: ingest producer not bound to the GPU's CPUs: %s" % e, file=sys.stderr)
        t0 = None
        stamps = []
        for b in range(-2, slabs):                             # (two untimed slabs first: the pipeline behind the queues starts from idle)
            if b == 0:
                pipe.wait()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if b >= 0:
                stamps.append(time.perf_counter())
            q = qs[b % 2]
            if pending[b % 2] is not None:
                pipe.wait(pending[b % 2])                      # the previous run on this queue's slab has finished
            q.reset()
            if native:
                counter = q.feed(n, counter)                   # the capture loop in native code, as the reference's is (src/capture.rs)
            else:
                for _ in range(n):
                    buf = q.acquire()
                    buf[0, 0, :] = (counter & 255, (counter >> 8) & 255, (counter >> 16) & 255, 255)
                    counter += 1
                    q.commit()
            ptr, cnt, _ = q.batch()
            assert cnt == n, "ingest dropped frames: %d of %d" % (cnt, n)
            pending[b % 2] = pipe.submit(ptr, cnt, stages=stages, grayscale=True, max_gap=15, anchors=anchors)
        pipe.wait()
        dt = time.perf_counter() - t0
        run.slab_s = [b_ - a_ for a_, b_ in zip(stamps, stamps[1:])]   # host time from one slab's hand-over to the next's (diagnostic)
        return qs, slabs * n, dt

    cores = os.cpu_count() or 8
    slots = max(4, min(32, cores // 4))
    affinity_before = os.sched_getaffinity(0)
    thr0 = cpu_throttled()
    qs, frames, dt = run(True, slots, frames_total)
    slab_s = sorted(run.slab_s)
    thr1 = cpu_throttled()
    for q in qs:
        q.close()
    qs, frames_py, dt_py = run(True, slots, max(2 * n, frames_total // 4), native=False)   # the same with a Python capture loop (three ctypes calls per frame)
    local_cpus = qs[0].local_cpus()
    q = qs[0]
    q.reset()
    k = min(64, n)
    t1 = time.perf_counter()
    for i in range(k):                                         # pageable source: one extra host copy per frame (bounded sample)
        src[i % len(src)][0, 0, 3] = 254 - (i & 1)             # alternate so consecutive CRCs differ
        q.push(src[i % len(src)])
    q.batch()
    dt_push = time.perf_counter() - t1
    for i in range(min(k, len(src))):
        src[i][0, 0, 3] = 255
    _, dup = qs[0].counts()
    for q in qs:
        q.close()
    qs, frames_full, dt_full = run(False, 4, max(2 * n, frames_total // 4))
    for q in qs:
        q.close()
    os.sched_setaffinity(0, affinity_before)                   # (the other legs run where they ran before)
    (_, _, rw, rh), (_, _, bw, bh) = smh.map_bounds(W, H), smh.button_bounds(W, H)
    # the host CRC alone, one thread, on frames that do not fit the cache together (what a worker does between memcpys)
    import ctypes as C
    lib = smh._lib.load()
    crc_level = lib.smhv_debug_crc32_host_level(None, 0, -1, None)
    import numpy as np
    bufs = [np.ascontiguousarray(src[i % len(src)]) for i in range(min(len(src), 24))]
    t1 = time.perf_counter()
    for _ in range(3):
        for b_ in bufs:
            lib.smhv_crc32_host(C.c_void_p(b_.ctypes.data), b_.nbytes)
    crc_one = 3 * sum(b_.nbytes for b_ in bufs) / (time.perf_counter() - t1) / 1e9
    # what THIS box's link gives pinned host memory -> HBM, for the figure above to be read against: copies of the packed size
    # (3.3 MB at 1080p) round robin on three streams (what the queue does), and one 256 MB copy (the link's best case)
    ceil_GBps = ceil_big_GBps = None
    try:
        pack = (rw * rh + bw * bh) * 4
        hsrc = torch.empty(pack * 24, dtype=torch.uint8, pin_memory=True)
        ddst = torch.empty(pack * 24, dtype=torch.uint8, device="cuda")
        sts = [torch.cuda.Stream() for _ in range(3)]
        for timed_round in (False, True):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 20 if timed_round else 2
            for r_ in range(reps):
                for i in range(24):
                    with torch.cuda.stream(sts[i % 3]):
                        ddst[i * pack:(i + 1) * pack].copy_(hsrc[i * pack:(i + 1) * pack], non_blocking=True)
            torch.cuda.synchronize()
            ceil_GBps = reps * 24 * pack / (time.perf_counter() - t1) / 1e9
        big = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
        dbig = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        for timed_round in (False, True):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(4):
                dbig.copy_(big, non_blocking=True)
            torch.cuda.synchronize()
            ceil_big_GBps = 4 * (256 << 20) / (time.perf_counter() - t1) / 1e9
        del hsrc, ddst, big, dbig
    except Exception as e:  # noqa: BLE001  (calibration only)
        print("bench.py: H2D ceiling not measured: %s" % e, file=sys.stderr)
    return {"frames_per_s": frames / dt, "frames": frames, "mode": "roi_upload", "staging_slots": slots, "host_cores": cores,
            "producer": "native capture loop (smhv_debug_ingest_feed: acquire, stamp pixel (0,0), commit -- the reference's capture thread is native code, src/capture.rs)",
            "python_producer_frames_per_s": frames_py / dt_py,
            "frames_per_s_by_slab": ({"median": n / slab_s[len(slab_s) // 2], "slowest": n / slab_s[-1], "fastest": n / slab_s[0], "slabs": len(slab_s),
                                      "what": "the same timed loop slab by slab (%d frames each; host time between consecutive hand-overs): a run whose total is "
                                              "well below its median slab had a few slow slabs, not a slow link" % n} if slab_s else None),
            "h2d_ceiling_GBps": {"packed_copies_three_streams": ceil_GBps, "one_256MB_copy": ceil_big_GBps,
                                 "what": "pinned host memory -> HBM on this box, measured here: copies of the queue's packed size round robin on three streams / one large copy"},
            "host_crc_GBps": frames * W * H * 4 / dt / 1e9, "host_crc_loop": {2: "VPCLMULQDQ (512-bit folding)", 1: "PCLMULQDQ", 0: "tables"}.get(crc_level, "?"),
            "host_crc_GBps_one_thread": crc_one, "uploaded_fraction": round((rw * rh + bw * bh) / float(W * H), 3),
            "h2d_GBps": frames * (rw * rh + bw * bh) * 4 / dt / 1e9,
            "full_upload_frames_per_s": frames_full / dt_full, "full_upload_h2d_GBps": frames_full * W * H * 4 / dt_full / 1e9,
            "push_frames_per_s": k / dt_push, "duplicates_dropped": dup,
            "cpu_quota_throttling_during_the_timed_loop": {"periods": thr1[0] - thr0[0], "ms": (thr1[1] - thr0[1]) / 1e3},
            "producer_and_workers_on": ("the %d CPUs next to the GPU (smhv_ingest_local_cpus)" % len(local_cpus)) if (local_cpus and affinity) else "any CPU (one NUMA node, sysfs does not say, or --ingest-affinity off)",
            "note": "PCIe-inclusive: pinned staging -> whole-frame CRC-32 on the host workers (dedupe rule of src/capture.rs:44-47) -> "
                    "packed ROI + button rows -> async H2D -> slab -> same pipeline, two slabs in flight; full_upload: whole frame H2D + "
                    "device CRC-32; push_frames_per_s adds the host memcpy from pageable memory (one thread)"}


def upload_synthetic(torch, synth, W, H, n, first, lines, distinct, device, keep_host):
    """n resident frames on `device`: generated in chunks through one pinned staging buffer (a 1024-frame shard is 8.5 GB:
    no host copy of the whole batch).  distinct < n: that many frames, tiled.  -> (device tensor, infos, host copy of the
    first `keep_host` frames, seconds spent in the H2D copies)."""
    import numpy as np
    k = n if distinct <= 0 else min(distinct, n)
    chunk = min(64, k)
    stage = torch.empty((chunk, H, W, 4), dtype=torch.uint8, pin_memory=True)
    d = torch.empty((n, H, W, 4), dtype=torch.uint8, device=device)
    infos, host = [], np.empty((min(keep_host, k), H, W, 4), np.uint8)
    h2d = 0.0
    t_gen = time.perf_counter()
    for c0 in range(0, k, chunk):
        c = min(chunk, k - c0)
        _, inf = synth.make_batch(W, H, c, first_idx=first + c0, n_lines=lines, out=stage.numpy()[:c])
        infos += inf
        if c0 < len(host):
            m = min(c, len(host) - c0)
            host[c0:c0 + m] = stage.numpy()[:m]
        t0 = time.perf_counter()
        d[c0:c0 + c].copy_(stage[:c], non_blocking=True)
        torch.cuda.synchronize(device)
        h2d += time.perf_counter() - t0
    for c0 in range(k, n, k):                                  # tile the distinct frames over the rest of the batch
        c = min(k, n - c0)
        d[c0:c0 + c].copy_(d[:c])
    infos = [infos[i % k] for i in range(n)]
    torch.cuda.synchronize(device)
    print("bench.py[%s]: %d distinct %dx%d frames generated and uploaded in %.1f s (%.1f s of it H2D), tiled to %d resident frames"
          % (device, k, W, H, time.perf_counter() - t_gen, h2d, n), file=sys.stderr)
    return d, infos, host, h2d


def sub_regions(steps, regions=1):
    """The K timed steps as `regions` regions (default ONE: exactly K steps between two barrier + synchronize brackets, as the
    driver's contract words it; rounds 1-5 cut them into five and reported the median -- every inner barrier drains the pipeline
    once, 5 ms of each 170 ms region).  --timed-regions N keeps the old form for a min / max."""
    k = max(1, min(regions, steps))
    return [steps * (i + 1) // k - steps * i // k for i in range(k)]


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def config0(args):
    """SURVEY 8(d) "Config 1" (BASELINE configs[0]): the reference's own test image through the CPU path, with the same
    frame's latency through the GPU trait path beside it.  One JSON line."""
    import numpy as np
    import torch  # noqa: F401  (HIP runtime first: see _lib.load)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures as fx
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    from oracle import oracle as orc   # this leg IS the CPU baseline; the GPU numbers beside it come from the product path

    frame_s, e, _g = fx.load_fixture("point_intersect_png")
    labels_s = [(300, 594, 433), (900, 594, 465)]               # the sample's scale labels as OCR would deliver them
    frame_y, info_y = synth.make_frame(1920, 1080, 0, n_lines=args.lines)
    cases = [("point_intersect.png (vision-common/samples, 2560x1440)", frame_s, labels_s, min(a[2] for a in labels_s)),
             ("synthetic 1920x1080 (frame 0 of the bench generator)", frame_y, info_y["anchors"], info_y["scales_start_y"])]
    have_gpu = torch.cuda.is_available()
    vision = smh.HipVision.init(0) if have_gpu else None
    state = smh.VisionState(lazy_map=False) if have_gpu else None   # (the trait-shaped sequence: crop_to_map returns the image)
    out_cases = []
    for name, frame, labels, start_y in cases:
        H, W = frame.shape[:2]
        info = dict(scales_start_y=start_y)
        split = cpu_stage_split(orc, [frame] * 5, [info] * 5)
        prev = omp_threads(1)
        for _ in range(2):
            ref = orc.process_frame(frame, stages=0xF, anchors=labels, scales_start_y=start_y)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):                                   # the whole call: + isolate_map_markers' own pass and the per-call buffers
            ref = orc.process_frame(frame, stages=0xF, anchors=labels, scales_start_y=start_y)
        cpu_ms = (time.perf_counter() - t0) / reps * 1e3
        if prev:
            omp_threads(prev)
        threads = cpu_threads(64)
        k = max(threads * 2, 8)
        batch = np.ascontiguousarray(np.broadcast_to(frame, (k,) + frame.shape))
        a = np.zeros((k, 3, 3), np.uint32)
        for j, s_ in enumerate(labels[:3]):
            a[:, j] = s_
        orc.process_batch(batch[:threads], threads, stages=0xF, anchors=a[:threads], n_anchors=len(labels), scales_start_y=start_y)
        t0 = time.perf_counter()
        orc.process_batch(batch, threads, stages=0xF, anchors=a, n_anchors=len(labels), scales_start_y=start_y)
        all_core = k / (time.perf_counter() - t0)
        c = {"frame": name, "size": [W, H], "rounds": int(ref["rounds"]), "lines": int(ref["n_lines"]), "ray_steps": int(ref["steps"]),
             "cpu_single_thread_ms_per_frame": cpu_ms, "cpu_single_thread_stage_ms": split,
             "cpu_all_core_frames_per_s": all_core, "cpu_threads": threads}
        if have_gpu:
            for _ in range(3):
                res = state.process(vision, frame, ocr_labels=labels)
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                res = state.process(vision, frame, ocr_labels=labels)
            c["gpu_trait_path_ms_per_frame"] = (time.perf_counter() - t0) / n * 1e3
            c["gpu_lines_equal_cpu"] = bool(res is not None and np.array_equal(res.markers, ref["lines"]) and res.meters_to_px_ratio == ref["mpx"])
        out_cases.append(c)
    first = out_cases[0]
    out = {"metric": "map frames/sec (single 2560x1440 sample screenshot, full CV pipeline), CPU path", "value": 1e3 / first["cpu_single_thread_ms_per_frame"],
           "unit": "frames/s", "n_gpus": 0, "steps": 5, "warmup": 2, "ms_per_step": first["cpu_single_thread_ms_per_frame"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32", "data": "vision-common/samples/point_intersect.png (committed fixture) + synthetic",
           "config": {"workload": CONFIGS[0]["name"], "baseline_config": 0, "path": "C oracle (port of vision-cpu: gcc -O2 -ffp-contract=off), one thread; "
                      "the Rust original cannot be built in this image"},
           "cases": out_cases,
           "cpu_baseline": {"value": first["cpu_all_core_frames_per_s"], "unit": "frames/s", "cores": first["cpu_threads"], "kind": "port", "cpu": cpu_model(),
                            "cpu_quota_cores": cpu_quota_cores(), "sample": "copies of the sample frame spread over the host threads"}}
    if vision is not None:
        vision.shutdown()
    emit(out)


def node_main(args, cfg, n, W, H, stages, rounds, custom):
    """--node: ONE process, every device through smhv_node_run + smhv_node_gather (the C ABI's multi-GPU form).
    Each pass = one asynchronous run on every device + one gather of all records to devices[0] (synchronous: the gather
    ends the pass, so each device has one batch in flight -- compare with value_depth1 of the one-process-per-GPU leg)."""
    import numpy as np
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    from squad_mortar_helper_amd.node import Node
    devs = list(range(args.gpus)) if args.force_device is None else [args.force_device]
    if args.force_device is not None and args.gpus > 1:
        print("bench.py --node: a communicator cannot hold one device twice; --force-device runs a world of one", file=sys.stderr)
    G = len(devs)
    frames, anchors, infos0 = [], [], None
    for i, dv in enumerate(devs):
        d, infos, _, _ = upload_synthetic(torch, synth, W, H, n, i * n, args.lines, args.distinct, torch.device("cuda", dv), 0)
        frames.append(d)
        anchors.append(smh.make_anchors([(x["scales_start_y"], x["anchors"]) for x in infos]) if stages & 0x8 else None)
        infos0 = infos0 or infos
    node = Node(devs, W, H, n, depth=2)
    ptrs = [d.data_ptr() for d in frames]
    counts = [n] * G

    def one_pass():
        node.run(ptrs, counts, stages=stages, anchors=anchors if stages & 0x8 else None)
        return node.gather()

    def step():
        for _ in range(rounds):
            one_pass()

    for _ in range(args.warmup):
        step()
    dts = []
    for k in sub_regions(args.steps, args.timed_regions):
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        dts.append((time.perf_counter() - t0, k))
    recs, tot = one_pass()
    assert tot == n * G, "node gather returned %d records, expected %d" % (tot, n * G)
    per = smh.results_to_dicts(recs[:tot])
    rates = [n * G * rounds * k / dt for dt, k in dts]
    value = median(rates)
    out = {"metric": "map frames/sec (%dx%d %s), whole job" % (W, H, "full CV pipeline" if stages == 0xF else "stages 0x%x" % stages),
           "value": value, "unit": "frames/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": n * G * rounds / value * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8/f32", "data": "synthetic",
           "value_min": min(rates), "value_max": max(rates), "value_whole_region": n * G * rounds * args.steps / sum(dt for dt, _ in dts),
           "config": {"workload": (cfg["name"] if not custom else "custom: %d x %dx%d frames, stages 0x%x" % (n, W, H, stages)) +
                                  "; ONE process, smhv_node_run + smhv_node_gather (ncclCommInitAll, one grouped ncclGather per pass)",
                      "baseline_config": args.config, "frames_per_gpu": n, "global_batch": n * G, "frame": [W, H], "stages": stages,
                      "passes_per_step": rounds, "frames_per_step": n * G * rounds, "parallelism": "frames block-sharded over %d devices, one process" % G,
                      "schedule": "smhv_node (one pass in flight per device, records gathered after every pass)"},
           "per_gpu_frames_per_s": value / G, "all_map_open": bool(all(r["map_open"] for r in per)),
           "lsd": {"rounds_per_frame": float(np.mean([r["rounds"] for r in per])), "lines_per_frame": float(np.mean([r["n_lines"] for r in per]))}}
    node.close()
    emit(out)


def traffic_child(args, n, W, H, stages):
    """What the traffic probe profiles: three plain passes over n resident frames (16 distinct, tiled), nothing else."""
    import numpy as np
    import torch
    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import synth
    k = min(16, n)
    frames, infos = synth.make_batch(W, H, k, first_idx=0, n_lines=args.lines)
    d = torch.from_numpy(frames).cuda()
    d = d.repeat((n + k - 1) // k, 1, 1, 1)[:n].contiguous()
    infos = [infos[i % k] for i in range(n)]
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos]) if stages & 8 else None
    vision = smh.HipVision.init(0)
    fb = smh.FrameBatch(vision, W, H, n)
    for _ in range(3):
        fb.run(d.data_ptr(), n, stages=stages, grayscale=True, max_gap=15, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fb.close()


def traffic_probe(args, n, W, H, stages, kname, timeout=150):
    """HBM bytes of the streaming kernel per launch, measured NOW on this box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    separate passes (nothing else combined with the counters but --kernel-trace: MI355X_MICROARCH.md, HBM / rocprofv3 section), each
    around a child `python3 bench.py --traffic-child` that runs three plain passes of this workload; reads = 2 x FETCH_SIZE (the
    guide's gfx950 correction for wide coalesced reads), units KB.  -> (bytes per launch, source text) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    # bench.py itself under a profiler (tools/profile_round.sh): a nested rocprofv3 would inherit the outer one's preloaded library, which
    # initialises the GPU in the launcher before it starts the child -- not attempted
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "bench.py is itself running under rocprofv3: no nested counter run"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="smh_traffic_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--traffic-child",
                   "--config", str(args.config), "--frames-per-gpu", str(n), "--width", str(W), "--height", str(H), "--stages", str(stages), "--lines", str(args.lines)]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout, text=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and kname in row.get("Kernel_Name", ""):
                            vals.append(float(row["Counter_Value"]))
            vals = [v for v in vals if v > 0.0]              # (a dispatch occasionally comes back with a counter of exactly 0: a dropped sample)
            if not vals:
                return None, "rocprofv3 --pmc %s gave no sample of %s (exit %d: %s)" % (counter, kname, r.returncode, (r.stderr or "")[-200:].replace("\n", " "))
            got[counter] = sum(vals) / len(vals) * 1024.0
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            return None, "traffic probe failed: %s" % e
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = 2.0 * got["FETCH_SIZE"], got["WRITE_SIZE"]
    return rd + wr, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around three plain passes of the workload; "
                     "reads = 2 x FETCH_SIZE (gfx950 correction) = %.1f MB, writes = %.1f MB per launch" % (rd / 1e6, wr / 1e6))


def main():
    args = parse_args()
    if args.config is None:
        args.config = 4 if args.gpus > 1 else 2
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.node and args.config != 0:
        sys.exit(self_launch(args))                    # before anything has touched the GPU
    if args.node:
        world = 1
    elif world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    cfg = CONFIGS[args.config]
    n = args.frames_per_gpu or cfg["frames"]
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    stages = cfg["stages"] if args.stages is None else args.stages
    rounds = args.rounds_per_step or cfg["rounds"]
    custom = (n, W, H, stages) != (cfg["frames"], cfg["width"], cfg["height"], cfg["stages"])
    if args.distinct is None:
        args.distinct = 256 if args.config == 4 else 0

    if args.traffic_child:
        return traffic_child(args, n, W, H, stages)
    if args.rendezvous_only:                           # the launch / rendezvous path alone (CPU test of the self-launch)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "rank_sum": float(t.item()), "baseline_config": args.config,
                              "frames_per_gpu": n, "global_batch": n * world}))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.config == 0:
        return config0(args)
    if args.node:
        return node_main(args, cfg, n, W, H, stages, rounds, custom)

    import numpy as np
    import torch

    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    nccl = args.dist_backend == "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if nccl:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    import squad_mortar_helper_amd as smh
    from squad_mortar_helper_amd import dist as sdist
    from squad_mortar_helper_amd import synth

    first = rank * n                                   # block shard of the global batch
    keep = max(args.cpu_sample, 8) if (rank == 0 and world == 1) else 8
    frames, infos, frames_host, h2d_s = upload_synthetic(torch, synth, W, H, n, first, args.lines, args.distinct, torch.device("cuda", local_rank), keep)

    vision = smh.HipVision.init(local_rank)
    # default: 12 batches in flight (3072 frames at 256 per batch = three per resident wave of the search service, from where on
    # its rate no longer grows; a pipeline that picks the batch-granular search has twelve hardware queues
    # for its slot streams), 8 for 1024-frame shards (config 4) -- each slot holds the output images of a whole batch, 5 MB
    # per 1080p frame
    depth = args.pipeline_depth if args.pipeline_depth > 0 else (8 if n >= 1024 else 12)
    idle_streams = [torch.cuda.Stream() for _ in range(max(0, args.idle_streams))]   # noqa: F841 (kept alive on purpose)
    if args.tile_cap > 0:
        smh._lib.load().smhv_debug_lsd_tile_cap(args.tile_cap)
    room = {"auto": 1 if (world > 1 and nccl) else 0, "on": 1, "off": 2}[args.room]
    pipe = smh.Pipeline(vision, W, H, n, depth, search=args.search, room_for_others=room)
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos]) if stages & 0x8 else None
    fptr = frames.data_ptr()

    # N > 1: the records of every pass are gathered to rank 0 on the slot's own stream (RCCL orders itself after the
    # work already enqueued there); all buffers are allocated once, up front.
    gather = None
    if world > 1:
        gather = sdist.RecordGather(dist, n, world, rank, device=("cuda" if nccl else "cpu"), slots=max(8, depth))
        rec_views = [sdist.device_records_view(b.device_ptrs()["results"], n) for b in pipe.slots]

    # A pass's records are complete when its slowest frame is (the library's line search is frame-granular: completion is
    # a host-visible counter, not a point on a stream), so the gather of pass k is issued when its slot comes round again
    # -- or at the next barrier -- behind a host wait for that slot; `hold` orders the slot's next pass behind the gather.
    pending = [False] * depth
    gather_stream = torch.cuda.Stream() if gather is not None else None
    submitted = [0]

    def flush_gather(slot):
        pipe.wait(slot)
        with torch.cuda.stream(gather_stream):
            gather.run(rec_views[slot] if nccl else rec_views[slot].cpu(), slot)
        pipe.hold(slot, gather_stream.cuda_stream)      # the gather reads the slot's records: its next pass waits for it
        pending[slot] = False

    def one_pass(p):
        if gather is not None and p is pipe:
            nxt = submitted[0] % depth
            if pending[nxt]:
                flush_gather(nxt)
        slot = p.submit(fptr, n, stages=stages, grayscale=True, max_gap=15, anchors=anchors)
        if gather is not None and p is pipe:
            submitted[0] += 1
            pending[slot] = True
        return slot

    def step(p=pipe):
        for _ in range(rounds):
            one_pass(p)

    def barrier():
        if gather is not None:
            for s_ in range(depth):
                k_ = (submitted[0] + s_) % depth           # oldest first
                if pending[k_]:
                    flush_gather(k_)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()           # all streams of the device

    def timed(p, steps):
        """`steps` steps as --timed-regions regions (default one), each bracketed by barrier + synchronize on both sides -> per-region
        (seconds MAX over ranks, steps)."""
        parts = sub_regions(steps, args.timed_regions)
        barrier()
        dts = []
        for k in parts:
            t0 = time.perf_counter()
            for _ in range(k):
                step(p)
            barrier()
            dts.append(time.perf_counter() - t0)
        t = torch.tensor(dts, dtype=torch.float64, device="cuda" if nccl else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [(float(x), k) for x, k in zip(t.tolist(), parts)]

    for _ in range(args.warmup):
        step()
    # a pipeline with SMHV_SEARCH_AUTO at depth >= 6 times both of its line searches on the workload before it settles on one
    # (24 x depth submissions, smh_runtime.cpp mode_control): that belongs to the warm-up, whatever --warmup says.  The number
    # of extra steps is computed, not polled: every rank must run the same number of passes (each gathers).
    extra_warmup = 0
    if depth >= 6 and args.search == "auto":
        extra_warmup = max(0, -(-(26 * depth - args.warmup * rounds) // rounds))
        for _ in range(extra_warmup):
            step()
        if extra_warmup and rank == 0:
            print("bench.py: %d extra warm-up step(s): the pipeline measures both line searches on the workload before it settles" % extra_warmup, file=sys.stderr)
    barrier()
    if not args.no_stage_timing:
        for b in pipe.slots:
            b.enable_timing(True)
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        watch = DeviceWatch("%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id) if hasattr(pr, "pci_bus_id") else None)
    except Exception:  # noqa: BLE001  (diagnostic only)
        watch = None
    if watch is not None and watch.files:
        with watch:
            regions = timed(pipe, args.steps)
    else:
        regions = timed(pipe, args.steps)
    stages_ms = None
    if not args.no_stage_timing:
        first = (args.warmup + extra_warmup) * rounds            # submissions before the timed region: slot = submission % depth
        per = [pipe.slots[(first + i) % depth].stage_ms() for i in range(min(depth, args.steps * rounds))]
        stages_ms = {k: float(np.mean([p[k] for p in per])) for k in per[0]}
        for b in pipe.slots:
            b.enable_timing(False)
    frames_per_step = n * world * rounds
    rates = [frames_per_step * k / dt for dt, k in regions]
    value = median(rates)
    dt_total = sum(dt for dt, _ in regions)

    svc_stats = pipe.search_stats()                     # the frame-granular line search's own counters (None: batch-granular search)
    svc_geo = pipe.peek() if svc_stats else None
    # ---- every slot of the pipeline must hold the same records (same frames, same stages): byte for byte ----
    used = min(depth, args.steps * rounds + args.warmup * rounds)
    slot_bytes = [bytes(pipe.slots[s].read_results(0, n)) for s in range(used)]
    slots_identical = all(sb == slot_bytes[0] for sb in slot_bytes)
    if not slots_identical:
        bad = [s for s in range(used) if slot_bytes[s] != slot_bytes[0]]
        raise SystemExit("bench.py: pipeline slots %s hold records that differ from slot 0's (same frames, same stages)" % bad)

    # ---- the same workload with ONE batch in flight, and one pass alone for the per-stage durations in isolation ----
    value_d1 = ms_d1 = iso_ms = d1_minmax = b2b_ms = None
    if not args.no_depth1 or not args.no_stage_timing:
        pipe1 = smh.Pipeline(vision, W, H, n, 1)
        if not args.no_depth1:
            k1 = max(2, args.steps // 2)
            for _ in range(2):
                step(pipe1)
            r1 = timed(pipe1, k1)
            rates1 = [frames_per_step * k / dt for dt, k in r1]
            value_d1 = median(rates1)
            d1_minmax = [min(rates1), max(rates1)]
            ms_d1 = frames_per_step / value_d1 * 1e3
            assert bytes(pipe1.slots[0].read_results(0, n)) == slot_bytes[0], "the depth-1 pipeline's records differ from the depth-%d pipeline's" % depth
        if not args.no_stage_timing:
            # one pass that runs ALONE and whole (a plain smhv_batch_run: the depth-1 pipeline cuts its submissions into
            # chunks, whose launches overlap) for the per-stage durations in isolation
            fb_iso = smh.FrameBatch(vision, W, H, n)
            fb_iso.enable_timing(True)
            # (four untimed launches first -- the first of a process loads the code object and touches cold page tables --, then the mean of
            # 24, each alone on the chip: single launches differ by +-8 % on one box, profiles/README.md)
            for k in range(4 + 24):
                fb_iso.run(fptr, n, stages=stages, grayscale=True, max_gap=15, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                if k == 3:
                    fb_iso.stage_ms()                      # (reading the stage times starts a new average)
            iso_ms = fb_iso.stage_ms()
            assert bytes(fb_iso.read_results(0, n)) == slot_bytes[0], "a plain smhv_batch_run's records differ from the pipeline's"
            fb_iso.close()
            if (stages & 0xC) and (stages & 0x3) and not args.no_back_to_back:
                # The streaming pass back to back with itself: plain runs WITHOUT the line search (debug knob) on four
                # streams, so that one launch's tail overlaps the next one's head -- the kernel's steady rate, which is what
                # the device-copy calibration below measures for a copy (ten copies back to back).  One launch alone pays a
                # fixed 0.08-0.11 ms on top ("the fixed cost of a streaming-pass launch").
                lib = smh._lib.load()
                fbs = [smh.FrameBatch(vision, W, H, n) for _ in range(4)]
                sts = [torch.cuda.Stream() for _ in fbs]
                try:
                    smh._lib.check(lib.smhv_debug_skip_line_search(1))
                    reps = 8
                    for timed_round in (False, True):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(reps if timed_round else 2):
                            for fbk, stk in zip(fbs, sts):
                                fbk.run(fptr, n, stages=stages, grayscale=True, max_gap=15, anchors=anchors, stream=stk.cuda_stream)
                        torch.cuda.synchronize()
                        b2b_ms = (time.perf_counter() - t0) * 1e3 / (reps * len(fbs))
                finally:
                    smh._lib.check(lib.smhv_debug_skip_line_search(0))
                    for fbk in fbs:
                        fbk.close()
        pipe1.close()

    # ---- result sanity + workload statistics (outside the timed region) ----
    fb = pipe.slots[0]
    recs = smh.results_to_dicts(fb.read_results(0, n))
    rounds_pf = float(np.mean([r["rounds"] for r in recs]))
    ray_steps = float(np.mean([r["ray_steps"] for r in recs]))
    n_lines = float(np.mean([r["n_lines"] for r in recs]))
    all_open = all(r["map_open"] for r in recs)
    gather_ok = None
    if gather is not None and rank == 0:
        last_slot = (args.warmup * rounds + args.steps * rounds - 1) % depth
        got = gather.records(last_slot)
        assert len(got) == n * world, "gather returned %d records, expected %d" % (len(got), n * world)
        sz = sdist.RECORD_BYTES
        gather_ok = bytes(got)[:n * sz] == slot_bytes[0]
        assert gather_ok, "rank 0's own block of the gather differs from its records"

    if world > 1:
        # every rank empties its C stdio buffer (RCCL's banner) NOW, so that nothing of the other ranks can land behind
        # rank 0's JSON line when they exit
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        sys.stdout.flush()
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    x, y, rw, rh = fb.roi
    bw, bh = fb.layout.button[2], fb.layout.button[3]
    kernel_bytes, full_bytes = algorithmic_bytes(rw, rh, bw, bh, stages)
    what = "full CV pipeline" if stages == 0xF else ("marker threshold + LSD" if stages == 0x1 else "stages 0x%x" % stages)
    out = {
        "metric": "map frames/sec (%dx%d %s), whole job" % (W, H, what),
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": frames_per_step / value * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/f32", "data": "synthetic",
        "config": {"workload": (cfg["name"] if not custom else "custom: %d x %dx%d frames, stages 0x%x" % (n, W, H, stages)) +
                               (", RCCL gather of result records" if world > 1 and args.config != 4 else ""),
                   "baseline_config": args.config, "frames_per_gpu": n, "global_batch": n * world, "frame": [W, H], "stages": stages,
                   "distinct_frames_per_gpu": (n if args.distinct <= 0 else min(args.distinct, n)),
                   "passes_per_step": rounds, "frames_per_step": frames_per_step,
                   "marker_lines_per_frame": args.lines, "parallelism": "frames block-sharded, dp%d" % world,
                   "pipeline_depth": depth, "room_for_others": bool(room == 1),
                   "service_workgroups": (svc_geo["service_workgroups"] if svc_geo else None), "service_waves_per_workgroup": (svc_geo["waves_per_workgroup"] if svc_geo else None),
                   "schedule": ("smhv_pipeline, frame-granular line search (k_lsd_service: one long-lived kernel pulls (slot, frame) items from a device ring)"
                                if svc_stats and svc_stats.get("mode") == "frame-granular" else "smhv_pipeline, batch-granular line search (one launch per batch)") +
                               ("; chosen by the pipeline's own measurement of both on this workload" if svc_stats and svc_stats.get("adaptive") else "")},
        "value_is": ("the %d timed steps as ONE region bracketed by barrier + synchronize" % args.steps if len(regions) == 1 else
                     "median of %d sub-regions of the %d timed steps (each bracketed by barrier + synchronize)" % (len(regions), args.steps)),
        "value_min": min(rates), "value_max": max(rates), "value_whole_region": frames_per_step * args.steps / dt_total,
        "timed_seconds": dt_total,
        "ms_per_pass": frames_per_step / value * 1e3 / rounds,
        "per_gpu_frames_per_s": value / world,
        "value_depth1": value_d1, "value_depth1_min_max": d1_minmax, "ms_per_pass_depth1": (ms_d1 / rounds if ms_d1 is not None else None),
        "h2d_seconds_for_batch": h2d_s,
        "search_service": svc_stats,
        "device_during_timed_region": (watch.summary() if watch is not None else None),
        "all_map_open": bool(all_open),
        "slots_identical": bool(slots_identical), "slots_compared": used,
    }
    if gather_ok is not None:
        out["gather_matches_rank0_records"] = bool(gather_ok)
    if stages_ms is not None:
        t_map = stages_ms["map_pass"] * 1e-3
        ach = n * kernel_bytes / t_map / 1e9 if t_map > 0 else 0.0
        traffic, tsrc = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
            if tj.get("frame") == [W, H] and stages == tj.get("stages", 0xF):
                traffic = tj["bytes_per_frame"] * n
                tsrc = "profiles/traffic.json (committed rocprofv3 PMC run %s: FETCH_SIZE x2 + WRITE_SIZE of the kernel, gfx950 correction)" % tj.get("run", "")
        except (OSError, ValueError, KeyError):
            pass
        kname = "k_map_brq_pass" if (stages & 0xC) and (stages & 0x3) else "k_map_pass"
        if world == 1 and not args.no_traffic_probe:
            live, why = traffic_probe(args, n, W, H, stages, kname)
            if live is not None:
                committed = traffic
                traffic, tsrc = live, why + ("; the committed PMC run (profiles/traffic.json) has %.1f MB" % (committed / 1e6) if committed else "")
            else:
                tsrc = (tsrc or "none") + " -- live probe: " + why
        # `achieved` / `frac`: the kernel's algorithmic bytes per launch / its launch duration -- of a launch that runs ALONE when the
        # isolated pass was timed (hipEvents on the launch's stream, live, after the timed region), which is the figure that
        # says something about the kernel.  The launches INSIDE the timed region of a deep pipeline overlap each other and the
        # search (their per-launch duration is a diagnostic: `in_pipeline`); what the whole pipeline moves per second is
        # `pipeline_hbm_frac`.
        out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                           "algorithmic_bytes_per_frame": kernel_bytes, "launch_ms": stages_ms["map_pass"],
                           "launch_is": "inside the timed region (launches of several batches overlap)",
                           "in_pipeline": {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "launch_ms": stages_ms["map_pass"],
                                           "note": "per-launch figure of OVERLAPPING launches (two streaming launches and the search run side by side "
                                                   "most of the time): a launch takes longer than the kernel needs while the pipeline as a whole moves "
                                                   "more bytes per second -- a diagnostic, not a roofline fraction"}}
        if iso_ms is not None and iso_ms["map_pass"] > 0:
            a2 = n * kernel_bytes / (iso_ms["map_pass"] * 1e-3) / 1e9
            out["roofline_isolated"] = {"kernel": kname, "achieved": a2, "frac": a2 / HBM_PEAK_GBS, "unit": "GB/s",
                                        "launch_ms": iso_ms["map_pass"], "stages_ms": iso_ms}
            out["roofline"].update({"achieved": a2, "frac": a2 / HBM_PEAK_GBS, "launch_ms": iso_ms["map_pass"],
                                    "launch_is": "a launch that runs alone (plain smhv_batch_run after the timed region; hipEvents on its stream; mean of 24 launches after 4 untimed ones)"})
            if b2b_ms is not None:
                a3 = n * kernel_bytes / (b2b_ms * 1e-3) / 1e9
                out["roofline_isolated"]["back_to_back"] = {
                    "ms_per_pass": b2b_ms, "achieved": a3, "frac": a3 / HBM_PEAK_GBS, "unit": "GB/s",
                    "what": "button + streaming pass + record kernel of plain runs on four streams, line search skipped "
                            "(smhv_debug_skip_line_search): wall time per pass; algorithmic bytes of the streaming pass only"}
            # calibration on THIS box (they differ by 10 %): a plain device-to-device copy moving the same number of bytes
            # (half read, half written); outside every timed region
            try:
                half = int(min(n, 256) * kernel_bytes // 2)
                ca = torch.empty(half, dtype=torch.uint8, device="cuda"); cb = torch.empty_like(ca)
                for _ in range(3):
                    cb.copy_(ca)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10):
                    cb.copy_(ca)
                e1.record(); torch.cuda.synchronize()
                copy_gbs = 2 * half / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
                out["roofline_isolated"]["device_copy_GBps"] = copy_gbs
                out["roofline_isolated"]["frac_of_device_copy"] = a2 / copy_gbs
                out["roofline"]["frac_isolated"] = a2 / HBM_PEAK_GBS
                out["roofline"]["frac_isolated_of_device_copy"] = a2 / copy_gbs
                del ca, cb
            except RuntimeError:
                pass
            # ... and a hand-written copy with the PASS'S OWN access pattern (smhv_debug_pattern_copy: every ROI quad loaded once
            # out of the full-width frame rows, ui / mask / ocr / scales rows stored with the pass's widths and pitches, no
            # arithmetic): what the memory system gives this pattern, alone and back to back on four streams like the pass
            if (stages & 0xC) and (stages & 0x3):
                lib = smh._lib.load()
                pc = [smh.FrameBatch(vision, W, H, n) for _ in range(4)]
                pst = [torch.cuda.Stream() for _ in pc]
                pat_bytes = n * (2 * rw * rh * 4 + rw * rh + 2 * (rw // 2) * (rh // 2))       # algorithmic bytes of the copy (no halo rows)
                try:
                    best = None
                    variants = {}
                    for rows in (4, 8, 12):                          # loads in flight per thread (the pass itself: 12)
                        for fbk, stk in zip(pc, pst):
                            smh._lib.check(lib.smhv_debug_pattern_copy(fbk._b, fptr, n, rows, stk.cuda_stream))
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        with torch.cuda.stream(pst[0]):
                            e0.record()
                            for _ in range(10):
                                smh._lib.check(lib.smhv_debug_pattern_copy(pc[0]._b, fptr, n, rows, pst[0].cuda_stream))
                            e1.record()
                        torch.cuda.synchronize()
                        a_ms = e0.elapsed_time(e1) / 10
                        t0 = time.perf_counter()
                        for _ in range(8):
                            for fbk, stk in zip(pc, pst):
                                smh._lib.check(lib.smhv_debug_pattern_copy(fbk._b, fptr, n, rows, stk.cuda_stream))
                        torch.cuda.synchronize()
                        bb = (time.perf_counter() - t0) * 1e3 / 32
                        variants[str(rows)] = {"alone_GBps": pat_bytes / (a_ms * 1e-3) / 1e9, "back_to_back_GBps": pat_bytes / (bb * 1e-3) / 1e9}
                        if best is None or a_ms < best[0]:
                            best = (a_ms, rows)
                    alone_ms = best[0]
                    b2b = pat_bytes / max(v["back_to_back_GBps"] for v in variants.values()) / 1e9 * 1e3
                    ri = out["roofline_isolated"]
                    ri["pattern_copy_GBps"] = pat_bytes / (alone_ms * 1e-3) / 1e9
                    ri["pattern_copy_ms"] = alone_ms
                    ri["pattern_copy_back_to_back_GBps"] = pat_bytes / (b2b * 1e-3) / 1e9
                    ri["pattern_copy_bytes"] = pat_bytes
                    ri["pattern_copy_variants"] = variants
                    ri["frac_of_pattern_copy"] = (n * kernel_bytes / (iso_ms["map_pass"] * 1e-3) / 1e9) / ri["pattern_copy_GBps"]
                    if b2b_ms is not None:
                        ri["back_to_back"]["frac_of_pattern_copy_back_to_back"] = (n * kernel_bytes / (b2b_ms * 1e-3) / 1e9) / ri["pattern_copy_back_to_back_GBps"]
                    ri["pattern_copy_what"] = ("k_pattern_copy, the best of 4 / 8 / 12 rows in flight per thread (pattern_copy_variants; the pass holds 12): one launch "
                                               "after the other on one stream (hipEvents) / four streams back to back (wall clock); its bytes are the pass's "
                                               "algorithmic bytes (no halo rows); same band height and work-item order as the pass.  A plain copy loop, not a bound: "
                                               "since the pass took 24-row bands in band-major order for launches that run alone it is up to 10 % FASTER than this copy")
                finally:
                    for fbk in pc:
                        fbk.close()
        out["stages_ms"] = stages_ms
        out["pipeline_algorithmic_GBps"] = value / world * full_bytes / 1e9
        out["pipeline_hbm_frac"] = value / world * full_bytes / 1e9 / HBM_PEAK_GBS
        out["roofline"]["pipeline_frac"] = out["pipeline_hbm_frac"]   # algorithmic bytes of the WHOLE pipeline per second / peak: the aggregate that means something at depth > 1
        # the same with the MEASURED bytes of a whole pass (every kernel; committed PMC run) and against what this box copies at
        try:
            if tj.get("frame") == [W, H] and stages == tj.get("stages", 0xF) and tj.get("pass_bytes") and tj.get("frames"):
                out["pipeline_traffic_GBps"] = value / world * tj["pass_bytes"] / tj["frames"] / 1e9
                copy = out.get("roofline_isolated", {}).get("device_copy_GBps")
                if copy:
                    out["pipeline_traffic_frac_of_device_copy"] = out["pipeline_traffic_GBps"] / copy
        except NameError:
            pass
        t_lsd = stages_ms["lsd"] * 1e-3
        out["lsd"] = {"rounds_per_frame": rounds_pf, "ray_steps_per_frame": ray_steps, "lines_per_frame": n_lines,
                      "ray_steps_per_s": (n * ray_steps / t_lsd) if t_lsd > 0 else None,
                      "time_share": stages_ms["lsd"] / max(sum(stages_ms.values()), 1e-9)}

    if args.ingest_frames > 0 and world == 1 and args.config == 2 and not custom:
        out["ingest"] = ingest_leg(smh, torch, vision, pipe, frames_host, anchors, stages, args.ingest_frames, W, H, n, affinity=args.ingest_affinity == "on")
    pipe.close()                                           # (its streams hold hardware queues the next leg's pipelines should get)
    if world == 1 and args.config == 2 and not custom and len(frames_host):
        out["trait_path"] = trait_path_leg(smh, vision, frames_host[0], infos[0]["anchors"])
    if args.side_probe > 0 and world == 1 and args.config == 2 and not custom and depth >= 3:
        out["co_residency"] = co_residency_leg(smh, torch, vision, W, H, n, depth, fptr, anchors, stages, args.side_probe, rccl_probe=args.gather_probe)
    if not args.no_real_samples and world == 1 and args.config == 2 and not custom:
        out["real_samples"] = real_samples_leg(smh, torch, vision, depth)

    if args.cpu_sample > 0 and world == 1:
        from oracle import oracle as orc   # CPU baseline leg only (checker, never the product path)
        k = min(args.cpu_sample, n, len(frames_host)) if n > 1 else min(args.cpu_sample, 16)
        if n == 1:                                         # config 1: more frames of the same kind for a stable figure
            sub, sinfo = synth.make_batch(W, H, k, first_idx=first, n_lines=args.lines)
        else:
            sub, sinfo = frames_host[:k], infos[:k]
        cores = cpu_threads(k)
        a = np.zeros((k, 3, 3), np.uint32)
        for i in range(k):
            for j, s_ in enumerate(sinfo[i]["anchors"][:3]):
                a[i, j] = s_
        kw = dict(stages=stages, anchors=a, n_anchors=len(sinfo[0]["anchors"]), scales_start_y=sinfo[0]["scales_start_y"])
        orc.process_batch(sub[:min(k, cores)], cores, stages=stages, anchors=a[:min(k, cores)], n_anchors=kw["n_anchors"], scales_start_y=kw["scales_start_y"])   # warm-up
        t0 = time.perf_counter()
        res = orc.process_batch(sub, cores, **kw)
        cdt = time.perf_counter() - t0
        same = n == 1 or all(records_equal_oracle(np, recs[i], res[i]) for i in range(k))
        split = cpu_stage_split(orc, sub[:8], sinfo[:8])
        out["cpu_baseline"] = {"value": k / cdt, "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
                               "cpu_quota_cores": cpu_quota_cores(),
                               "single_thread_stage_ms": split, "single_thread_frames_per_s": 1e3 / max(sum(split.values()), 1e-9),
                               "gpu_records_equal_cpu": bool(same),
                               "sample": "%d frames of rank 0's workload, same stages, C oracle (gcc -O2, -ffp-contract=off), frames parallel across "
                                         "%d threads; lines (bit-exact), round and mask-pixel counts and m/px of these frames match the GPU records: %s; "
                                         "stage split: median of 8 frames on one thread after an untimed pass" % (k, cores, same)}
        if not same:
            emit(out)
            raise SystemExit("bench.py: GPU records differ from the CPU oracle on the sampled frames")
    if world > 1:
        dist.destroy_process_group()
    emit(out)


if __name__ == "__main__":
    main()
