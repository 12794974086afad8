// smh_runtime.cpp -- host runtime behind the C ABI of include/smh_vision_hip.h.
//
// Plays the role of the reference's Rust-CUDA host side (vision-gpu/src/{lib,cuda,gpuimage}.rs):
// device context, per-frame-size buffer set (GpuMemory, vision-gpu/src/lib.rs:33-104), streams for
// the two concurrent branches of VisionState::process (src/vision/mod.rs:219-223), pinned staging
// for the images the host-side OCR / scale scan consume, and error reporting as status codes.
// It is written directly on the HIP runtime API; there is no CPU fallback in this library: if the
// device or the gfx950 code object is missing every entry point fails with an error code and the
// caller does what the reference does on plugin failure (falls back to its own CPU back-end,
// src/vision/hardware.rs:73-76).
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sched.h>
#include <pthread.h>
#include <cctype>
#include <mutex>
#include <vector>
#include <new>
#include <string>

#include "smh_consts.h"
#include "smh_kernels.h"

using namespace smh;

// ------------------------------------------------------------------------------------------------
// errors / logging
// ------------------------------------------------------------------------------------------------
static thread_local std::string t_last_error;

static int fail(int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	t_last_error = buf;
	return code;
}

// (for the other host translation units of the library)
extern "C" int smhv_internal_fail(int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	t_last_error = buf;
	return code;
}

#define HIPCHK(expr)                                                                                          \
	do {                                                                                                      \
		hipError_t _e = (expr);                                                                               \
		if (_e != hipSuccess) return fail(SMHV_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
	} while (0)

// Waits of the pipelined paths.  hipEventSynchronize / hipStreamSynchronize may put the thread to sleep on an interrupt --
// whether they do depends on process-wide state the library does not own (measured: a pipeline created before anything else
// had used the device waited that way and ran the sample screenshots at 107 k frames/s; the same pipeline created after one
// unrelated 16-byte torch copy ran at 203 k, depth 4, batch-granular search).  A slot of a pipeline is due within a
// millisecond, so these waits poll first (hipEventQuery / hipStreamQuery, ~1 us per look, yielding the core between looks)
// and fall back to the blocking call only after SMH_SPIN_WAIT_US.
#define SMH_SPIN_WAIT_US 20000
static hipError_t wait_event(hipEvent_t ev) {
	hipError_t e = hipEventQuery(ev);
	if (e != hipErrorNotReady) return e;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (uint32_t k = 0;; ++k) {
		e = hipEventQuery(ev);
		if (e != hipErrorNotReady) return e;
		if ((k & 15u) == 15u) {
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if ((t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000 > SMH_SPIN_WAIT_US) break;
			sched_yield();
		}
	}
	return hipEventSynchronize(ev);
}
static hipError_t wait_stream(hipStream_t st) {
	hipError_t e = hipStreamQuery(st);
	if (e != hipErrorNotReady) return e;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (uint32_t k = 0;; ++k) {
		e = hipStreamQuery(st);
		if (e != hipErrorNotReady) return e;
		if ((k & 15u) == 15u) {
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if ((t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000 > SMH_SPIN_WAIT_US) break;
			sched_yield();
		}
	}
	return hipStreamSynchronize(st);
}

extern "C" SMHV_API const char *smhv_last_error(void) { return t_last_error.c_str(); }

// ------------------------------------------------------------------------------------------------
// geometry: vision-common/src/screen.rs + consts/mod.rs + vision-cpu/src/lib.rs:137-145
// ------------------------------------------------------------------------------------------------
static uint32_t screen_h(double frac, uint32_t H) {
	// RelativeBound::ScreenH(h) => (h * screen_size[1] as f64).round() as u32   (screen.rs:58-65)
	double r = std::round(frac * (double)H);
	if (!(r > 0.0)) return 0u;
	if (r >= 4294967296.0) return 0xFFFFFFFFu;
	return (uint32_t)r;
}

extern "C" SMHV_API int smhv_map_bounds(uint32_t W, uint32_t H, uint32_t out[4]) {
	if (!out) return fail(SMHV_E_INVALID, "null output");
	const uint32_t w = screen_h(SMH_MAP_W, H), h = screen_h(SMH_MAP_H, H);
	const uint32_t x = screen_h(SMH_MAP_X, H), yb = screen_h(SMH_MAP_Y_BOTTOM, H);
	if ((uint64_t)yb + h > H || w > W) return fail(SMHV_E_GEOMETRY, "frame %ux%u: map bounds underflow", W, H);
	const uint32_t y = H - yb - h;
	const uint32_t w2 = W - w;          // "Map fills remaining space"
	if ((uint64_t)x + w2 > W) return fail(SMHV_E_GEOMETRY, "frame %ux%u: map bounds underflow", W, H);
	const uint32_t x2 = W - x - w2;
	// par_crop_into panics when x + w >= width || y + h >= height (util/src/image.rs:71-77)
	if ((uint64_t)x2 + w2 >= W || (uint64_t)y + h >= H || w2 < 8 || h < 8)
		return fail(SMHV_E_GEOMETRY, "frame %ux%u: map crop (%u,%u,%u,%u) is outside the frame", W, H, x2, y, w2, h);
	out[0] = x2; out[1] = y; out[2] = w2; out[3] = h;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_button_bounds(uint32_t W, uint32_t H, uint32_t out[4]) {
	if (!out) return fail(SMHV_E_INVALID, "null output");
	const uint32_t w = screen_h(SMH_BTN_W, H), h = screen_h(SMH_BTN_H, H);
	const uint32_t xr = screen_h(SMH_BTN_X_RIGHT, H), yb = screen_h(SMH_BTN_Y_BOTTOM, H);
	if ((uint64_t)xr + w > W || (uint64_t)yb + h > H || w == 0 || h == 0)
		return fail(SMHV_E_GEOMETRY, "frame %ux%u: button bounds underflow", W, H);
	out[0] = W - xr - w; out[1] = H - yb - h; out[2] = w; out[3] = h;
	return SMHV_OK;
}

static int compute_geom(uint32_t W, uint32_t H, Geom *g) {
	uint32_t m[4], bt[4];
	int rc = smhv_map_bounds(W, H, m);
	if (rc) return rc;
	rc = smhv_button_bounds(W, H, bt);
	if (rc) return rc;
	memset(g, 0, sizeof *g);
	g->W = W; g->H = H;
	g->rx = m[0]; g->ry = m[1]; g->rw = m[2]; g->rh = m[3];
	g->bx = bt[0]; g->by = bt[1]; g->bw = bt[2]; g->bh = bt[3];
	g->qw = g->rw / 2; g->qh = g->rh / 2;            // brq = bottom right quadrant (lib.rs:143-145)
	g->qx = g->rx + g->qw; g->qy = g->ry + g->qh;
	if (g->qw < SMH_OCR_DILATE_RADIUS || g->qh < SMH_OCR_DILATE_RADIUS) return fail(SMHV_E_GEOMETRY, "frame %ux%u too small", W, H);
	g->m_ax = g->rx & ~3u; g->m_xoff = g->rx - g->m_ax;
	g->m_quads = (g->rw + g->m_xoff + 3u) / 4u;
	g->m_block = (g->m_quads + 63u) & ~63u;
	g->q_ax = g->qx & ~3u; g->q_xoff = g->qx - g->q_ax;
	g->q_quads = (g->qw + g->q_xoff + 3u) / 4u;
	g->q_block = (g->q_quads + 63u) & ~63u;
	if (g->m_block > 1024u) return fail(SMHV_E_GEOMETRY, "frame %ux%u: map ROI wider than 4096 px is not supported", W, H);
	const uint32_t mq = (g->m_quads + 15u) & ~15u, qq = (g->q_quads + 15u) & ~15u;
	g->frame_bytes = (uint64_t)W * H * 4;
	g->ui_pitch = (uint64_t)mq * 16; g->ui_stride = g->ui_pitch * g->rh;
	g->mask_pitch = (uint64_t)mq * 4; g->mask_stride = g->mask_pitch * g->rh;
	g->bits_pitch_w = mq / 8; g->bits_stride_w = (uint64_t)g->bits_pitch_w * g->rh;
	g->ocr_pitch = (uint64_t)qq * 4; g->ocr_stride = g->ocr_pitch * g->qh;
	return SMHV_OK;
}

// ------------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------------
struct smhv_batch {
	smhv_ctx *ctx = nullptr;
	Geom g{};
	uint32_t max_frames = 0;
	uint8_t *d_ui = nullptr, *d_mask = nullptr, *d_ocr = nullptr, *d_scales = nullptr;
	uint32_t *d_bits = nullptr, *d_bars = nullptr, *d_tiled = nullptr;
	uint8_t *d_occ = nullptr;
	FrameAux *d_aux = nullptr;
	smhv_frame_result *d_results = nullptr;   // max_frames (+3 spare records for the per-frame trait path)
	smhv_anchors *d_anchors = nullptr;
	BatchError *h_err = nullptr, *d_err = nullptr;   // error mailbox: pinned host memory and its device address (smh_kernels.h)
	FarmFrame *d_farm = nullptr;                     // late-helper exchange of k_lsd_tile (one entry per frame)
	// k_lsd cooperation (smh_kernels.h): [LsdCtl][LsdCoop x n] zeroed per launch, request rings, result caches
	uint8_t *d_lsd_ctl = nullptr;
	uint32_t *d_lsd_req = nullptr;
	LsdCacheEntry *d_lsd_cache = nullptr;
	uint32_t lsd_epoch = 0;
	// pinned staging buffers for the per-run anchor upload, each with the event of the copy that last read it
	struct AnchorStage { smhv_anchors *h = nullptr; hipEvent_t done = nullptr; };
	std::vector<AnchorStage> anchor_stage;
	// per-stage hipEvent ring: up to TIMING_RING timed runs are kept so a benchmark loop can read the
	// average stage durations afterwards without synchronising between steps
	static constexpr int TIMING_RING = 64;
	static constexpr int TIMING_EVENTS = 10;              // start/end of button, map pass, brq pass, lsd, scale ratio
	bool timing = false;
	hipEvent_t (*ev)[TIMING_EVENTS] = nullptr;
	uint64_t timed_runs = 0;
	// The scales branch (brq pass + scale ratio) runs on its own stream beside the markers branch, like the two
	// concurrent branches of VisionState::process (src/vision/mod.rs:219-223): fork after the button test, join
	// before the record is finalised.
	hipStream_t s_scales = nullptr;
	hipStream_t s_scales_ext = nullptr;       // caller-provided stream for the scales branch (smhv_batch_set_scales_stream)
	hipEvent_t ev_map_done = nullptr;         // recorded after the streaming pass of every run (smhv_batch_wait_map_pass)
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	// the per-mode k_lsd kernels of frames larger than 1080p run side by side (smh_kernels.h, LsdFork)
	LsdFork lsd_fork{};
	// threads per workgroup of the line search (k_lsd_tile): 1024 for a batch that has the chip to itself, 512 for the batches of
	// a pipeline (two workgroups per CU, and room for the streaming pass of the other batches beside them)
	uint32_t lsd_late_kc = 0;             // late helpers of k_lsd_tile: thousands of cycles a frame works alone before it asks (0 = none)
	uint32_t lsd_bs = 1024;
	bool lsd_prefer_classic = false;      // set by smhv_pipeline_create where the workgroup-synchronous k_lsd measures faster
	LaunchTuning tune{0u, 0u, 0u, 0u};        // occupancy policy of a pipelined batch (smh_kernels.h); all zero for a batch that runs alone
	// probe of a pipeline that adapts its policy to the workload: start of the streaming pass, its end, end of the line search
	hipEvent_t ev_probe[3] = {nullptr, nullptr, nullptr};
	bool probe = false, probe_valid = false;
};

struct smhv_ctx {
	int device = 0;
	smhv_log_fn log = nullptr;
	hipStream_t s_main = nullptr, s_markers = nullptr, s_scales = nullptr;
	// crop_to_map's ui_map travels to pinned host memory on a stream of its own while the two branches run (the reference hands
	// its caller a pinned image that is copied when somebody looks at it: PinnedGpuImage, vision-gpu/src/gpuimage.rs:117-166);
	// two buffers in turn, so that the map of frame k stays readable while frame k + 1 is processed
	hipStream_t s_ui = nullptr;
	hipEvent_t ev_map = nullptr, ev_ui[2] = {nullptr, nullptr};
	hipEvent_t ev_ui_part[3] = {nullptr, nullptr, nullptr};     // the eager crop_to_map: the ui_map leaves the device in four row blocks (the last one's event is ev_ui)
	uint8_t *h_ui[2] = {nullptr, nullptr};
	size_t h_ui_cap[2] = {0, 0};
	uint8_t *d_ui_tight = nullptr;                               // the ui_map packed tightly on the device: it leaves as contiguous copies (k_pack_rows)
	size_t d_ui_tight_cap = 0;
	uint32_t ui_turn = 0;
	bool ui_pending = false, minimap_cached = false;
	// current frame (per-call trait path); ~ GpuMemory
	uint32_t W = 0, H = 0;
	bool have_frame = false;
	uint8_t *d_frame = nullptr;        // owned copy of the uploaded frame
	size_t d_frame_cap = 0;
	const uint8_t *frame_ptr = nullptr; // d_frame or the caller's device pointer
	smhv_batch *fb = nullptr;          // single-frame buffer set
	// pinned staging
	uint8_t *h_ocr = nullptr, *h_scales = nullptr;
	// pinned staging for pitched device images (one per concurrent branch + one for the batch read-back, see copy_image_d2h)
	uint8_t *h_stage[3] = {nullptr, nullptr, nullptr};
	size_t h_stage_cap[3] = {0, 0, 0};
	smhv_frame_result *h_res = nullptr;
	FrameAux *h_aux = nullptr;
	uint32_t *h_bars = nullptr;
	// per-frame state
	bool cropped = false, map_open = false, isolated = false, mask_valid = false, scales_valid = false;
	// sector culling tables of k_lsd, one per gap threshold T = ceil(max_gap) seen so far (built on first use)
	static constexpr int SECTOR_CACHE = 8;
	uint32_t sector_T[SECTOR_CACHE] = {};
	uint32_t *sector_tab[SECTOR_CACHE] = {};
	int sector_n = 0;
	float *d_ray_off = nullptr;         // Buffers::ray_off (built at init, rebuilt by smhv_set_ray_table)
	uint32_t *d_side = nullptr;         // smhv_debug_side_kernel's output words
	// host wall time of every trait call, summed (smhv_trait_times: the per-call path's counterpart of the reference's Timeshares
	// waterfall, vision-common/src/debug.rs:3-30); two threads call in, hence atomics
	std::atomic<uint64_t> tt_ns[SMHV_TRAIT_CALLS] = {}, tt_calls[SMHV_TRAIT_CALLS] = {};
	std::mutex mu;                      // serialises (re)allocation only
	// Lifetime: batches and ingest queues hold a reference, so smhv_shutdown with children still alive releases the
	// context's own resources and marks it closed, and the object itself goes with the last child (their destroy
	// functions only need `device`).  Every entry point that takes a child checks `closed` first.
	std::atomic<int> refs{1};
	std::atomic<bool> closed{false};
};

// Open contexts (smhv_set_ray_table rebuilds the offset tables of every context on the device whose ray directions change)
static std::mutex g_ctx_mu;
static std::vector<smhv_ctx *> g_ctx_open;

static void ctx_release(smhv_ctx *c) {
	if (c && c->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete c;
}
#define CTX_OPEN(c)                                                                                          \
	do {                                                                                                     \
		if (!(c) || (c)->closed.load(std::memory_order_acquire)) return fail(SMHV_E_INVALID, "the context was shut down"); \
	} while (0)

static void logf(smhv_ctx *c, int lvl, const char *fmt, ...) {
	if (!c || !c->log) return;
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	c->log(lvl, buf);
}

// Smallest cyclic run of 64-ray units covering mask m (a superset only casts more rays) as the byte code of
// Buffers::cull_tab: first | (n - 1) << 6 for n <= 4, else 0xFF = every unit.  m != 0.
static uint8_t unit_code(unsigned long long m) {
	const uint32_t NU = 57;                                   // units per candidate
	m &= (1ull << NU) - 1ull;
	uint32_t first = 0, cnt = NU, best_gap = 0;               // complement of the longest cyclic zero gap
	for (uint32_t st = 0; st < NU; ++st) {
		if (!((m >> st) & 1ull)) continue;
		uint32_t gap = 0;
		while (gap < NU - 1 && !((m >> ((st + 1 + gap) % NU)) & 1ull)) ++gap;
		if (gap > best_gap) { best_gap = gap; first = (st + 1 + gap) % NU; cnt = NU - gap; }
	}
	if (best_gap == 0 || cnt > 4) return 0xFF;
	return (uint8_t)(first | ((cnt - 1) << 6));               // first <= 56: never 0xFF
}

// Device table for max_gap (nullptr => k_lsd casts every ray).  Built once per distinct threshold: the kernel
// fills the dense (2R+1)^2 table, the host condenses it into per-word cells (smh_kernels.h, Buffers::cull_tab).
static int sector_table_for(smhv_ctx *c, uint32_t max_gap, hipStream_t s, Buffers *bf) {
	bf->cull_tab = nullptr;
	if (max_gap == 0 || max_gap > 49) return SMHV_OK;        // nothing to cull
	std::lock_guard<std::mutex> lk(c->mu);
	for (int i = 0; i < c->sector_n; ++i)
		if (c->sector_T[i] == max_gap) { bf->cull_tab = c->sector_tab[i]; return SMHV_OK; }
	if (c->sector_n == smhv_ctx::SECTOR_CACHE) return SMHV_OK;   // cache full: fall back to casting every ray
	unsigned long long *d_dense = nullptr;
	HIPCHK(hipMalloc((void **)&d_dense, sizeof(unsigned long long) * SMH_SECTOR_ENTRIES));
	std::vector<unsigned long long> dense(SMH_SECTOR_ENTRIES);
	hipError_t e = launch_build_sector_table(d_dense, max_gap, s);
	if (e == hipSuccess) e = hipMemcpyAsync(dense.data(), d_dense, sizeof(unsigned long long) * SMH_SECTOR_ENTRIES, hipMemcpyDeviceToHost, s);
	if (e == hipSuccess) e = hipStreamSynchronize(s);
	(void)hipFree(d_dense);
	if (e != hipSuccess) return fail(SMHV_E_HIP, "sector table: %s", hipGetErrorString(e));
	std::vector<uint32_t> cells(SMH_CULL_TAB_WORDS, 0u);
	uint8_t *codes = (uint8_t *)&cells[SMH_CULL_CELLS];
	for (uint32_t row = 0; row < SMH_SECTOR_DIM; ++row)
		for (uint32_t col = 0; col < SMH_SECTOR_DIM; ++col) {
			const unsigned long long m = dense[row * SMH_SECTOR_DIM + col] & ((1ull << 57) - 1ull);
			if (!m) continue;
			cells[row * 4 + (col >> 5)] |= 1u << (col & 31u);
			codes[row * SMH_SECTOR_DIM + col] = unit_code(m);
		}
	uint32_t *d = nullptr;
	HIPCHK(hipMalloc((void **)&d, sizeof(uint32_t) * cells.size()));
	e = hipMemcpy(d, cells.data(), sizeof(uint32_t) * cells.size(), hipMemcpyHostToDevice);
	if (e != hipSuccess) { (void)hipFree(d); return fail(SMHV_E_HIP, "sector table: %s", hipGetErrorString(e)); }
	c->sector_T[c->sector_n] = max_gap; c->sector_tab[c->sector_n] = d; c->sector_n++;
	bf->cull_tab = d;
	return SMHV_OK;
}

static Buffers make_buffers(smhv_batch *b, const uint8_t *frames, uint32_t result_slot) {
	Buffers bf;
	bf.err = b->d_err;
	bf.farm = b->d_farm; bf.rec_stages = 0u; bf.rec_bars = nullptr; bf.lsd_flags = 0u; bf.lsd_late_kc = 0u;
	bf.cull_tab = nullptr;
	bf.ray_off = b->ctx->d_ray_off;
	bf.frames = frames;
	bf.ui = b->d_ui; bf.mask = b->d_mask; bf.ocr = b->d_ocr; bf.scales = b->d_scales;
	bf.bits = b->d_bits; bf.aux = b->d_aux;
	bf.tiled = b->d_tiled; bf.occ = b->d_occ;
	bf.results = b->d_results + result_slot;
	bf.anchors = b->d_anchors;
	bf.co.ctl = nullptr;                                     // k_lsd cooperation is opt-in: smhv_batch_run sets it for SMHV_STAGE_LSD_HELPERS
	bf.co.coop = (LsdCoop *)(b->d_lsd_ctl + sizeof(LsdCtl));
	bf.co.req = b->d_lsd_req;
	bf.co.cache = b->d_lsd_cache;
	// a fresh epoch per set of launches: cache entries of earlier launches read as empty slots (0 = never used)
	if (++b->lsd_epoch == 0) {
		(void)hipMemset(b->d_lsd_cache, 0, sizeof(LsdCacheEntry) * SMH_LSD_CACHE_SLOTS * (size_t)b->max_frames);
		b->lsd_epoch = 1;
	}
	bf.co.epoch = b->lsd_epoch;
	return bf;
}

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" SMHV_API int smhv_init(int device, smhv_log_fn log, smhv_ctx **out) {
	if (!out) return fail(SMHV_E_INVALID, "null output");
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) return fail(SMHV_E_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
	if (device < 0 || device >= count) return fail(SMHV_E_NO_DEVICE, "device %d out of range (%d devices)", device, count);
	HIPCHK(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(SMHV_E_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
	smhv_ctx *c = new (std::nothrow) smhv_ctx();
	if (!c) return fail(SMHV_E_INVALID, "out of host memory");
	c->device = device; c->log = log;
	hipError_t he = hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking);
	if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->s_markers, hipStreamNonBlocking);
	if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->s_scales, hipStreamNonBlocking);
	if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->s_ui, hipStreamNonBlocking);
	if (he == hipSuccess) he = hipEventCreateWithFlags(&c->ev_map, hipEventDisableTiming);
	for (int i = 0; i < 2 && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&c->ev_ui[i], hipEventDisableTiming);
	for (int i = 0; i < 3 && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&c->ev_ui_part[i], hipEventDisableTiming);
	if (he == hipSuccess) he = hipHostMalloc((void **)&c->h_res, sizeof(smhv_frame_result) * 4);
	if (he == hipSuccess) he = hipHostMalloc((void **)&c->h_aux, sizeof(FrameAux));
	if (he == hipSuccess) he = hipHostMalloc((void **)&c->h_bars, sizeof(uint32_t) * SMHV_MAX_SCALES * 4 + sizeof(smhv_anchors));   // (+ the anchors of calc_meters_to_px_ratio)
	if (he == hipSuccess) he = hipMalloc((void **)&c->d_ray_off, sizeof(float) * 2 * SMH_LSD_RAYS * (SMH_RAY_OFF_BATCHES + 1));
	if (he == hipSuccess) he = launch_build_ray_offsets(c->d_ray_off, c->s_main);
	if (he == hipSuccess) he = hipStreamSynchronize(c->s_main);
	if (he != hipSuccess) {
		smhv_shutdown(c);                                    // releases whatever was created
		return fail(SMHV_E_HIP, "context setup failed: %s", hipGetErrorString(he));
	}
	logf(c, 3, "smh_vision_hip ready on device %d (%s, %d CUs)", device, prop.gcnArchName, prop.multiProcessorCount);
	{ std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctx_open.push_back(c); }
	*out = c;
	return SMHV_OK;
}

extern "C" SMHV_API void smhv_shutdown(smhv_ctx *c) {
	if (!c || c->closed.exchange(true, std::memory_order_acq_rel)) return;
	{
		std::lock_guard<std::mutex> lk(g_ctx_mu);
		for (size_t i = 0; i < g_ctx_open.size(); ++i) if (g_ctx_open[i] == c) { g_ctx_open.erase(g_ctx_open.begin() + (long)i); break; }
	}
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	if (c->fb) { smhv_batch_destroy(c->fb); c->fb = nullptr; }
	c->have_frame = false; c->cropped = false; c->map_open = false; c->mask_valid = false; c->scales_valid = false;
	for (int i = 0; i < c->sector_n; ++i) (void)hipFree(c->sector_tab[i]);
	c->sector_n = 0;
	if (c->d_ray_off) { (void)hipFree(c->d_ray_off); c->d_ray_off = nullptr; }
	if (c->d_side) { (void)hipFree(c->d_side); c->d_side = nullptr; }
	if (c->d_frame) (void)hipFree(c->d_frame);
	if (c->h_ocr) (void)hipHostFree(c->h_ocr);
	if (c->h_scales) (void)hipHostFree(c->h_scales);
	for (int i = 0; i < 3; ++i) { if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]); c->h_stage[i] = nullptr; c->h_stage_cap[i] = 0; }
	if (c->h_res) (void)hipHostFree(c->h_res);
	if (c->h_aux) (void)hipHostFree(c->h_aux);
	if (c->h_bars) (void)hipHostFree(c->h_bars);
	if (c->s_main) (void)hipStreamDestroy(c->s_main);
	if (c->s_markers) (void)hipStreamDestroy(c->s_markers);
	if (c->s_scales) (void)hipStreamDestroy(c->s_scales);
	if (c->s_ui) (void)hipStreamDestroy(c->s_ui);
	if (c->ev_map) (void)hipEventDestroy(c->ev_map);
	for (int i = 0; i < 3; ++i) { if (c->ev_ui_part[i]) (void)hipEventDestroy(c->ev_ui_part[i]); c->ev_ui_part[i] = nullptr; }
	for (int i = 0; i < 2; ++i) { if (c->ev_ui[i]) (void)hipEventDestroy(c->ev_ui[i]); if (c->h_ui[i]) (void)hipHostFree(c->h_ui[i]); c->ev_ui[i] = nullptr; c->h_ui[i] = nullptr; c->h_ui_cap[i] = 0; }
	if (c->d_ui_tight) (void)hipFree(c->d_ui_tight);
	c->d_ui_tight = nullptr; c->d_ui_tight_cap = 0;
	c->s_ui = nullptr; c->ev_map = nullptr;
	c->d_frame = nullptr; c->d_frame_cap = 0; c->h_ocr = c->h_scales = nullptr; c->h_res = nullptr; c->h_aux = nullptr; c->h_bars = nullptr;
	c->s_main = c->s_markers = c->s_scales = nullptr;
	logf(c, 3, "smh_vision_hip shut down");
	c->log = nullptr;
	ctx_release(c);
}

extern "C" SMHV_API int smhv_thread_ctx(smhv_ctx *c) {
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	return SMHV_OK;
}

static std::atomic<bool> g_skip_lsd{false};
extern "C" SMHV_API int smhv_debug_lsd_classic(int on) {
	lsd_set_classic(on != 0);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_debug_lsd_tile_cap(uint32_t cap) {
	lsd_set_tile_cap(cap);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_debug_skip_line_search(int on) {
	g_skip_lsd.store(on != 0, std::memory_order_relaxed);
	return SMHV_OK;
}

static std::atomic<bool> g_no_host_atomics{false};
extern "C" SMHV_API int smhv_debug_no_host_atomics(int on) {
	g_no_host_atomics.store(on != 0, std::memory_order_relaxed);
	return SMHV_OK;
}
// A kernel with the resource footprint of a collective's kernel (21 KB of LDS, 280 VGPRs per 256-thread workgroup) that does
// next to nothing, asynchronous on `stream`: the co-residency probe of the tests and of bench.py --side-probe.
extern "C" SMHV_API int smhv_debug_side_kernel(smhv_ctx *c, uint32_t workgroups, void *stream) {
	if (!c || workgroups == 0 || workgroups > 1024u) return fail(SMHV_E_INVALID, "side_kernel: 1..1024 workgroups");
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	if (!c->d_side) HIPCHK(hipMalloc((void **)&c->d_side, 1024u * sizeof(uint32_t)));
	HIPCHK(launch_side_probe(c->d_side, workgroups, 64u, (hipStream_t)stream));
	return SMHV_OK;
}
extern "C" SMHV_API int smhv_debug_lsd_threads(uint32_t threads) {
	if (threads != 0 && (threads < 128 || threads > 1024 || threads % 64)) return fail(SMHV_E_INVALID, "lsd_threads: 0 or a multiple of 64 from 128 to 1024");
	lsd_set_threads(threads);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_debug_lsd_spin_limit(uint32_t polls) {
	lsd_set_spin_limit(polls);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_set_ray_table(smhv_ctx *c, const float *dx, const float *dy) {
	if (!c || !dx || !dy) return fail(SMHV_E_INVALID, "bad arguments");
	CTX_OPEN(c);
	// sanity: unit vectors at 0.1 degree steps (the sector culling tables assume that geometry)
	for (int i = 0; i < SMH_LSD_RAYS; ++i) {
		const double a = (double)i / 10.0 * 3.14159265358979323846 / 180.0;
		if (!(std::fabs((double)dx[i] - std::cos(a)) < 1e-5 && std::fabs((double)dy[i] - std::sin(a)) < 1e-5))
			return fail(SMHV_E_INVALID, "ray table entry %d is not (cos, sin) of %.1f degrees", i, i / 10.0);
	}
	// The directions are a device global of the code object, the accumulated offsets derived from them a buffer of each
	// context: both change together, for every open context on this device, or a line search would mix two tables.
	std::lock_guard<std::mutex> lk(g_ctx_mu);                 // (smhv_init / smhv_shutdown of other contexts wait)
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipDeviceSynchronize());                           // no k_lsd launch may be reading the table
	HIPCHK(set_ray_table(dx, dy));
	for (smhv_ctx *o : g_ctx_open)
		if (o->device == c->device && o->d_ray_off) HIPCHK(launch_build_ray_offsets(o->d_ray_off, c->s_main));
	HIPCHK(hipStreamSynchronize(c->s_main));
	return SMHV_OK;
}

// ------------------------------------------------------------------------------------------------
// batch objects
// ------------------------------------------------------------------------------------------------
static hipError_t create_stream(hipStream_t *st, const uint32_t *cu_mask) {
	// (hipExtStreamCreateWithCUMask streams are non-blocking with respect to the null stream as well)
	return cu_mask ? hipExtStreamCreateWithCUMask(st, 8, cu_mask) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

extern "C" SMHV_API int smhv_batch_create(smhv_ctx *c, uint32_t W, uint32_t H, uint32_t max_frames, smhv_batch **out) {
	if (!c || !out || max_frames == 0) return fail(SMHV_E_INVALID, "bad arguments");
	*out = nullptr;
	CTX_OPEN(c);
	Geom g;
	int rc = compute_geom(W, H, &g);
	if (rc) return rc;
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = new (std::nothrow) smhv_batch();
	if (!b) return fail(SMHV_E_INVALID, "out of host memory");
	c->refs.fetch_add(1, std::memory_order_relaxed);
	b->ctx = c; b->g = g; b->max_frames = max_frames;
	const size_t n = max_frames;
#define ALLOC0(ptr, bytes)                                          \
	do {                                                            \
		hipError_t _e = hipMalloc((void **)&(ptr), (bytes));        \
		if (_e == hipSuccess) _e = hipMemset((ptr), 0, (bytes));    \
		if (_e != hipSuccess) { smhv_batch_destroy(b); return fail(SMHV_E_HIP, "allocating %zu bytes: %s", (size_t)(bytes), hipGetErrorString(_e)); } \
	} while (0)
	ALLOC0(b->d_ui, g.ui_stride * n);
	ALLOC0(b->d_mask, g.mask_stride * n);
	ALLOC0(b->d_bits, g.bits_stride_w * 4 * n);
	ALLOC0(b->d_tiled, tiled_stride_w(g) * 4 * n);
	ALLOC0(b->d_occ, occ_stride(g) * n);
	ALLOC0(b->d_ocr, g.ocr_stride * n);
	ALLOC0(b->d_scales, g.ocr_stride * n);      // zero-initialised like GrayImage::new (lib.rs:86)
	ALLOC0(b->d_aux, sizeof(FrameAux) * n);
	ALLOC0(b->d_results, sizeof(smhv_frame_result) * (n + 3));
	ALLOC0(b->d_anchors, sizeof(smhv_anchors) * n);
	ALLOC0(b->d_bars, sizeof(uint32_t) * SMHV_MAX_SCALES * 4 * n);
	ALLOC0(b->d_lsd_ctl, lsd_coop_ctl_bytes(max_frames));
	ALLOC0(b->d_lsd_req, sizeof(uint32_t) * SMH_LSD_REQ_CAP * n);
	ALLOC0(b->d_farm, sizeof(FarmFrame) * n);
	ALLOC0(b->d_lsd_cache, sizeof(LsdCacheEntry) * SMH_LSD_CACHE_SLOTS * n);
#undef ALLOC0
	{
		// the error mailbox lives in host memory: the device writes it only when a frame fails, the host reads it for free
		hipError_t e = hipHostMalloc((void **)&b->h_err, sizeof(BatchError), hipHostMallocMapped | hipHostMallocCoherent);
		if (e == hipSuccess) { memset(b->h_err, 0, sizeof(BatchError)); e = hipHostGetDevicePointer((void **)&b->d_err, b->h_err, 0); }
		if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_map_done, hipEventDisableTiming);
		// Only for frame sizes that need them: a process has few hardware queues, and every extra stream makes it more
		// likely that two independent branches share one (measured: 10 % off the pipelined 1080p throughput).
		if (max_frames > 1 && !lsd_rows_only(b->g)) {
			if (e == hipSuccess) e = create_stream(&b->lsd_fork.s1, nullptr);
			if (e == hipSuccess) e = create_stream(&b->lsd_fork.s2, nullptr);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&b->lsd_fork.fork, hipEventDisableTiming);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&b->lsd_fork.join1, hipEventDisableTiming);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&b->lsd_fork.join2, hipEventDisableTiming);
		}
		if (e != hipSuccess) { smhv_batch_destroy(b); return fail(SMHV_E_HIP, "branch streams: %s", hipGetErrorString(e)); }
	}
	*out = b;
	return SMHV_OK;
}

extern "C" SMHV_API void smhv_batch_destroy(smhv_batch *b) {
	if (!b) return;
	if (b->ctx) (void)hipSetDevice(b->ctx->device);
	(void)hipDeviceSynchronize();
	void *ptrs[] = {b->d_ui, b->d_mask, b->d_bits, b->d_tiled, b->d_occ, b->d_ocr, b->d_scales, b->d_aux, b->d_results, b->d_anchors, b->d_bars, b->d_farm,
	                b->d_lsd_ctl, b->d_lsd_req, b->d_lsd_cache};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (b->h_err) (void)hipHostFree(b->h_err);
	for (auto &a : b->anchor_stage) {
		if (a.h) (void)hipHostFree(a.h);
		if (a.done) (void)hipEventDestroy(a.done);
	}
	if (b->ev) {
		for (int r = 0; r < smhv_batch::TIMING_RING; ++r)
			for (int i = 0; i < smhv_batch::TIMING_EVENTS; ++i) if (b->ev[r][i]) (void)hipEventDestroy(b->ev[r][i]);
		delete[] b->ev;
	}
	if (b->s_scales) (void)hipStreamDestroy(b->s_scales);
	if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
	if (b->ev_join) (void)hipEventDestroy(b->ev_join);
	if (b->ev_map_done) (void)hipEventDestroy(b->ev_map_done);
	for (auto e : b->ev_probe) if (e) (void)hipEventDestroy(e);
	if (b->lsd_fork.s1) (void)hipStreamDestroy(b->lsd_fork.s1);
	if (b->lsd_fork.s2) (void)hipStreamDestroy(b->lsd_fork.s2);
	if (b->lsd_fork.fork) (void)hipEventDestroy(b->lsd_fork.fork);
	if (b->lsd_fork.join1) (void)hipEventDestroy(b->lsd_fork.join1);
	if (b->lsd_fork.join2) (void)hipEventDestroy(b->lsd_fork.join2);
	ctx_release(b->ctx);
	delete b;
}

// Frames that failed since the last report (smh_vision_hip.h, SMHV_FRAME_*).  The caller has synchronised with the run.
static int batch_check_errors(smhv_batch *b, const char *what) {
	BatchError *e = b->h_err;
	if (!e) return SMHV_OK;
	const uint32_t count = __atomic_load_n(&e->count, __ATOMIC_ACQUIRE);
	if (count == 0) return SMHV_OK;
	const BatchError snap = *e;
	memset(e, 0, sizeof *e);                                  // reported once
	logf(b->ctx, 1, "%s: %u frame(s) failed, first: frame %u status %u", what, count, snap.frame, snap.status);
	if (snap.status == SMHV_FRAME_LSD_STUCK)
		return fail(SMHV_E_STATE, "%s: %u frame(s) without marker lines (status != 0 in their records); first: frame %u, line search gave up "
		            "(SMHV_FRAME_LSD_STUCK: head %u tail %u list %u/%u lines %u rounds %u spec %u head-state %u)", what, count, snap.frame,
		            snap.info[0], snap.info[1], snap.info[2], snap.info[3], snap.info[4], snap.info[5], snap.info[6], snap.info[7]);
	return fail(SMHV_E_STATE, "%s: %u frame(s) failed; first: frame %u, status %u", what, count, snap.frame, snap.status);
}
extern "C" int smhv_internal_batch_check(smhv_batch *b, const char *what) { return b ? batch_check_errors(b, what) : SMHV_OK; }   // smh_node.cpp

extern "C" SMHV_API int smhv_batch_layout_get(smhv_batch *b, smhv_batch_layout *o) {
	if (!b || !o) return fail(SMHV_E_INVALID, "bad arguments");
	const Geom &g = b->g;
	memset(o, 0, sizeof *o);
	o->frame_w = g.W; o->frame_h = g.H;
	o->roi[0] = g.rx; o->roi[1] = g.ry; o->roi[2] = g.rw; o->roi[3] = g.rh;
	o->button[0] = g.bx; o->button[1] = g.by; o->button[2] = g.bw; o->button[3] = g.bh;
	o->brq_w = g.qw; o->brq_h = g.qh;
	o->ui_pitch = g.ui_pitch; o->ui_stride = g.ui_stride; o->ui_offset = (uint64_t)g.m_xoff * 4;
	o->mask_pitch = g.mask_pitch; o->mask_stride = g.mask_stride; o->mask_offset = g.m_xoff;
	o->ocr_pitch = g.ocr_pitch; o->ocr_stride = g.ocr_stride; o->ocr_offset = g.q_xoff;
	o->scales_pitch = g.ocr_pitch; o->scales_stride = g.ocr_stride; o->scales_offset = g.q_xoff;
	o->bits_pitch_words = g.bits_pitch_w; o->bits_stride = g.bits_stride_w * 4; o->bits_xoff = g.m_xoff;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_enable_timing(smhv_batch *b, int enable) {
	if (!b) return fail(SMHV_E_INVALID, "null batch");
	if (enable && !b->ev) {
		HIPCHK(hipSetDevice(b->ctx->device));
		b->ev = new hipEvent_t[smhv_batch::TIMING_RING][smhv_batch::TIMING_EVENTS]();   // zeroed: destroy skips what was never created
		for (int r = 0; r < smhv_batch::TIMING_RING; ++r)
			for (int i = 0; i < smhv_batch::TIMING_EVENTS; ++i) HIPCHK(hipEventCreate(&b->ev[r][i]));
	}
	b->timing = enable != 0;
	b->timed_runs = 0;
	return SMHV_OK;
}

// s: the streaming kernels (button test, the fused map / quadrant pass); sl: the line-segment search and the record kernel.
// sl == s for a plain smhv_batch_run and for every pipeline (sl != s: the line search on a stream of its own, waiting for the streaming pass).
// svc != null: the batch belongs to a pipeline with a frame-granular search service (smh_kernels.h): the streaming side ends
// with the publication of the frames, and the service's waves search them and write their records.
// s_pro: the pipeline's prologue stream -- the anchor upload and the button test of a submission run there, ahead of time, so
// that the chain on a streaming stream is pass -> publication -> pass: the button test (45 us inside a busy pipeline, plus a
// hand-over) is off it.
struct SvcPublish { SvcCtl *ctl; unsigned long long *ring; SvcSlot *slots; uint32_t slot, seq, ring_log2; const uint32_t *cull_tab; bool have_cull; hipStream_t s_pro; hipEvent_t ev_pro;
                    bool *published; };   // <- set once k_svc_publish has been enqueued (from then on the device WILL complete the submission)
static int batch_run_impl(smhv_batch *b, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                          const smhv_anchors *anchors, hipStream_t s, hipStream_t sl, const SvcPublish *svc = nullptr) {
	if (!b || !d_frames || n == 0 || n > b->max_frames) return fail(SMHV_E_INVALID, "bad arguments (n=%u, capacity %u)", n, b ? b->max_frames : 0);
	CTX_OPEN(b->ctx);
	if ((stages & (SMHV_STAGE_ALL | SMHV_STAGE_MINIMAP)) == 0) return fail(SMHV_E_INVALID, "no stage selected");
	stages &= SMHV_STAGE_ALL | SMHV_STAGE_MINIMAP | SMHV_STAGE_EXACT_STATS | SMHV_STAGE_LSD_HELPERS;
	HIPCHK(hipSetDevice(b->ctx->device));
	const Geom &g = b->g;
	Buffers bf = make_buffers(b, (const uint8_t *)d_frames, 0);
	// the helper exchange of k_lsd_tile tags its words with 16 bits of the launch epoch: start every 65,536th launch of a batch
	// from a clean slate, so that a stale word of the launch 65,536 ago can never read as this launch's
	if ((bf.co.epoch & 0xFFFFu) == 0u && b->d_farm) HIPCHK(hipMemsetAsync(b->d_farm, 0, sizeof(FarmFrame) * (size_t)b->max_frames, s));
	if (stages & SMHV_STAGE_LSD_HELPERS) bf.co.ctl = (LsdCtl *)b->d_lsd_ctl;
	if (svc && svc->have_cull) bf.cull_tab = svc->cull_tab;     // (looked up by the caller: the service is launched with it)
	else if ((stages & SMHV_STAGE_MARKERS) && !(stages & SMHV_STAGE_EXACT_STATS)) {
		int rc = sector_table_for(b->ctx, max_gap, s, &bf);
		if (rc) return rc;
	}
	const bool scales = (stages & SMHV_STAGE_SCALES) && anchors;
	const hipStream_t sb = (svc && svc->s_pro) ? svc->s_pro : s;   // where the anchor upload and the button test go
	if (scales) {
		// Pinned staging for the anchor upload (a pageable source would make hipMemcpyAsync synchronous).  A staging
		// buffer is reused once the copy that read it has completed; if none is free another one is allocated, so this call
		// never waits for the device.
		smhv_batch::AnchorStage *st = nullptr;
		for (auto &a : b->anchor_stage)
			if (hipEventQuery(a.done) == hipSuccess) { st = &a; break; }
		if (!st) {
			smhv_batch::AnchorStage a;
			hipError_t e = hipHostMalloc((void **)&a.h, sizeof(smhv_anchors) * b->max_frames);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&a.done, hipEventDisableTiming);
			if (e != hipSuccess) { if (a.h) (void)hipHostFree(a.h); return fail(SMHV_E_HIP, "anchor staging: %s", hipGetErrorString(e)); }
			b->anchor_stage.push_back(a);
			st = &b->anchor_stage.back();
		}
		memcpy(st->h, anchors, sizeof(smhv_anchors) * n);
		HIPCHK(hipMemcpyAsync(b->d_anchors, st->h, sizeof(smhv_anchors) * n, hipMemcpyHostToDevice, sb));
		HIPCHK(hipEventRecord(st->done, sb));
	}
	const bool t = b->timing;
	hipEvent_t *ev = t ? b->ev[b->timed_runs % smhv_batch::TIMING_RING] : nullptr;
#define STAGE_BEGIN(i, st) do { if (t) HIPCHK(hipEventRecord(ev[2 * (i)], (st))); } while (0)
#define STAGE_END(i, st) do { if (t) HIPCHK(hipEventRecord(ev[2 * (i) + 1], (st))); } while (0)
	STAGE_BEGIN(0, sb);
	HIPCHK(launch_button(g, bf, n, 0, sb));
	STAGE_END(0, sb);
	if (sb != s) {
		HIPCHK(hipEventRecord(svc->ev_pro, sb));
		HIPCHK(hipStreamWaitEvent(s, svc->ev_pro, 0));
	}
	// ---- ONE streaming pass: ui_map, marker mask + dilation, and -- on the pixels it has loaded anyway -- the two
	// bottom-right-quadrant images (ocr_preprocess, find_scales_preprocess).  Everything runs on the caller's stream: the
	// schedule does not depend on how HIP happens to map extra streams onto hardware queues.
	uint32_t mflags = 0, qflags = 0;
	if (stages & SMHV_STAGE_MARKERS) mflags |= MAP_MASK;
	if (stages & SMHV_STAGE_UI_MAP) mflags |= MAP_UI;
	if (stages & SMHV_STAGE_OCR) qflags |= BRQ_OCR;
	if (scales) qflags |= BRQ_SCALES;
	if (b->probe) HIPCHK(hipEventRecord(b->ev_probe[0], s));
	STAGE_BEGIN(1, s);
	if (mflags && qflags) HIPCHK(launch_map_brq_pass(g, bf, n, mflags, qflags, grayscale, 0, 1, s, &b->tune));
	else if (mflags) HIPCHK(launch_map_pass(g, bf, n, mflags, grayscale, s, true, b->tune.map_overlapped != 0u));
	STAGE_END(1, s);
	STAGE_BEGIN(2, s);
	if (qflags && !mflags) HIPCHK(launch_brq_pass(g, bf, n, qflags, 0, 1, s));
	STAGE_END(2, s);
	if (!svc) HIPCHK(hipEventRecord(b->ev_map_done, s));        // (every event recorded between two kernels of a chain lengthens their hand-over)
	if (b->probe) HIPCHK(hipEventRecord(b->ev_probe[1], s));
	if (sl != s) HIPCHK(hipStreamWaitEvent(sl, b->ev_map_done, 0));
	const bool skip_lsd = g_skip_lsd.load(std::memory_order_relaxed);   // diagnostic (smhv_debug_skip_line_search): the streaming pass with every output, no search
	if (svc) {
		// ---- frame-granular: publish the frames; the service searches them and writes the records (minimap first: its kernel
		// needs nothing of the search and the record keeps what it wrote) ----
		if (stages & SMHV_STAGE_MINIMAP) HIPCHK(launch_find_minimap(g, bf, n, s));
		bf.rec_stages = SMH_REC_ON | (scales ? stages : (stages & ~SMHV_STAGE_SCALES));
		bf.rec_bars = b->d_bars;
		STAGE_BEGIN(3, s);
		HIPCHK(launch_svc_publish(svc->ctl, svc->ring, svc->slots, svc->slot, bf, n, svc->seq, svc->ring_log2, s));
		if (svc->published) *svc->published = true;
		STAGE_END(3, s);
		STAGE_BEGIN(4, s);
		STAGE_END(4, s);
		if (t) b->timed_runs++;
		return SMHV_OK;
	}
	STAGE_BEGIN(3, sl);
	// late helpers: workgroups of k_lsd_tile that have finished their frame help one that is still at work (smh_kernels.h,
	// FarmFrame::want).  smhv_pipeline switches them on for search-bound workloads (smhv_pipeline_options::late_helpers)
	const uint32_t late_kc = b->lsd_late_kc;
	if (late_kc > 0u) { bf.lsd_flags |= SMH_LSD_LATE_HELP; bf.lsd_late_kc = late_kc; }
	// The workgroups of k_lsd_tile write their frames' records themselves (scale ratio + derived marker outputs, smh_record.inc):
	// one kernel less in the batch's chain on its hardware queue (with stage timing on, the record's share is then inside the
	// search's and stage 4 reads zero).  Not with the minimap stage: its kernel comes in between.
	bool record_fused = false;
	if (!(stages & SMHV_STAGE_MINIMAP)) {
		bf.rec_stages = SMH_REC_ON | (scales ? stages : (stages & ~SMHV_STAGE_SCALES));
		bf.rec_bars = b->d_bars;
	}
	if ((stages & SMHV_STAGE_MARKERS) && !skip_lsd) HIPCHK(launch_lsd(g, bf, n, (float)max_gap, 0, 0.0f, 0.0f, sl, b->lsd_fork.s1 ? &b->lsd_fork : nullptr, b->lsd_bs, b->lsd_prefer_classic,
	                                                      (mflags && qflags) ? b->tune.lsd_tile_limit : 0u,   // (the limit makes room for the fused pass's reservation: no fused pass, no limit)
	                                                      &record_fused));
	STAGE_END(3, sl);
	if (b->probe) { HIPCHK(hipEventRecord(b->ev_probe[2], sl)); b->probe_valid = (stages & SMHV_STAGE_MARKERS) && mflags; }
	if (stages & SMHV_STAGE_MINIMAP) HIPCHK(launch_find_minimap(g, bf, n, sl));
	STAGE_BEGIN(4, sl);
	if (record_fused) { /* written by the search's own workgroups */ }
	else if (scales) HIPCHK(launch_scales_finalize(g, bf, n, stages, b->d_bars, sl));
	else HIPCHK(launch_finalize(g, bf, n, stages & ~SMHV_STAGE_SCALES, sl));
	STAGE_END(4, sl);
#undef STAGE_BEGIN
#undef STAGE_END
	if (t) b->timed_runs++;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_run(smhv_batch *b, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                                       const smhv_anchors *anchors, void *stream) {
	return batch_run_impl(b, d_frames, n, stages, grayscale, max_gap, anchors, (hipStream_t)stream, (hipStream_t)stream);
}

extern "C" SMHV_API int smhv_debug_pattern_copy(smhv_batch *b, const void *d_frames, uint32_t n, uint32_t rows_in_flight, void *stream) {
	if (!b || !d_frames || n == 0 || n > b->max_frames || (rows_in_flight != 0u && rows_in_flight != 4u && rows_in_flight != 8u && rows_in_flight != 12u))
		return fail(SMHV_E_INVALID, "bad arguments (rows in flight: 0 = 4, 4, 8, 12)");
	CTX_OPEN(b->ctx);
	HIPCHK(hipSetDevice(b->ctx->device));
	const Buffers bf = make_buffers(b, (const uint8_t *)d_frames, 0);
	HIPCHK(launch_pattern_copy(b->g, bf, n, rows_in_flight, (hipStream_t)stream));
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_wait_map_pass(smhv_batch *b, void *stream) {
	if (!b) return fail(SMHV_E_INVALID, "null batch");
	HIPCHK(hipSetDevice(b->ctx->device));
	HIPCHK(hipStreamWaitEvent((hipStream_t)stream, b->ev_map_done, 0));   // no-op before the first run
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_set_scales_stream(smhv_batch *b, void *stream) {
	if (!b) return fail(SMHV_E_INVALID, "null batch");
	(void)stream;                                             // kept for source compatibility: there is no scales branch stream any more
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_stage_ms(smhv_batch *b, float ms[5]) {
	if (!b || !ms) return fail(SMHV_E_INVALID, "bad arguments");
	if (!b->ev || b->timed_runs == 0) return fail(SMHV_E_STATE, "no timed smhv_batch_run since timing was enabled");
	const uint64_t runs = b->timed_runs < (uint64_t)smhv_batch::TIMING_RING ? b->timed_runs : (uint64_t)smhv_batch::TIMING_RING;
	double acc[5] = {0, 0, 0, 0, 0};
	for (uint64_t r = 0; r < runs; ++r) {
		hipEvent_t *ev = b->ev[(b->timed_runs - 1 - r) % smhv_batch::TIMING_RING];
		for (int i = 0; i < 5; ++i) {
			float m = 0;
			HIPCHK(hipEventSynchronize(ev[2 * i + 1]));
			HIPCHK(hipEventElapsedTime(&m, ev[2 * i], ev[2 * i + 1]));
			acc[i] += m;
		}
	}
	for (int i = 0; i < 5; ++i) ms[i] = (float)(acc[i] / (double)runs);
	b->timed_runs = 0;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_lsd_coop_stats(smhv_batch *b, uint32_t first, uint32_t n, uint32_t *out) {
	if (!b || !out || (uint64_t)first + n > b->max_frames) return fail(SMHV_E_INVALID, "bad arguments");
	HIPCHK(hipSetDevice(b->ctx->device));
	HIPCHK(hipDeviceSynchronize());
	std::vector<LsdCoop> h(n);
	HIPCHK(hipMemcpy(h.data(), b->d_lsd_ctl + sizeof(LsdCtl) + sizeof(LsdCoop) * (size_t)first, sizeof(LsdCoop) * n, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < n; ++i) { out[4 * i] = h[i].stat_groups; out[4 * i + 1] = h[i].stat_hits; out[4 * i + 2] = h[i].stat_casts; out[4 * i + 3] = h[i].req_tail; }
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_device_ptrs(smhv_batch *b, void **r, void **ui, void **mask, void **ocr, void **scales, void **bits) {
	if (!b) return fail(SMHV_E_INVALID, "null batch");
	if (r) *r = b->d_results;
	if (ui) *ui = b->d_ui;
	if (mask) *mask = b->d_mask;
	if (ocr) *ocr = b->d_ocr;
	if (scales) *scales = b->d_scales;
	if (bits) *bits = b->d_bits;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_tile_mask(smhv_batch *b, void **d_tiled, void **d_occ, uint32_t geometry[4]) {
	if (!b) return fail(SMHV_E_INVALID, "null batch");
	if (d_tiled) *d_tiled = b->d_tiled;
	if (d_occ) *d_occ = b->d_occ;
	if (geometry) { geometry[0] = tiled_rows(b->g); geometry[1] = b->g.bits_pitch_w; geometry[2] = occ_pitch(b->g); geometry[3] = b->g.m_xoff; }
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_read_tile_mask(smhv_batch *b, uint32_t frame, uint32_t *tiled, uint8_t *occ, uint32_t *bits, int *written) {
	if (!b || frame >= b->max_frames) return fail(SMHV_E_INVALID, "bad arguments");
	HIPCHK(hipSetDevice(b->ctx->device));
	HIPCHK(hipDeviceSynchronize());
	const Geom &g = b->g;
	if (tiled) HIPCHK(hipMemcpy(tiled, b->d_tiled + (size_t)frame * tiled_stride_w(g), tiled_stride_w(g) * 4, hipMemcpyDeviceToHost));
	if (occ) HIPCHK(hipMemcpy(occ, b->d_occ + (size_t)frame * occ_stride(g), occ_stride(g), hipMemcpyDeviceToHost));
	if (bits) HIPCHK(hipMemcpy(bits, b->d_bits + (size_t)frame * g.bits_stride_w, g.bits_stride_w * 4, hipMemcpyDeviceToHost));
	if (written) {
		FrameAux a;
		HIPCHK(hipMemcpy(&a, b->d_aux + frame, sizeof a, hipMemcpyDeviceToHost));
		*written = a.tiles ? 1 : 0;
	}
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_read_results(smhv_batch *b, uint32_t first, uint32_t n, smhv_frame_result *out) {
	if (!b || !out || (uint64_t)first + n > b->max_frames) return fail(SMHV_E_INVALID, "bad arguments");
	HIPCHK(hipSetDevice(b->ctx->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(out, b->d_results + first, sizeof(smhv_frame_result) * n, hipMemcpyDeviceToHost));
	return batch_check_errors(b, "batch_read_results");       // (the records are in `out` either way)
}

// Tight host copy (rows x width bytes) of a pitched device image.  hipMemcpy2D turns into one small transfer per row
// when the row width is not a multiple of 4 bytes (measured: 4 ms for the 657 x 548 ocr image, 10 us per row), so the
// padded rows come over in ONE transfer into pinned staging `slot` and are repacked on the host.
static int copy_image_d2h(smhv_ctx *c, int slot, uint8_t *dst, const uint8_t *d_rows, size_t pitch, size_t xoff, size_t width, size_t rows, hipStream_t s) {
	const size_t bytes = pitch * rows;
	if (c->h_stage_cap[slot] < bytes) {
		if (c->h_stage[slot]) { (void)hipHostFree(c->h_stage[slot]); c->h_stage[slot] = nullptr; c->h_stage_cap[slot] = 0; }
		HIPCHK(hipHostMalloc((void **)&c->h_stage[slot], bytes, hipHostMallocDefault));
		c->h_stage_cap[slot] = bytes;
	}
	HIPCHK(hipMemcpyAsync(c->h_stage[slot], d_rows, bytes, hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	for (size_t r = 0; r < rows; ++r) memcpy(dst + r * width, c->h_stage[slot] + r * pitch + xoff, width);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_batch_read_image(smhv_batch *b, int which, uint32_t frame, uint8_t *out) {
	if (!b || !out || frame >= b->max_frames) return fail(SMHV_E_INVALID, "bad arguments");
	CTX_OPEN(b->ctx);                                         // the staging buffers belong to the context
	const Geom &g = b->g;
	HIPCHK(hipSetDevice(b->ctx->device));
	HIPCHK(hipDeviceSynchronize());
	switch (which) {
	case 100:
		HIPCHK(hipMemcpy2D(out, (size_t)g.rw * 4, b->d_ui + frame * g.ui_stride + (size_t)g.m_xoff * 4, g.ui_pitch, (size_t)g.rw * 4, g.rh, hipMemcpyDeviceToHost));
		break;
	case SMHV_VIEW_LSD_INPUT: {
		std::lock_guard<std::mutex> lk(b->ctx->mu);
		return copy_image_d2h(b->ctx, 2, out, b->d_mask + frame * g.mask_stride, g.mask_pitch, g.m_xoff, g.rw, g.rh, b->ctx->s_main);
	}
	case SMHV_VIEW_OCR_INPUT: {
		std::lock_guard<std::mutex> lk(b->ctx->mu);
		return copy_image_d2h(b->ctx, 2, out, b->d_ocr + frame * g.ocr_stride, g.ocr_pitch, g.q_xoff, g.qw, g.qh, b->ctx->s_main);
	}
	case SMHV_VIEW_FIND_SCALES_INPUT: {
		std::lock_guard<std::mutex> lk(b->ctx->mu);
		return copy_image_d2h(b->ctx, 2, out, b->d_scales + frame * g.ocr_stride, g.ocr_pitch, g.q_xoff, g.qw, g.qh, b->ctx->s_main);
	}
	default:
		return fail(SMHV_E_INVALID, "unsupported image id %d", which);
	}
	return SMHV_OK;
}

// ------------------------------------------------------------------------------------------------
// pipeline: `depth` batches in flight, each on its own stream; the library owns the schedule
// ------------------------------------------------------------------------------------------------
static std::atomic<int> g_own_queues[64];  // per device: streams with a hardware queue of their own, all live pipelines of the process (pipeline_create_impl)
struct smhv_pipeline {
	smhv_ctx *ctx = nullptr;
	uint32_t depth = 0;
	smhv_pipeline_options opt{};        // as given to smhv_pipeline_create_ex (all zero: the defaults)
	std::vector<smhv_batch *> batch;
	std::vector<hipStream_t> stream;    // batch-granular search: stream[slot] carries the slot's whole pass
	std::vector<hipEvent_t> done;       // end of the slot's most recent submission
	std::vector<hipEvent_t> hold;       // smhv_pipeline_hold: a consumer of the slot's outputs on some other stream
	std::vector<char> held;
	std::vector<hipStream_t> last_sl;   // stream on which the slot's most recent record kernel ran
	hipEvent_t ev_after = nullptr;
	// The occupancy policy suits a pipeline whose streaming pass and line search take comparable time (the synthetic
	// workload: 0.9 and 1.1 ms per launch).  Real screenshots in small batches are search-bound (a 128-frame launch waits
	// 4-5 ms for its slowest frame while its streaming pass takes 0.5): one search workgroup per CU is a cut for them.  So the
	// pipeline measures both (three events per submission, read when the slot comes round again) and switches the policy off
	// while the search takes more than SMH_ADAPT_OFF x the streaming pass, back on below SMH_ADAPT_ON x.
	LaunchTuning tuning{0u, 0u, 0u, 0u};
	bool adapt = false, tune_on = true;
	float ratio_ema = 0.0f;
	uint64_t submitted = 0;             // submissions so far
	uint64_t round_start = 0;           // index of the first submission after the pipeline last ran empty
	// ---- frame-granular search service (smh_kernels.h, smh_service.inc): depth >= 3 ----
	// stream[] then holds svc_streams streams that the submissions' streaming sides take in turn, s_search carries the
	// service kernel (a stream with its own hardware queue: nothing else may queue behind a kernel that lives for seconds)
	bool svc = false;
	uint32_t svc_compact = 0;           // the service's tile stores use the compact index (svc_waves_for)
	uint32_t svc_streams = 0, svc_waves = 0, svc_part_words = 0, svc_tile_cap = 0, svc_list_cap = 0, svc_lds = 0, svc_wgs = 0, svc_ring_log2 = 0;
	SvcCtl *d_svc_ctl = nullptr;
	unsigned long long *d_svc_ring = nullptr;
	SvcSlot *d_svc_slots = nullptr;
	SvcHost *h_svc = nullptr, *d_svc_host = nullptr;
	SvcRemote *d_svc_remote = nullptr;  // help across workgroups: one exchange block and one tile-store block per wave of the service
	uint32_t *d_svc_store = nullptr;
	uint32_t svc_store_words = 0;
	hipStream_t s_search = nullptr, s_pro = nullptr;   // (s_pro: anchor uploads and button tests, ahead of the streaming streams)
	std::mutex peek_mu;                 // smhv_debug_pipeline_peek (any thread)
	hipStream_t s_peek = nullptr;
	SvcCtl *h_peek = nullptr;
	std::vector<hipEvent_t> ev_pub;     // per slot: the slot's items have been published
	std::vector<hipEvent_t> ev_pro;     // per slot: its button test has run
	std::vector<uint32_t> seq;          // per slot: sequence number of its most recent submission (0: none yet)
	std::vector<hipStream_t> slot_st;   // per slot: the stream its most recent streaming side ran on
	uint32_t seq_counter = 0, svc_epoch = 0;
	bool svc_key_valid = false;         // the running / next service's sector table and gap threshold
	const uint32_t *svc_cull = nullptr;
	uint32_t svc_max_gap = 0;
	// ---- SMHV_SEARCH_AUTO at depth >= SMH_SVC_AUTO_DEPTH: both searches, and the pipeline MEASURES which one this workload
	// runs faster on.  The frame-granular service costs a third of the wave-time per frame but a frame is one wave's work from
	// start to end; a batch whose heaviest frame takes longer than the pipeline has slots for is faster on the batch-granular
	// search, eight waves per frame (measured, depth 12: synthetic 1080p scene 519 k against 419 k frames/s for the service;
	// the reference's 1440p screenshots, 0-372 rounds per frame, 163 k against 288 k for the batch-granular search).  Both
	// write byte-identical records, so the choice is a matter of speed only: the pipeline times a window of submissions in
	// each mode (host clock over 2 x depth submissions, after a warm-up), keeps the faster one, and looks again every
	// SMH_MODE_RECHECK submissions or when the shape of the submissions changes.  Submissions of both kinds may be in flight
	// at once (a slot remembers which kind its last submission was).
	std::vector<hipStream_t> svc_stream;   // the streaming streams of frame-granular submissions (stream[]: one per slot, batch-granular)
	std::vector<uint8_t> slot_frame;       // per slot: its most recent submission went to the service
	bool adaptive = false, mode_frame = false;
	uint32_t own_queues = 0;               // streams of this pipeline that were given a hardware queue of their own
	struct ModeCtl {
		uint32_t phase = 0, count = 0, settle = 0, key_n = 0, key_stages = 0, key_gap = 0, decisions = 0;
		uint32_t cand_n = 0, cand_stages = 0, cand_gap = 0, cand_count = 0;   // a shape other than the settled one, and for how many submissions in a row
		uint32_t remeasured = 0;               // measurements started by a change of shape since the last periodic one (capped)
		bool window_mixed = false;             // the shape changed inside a measurement window: that window does not count
		uint64_t frames = 0;
		struct timespec t0{};
		double rate[2] = {0.0, 0.0};       // frames/s measured in the last window of [0] batch-granular, [1] frame-granular
	} mc;
};
static int svc_wait_slot(smhv_pipeline *p, uint32_t slot);
// wait for the slot's most recent submission, whichever search it went to
static int slot_wait(smhv_pipeline *p, uint32_t slot) {
	if (p->svc && p->slot_frame[slot]) return svc_wait_slot(p, slot);
	HIPCHK(wait_event(p->done[slot]));
	return SMHV_OK;
}
static int svc_launch(smhv_pipeline *p, uint32_t slot);
static int svc_submit(smhv_pipeline *p, uint32_t slot, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                      const smhv_anchors *anchors, void *after_stream, uint32_t *slot_out);

extern "C" SMHV_API void smhv_pipeline_destroy(smhv_pipeline *p) {
	if (!p) return;
	if (p->ctx) (void)hipSetDevice(p->ctx->device);
	if (p->svc) for (uint32_t i = 0; i < p->depth && i < p->slot_frame.size(); ++i) if (p->slot_frame[i]) (void)svc_wait_slot(p, i);   // (a stalled service is relaunched by the wait)
	(void)hipDeviceSynchronize();                             // the service closes by itself once every submission is complete
	for (auto b : p->batch) if (b) smhv_batch_destroy(b);
	for (auto e : p->done) if (e) (void)hipEventDestroy(e);
	for (auto e : p->hold) if (e) (void)hipEventDestroy(e);
	if (p->ev_after) (void)hipEventDestroy(p->ev_after);
	for (auto st : p->stream) if (st) (void)hipStreamDestroy(st);
	for (auto st : p->svc_stream) if (st) (void)hipStreamDestroy(st);
	if (p->ctx) g_own_queues[(uint32_t)p->ctx->device & 63u].fetch_sub((int)p->own_queues, std::memory_order_relaxed);
	if (p->s_search) (void)hipStreamDestroy(p->s_search);
	if (p->s_pro) (void)hipStreamDestroy(p->s_pro);
	if (p->s_peek) (void)hipStreamDestroy(p->s_peek);
	if (p->h_peek) (void)hipHostFree(p->h_peek);
	for (auto e : p->ev_pub) if (e) (void)hipEventDestroy(e);
	for (auto e : p->ev_pro) if (e) (void)hipEventDestroy(e);
	if (p->d_svc_ctl) (void)hipFree(p->d_svc_ctl);
	if (p->d_svc_ring) (void)hipFree(p->d_svc_ring);
	if (p->d_svc_slots) (void)hipFree(p->d_svc_slots);
	if (p->d_svc_remote) (void)hipFree(p->d_svc_remote);
	if (p->d_svc_store) (void)hipFree(p->d_svc_store);
	if (p->h_svc) (void)hipHostFree(p->h_svc);
	ctx_release(p->ctx);
	delete p;
}

// Occupancy policy of the batches of a deep pipeline (measured on MI355X, DESIGN.md section 7: 256 x 1080p, depth 4, 389 k ->
// 453 k frames/s).  The streaming pass saturates HBM with two of its 4-wave workgroups per CU; the third and fourth only queue
// up in the memory system, and the wave slots and registers they hold are what the line searches of the other batches in
// flight are short of.  So each streaming workgroup RESERVES LDS it does not use: R bytes with 4 R > 160 KB (a fourth
// workgroup never fits: three measured like two, four are the uncapped kernel) and 2 R + L <= 160 KB, L = the LDS of one
// line-search workgroup with its tile store limited to
// SMH_PIPE_TILE_LIMIT tiles (200 up to 1080p, 320 above; a frame with more is searched on the mask in global memory).  The grid of
// the streaming pass is capped as well (its workgroups walk the items with a grid stride).  Frame sizes whose tile index
// leaves no room for that (4K and up) get no policy.
#define SMH_PIPE_TILE_LIMIT(g) ((g).rh > 900u ? 320u : 200u)   // a 1080p scene has 36-126 mask tiles, the 1440p screenshots up to 261
// ... and of the search service's waves: 272 above 1080p is what lets THREE waves of a service workgroup (36.6 KB each) fit a
// CU at 1440p instead of two (the reference's 1440p screenshots have up to 261 mask tiles)
#define SMH_SVC_TILE_LIMIT(g) ((g).rh > 900u ? 272u : 200u)
#define SMH_PIPE_MAP_GRID 1024u
#define SMH_LDS_PER_CU 163840u
#define SMH_ADAPT_OFF 3.5f              // line search / streaming pass, launch durations: above -> no occupancy policy
#define SMH_ADAPT_ON 1.6f               // ... below -> policy on again
#define SMH_LATE_KC_SEARCH_BOUND 400u   // late helpers of a search-bound pipeline: a frame asks for help after 0.17 ms
static LaunchTuning pipeline_tuning(const Geom &g) {
	LaunchTuning t{0u, 0u, 0u, 0u, 0u};
	const uint32_t lsd = (lsd_tile_lds_bytes(g, SMH_PIPE_TILE_LIMIT(g)) + 1023u) & ~1023u;    // (allocation granularity: be generous)
	if (lsd + 2048u >= SMH_LDS_PER_CU) return t;
	const uint32_t r = ((SMH_LDS_PER_CU - lsd - 2048u) / 2u) & ~1023u;                           // two streaming workgroups beside one line search
	// (a line search too large for "3 R > 160 KB" still gets a cap of three: measured a little below the cap of two)
	if (4u * r <= SMH_LDS_PER_CU || r <= map_brq_lds_bytes(g)) return t;                        // a fourth streaming workgroup would still fit: no cap at all
	t.map_lds_total = r; t.map_grid_cap = SMH_PIPE_MAP_GRID; t.lsd_tile_limit = SMH_PIPE_TILE_LIMIT(g);
	return t;
}

// Defaults of smhv_pipeline_options (all zero): which search a pipeline gets is decided from what was measured on MI355X
// (DESIGN.md).  One box, frames/s batch-granular / frame-granular: 256 x 1080p at depth 5 / 6 / 7 / 8: 340 / 328, 377 / 376,
// 407 / 421, 409 / 462 k; 128 x 1440p: 209 / 202, 233 / 232, 251 / 254, 254 / 273 k.  Below depth 6 the batch-granular search;
// from there on the pipeline has both and measures (mode_control).
#define SMH_SVC_AUTO_DEPTH 6u
#define SMH_OWN_QUEUES 16u
#define SMH_OWN_QUEUES_PROCESS 20u
static int pipeline_create_impl(smhv_ctx *c, uint32_t W, uint32_t H, uint32_t max_frames, uint32_t depth, const smhv_pipeline_options *opt_in, smhv_pipeline **out) {
	if (!c || !out || max_frames == 0 || depth == 0 || depth > SVC_MAX_SLOTS) return fail(SMHV_E_INVALID, "pipeline_create: bad arguments (depth 1..%u)", SVC_MAX_SLOTS);
	*out = nullptr;
	smhv_pipeline_options opt;
	memset(&opt, 0, sizeof opt);
	if (opt_in) {
		if (opt_in->size < 2 * sizeof(uint32_t) || opt_in->size > 4096u) return fail(SMHV_E_INVALID, "pipeline_create_ex: options.size is not set");
		memcpy(&opt, opt_in, opt_in->size < sizeof opt ? opt_in->size : sizeof opt);   // (a caller built against an older, shorter struct)
	}
	if (opt.search > SMHV_SEARCH_FRAME || opt.occupancy_policy > 2u || opt.late_helpers > 2u || opt.streams > 8u || opt.room_for_others > 2u)
		return fail(SMHV_E_INVALID, "pipeline_create_ex: bad option value");
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	Geom g0;
	{ int rc = compute_geom(W, H, &g0); if (rc) return rc; }
	smhv_pipeline *p = new (std::nothrow) smhv_pipeline();
	if (!p) return fail(SMHV_E_INVALID, "out of host memory");
	c->refs.fetch_add(1, std::memory_order_relaxed);
	p->ctx = c; p->depth = depth; p->opt = opt;
	// Frame-granular search (smh_kernels.h): depth >= 3 and a frame size whose tile store fits beside the streaming pass
	if (opt.search != SMHV_SEARCH_BATCH && depth >= 3 && (opt.search == SMHV_SEARCH_FRAME || depth >= SMH_SVC_AUTO_DEPTH) && max_frames < (1u << 24)) {
		p->svc_waves = svc_waves_for(g0, SMH_SVC_TILE_LIMIT(g0), &p->svc_part_words, &p->svc_tile_cap, &p->svc_list_cap, &p->svc_lds, &p->svc_compact);
		p->svc = p->svc_waves > 0u;
	}
	if (opt.search == SMHV_SEARCH_FRAME && !p->svc) {
		ctx_release(c);
		delete p;
		return fail(SMHV_E_INVALID, "pipeline_create_ex: the frame-granular search needs depth >= 3 and a frame size whose mask tiles fit the LDS beside the streaming pass (%ux%u, depth %u)", W, H, depth);
	}
	if (p->svc) {
		// The service's life cycle is a handshake of 64-bit system-scope atomics on mapped host memory (SvcHost).  A platform that
		// does not route them (no PCIe atomics: some hosts, VMs, passthrough) would leave a launch that can never close: ask the
		// runtime, then try one.  Without them: the batch-granular search (an explicit SMHV_SEARCH_FRAME is an error).
		bool atomics_ok = false;
		int native = 0;
		hipError_t ea = hipHostMalloc((void **)&p->h_svc, sizeof(SvcHost), hipHostMallocMapped | hipHostMallocCoherent);
		if (ea == hipSuccess) { memset(p->h_svc, 0, sizeof(SvcHost)); ea = hipHostGetDevicePointer((void **)&p->d_svc_host, p->h_svc, 0); }
		if (ea == hipSuccess && hipDeviceGetAttribute(&native, hipDeviceAttributeHostNativeAtomicSupported, c->device) != hipSuccess) { native = -1; (void)hipGetLastError(); }
		if (ea == hipSuccess) ea = svc_probe_host_atomics(p->h_svc, p->d_svc_host, &atomics_ok);
		if (g_no_host_atomics.load(std::memory_order_relaxed)) atomics_ok = false;
		if (ea != hipSuccess || !atomics_ok) {
			(void)hipGetLastError();
			logf(c, 2, "pipeline: device-side atomics on mapped host memory %s (hipDeviceAttributeHostNativeAtomicSupported = %d): no frame-granular search",
			     ea != hipSuccess ? hipGetErrorString(ea) : "do not take effect", native);
			if (p->h_svc) { (void)hipHostFree(p->h_svc); p->h_svc = nullptr; p->d_svc_host = nullptr; }
			p->svc = false;
			if (opt.search == SMHV_SEARCH_FRAME) {
				ctx_release(c);
				delete p;
				return fail(SMHV_E_INVALID, "pipeline_create_ex: the frame-granular search needs device-side 64-bit atomics on mapped host memory (PCIe atomics), which this platform does not provide");
			}
		}
	}
	p->batch.assign(depth, nullptr); p->done.assign(depth, nullptr); p->hold.assign(depth, nullptr); p->held.assign(depth, 0); p->last_sl.assign(depth, nullptr);
	hipError_t e = hipSuccess;
	{
		// Every stream of the pipeline gets a hardware queue of its own: a stream created with a CU mask (here: all CUs) does,
		// the mask being a property of the queue.  Ordinary streams are dealt onto the process's four hardware queues when they
		// are first USED, in an order that depends on what else the process has done with the device by then -- measured: a
		// depth-4 pipeline created before the process's first other use of the device had two of its four chains on one
		// queue (kernel trace: 1111 / 555 / 555 kernels on three queues) and ran the sample screenshots at 104 k frames/s, the
		// same pipeline created after one unrelated 16-byte copy had four queues and 203 k.  (The first streams of a pipeline, within a budget:
		// below.)  The search service's kernel lives as
		// long as the pipeline is busy: whatever shared its queue would wait that long.
		p->svc_streams = p->svc ? std::min<uint32_t>(depth, opt.streams ? opt.streams : 2u) : 0u;
		p->adaptive = p->svc && opt.search == SMHV_SEARCH_AUTO;
		p->mode_frame = p->svc;
		p->slot_frame.assign(depth, 0);
		const uint32_t ns = (!p->svc || p->adaptive) ? depth : 0u;   // one stream per slot for batch-granular submissions
		const uint32_t full[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
		p->stream.assign(ns, nullptr);
		// (a budget: SMH_OWN_QUEUES queues per pipeline, the service's included, and SMH_OWN_QUEUES_PROCESS for all live pipelines
		// of the process on the same device -- with 21 queues the hardware scheduler started to time-slice them: stalls of 80 ms
		// in the kernel trace of a depth-16 pipeline, and a second 16-queue pipeline beside the first ran at half its rate.
		// Streams beyond the budget are ordinary ones; the search kernel's stream always has its own queue.)
		auto own_queue = [p]() {
			if (p->own_queues >= SMH_OWN_QUEUES) return false;
			std::atomic<int> &cnt = g_own_queues[(uint32_t)p->ctx->device & 63u];
			if (cnt.fetch_add(1, std::memory_order_relaxed) >= (int)SMH_OWN_QUEUES_PROCESS) { cnt.fetch_sub(1, std::memory_order_relaxed); return false; }
			p->own_queues++;
			return true;
		};
		if (p->svc) {
			p->svc_stream.assign(p->svc_streams, nullptr);
			(void)own_queue();
			e = create_stream(&p->s_search, full);
			for (uint32_t i = 0; i < p->svc_streams && e == hipSuccess; ++i) e = create_stream(&p->svc_stream[i], own_queue() ? full : nullptr);
			if (e == hipSuccess && !(opt.flags & SMHV_PIPE_NO_PROLOGUE)) e = create_stream(&p->s_pro, own_queue() ? full : nullptr);
		}
		for (uint32_t i = 0; i < ns && e == hipSuccess; ++i) e = create_stream(&p->stream[i], own_queue() ? full : nullptr);
	}
	if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_after, hipEventDisableTiming);
	for (uint32_t i = 0; i < depth && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming);
	for (uint32_t i = 0; i < depth && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&p->hold[i], hipEventDisableTiming);
	if (p->svc && e == hipSuccess) {
		p->ev_pub.assign(depth, nullptr); p->ev_pro.assign(depth, nullptr); p->seq.assign(depth, 0u); p->slot_st.assign(depth, nullptr);
		for (uint32_t i = 0; i < depth && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&p->ev_pub[i], hipEventDisableTiming);
		for (uint32_t i = 0; i < depth && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&p->ev_pro[i], hipEventDisableTiming);
		uint32_t lg = 8;
		while ((1ull << lg) < (uint64_t)depth * max_frames) ++lg;
		p->svc_ring_log2 = lg;
		int cus = 0;
		if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
		// One service workgroup per CU where four waves of it fit beside the streaming pass (up to 1080p).  Larger frames (three
		// waves per workgroup at 1440p, most of a CU's LDS all the same) do better with the service on five eighths of the CUs
		// and the others left to the streaming pass alone -- measured at 128 x 1440p, depth 12, one box, three waves per
		// workgroup: 256 / 224 / 192 / 176 / 160 / 144 / 128 / 112 workgroups: 225 / 239 / 254 / 261 / 271 / 272 / 251 / 227 k
		// frames/s (two waves per workgroup: 229 k with 256, 252 k with 192; 1080p, four waves: 540 / 531 / 514 / - / 490 k: every CU).
		// room_for_others: an eighth of the CUs stays without a service workgroup (measured with a 21 KB-LDS / 280-VGPR probe kernel
		// launched once per submission beside a saturated 1080p pipeline: 224 of 256 workgroups -- every probe on the chip within
		// 0.9 ms, the pipeline 0.3 % slower than without probes; 240 or 248 -- sometimes fine, sometimes seconds)
		const int svc_cus = opt.room_for_others == 1u ? cus - std::max(cus / 8, 1) : cus;
		// (above 1080p -- the frame sizes whose tile stores sit behind the compact index -- a service workgroup takes 130 KB of its CU's
		// LDS whatever its waves: 128 x 1440p at depth 12, four waves per workgroup, 256 / 224 / 192 / 176 / 160 / 144 / 128 / 112
		// workgroups: 212 / 225 / 245 / 255 / 263 / 271 / 267 / 243 k frames/s on the synthetic scene, 225 / 247 / 260 / 260 / 256 / 244 /
		// 226 / 211 k on the reference's screenshots in round 5.  Round 6 -- the tile store built from the pass's tile-major mask, a
		// frame 20 % cheaper -- 192 / 160 / 144 / 128 / 112 / 96 workgroups: - / 269 / 275 / 282 / 293 / 262 k synthetic, 261 / 276 / 273 / 253 /
		// 231 / - k on the screenshots; and with the walk over the bit rows (those frame sizes keep it: bands of whole tile rows for the tile-major mask
		// would cost the pass a band, band_rows_for) 144 against 160 workgroups: 296-297 against 287-288 k synthetic, 272-274 against 276 k on the
		// screenshots: nine sixteenths of the CUs)
		const bool every_cu = p->svc_waves >= 4u && !p->svc_compact;
		p->svc_wgs = opt.service_workgroups ? opt.service_workgroups : (uint32_t)std::max(every_cu ? svc_cus : std::min(cus * 9 / 16, svc_cus), 1);
		if (e == hipSuccess) e = hipMalloc((void **)&p->d_svc_ctl, sizeof(SvcCtl));
		if (e == hipSuccess) e = hipMemset(p->d_svc_ctl, 0, sizeof(SvcCtl));
		if (e == hipSuccess) e = hipMalloc((void **)&p->d_svc_ring, sizeof(unsigned long long) << lg);
		if (e == hipSuccess) e = hipMemset(p->d_svc_ring, 0, sizeof(unsigned long long) << lg);
		if (e == hipSuccess) e = hipMalloc((void **)&p->d_svc_slots, sizeof(SvcSlot) * depth);
		if (e == hipSuccess) e = hipMemset(p->d_svc_slots, 0, sizeof(SvcSlot) * depth);
		if (!(opt.flags & (SMHV_PIPE_NO_REMOTE_HELP | SMHV_PIPE_NO_TEAM_HELP)) && (uint64_t)p->svc_wgs * p->svc_waves <= 0xFFFFu) {
			// help across workgroups (smh_kernels.h): per wave of the launch an exchange block and room for its frame's tile store
			const size_t owners = (size_t)p->svc_wgs * p->svc_waves;
			p->svc_store_words = svc_store_words_for(g0, p->svc_tile_cap, p->svc_compact);
			if (e == hipSuccess) e = hipMalloc((void **)&p->d_svc_remote, sizeof(SvcRemote) * owners);
			if (e == hipSuccess) e = hipMemset(p->d_svc_remote, 0, sizeof(SvcRemote) * owners);
			if (e == hipSuccess) e = hipMalloc((void **)&p->d_svc_store, sizeof(uint32_t) * owners * p->svc_store_words);
		}
	}
	if (e != hipSuccess) { smhv_pipeline_destroy(p); return fail(SMHV_E_HIP, "pipeline streams / events: %s", hipGetErrorString(e)); }
	for (uint32_t i = 0; i < depth; ++i) {
		int rc = smhv_batch_create(c, W, H, max_frames, &p->batch[i]);
		if (rc) { smhv_pipeline_destroy(p); return rc; }
		// Batch-granular search, measured on MI355X: frames whose mask window fits the LDS (<= 1080p) -- depth 1:
		// k_lsd with helper workgroups, depth 2: k_lsd, depth >= 3: k_lsd_tile with 512-thread workgroups; larger frames:
		// k_lsd_tile always.
		p->batch[i]->lsd_bs = depth >= 2 ? 512u : 1024u;
		p->batch[i]->lsd_prefer_classic = depth == 2 && lsd_rows_only(p->batch[i]->g);
		if (p->svc && !p->adaptive) {
			// no occupancy policy: the service's resident waves (one per SIMD, 136-168 registers each) are what caps the streaming
			// pass beside them -- at three workgroups per CU in its two-set form (112 registers), two in the three-set form (128).
			// Round 5 ran the three-set form up to 1080p (the third workgroup cost the search more than it gave the pass); since the
			// service builds its tile stores from the pass's tile-major mask it has the slack, and two sets are ahead everywhere
			// (256 x 1080p, one box: 571-579 against 529-530 k frames/s).  SMHV_PIPE_THREE_LOAD_SETS: the old form, for A/B.
			// The streaming waves go first on their SIMD
			p->batch[i]->tune.map_prio = (opt.flags & SMHV_PIPE_NO_STREAM_PRIORITY) ? 0u : 1u;
			p->batch[i]->tune.map_deep = (opt.flags & SMHV_PIPE_THREE_LOAD_SETS) ? 1u : 0u;   // (a service workgroup on every CU: launch_map_brq_pass)
			p->batch[i]->tune.map_overlapped = 1u;
		} else if (depth >= 3) {                                  // (an adaptive pipeline sets the tuning of a slot per submission)
			if (opt.occupancy_policy != 2u) p->tuning = pipeline_tuning(p->batch[i]->g);
			p->batch[i]->tune = p->tuning;
			p->batch[i]->tune.map_overlapped = 1u;
			if (opt.late_helpers == 1u) p->batch[i]->lsd_late_kc = 20u;           // every frame asks at once
			if (opt.occupancy_policy == 0u) {
				p->adapt = true;
				hipError_t e2 = hipSuccess;
				for (int k = 0; k < 3 && e2 == hipSuccess; ++k) e2 = hipEventCreate(&p->batch[i]->ev_probe[k]);
				if (e2 != hipSuccess) { smhv_pipeline_destroy(p); return fail(SMHV_E_HIP, "pipeline probe events: %s", hipGetErrorString(e2)); }
				p->batch[i]->probe = true;
			}
		}
	}
	*out = p;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_pipeline_create(smhv_ctx *c, uint32_t W, uint32_t H, uint32_t max_frames, uint32_t depth, smhv_pipeline **out) {
	return pipeline_create_impl(c, W, H, max_frames, depth, nullptr, out);
}

extern "C" SMHV_API int smhv_pipeline_create_ex(smhv_ctx *c, uint32_t W, uint32_t H, uint32_t max_frames, uint32_t depth, const smhv_pipeline_options *opt, smhv_pipeline **out) {
	return pipeline_create_impl(c, W, H, max_frames, depth, opt, out);
}

// ---- the host side of the frame-granular search service --------------------------------------------------------------
static inline bool svc_alive(smhv_pipeline *p) { return (uint32_t)__atomic_load_n(&p->h_svc->state, __ATOMIC_ACQUIRE) != 0u; }

// Launch the service kernel behind the publication of `slot`'s items.  The alive flag is set first: from then on only the
// device clears it (smh_service.inc, svc_pop).
static int svc_launch(smhv_pipeline *p, uint32_t slot) {
	SvcParams sp;
	sp.ctl = p->d_svc_ctl; sp.ring = p->d_svc_ring; sp.slots = p->d_svc_slots; sp.host = p->d_svc_host;
	sp.cull_tab = p->svc_cull; sp.ray_off = p->ctx->d_ray_off; sp.max_gap = (float)p->svc_max_gap;
	sp.tile_cap = p->svc_tile_cap; sp.list_cap = p->svc_list_cap; sp.part_words = p->svc_part_words; sp.ring_log2 = p->svc_ring_log2;
	sp.epoch = ++p->svc_epoch;
	if (sp.epoch == 0u) sp.epoch = ++p->svc_epoch;
	sp.idle_short = p->opt.idle_close_us ? p->opt.idle_close_us * 2u : 100u;   // x 1024 cycles: ~45 us without work and nothing outstanding
	sp.flags = (p->opt.flags & SMHV_PIPE_NO_TEAM_HELP) ? 8u : 0u;
	sp.idle_long = 50000u;                                               // ~20 ms without work, nobody at work: the streaming side is stuck
	if (p->opt.flags & SMHV_PIPE_HELP_FIRST) sp.flags |= 16u;
	if (p->opt.flags & SMHV_PIPE_WALK_BIT_ROWS) sp.flags |= 32u;

	sp.remote = p->d_svc_remote; sp.remote_store = p->d_svc_store; sp.remote_store_words = p->svc_store_words;
	sp.remote_after = p->opt.remote_after ? p->opt.remote_after : 24u;
	sp.remote_tickets = std::min<uint32_t>(p->opt.remote_tickets ? p->opt.remote_tickets : 3u, 16u);
	sp.remote_last_div = p->opt.remote_last ? p->opt.remote_last : 6u;
	sp.compact = p->svc_compact;
	__atomic_fetch_or(&p->h_svc->state, (unsigned long long)sp.epoch, __ATOMIC_ACQ_REL);   // (the low half is 0: only then is this called)
	hipError_t e = hipStreamWaitEvent(p->s_search, p->ev_pub[slot], 0);
	if (e == hipSuccess) e = launch_lsd_service(p->batch[slot]->g, sp, p->svc_wgs, p->svc_waves, p->svc_lds, p->s_search);
	if (e != hipSuccess) {
		__atomic_fetch_and(&p->h_svc->state, 0xFFFFFFFF00000000ull, __ATOMIC_ACQ_REL);
		return fail(SMHV_E_HIP, "launching the line-search service: %s", hipGetErrorString(e));
	}
	p->h_svc->launches++;
	return SMHV_OK;
}

// Host wait for the slot's most recent submission: the wave that finishes its last frame stores the submission's sequence
// number into mapped host memory.
static int svc_wait_slot(smhv_pipeline *p, uint32_t slot) {
	const uint32_t target = p->seq[slot];
	if (target == 0u) return SMHV_OK;
	volatile uint32_t *flag = &p->h_svc->done_seq[slot];
	uint64_t spins = 0;
	struct timespec t0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	// (sequence numbers only grow, modulo 2^32: "reached" is a signed difference, so a slot whose flag has moved PAST the number
	// waited for -- a submission the host gave up on but the device completed -- still reads as complete)
	auto reached = [flag, target]() { return (int32_t)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - target) >= 0; };
	while (!reached()) {
		if (!svc_alive(p)) {
			// no launch alive with work outstanding: the service gave way (its streaming side did not move for idle_long): again
			if (reached()) break;
			int rc = svc_launch(p, slot);
			if (rc) return rc;
		}
		if (++spins < 4000u) { __builtin_ia32_pause(); continue; }
		struct timespec ts = {0, 20000};
		nanosleep(&ts, nullptr);
		if ((spins & 1023u) == 0u) {
			struct timespec t1;
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if (t1.tv_sec - t0.tv_sec > 60) {
				const unsigned long long st = __atomic_load_n(&p->h_svc->state, __ATOMIC_ACQUIRE);
				return fail(SMHV_E_STATE, "pipeline_wait: slot %u (sequence %u, last completed %u) did not complete within 60 s: line-search service stalled "
				            "(submissions %u, launch alive %u, launches %u)", slot, target, *flag, (uint32_t)(st >> 32), (uint32_t)st, p->h_svc->launches);
			}
		}
	}
	return SMHV_OK;
}

static int svc_submit(smhv_pipeline *p, uint32_t slot, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                      const smhv_anchors *anchors, void *after_stream, uint32_t *slot_out) {
	smhv_batch *b = p->batch[slot];
	if (!d_frames || n == 0 || n > b->max_frames) return fail(SMHV_E_INVALID, "bad arguments (n=%u, capacity %u)", n, b->max_frames);
	// the slot's previous submission owns its buffers until its last frame has been counted off
	int rc = slot_wait(p, slot);
	if (rc) return rc;
	if (batch_check_errors(b, "pipeline_submit (the slot's previous submission, never waited for)") != SMHV_OK) logf(p->ctx, 2, "%s", t_last_error.c_str());
	hipStream_t st = p->svc_stream[p->submitted % p->svc_streams];
	if (p->adaptive) { b->tune = LaunchTuning{0u, 0u, 0u, (p->opt.flags & SMHV_PIPE_NO_STREAM_PRIORITY) ? 0u : 1u, (p->opt.flags & SMHV_PIPE_THREE_LOAD_SETS) ? 1u : 0u, 1u}; b->probe = false; b->lsd_late_kc = 0u; }
	// the sector table of this gap threshold is a launch parameter of the service: a submission with another one waits for
	// the service to finish what it has and close (a host that alternates thresholds pays a drain per change)
	Buffers probe{};
	const bool want_cull = (stages & SMHV_STAGE_MARKERS) && !(stages & SMHV_STAGE_EXACT_STATS);
	if (want_cull) { rc = sector_table_for(p->ctx, max_gap, st, &probe); if (rc) return rc; }
	if (!p->svc_key_valid || p->svc_cull != probe.cull_tab || p->svc_max_gap != max_gap) {
		for (uint32_t i = 0; i < p->depth; ++i) { rc = svc_wait_slot(p, i); if (rc) return rc; }
		struct timespec td0;
		clock_gettime(CLOCK_MONOTONIC, &td0);
		for (uint64_t spins = 0; svc_alive(p); ++spins) {      // (everything is complete: the launch closes within idle_short; bounded like svc_wait_slot)
			struct timespec ts = {0, 20000}, td1;
			nanosleep(&ts, nullptr);
			if ((spins & 1023u) == 1023u) {
				clock_gettime(CLOCK_MONOTONIC, &td1);
				if (td1.tv_sec - td0.tv_sec > 60) {
					const unsigned long long stw = __atomic_load_n(&p->h_svc->state, __ATOMIC_ACQUIRE);
					return fail(SMHV_E_STATE, "pipeline_submit: the line-search service did not close within 60 s of its last submission (another gap threshold needs a new launch; "
					            "submissions %u, launch alive %u, launches %u)", (uint32_t)(stw >> 32), (uint32_t)stw, p->h_svc->launches);
				}
			}
		}
		p->svc_key_valid = true; p->svc_cull = probe.cull_tab; p->svc_max_gap = max_gap;
	}
	// (what the submission has to wait for gates its first kernel: the button test, on the prologue stream when there is one;
	// the streaming stream waits for that test)
	const hipStream_t s_first = p->s_pro ? p->s_pro : st;
	if (after_stream) {
		HIPCHK(hipEventRecord(p->ev_after, (hipStream_t)after_stream));
		HIPCHK(hipStreamWaitEvent(s_first, p->ev_after, 0));
	}
	if (p->held[slot]) {
		HIPCHK(hipStreamWaitEvent(s_first, p->hold[slot], 0));
		p->held[slot] = 0;
	}
	uint32_t seq = ++p->seq_counter;
	if (seq == 0u) seq = ++p->seq_counter;
	bool published = false;
	SvcPublish pub{p->d_svc_ctl, p->d_svc_ring, p->d_svc_slots, slot, seq, p->svc_ring_log2, probe.cull_tab, true, p->s_pro, p->ev_pro[slot], &published};
	// The submission is counted BEFORE its kernels are enqueued: from here on the service does not regard itself as drained
	// (were it counted afterwards, its items could be there -- and a wave at work on them -- while the count still said
	// "everything complete", and the service would close under that wave).
	__atomic_fetch_add(&p->h_svc->state, 1ull << 32, __ATOMIC_ACQ_REL);
	rc = batch_run_impl(b, d_frames, n, stages, grayscale, max_gap, anchors, st, st, &pub);
	if (rc && !published) {
		__atomic_fetch_sub(&p->h_svc->state, 1ull << 32, __ATOMIC_ACQ_REL);   // nothing of it was published: take the count back
		return rc;
	}
	// (an error AFTER the publication kernel went in -- a failing event record, a sticky error of some earlier call: the device
	// completes the submission all the same, so it stays counted and the slot waits for ITS sequence number; the error is
	// still the caller's)
	const int rc_late = rc;
	p->seq[slot] = seq;
	p->slot_st[slot] = st;
	p->last_sl[slot] = st;
	p->slot_frame[slot] = 1;
	// whoever finds no launch alive starts one, ordered behind this submission's items (the event is recorded only then: a
	// marker between two kernels of the streaming chain costs every submission a slower hand-over)
	if (!svc_alive(p)) {
		HIPCHK(hipEventRecord(p->ev_pub[slot], st));
		rc = svc_launch(p, slot);
		if (rc) return rc;
	}
	p->submitted++;
	if (slot_out) *slot_out = slot;
	return rc_late;
}


// The mode controller of an adaptive pipeline (smhv_pipeline::ModeCtl): called at the top of every submission.
//   phase 0  warm-up in the current mode (2 depth submissions)
//   phase 1  measure the current mode: frames submitted / host-clock time over W = 8 depth submissions -> rate[mode].  A full
//            pipeline lets a submission in when its slot's previous one has completed, so over a long window the cadence of
//            the calls IS the throughput; over a short one it is not (the calls return in bursts: 65 to 2343 k frames/s over
//            windows of 16 submissions of a pipeline that runs at 420 k).  Draining the pipeline around the window instead
//            measures the ramps: a batch-granular pipeline restarts staggered, one streaming pass after the other.
//   phase 2  the other mode, warm-up (6 depth submissions: the other search's submissions drain, and a batch-granular pipeline
//            also has to find its occupancy policy)
//   phase 3  measure it the same way -> rate[other]; keep the faster mode (the batch-granular search has to be 3 % ahead: on a
//            tie -- a host that submits slower than either search runs, batches of one frame -- the service is the cheaper
//            one for the host, one small publication kernel per submission instead of a search launch)
//   phase 4  settled for SMH_MODE_RECHECK submissions, or until the submissions change shape for good: frames per submission by
//            more than 25 %, stages or gap threshold, for `depth` submissions in a row (at most SMH_MODE_REMEASURE_CAP such
//            measurements between two periodic ones); a window inside which the shape changed is measured again
#define SMH_MODE_RECHECK 16384u
// A submission's shape as the controller sees it: frames per submission in buckets of +-25 % (a host whose batches vary -- ingest
// slabs after duplicate drops, a short last batch -- is ONE workload), stages and gap threshold exactly.
static inline bool same_shape(uint32_t n_a, uint32_t st_a, uint32_t gap_a, uint32_t n_b, uint32_t st_b, uint32_t gap_b) {
	if (st_a != st_b || gap_a != gap_b) return false;
	const uint64_t lo = std::min(n_a, n_b), hi = std::max(n_a, n_b);
	return hi * 4u <= lo * 5u;                                 // within 25 %
}
#define SMH_MODE_REMEASURE_CAP 4u       // measurements a change of shape may start between two periodic ones
static int mode_control(smhv_pipeline *p, uint32_t n, uint32_t stages, uint32_t max_gap) {
	smhv_pipeline::ModeCtl &m = p->mc;
	const uint32_t W = 8u * p->depth;
	auto begin = [&m]() { m.count = 0; m.frames = 0; m.window_mixed = false; clock_gettime(CLOCK_MONOTONIC, &m.t0); };
	auto rate = [&m]() {
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		const double dt = (double)(t1.tv_sec - m.t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - m.t0.tv_nsec);
		return dt > 0.0 ? (double)m.frames / dt : 0.0;
	};
	const bool same = m.key_n == 0u || same_shape(n, stages, max_gap, m.key_n, m.key_stages, m.key_gap);
	if (m.key_n == 0u) { m.key_n = n; m.key_stages = stages; m.key_gap = max_gap; }
	if (m.phase == 4) {
		// settled: a new shape has to PERSIST (depth submissions in a row) before it starts a new measurement, and only
		// SMH_MODE_REMEASURE_CAP of those between two periodic ones -- a host that alternates shapes keeps what it has
		if (same) m.cand_count = 0;
		else {
			if (m.cand_count && same_shape(n, stages, max_gap, m.cand_n, m.cand_stages, m.cand_gap)) m.cand_count++;
			else { m.cand_n = n; m.cand_stages = stages; m.cand_gap = max_gap; m.cand_count = 1; }
			if (m.cand_count >= p->depth && m.remeasured < SMH_MODE_REMEASURE_CAP) {
				m.key_n = m.cand_n; m.key_stages = m.cand_stages; m.key_gap = m.cand_gap;
				m.cand_count = 0; m.remeasured++;
				m.phase = 0; m.count = 0;
			}
		}
	} else if (!same) m.window_mixed = true;                   // (a window that saw two shapes is measured again)
	int rc = SMHV_OK;
	switch (m.phase) {
	case 0:
		if (m.count >= 2u * p->depth) { m.phase = 1; begin(); }
		break;
	case 1:
		if (m.count >= W) {
			if (m.window_mixed) { m.key_n = n; m.key_stages = stages; m.key_gap = max_gap; begin(); break; }
			m.rate[p->mode_frame ? 1 : 0] = rate(); p->mode_frame = !p->mode_frame; m.phase = 2; m.count = 0;
		}
		break;
	case 2:
		if (m.count >= 6u * p->depth) { m.phase = 3; begin(); }
		break;
	case 3:
		if (m.count >= W) {
			if (m.window_mixed) { m.key_n = n; m.key_stages = stages; m.key_gap = max_gap; begin(); break; }
			m.rate[p->mode_frame ? 1 : 0] = rate();
			p->mode_frame = !(m.rate[0] > 1.03 * m.rate[1]);
			m.decisions++;
			m.phase = 4; m.count = 0; m.settle = SMH_MODE_RECHECK;
			logf(p->ctx, 3, "pipeline: frame-granular search %.0f frames/s, batch-granular %.0f -> %s", m.rate[1], m.rate[0], p->mode_frame ? "frame-granular" : "batch-granular");
		}
		break;
	default:
		if (m.count >= m.settle) { m.phase = 1; m.remeasured = 0; begin(); }
		break;
	}
	return rc;
}

extern "C" SMHV_API int smhv_pipeline_submit(smhv_pipeline *p, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                                             const smhv_anchors *anchors, void *after_stream, uint32_t *slot_out) {
	if (!p) return fail(SMHV_E_INVALID, "null pipeline");
	CTX_OPEN(p->ctx);
	HIPCHK(hipSetDevice(p->ctx->device));
	const uint32_t slot = (uint32_t)(p->submitted % p->depth);
	if (p->adaptive) { int rc = mode_control(p, n, stages, max_gap); if (rc) return rc; }
	if (p->svc && p->mode_frame) {
		int rc = svc_submit(p, slot, d_frames, n, stages, grayscale, max_gap, anchors, after_stream, slot_out);
		if (!rc && p->adaptive) { p->mc.count++; p->mc.frames += n; }
		return rc;
	}
	// the slot's previous submission (depth submissions ago) owns its output buffers until it has finished: this is the
	// only place the call can wait, and only when more than `depth` submissions would be in flight
	{ int rc = slot_wait(p, slot); if (rc) return rc; }
	// frames the slot's previous submission gave up, if nobody waited for it: reported (logged) now, with that submission,
	// not by some later wait with the wrong run's frame index
	if (batch_check_errors(p->batch[slot], "pipeline_submit (the slot's previous submission, never waited for)") != SMHV_OK)
		logf(p->ctx, 2, "%s", t_last_error.c_str());
	if (p->adapt) {
		smhv_batch *bb = p->batch[slot];
		if (bb->probe_valid) {                                // the slot's previous submission has finished: its three events are there
			float map_ms = 0.0f, lsd_ms = 0.0f;
			if (hipEventElapsedTime(&map_ms, bb->ev_probe[0], bb->ev_probe[1]) == hipSuccess &&
			    hipEventElapsedTime(&lsd_ms, bb->ev_probe[1], bb->ev_probe[2]) == hipSuccess && map_ms > 0.0f) {
				const float ratio = lsd_ms / map_ms;
				p->ratio_ema = p->ratio_ema == 0.0f ? ratio : 0.75f * p->ratio_ema + 0.25f * ratio;
				if (p->tune_on && p->ratio_ema > SMH_ADAPT_OFF) p->tune_on = false;
				else if (!p->tune_on && p->ratio_ema < SMH_ADAPT_ON) p->tune_on = true;
			}
			bb->probe_valid = false;
		}
		bb->tune = p->tune_on ? p->tuning : LaunchTuning{0u, 0u, 0u, 0u, 0u, 0u};
		bb->tune.map_overlapped = 1u;
		// ... and a search-bound pipeline lets the workgroups of k_lsd_tile that have finished their frame help the ones still at
		// work after SMH_LATE_KC_SEARCH_BOUND thousand cycles (sample screenshots, batch 128: 96 -> 106 k frames/s at depth 4,
		// 127 -> 138 k at depth 8; the synthetic pipeline, ratio 1.3, loses 3 % with them and keeps them off)
		bb->lsd_late_kc = p->opt.late_helpers == 2u ? 0u : (p->opt.late_helpers == 1u ? 20u : (p->tune_on ? 0u : SMH_LATE_KC_SEARCH_BOUND));
		// a sample every fourth round of the slots is plenty for a running average, and the three timed events sit in the batch's
		// chain on its hardware queue (the hand-over before the search: 33 us with them, 14 without)
		bb->probe = (p->submitted / p->depth) % 4u == 0u || (p->adaptive && p->mc.phase != 4u);
	}
	if (p->adaptive && !p->adapt) {                           // (a fixed policy: what the slot's last frame-granular submission changed)
		p->batch[slot]->tune = p->tuning;
		p->batch[slot]->tune.map_overlapped = 1u;
		p->batch[slot]->lsd_late_kc = p->opt.late_helpers == 1u ? 20u : 0u;
	}
	hipStream_t st, sl;
	{
		st = sl = p->stream[slot];
		bool idle = true;
		for (uint32_t i = 0; i < p->depth && idle; ++i) idle = hipEventQuery(p->done[i]) == hipSuccess;
		if (idle) p->round_start = p->submitted;
		// Stagger: in the first round after the device ran empty, submission k starts its streaming pass when submission k-1
		// has finished its own, so the batches run half a period apart from the outset -- the streaming pass of one underneath
		// the line-segment search of the other -- instead of locking into streaming together and then searching together.
		const uint64_t k = p->submitted - p->round_start;
		if (k > 0 && k < p->depth) {
			const uint32_t prev = (uint32_t)((p->submitted - 1) % p->depth);
			HIPCHK(hipStreamWaitEvent(st, p->batch[prev]->ev_map_done, 0));
		}
	}
	if (after_stream) {                                       // e.g. the producer of d_frames
		HIPCHK(hipEventRecord(p->ev_after, (hipStream_t)after_stream));
		HIPCHK(hipStreamWaitEvent(st, p->ev_after, 0));
	}
	if (p->held[slot]) {                                      // a consumer of the slot's previous outputs (smhv_pipeline_hold)
		HIPCHK(hipStreamWaitEvent(st, p->hold[slot], 0));
		p->held[slot] = 0;
	}
	// one batch in flight: nothing else can use the CUs of the frames that finish early, so they help the heavy frames (the
	// helper scheme belongs to the workgroup-synchronous k_lsd and to frames whose mask window fits the LDS: <= 1080p;
	// larger frames are better off on the tile kernel)
	if (p->depth == 1 && lsd_rows_only(p->batch[slot]->g)) stages |= SMHV_STAGE_LSD_HELPERS;
	int rc = batch_run_impl(p->batch[slot], d_frames, n, stages, grayscale, max_gap, anchors, st, sl);
	if (rc) return rc;
	HIPCHK(hipEventRecord(p->done[slot], sl));
	p->last_sl[slot] = sl;
	p->slot_frame[slot] = 0;
	p->submitted++;
	if (p->adaptive) { p->mc.count++; p->mc.frames += n; }
	if (slot_out) *slot_out = slot;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_pipeline_wait(smhv_pipeline *p, uint32_t slot) {
	if (!p || slot >= p->depth) return fail(SMHV_E_INVALID, "pipeline_wait: bad arguments");
	HIPCHK(hipSetDevice(p->ctx->device));
	{ int rc = slot_wait(p, slot); if (rc) return rc; }
	return batch_check_errors(p->batch[slot], "pipeline_wait");
}

extern "C" SMHV_API int smhv_pipeline_wait_all(smhv_pipeline *p) {
	if (!p) return fail(SMHV_E_INVALID, "null pipeline");
	HIPCHK(hipSetDevice(p->ctx->device));
	for (uint32_t i = 0; i < p->depth; ++i) { int r = slot_wait(p, i); if (r) return r; }
	int rc = SMHV_OK;
	for (uint32_t i = 0; i < p->depth; ++i) {                 // every slot is checked (and cleared); the first failure is the one returned
		std::string keep = t_last_error;
		const int r = batch_check_errors(p->batch[i], "pipeline_wait_all");
		if (r && !rc) rc = r;
		else if (r) t_last_error = keep;
	}
	return rc;
}

extern "C" SMHV_API int smhv_debug_pipeline_stats(smhv_pipeline *p, uint64_t out[32]) {
	if (!p || !out) return fail(SMHV_E_INVALID, "bad arguments");
	memset(out, 0, sizeof(uint64_t) * 32);
	if (!p->svc) return SMHV_OK;
	HIPCHK(hipSetDevice(p->ctx->device));
	HIPCHK(hipDeviceSynchronize());                           // (the service closes by itself once nothing is outstanding)
	SvcCtl c;
	HIPCHK(hipMemcpy(&c, p->d_svc_ctl, sizeof c, hipMemcpyDeviceToHost));
	out[0] = 1u; out[1] = p->h_svc->launches; out[2] = c.stat_items; out[3] = c.stat_waves; out[4] = c.stat_busy; out[5] = c.stat_life;
	out[6] = (uint64_t)p->svc_wgs * p->svc_waves; out[7] = c.completed;
	for (int k = 0; k < 4; ++k) out[8 + k] = c.stat_phase[k];
	out[12] = c.stat_help;
	out[16] = c.stat_requests; out[17] = c.stat_attached; out[18] = c.stat_remote; out[19] = (uint64_t)(int64_t)c.help_avail;
	out[20] = c.stat_t_wait; out[21] = c.stat_t_sub; out[22] = c.stat_t_last;
	out[23] = c.stat_h_polls; out[24] = c.stat_h_empty; out[25] = c.stat_h_lost; out[26] = c.stat_h_idle_exit; out[27] = c.stat_h_closed_exit; out[28] = c.stat_h_cycles; out[29] = c.stat_h_cast_cycles;
	out[13] = (p->adaptive ? 2u : 0u) | (p->mode_frame ? 1u : 0u) | (p->mc.phase == 4u ? 4u : 0u); out[14] = (uint64_t)p->mc.rate[1]; out[15] = (uint64_t)p->mc.rate[0];
	return SMHV_OK;
}

// (no device-wide synchronisation: usable while a pipeline is stuck)
extern "C" SMHV_API int smhv_debug_pipeline_peek(smhv_pipeline *p, uint64_t out[16]) {
	if (!p || !out) return fail(SMHV_E_INVALID, "bad arguments");
	memset(out, 0, sizeof(uint64_t) * 16);
	if (!p->svc) return SMHV_OK;
	HIPCHK(hipSetDevice(p->ctx->device));
	const unsigned long long st = __atomic_load_n(&p->h_svc->state, __ATOMIC_ACQUIRE);
	out[0] = st >> 32; out[1] = (uint32_t)st; out[2] = p->h_svc->launches; out[3] = p->seq_counter;
	out[14] = ((uint64_t)p->svc_wgs << 32) | p->svc_waves; out[15] = ((uint64_t)p->svc_part_words << 32) | p->svc_lds;   // geometry of the service's launches
	// (callable from another thread while a wait is stuck: its stream and staging block belong to the pipeline, under a lock)
	std::lock_guard<std::mutex> lk(p->peek_mu);
	if (!p->s_peek) HIPCHK(hipStreamCreateWithFlags(&p->s_peek, hipStreamNonBlocking));
	if (!p->h_peek) HIPCHK(hipHostMalloc((void **)&p->h_peek, sizeof(SvcCtl)));
	SvcCtl *h_ctl = p->h_peek;
	HIPCHK(hipMemcpyAsync(h_ctl, p->d_svc_ctl, sizeof(SvcCtl), hipMemcpyDeviceToHost, p->s_peek));
	HIPCHK(hipStreamSynchronize(p->s_peek));
	out[4] = (uint64_t)(int64_t)h_ctl->avail; out[5] = h_ctl->head; out[6] = h_ctl->reserve; out[7] = h_ctl->closing; out[8] = h_ctl->completed; out[9] = h_ctl->busy;
	for (uint32_t i = 0; i < 4 && i < p->depth; ++i) out[10 + i] = ((uint64_t)p->seq[i] << 32) | __atomic_load_n(&p->h_svc->done_seq[i], __ATOMIC_ACQUIRE);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_pipeline_hold(smhv_pipeline *p, uint32_t slot, void *stream) {
	if (!p || slot >= p->depth) return fail(SMHV_E_INVALID, "pipeline_hold: bad arguments");
	HIPCHK(hipSetDevice(p->ctx->device));
	HIPCHK(hipEventRecord(p->hold[slot], (hipStream_t)stream));
	p->held[slot] = 1;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_pipeline_slot(smhv_pipeline *p, uint32_t slot, smhv_batch **batch, void **stream) {
	if (!p || slot >= p->depth) return fail(SMHV_E_INVALID, "pipeline_slot: bad arguments");
	if (batch) *batch = p->batch[slot];
	if (stream && p->svc && (p->slot_frame[slot] || p->stream.empty())) {
		// frame-granular search: a submission's completion is not a point on a stream; the call waits for the slot's records
		// on the host, after which any stream will do
		HIPCHK(hipSetDevice(p->ctx->device));
		int rc = svc_wait_slot(p, slot);
		if (rc) return rc;
		*stream = (void *)(p->slot_st[slot] ? p->slot_st[slot] : p->svc_stream[0]);
		return SMHV_OK;
	}
	// the stream on which the slot's most recent record kernel runs (what a consumer has to order itself after)
	if (stream) *stream = (void *)(p->last_sl[slot] ? p->last_sl[slot] : p->stream[slot]);
	return SMHV_OK;
}

// ------------------------------------------------------------------------------------------------
// per-frame trait surface
// ------------------------------------------------------------------------------------------------
static int ensure_frame_buffers(smhv_ctx *c, uint32_t w, uint32_t h) {
	if (c->fb && c->W == w && c->H == h) return SMHV_OK;
	// dimensions changed: (re)allocate, like GpuMemory::update (vision-gpu/src/lib.rs:106-138)
	std::lock_guard<std::mutex> lk(c->mu);
	Geom g;
	int rc = compute_geom(w, h, &g);
	if (rc) return rc;
	HIPCHK(hipDeviceSynchronize());
	if (c->fb) { smhv_batch_destroy(c->fb); c->fb = nullptr; }
	if (c->h_ocr) { (void)hipHostFree(c->h_ocr); c->h_ocr = nullptr; }
	if (c->h_scales) { (void)hipHostFree(c->h_scales); c->h_scales = nullptr; }
	rc = smhv_batch_create(c, w, h, 1, &c->fb);
	if (rc) return rc;
	HIPCHK(hipHostMalloc((void **)&c->h_ocr, (size_t)g.qw * g.qh));
	HIPCHK(hipHostMalloc((void **)&c->h_scales, (size_t)g.qw * g.qh));
	memset(c->h_scales, 0, (size_t)g.qw * g.qh);
	memset(c->h_ocr, 0, (size_t)g.qw * g.qh);
	c->W = w; c->H = h;
	logf(c, 3, "allocated buffers for %ux%u frames (map ROI %u,%u %ux%u)", w, h, g.rx, g.ry, g.rw, g.rh);
	return SMHV_OK;
}

struct TraitTimer {                                         // one per trait call: its wall time goes to the context's table
	smhv_ctx *c; int idx; struct timespec t0;
	TraitTimer(smhv_ctx *c_, int idx_) : c(c_), idx(idx_) { clock_gettime(CLOCK_MONOTONIC, &t0); }
	~TraitTimer() {
		if (!c) return;
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		c->tt_ns[idx].fetch_add((uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec)), std::memory_order_relaxed);
		c->tt_calls[idx].fetch_add(1u, std::memory_order_relaxed);
	}
};
extern "C" SMHV_API int smhv_trait_times(smhv_ctx *c, uint64_t ns[SMHV_TRAIT_CALLS], uint64_t calls[SMHV_TRAIT_CALLS], int reset) {
	if (!c) return fail(SMHV_E_INVALID, "null context");
	for (int i = 0; i < SMHV_TRAIT_CALLS; ++i) {
		if (ns) ns[i] = c->tt_ns[i].load(std::memory_order_relaxed);
		if (calls) calls[i] = c->tt_calls[i].load(std::memory_order_relaxed);
		if (reset) { c->tt_ns[i].store(0, std::memory_order_relaxed); c->tt_calls[i].store(0, std::memory_order_relaxed); }
	}
	return SMHV_OK;
}

static void reset_frame_state(smhv_ctx *c) {
	c->have_frame = true; c->cropped = false; c->map_open = false; c->isolated = false; c->mask_valid = false; c->minimap_cached = false;
}

extern "C" SMHV_API int smhv_load_frame(smhv_ctx *c, const uint8_t *bgra, uint32_t w, uint32_t h) {
	TraitTimer tt_(c, SMHV_T_LOAD_FRAME);
	if (!c || !bgra || w == 0 || h == 0) return fail(SMHV_E_INVALID, "bad arguments");
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	int rc = ensure_frame_buffers(c, w, h);
	if (rc) return rc;
	const size_t bytes = (size_t)w * h * 4;
	if (c->d_frame_cap < bytes) {
		HIPCHK(hipDeviceSynchronize());
		if (c->d_frame) (void)hipFree(c->d_frame);
		c->d_frame = nullptr; c->d_frame_cap = 0;
		HIPCHK(hipMalloc((void **)&c->d_frame, bytes));
		c->d_frame_cap = bytes;
	}
	// Only the rows the pipeline reads are uploaded: the map ROI rows and the button rows below
	// them (the reference uploads the whole frame although 61 % of it is never inspected).
	const Geom &g = c->fb->g;
	const uint32_t y0 = g.ry < g.by ? g.ry : g.by;
	const uint32_t y1 = (g.ry + g.rh > g.by + g.bh) ? g.ry + g.rh : g.by + g.bh;
	const size_t off = (size_t)y0 * w * 4, len = (size_t)(y1 - y0) * w * 4;
	HIPCHK(hipMemcpyAsync(c->d_frame + off, bgra + off, len, hipMemcpyHostToDevice, c->s_main));
	HIPCHK(wait_stream(c->s_main));              // (the caller's buffer is its own again when the call returns)
	c->frame_ptr = c->d_frame;
	reset_frame_state(c);
	return SMHV_OK;
}

// The sub-view case of load_frame (vision-gpu/src/lib.rs:175-179: `frame.inner().bounds() != frame.bounds()`): a VisionFrame
// is a rectangle of a parent image (util/src/image.rs:238-262); the reference repacks it on the host and uploads the copy.
// Here the rows the pipeline reads go straight from the parent into the tight device frame with one pitched copy.
extern "C" SMHV_API int smhv_load_frame_view(smhv_ctx *c, const uint8_t *parent_bgra, uint32_t parent_w, uint32_t parent_h,
                                              uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
	if (!c || !parent_bgra || w == 0 || h == 0) return fail(SMHV_E_INVALID, "bad arguments");
	if ((uint64_t)x + w > parent_w || (uint64_t)y + h > parent_h)
		return fail(SMHV_E_INVALID, "load_frame_view: the view %ux%u+%u+%u does not lie inside its %ux%u parent", w, h, x, y, parent_w, parent_h);
	if (x == 0 && w == parent_w) return smhv_load_frame(c, parent_bgra + (size_t)y * parent_w * 4, w, h);   // full rows: already tight
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	int rc = ensure_frame_buffers(c, w, h);
	if (rc) return rc;
	const size_t bytes = (size_t)w * h * 4;
	if (c->d_frame_cap < bytes) {
		HIPCHK(hipDeviceSynchronize());
		if (c->d_frame) (void)hipFree(c->d_frame);
		c->d_frame = nullptr; c->d_frame_cap = 0;
		HIPCHK(hipMalloc((void **)&c->d_frame, bytes));
		c->d_frame_cap = bytes;
	}
	const Geom &g = c->fb->g;
	const uint32_t y0 = g.ry < g.by ? g.ry : g.by;
	const uint32_t y1 = (g.ry + g.rh > g.by + g.bh) ? g.ry + g.rh : g.by + g.bh;
	const uint8_t *src = parent_bgra + ((size_t)(y + y0) * parent_w + x) * 4;
	HIPCHK(hipMemcpy2DAsync(c->d_frame + (size_t)y0 * w * 4, (size_t)w * 4, src, (size_t)parent_w * 4, (size_t)w * 4, y1 - y0, hipMemcpyHostToDevice, c->s_main));
	HIPCHK(wait_stream(c->s_main));
	c->frame_ptr = c->d_frame;
	reset_frame_state(c);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_load_frame_device(smhv_ctx *c, const void *d_bgra, uint32_t w, uint32_t h) {
	TraitTimer tt_(c, SMHV_T_LOAD_FRAME);
	if (!c || !d_bgra || w == 0 || h == 0) return fail(SMHV_E_INVALID, "bad arguments");
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	int rc = ensure_frame_buffers(c, w, h);
	if (rc) return rc;
	c->frame_ptr = (const uint8_t *)d_bgra;
	reset_frame_state(c);
	return SMHV_OK;
}

static int require_open(smhv_ctx *c, const char *what);
extern "C" SMHV_API int smhv_crop_to_map(smhv_ctx *c, int grayscale, int *map_open, uint32_t roi[4], uint8_t *ui_rgba) {
	TraitTimer tt_(c, SMHV_T_CROP_TO_MAP);
	if (!c || !map_open) return fail(SMHV_E_INVALID, "bad arguments");
	if (!c->have_frame) return fail(SMHV_E_INVALID, "crop_to_map called before load_frame");
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	const Geom &g = b->g;
	Buffers bf = make_buffers(b, c->frame_ptr, 0);
	hipStream_t s = c->s_main;
	// ONE host wait per call: the button test, the pass over the ROI (ui_map and, speculatively, the marker mask -- the
	// reference's cropped_map / isolate / mask sequence re-reads the crop three times; mask_marker_lines then only has to
	// publish it) and the minimap walk (src/vision/mod.rs:85 asks for it next) are enqueued back to back, then the few bytes
	// that decide how the call returns.  On a frame whose map is closed the pass ran for nothing (~20 us of the GPU, nobody
	// waits for it).
	HIPCHK(launch_button(g, bf, 1, 0, s));
	HIPCHK(launch_map_pass(g, bf, 1, MAP_UI | MAP_MASK, grayscale, s));
	HIPCHK(hipEventRecord(c->ev_map, s));
	Buffers bm = make_buffers(b, c->frame_ptr, 3);             // the minimap's own record slot
	HIPCHK(launch_find_minimap(g, bm, 1, s));
	HIPCHK(hipMemcpyAsync(c->h_aux, b->d_aux, sizeof(FrameAux), hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(&c->h_res[3], b->d_results + 3, sizeof(smhv_frame_result), hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	c->cropped = true; c->isolated = false; c->mask_valid = false; c->ui_pending = false;
	c->map_open = c->h_aux->open != 0;
	*map_open = c->map_open ? 1 : 0;
	if (!c->map_open) return SMHV_OK;          // Ok(None)
	c->minimap_cached = true;
	if (roi) { roi[0] = g.rx; roi[1] = g.ry; roi[2] = g.rw; roi[3] = g.rh; }
	c->mask_valid = true;
	// ui_map: to pinned host memory on a stream of its own, behind the pass -- the call does not wait for those 3.2 MB (1080p);
	// smhv_ui_map does, when somebody wants to look at the image, usually after the two branches have been started
	const uint32_t t = (c->ui_turn++) & 1u;
	const size_t ui_bytes = (size_t)g.rw * g.rh * 4;
	if (c->h_ui_cap[t] < ui_bytes) {
		if (c->h_ui[t]) { HIPCHK(hipEventSynchronize(c->ev_ui[t])); (void)hipHostFree(c->h_ui[t]); c->h_ui[t] = nullptr; c->h_ui_cap[t] = 0; }
		HIPCHK(hipHostMalloc((void **)&c->h_ui[t], ui_bytes, hipHostMallocDefault));
		c->h_ui_cap[t] = ui_bytes;
	}
	if (c->d_ui_tight_cap < ui_bytes) {
		HIPCHK(wait_stream(c->s_ui));
		if (c->d_ui_tight) (void)hipFree(c->d_ui_tight);
		c->d_ui_tight = nullptr; c->d_ui_tight_cap = 0;
		HIPCHK(hipMalloc((void **)&c->d_ui_tight, ui_bytes));
		c->d_ui_tight_cap = ui_bytes;
	}
	HIPCHK(hipStreamWaitEvent(c->s_ui, c->ev_map, 0));
	// packed tightly on the device first (3 us), so that it crosses PCIe as contiguous copies: the pitched copy this replaces took
	// 0.2 ms in a fresh process and 0.6 ms in one that had destroyed a pipeline, the contiguous one 0.07 ms in both
	const size_t row_bytes = (size_t)g.rw * 4;
	HIPCHK(launch_pack_rows(b->d_ui + (size_t)g.m_xoff * 4, g.ui_pitch, c->d_ui_tight, (uint32_t)row_bytes, g.rh, c->s_ui));
	if (!ui_rgba) {
		HIPCHK(hipMemcpyAsync(c->h_ui[t], c->d_ui_tight, ui_bytes, hipMemcpyDeviceToHost, c->s_ui));
		HIPCHK(hipEventRecord(c->ev_ui[t], c->s_ui));
		c->ui_pending = true;
		return SMHV_OK;
	}
	// The eager form -- what the trait's `crop_to_map -> Option<(RgbaImage, [u32; 4])>` amounts to (vision-common/src/lib.rs:47): the
	// image in the caller's memory when the call returns.  It leaves the device in four row blocks, and the host copies block k
	// out of the pinned buffer while block k + 1 is still crossing PCIe: the call's 3.2 MB host copy hides behind the transfer
	// instead of following it.
	const uint32_t parts = 4, rows_per = (g.rh + parts - 1) / parts;
	for (uint32_t k = 0; k < parts; ++k) {
		const uint32_t r0 = k * rows_per, r1 = std::min(g.rh, r0 + rows_per);
		if (r1 > r0) HIPCHK(hipMemcpyAsync(c->h_ui[t] + (size_t)r0 * row_bytes, c->d_ui_tight + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes, hipMemcpyDeviceToHost, c->s_ui));
		HIPCHK(hipEventRecord(k + 1 < parts ? c->ev_ui_part[k] : c->ev_ui[t], c->s_ui));
	}
	c->ui_pending = true;
	for (uint32_t k = 0; k < parts; ++k) {
		const uint32_t r0 = k * rows_per, r1 = std::min(g.rh, r0 + rows_per);
		HIPCHK(wait_event(k + 1 < parts ? c->ev_ui_part[k] : c->ev_ui[t]));
		if (r1 > r0) memcpy(ui_rgba + (size_t)r0 * row_bytes, c->h_ui[t] + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
	}
	return SMHV_OK;
}

// The ui_map of the frame crop_to_map last ran on, in pinned host memory owned by the context: rw x rh RGBA8, tightly packed,
// readable until the SECOND crop_to_map after this one's (two buffers take turns).  Waits for the copy crop_to_map started.
extern "C" SMHV_API int smhv_ui_map(smhv_ctx *c, const uint8_t **rgba, uint32_t *w, uint32_t *h) {
	TraitTimer tt_(c, SMHV_T_UI_MAP);
	int rc = require_open(c, "ui_map");
	if (rc) return rc;
	if (!rgba) return fail(SMHV_E_INVALID, "null output");
	if (!c->ui_pending) return fail(SMHV_E_STATE, "ui_map: crop_to_map has not produced a map for this frame");
	HIPCHK(hipSetDevice(c->device));
	const uint32_t t = (c->ui_turn - 1u) & 1u;
	HIPCHK(wait_event(c->ev_ui[t]));
	*rgba = c->h_ui[t];
	if (w) *w = c->fb->g.rw;
	if (h) *h = c->fb->g.rh;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_red_pixels(smhv_ctx *c, uint32_t *count) {
	if (!c || !count) return fail(SMHV_E_INVALID, "bad arguments");
	if (!c->cropped) return fail(SMHV_E_STATE, "crop_to_map has not run for this frame");
	*count = c->h_aux->red;
	return SMHV_OK;
}

static int require_open(smhv_ctx *c, const char *what) {
	if (!c) return fail(SMHV_E_INVALID, "null context");
	if (!c->have_frame || !c->cropped) return fail(SMHV_E_INVALID, "%s called before load_frame/crop_to_map", what);
	if (!c->map_open) return fail(SMHV_E_STATE, "%s: the map is closed for this frame (crop_to_map returned None)", what);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ocr_preprocess(smhv_ctx *c, const uint8_t **out, size_t *len) {
	TraitTimer tt_(c, SMHV_T_OCR_PREPROCESS);
	int rc = require_open(c, "ocr_preprocess");
	if (rc) return rc;
	if (!out || !len) return fail(SMHV_E_INVALID, "null output");
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	const Geom &g = b->g;
	Buffers bf = make_buffers(b, c->frame_ptr, 1);
	hipStream_t s = c->s_scales;
	HIPCHK(launch_brq_pass(g, bf, 1, BRQ_OCR, 0, 0, s));
	rc = copy_image_d2h(c, 1, c->h_ocr, b->d_ocr, g.ocr_pitch, g.q_xoff, g.qw, g.qh, s);
	if (rc) return rc;
	*out = c->h_ocr; *len = (size_t)g.qw * g.qh;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_find_scales_preprocess(smhv_ctx *c, uint32_t scales_start_y, const uint8_t **out, uint32_t *w, uint32_t *h) {
	TraitTimer tt_(c, SMHV_T_FIND_SCALES_PREPROCESS);
	int rc = require_open(c, "find_scales_preprocess");
	if (rc) return rc;
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	const Geom &g = b->g;
	if (scales_start_y > g.qh) return fail(SMHV_E_INVALID, "scales_start_y %u is below the %u-row quadrant", scales_start_y, g.qh);
	Buffers bf = make_buffers(b, c->frame_ptr, 1);
	hipStream_t s = c->s_scales;
	HIPCHK(launch_brq_pass(g, bf, 1, BRQ_SCALES, scales_start_y, 0, s));
	rc = copy_image_d2h(c, 1, c->h_scales, b->d_scales, g.ocr_pitch, g.q_xoff, g.qw, g.qh, s);
	if (rc) return rc;
	c->scales_valid = true;
	if (out) *out = c->h_scales;
	if (w) *w = g.qw;
	if (h) *h = g.qh;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_isolate_map_markers(smhv_ctx *c) {
	TraitTimer tt_(c, SMHV_T_ISOLATE_MAP_MARKERS);
	int rc = require_open(c, "isolate_map_markers");
	if (rc) return rc;
	// The isolated crop is only observable through DebugView::LSDPreprocess; it is materialised
	// there on demand.  The marker predicate is idempotent under isolation, so the mask is the same.
	c->isolated = true;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_mask_marker_lines(smhv_ctx *c) {
	TraitTimer tt_(c, SMHV_T_MASK_MARKER_LINES);
	int rc = require_open(c, "mask_marker_lines");
	if (rc) return rc;
	if (c->mask_valid) return SMHV_OK;
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	Buffers bf = make_buffers(b, c->frame_ptr, 0);
	HIPCHK(launch_map_pass(b->g, bf, 1, MAP_MASK, 1, c->s_markers));
	HIPCHK(wait_stream(c->s_markers));
	c->mask_valid = true;
	return SMHV_OK;
}

static int require_mask(smhv_ctx *c, const char *what) {
	int rc = require_open(c, what);
	if (rc) return rc;
	if (!c->mask_valid) return fail(SMHV_E_INVALID, "%s called before mask_marker_lines", what);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_get_lsd_image(smhv_ctx *c, uint8_t *out, uint32_t *w, uint32_t *h) {
	int rc = require_mask(c, "get_lsd_image");
	if (rc) return rc;
	HIPCHK(hipSetDevice(c->device));
	const Geom &g = c->fb->g;
	if (w) *w = g.rw;
	if (h) *h = g.rh;
	if (out) {
		rc = copy_image_d2h(c, 0, out, c->fb->d_mask, g.mask_pitch, g.m_xoff, g.rw, g.rh, c->s_markers);
		if (rc) return rc;
	}
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_find_longest_line(smhv_ctx *c, float px, float py, float max_gap, smhv_line *line, float *len_sq) {
	TraitTimer tt_(c, SMHV_T_FIND_LONGEST_LINE);
	int rc = require_mask(c, "find_longest_line");
	if (rc) return rc;
	if (!line || !len_sq) return fail(SMHV_E_INVALID, "null output");
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	Buffers bf = make_buffers(b, c->frame_ptr, 2);
	hipStream_t s = c->s_markers;
	HIPCHK(launch_lsd(b->g, bf, 1, max_gap, 1, px, py, s, nullptr));
	HIPCHK(hipMemcpyAsync(&c->h_res[2], b->d_results + 2, sizeof(smhv_frame_result), hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	*line = c->h_res[2].lines[0];
	*len_sq = (float)c->h_res[2].length_px[0];
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_find_marker_lines(smhv_ctx *c, uint32_t max_gap, smhv_line out[SMHV_MAX_LINES], uint32_t *n) {
	TraitTimer tt_(c, SMHV_T_FIND_MARKER_LINES);
	int rc = require_mask(c, "find_marker_lines");
	if (rc) return rc;
	if (!out || !n) return fail(SMHV_E_INVALID, "null output");
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	Buffers bf = make_buffers(b, c->frame_ptr, 0);
	hipStream_t s = c->s_markers;
	rc = sector_table_for(c, max_gap, s, &bf);
	if (rc) return rc;
	// one frame, nothing beside it: the workgroup-synchronous kernel is as fast or faster (2.4 against 2.8 ms on the heaviest sample)
	HIPCHK(launch_lsd(b->g, bf, 1, (float)max_gap, 0, 0.0f, 0.0f, s, nullptr, 1024u, true));
	HIPCHK(launch_finalize(b->g, bf, 1, SMHV_STAGE_MARKERS, s));
	HIPCHK(hipMemcpyAsync(&c->h_res[0], b->d_results, sizeof(smhv_frame_result), hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	*n = c->h_res[0].n_lines;
	memcpy(out, c->h_res[0].lines, sizeof(smhv_line) * SMHV_MAX_LINES);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_lsd_stats(smhv_ctx *c, uint32_t max_gap, int exact, uint32_t *rounds, uint64_t *ray_steps) {
	int rc = require_mask(c, "lsd_stats");
	if (rc) return rc;
	if (!rounds || !ray_steps) return fail(SMHV_E_INVALID, "null output");
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	Buffers bf = make_buffers(b, c->frame_ptr, 2);          // scratch record: does not disturb find_marker_lines' result
	hipStream_t s = c->s_markers;
	if (!exact) {
		rc = sector_table_for(c, max_gap, s, &bf);
		if (rc) return rc;
	}
	// one frame, nothing beside it: the workgroup-synchronous kernel is as fast or faster (2.4 against 2.8 ms on the heaviest sample)
	HIPCHK(launch_lsd(b->g, bf, 1, (float)max_gap, 0, 0.0f, 0.0f, s, nullptr, 1024u, true));
	HIPCHK(hipMemcpyAsync(&c->h_res[2], b->d_results + 2, sizeof(smhv_frame_result), hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	*rounds = c->h_res[2].rounds;
	*ray_steps = c->h_res[2].ray_steps;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_calc_meters_to_px_ratio(smhv_ctx *c, const uint32_t *scales, uint32_t n, double *ratio, int *has, uint32_t *bars) {
	TraitTimer tt_(c, SMHV_T_CALC_METERS_TO_PX_RATIO);
	int rc = require_open(c, "calc_meters_to_px_ratio");
	if (rc) return rc;
	if (!ratio || !has || (n && !scales)) return fail(SMHV_E_INVALID, "null argument");
	if (n > SMHV_MAX_SCALES) return fail(SMHV_E_INVALID, "at most %d scales", SMHV_MAX_SCALES);
	if (!c->scales_valid) return fail(SMHV_E_INVALID, "calc_meters_to_px_ratio called before find_scales_preprocess");
	*has = 0; *ratio = 0.0;
	if (n == 0) return SMHV_OK;                 // scales.is_empty() => None (mpx_ratio.rs:80-82)
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	smhv_anchors an;
	memset(&an, 0, sizeof an);
	an.n = n; an.scales_start_y = 0;
	memcpy(an.scales, scales, sizeof(uint32_t) * 3 * n);
	Buffers bf = make_buffers(b, c->frame_ptr, 1);
	hipStream_t s = c->s_scales;
	// (the anchors travel through pinned memory of the context: the copy is asynchronous and nothing waits for it but the kernel)
	smhv_anchors *h_an = (smhv_anchors *)(c->h_bars + SMHV_MAX_SCALES * 4);
	*h_an = an;
	HIPCHK(hipMemcpyAsync(b->d_anchors, h_an, sizeof an, hipMemcpyHostToDevice, s));
	HIPCHK(launch_scale_ratio(b->g, bf, 1, b->d_bars, s));
	HIPCHK(hipMemcpyAsync(&c->h_res[1], b->d_results + 1, sizeof(smhv_frame_result), hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(c->h_bars, b->d_bars, sizeof(uint32_t) * SMHV_MAX_SCALES * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	*has = c->h_res[1].has_mpx ? 1 : 0;
	*ratio = c->h_res[1].mpx;
	if (bars) memcpy(bars, c->h_bars, sizeof(uint32_t) * 4 * n);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_find_minimap(smhv_ctx *c, uint32_t rect[4], int *found) {
	TraitTimer tt_(c, SMHV_T_FIND_MINIMAP);
	int rc = require_open(c, "find_minimap");
	if (rc) return rc;
	if (!rect || !found) return fail(SMHV_E_INVALID, "null output");
	if (c->minimap_cached) {                                   // crop_to_map ran the walk with its own pass and fetched the record
		*found = c->h_res[3].has_minimap ? 1 : 0;
		memcpy(rect, c->h_res[3].minimap, sizeof(uint32_t) * 4);
		return SMHV_OK;
	}
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	Buffers bf = make_buffers(b, c->frame_ptr, 3);          // its own record slot: runs on the crop stream
	hipStream_t s = c->s_main;
	HIPCHK(launch_find_minimap(b->g, bf, 1, s));
	HIPCHK(hipMemcpyAsync(&c->h_res[3], b->d_results + 3, sizeof(smhv_frame_result), hipMemcpyDeviceToHost, s));
	HIPCHK(wait_stream(s));
	*found = c->h_res[3].has_minimap ? 1 : 0;
	memcpy(rect, c->h_res[3].minimap, sizeof(uint32_t) * 4);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_get_debug_view(smhv_ctx *c, int which, uint8_t *rgba, uint32_t *w, uint32_t *h) {
	TraitTimer tt_(c, SMHV_T_GET_DEBUG_VIEW);
	if (!c) return fail(SMHV_E_INVALID, "null context");
	if (which == SMHV_VIEW_NONE) { if (w) *w = 0; if (h) *h = 0; return SMHV_OK; }
	if (which < 0 || which > SMHV_VIEW_CROPPED_BRQ) return fail(SMHV_E_INVALID, "unknown debug view %d", which);
	if (!c->have_frame || !c->fb) return fail(SMHV_E_INVALID, "get_debug_view called before load_frame");
	HIPCHK(hipSetDevice(c->device));
	smhv_batch *b = c->fb;
	const Geom &g = b->g;
	const bool brq = which == SMHV_VIEW_OCR_INPUT || which == SMHV_VIEW_FIND_SCALES_INPUT || which == SMHV_VIEW_CROPPED_BRQ;
	const uint32_t vw = brq ? g.qw : g.rw, vh = brq ? g.qh : g.rh;
	if (w) *w = vw;
	if (h) *h = vh;
	if (!rgba) return SMHV_OK;
	uint8_t *d_tmp = nullptr;
	HIPCHK(hipMalloc((void **)&d_tmp, (size_t)vw * vh * 4));
	Buffers bf = make_buffers(b, c->frame_ptr, 0);
	hipError_t e = launch_debug_view(g, bf, 0, which, c->isolated ? 1 : 0, d_tmp, c->s_main);
	if (e == hipSuccess) e = hipMemcpyAsync(rgba, d_tmp, (size_t)vw * vh * 4, hipMemcpyDeviceToHost, c->s_main);
	if (e == hipSuccess) e = hipStreamSynchronize(c->s_main);
	(void)hipFree(d_tmp);
	if (e != hipSuccess) return fail(SMHV_E_HIP, "debug view: %s", hipGetErrorString(e));
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_debug_marker_table(smhv_ctx *c, uint32_t *bits) {
	if (!c || !bits) return fail(SMHV_E_INVALID, "bad arguments");
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	uint32_t *d = nullptr;
	const size_t bytes = ((size_t)1 << 24) / 8;
	HIPCHK(hipMalloc((void **)&d, bytes));
	hipError_t e = launch_marker_table(d, c->s_main);
	if (e == hipSuccess) e = hipMemcpyAsync(bits, d, bytes, hipMemcpyDeviceToHost, c->s_main);
	if (e == hipSuccess) e = hipStreamSynchronize(c->s_main);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(SMHV_E_HIP, "marker table: %s", hipGetErrorString(e));
	return SMHV_OK;
}

// ------------------------------------------------------------------------------------------------
// ingest queue: the step in front of load_frame.  The reference's capture thread
// (src/capture.rs:33-63) hashes every captured frame with crc32fast and hands it on only when the CRC
// differs from the previous capture's; here the capture source writes into pinned staging buffers, each
// frame is uploaded with an asynchronous copy on the queue's own stream (so uploads overlap the batch
// compute of earlier frames), its CRC-32 is computed on the device (k_crc32), and frames whose CRC
// differs from the last accepted one are appended to a device-resident slab that smhv_batch_run takes.
// ------------------------------------------------------------------------------------------------
struct smhv_ingest {
	smhv_ctx *ctx = nullptr;
	uint32_t W = 0, H = 0, slots = 0, capacity = 0;
	uint32_t workers_opt = 0;                                // SMHV_INGEST_WORKERS(n) of the flags: hashing threads (0: the library's choice)
	size_t frame_bytes = 0;
	hipStream_t s = nullptr;
	hipStream_t s_alt = nullptr;                             // region-of-interest mode: the packed uploads take s and s_alt in turn (two copy engines)
	std::vector<uint8_t *> h_stage, d_stage, d_raw;           // d_raw: a slot's upload in a decoder's layout (allocated on first use)
	std::vector<hipEvent_t> done;
	uint32_t *d_acc = nullptr, *h_acc = nullptr;            // one CRC accumulator per slot (device / pinned host)
	uint32_t *d_x_local = nullptr, *d_x_wg = nullptr;
	uint32_t wgs = 0, rounds = 0, x_skip = 0, len_term = 0;
	uint64_t head = 0, tail = 0;                             // frames acquired / resolved (slot = index % slots)
	bool acquired = false;
	uint8_t *d_slab = nullptr;
	uint32_t count = 0, last_crc = 0;                        // capture.rs:34 `let mut last_frame_crc32 = 0;`
	uint64_t n_new = 0, n_dup = 0;
	// ---- SMHV_INGEST_ROI_UPLOAD: hash on the host, upload only what the pipeline reads (smh_crc_host.cpp) ----
	// A committed frame goes to a worker thread: CRC-32 of the whole staging buffer (what the reference's capture thread
	// hashes), then the map-ROI rows and the button rows packed into a second pinned buffer.  Frames are resolved in order by
	// the producer thread as before: a duplicate costs the PCIe link nothing, a new frame one contiguous upload of the packed
	// rows (39 % of a 1080p frame) and two pitched device copies into its place in the slab.
	bool roi = false;
	Geom g{};
	size_t roi_row_bytes = 0, btn_row_bytes = 0, pack_bytes = 0;
	std::vector<uint8_t *> h_pack, d_pack;
	std::vector<uint32_t> h_crc;
	std::vector<int> crc_state;                              // 0 idle, 1 queued / being hashed, 2 done  (guarded by mu)
	std::vector<std::thread> workers;
	std::string local_cpus;                                  // the CPUs next to the GPU as a Linux cpulist ("": one node, or sysfs does not say)
	size_t n_workers = 0;                                    // hashing threads (set before they start)
	bool pin_workers = false;                                // the hashing threads run on them (unless SMHV_INGEST_NO_AFFINITY)
	std::mutex mu;
	std::condition_variable cv_job, cv_done;
	std::deque<uint32_t> jobs;
	bool stop = false;
};
namespace smh { uint32_t crc32_host_update(uint32_t st, const uint8_t *p, size_t n); }   // smh_crc_host.cpp

// cores this process may use: hardware threads, capped by the cgroup CPU quota (v2: cpu.max, v1: cfs_quota_us / cfs_period_us)
static uint32_t usable_cores() {
	uint32_t n = std::max(1u, std::thread::hardware_concurrency());
	long long quota = -1, period = 0;
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char a[32] = {0};
		if (fscanf(f, "%31s %lld", a, &period) == 2 && strcmp(a, "max") != 0) quota = atoll(a);
		fclose(f);
	} else {
		if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
		if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
	}
	if (quota > 0 && period > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (quota + period - 1) / period));
	return n;
}

// The CPUs on the GPU's side of a multi-socket host (the NUMA node of its PCI device: sysfs), as the kernel's cpulist text; ""
// on a single-node host or where sysfs does not say.  Measured on a two-socket MI355X box (16-core quota, 256 x 1080p, the queue
// alone): producer and hashing threads on the GPU's socket 11.8-11.9 k frames/s, on the other socket 7.3 k, left to the
// scheduler 9.4-11.5 k from run to run -- the staging buffers are pinned next to the GPU, and a thread on the far socket hashes
// and packs them across the socket link.
static std::string gpu_local_cpulist(int device) {
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return "";
	for (char *c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);
	auto read_line = [](const std::string &path) -> std::string {
		std::string out;
		if (FILE *f = fopen(path.c_str(), "r")) { char buf[4096] = {0}; if (fgets(buf, sizeof buf, f)) out = buf; fclose(f); }
		while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
		return out;
	};
	const std::string dev = std::string("/sys/bus/pci/devices/") + bdf;
	const std::string node = read_line(dev + "/numa_node");
	if (node.empty() || atoi(node.c_str()) < 0) return "";
	if (read_line("/sys/devices/system/node/node1/cpulist").empty() && atoi(node.c_str()) == 0) return "";   // one node: nothing to choose
	return read_line(dev + "/local_cpulist");
}
// "64-127,192-255" -> cpu_set_t; false when the text holds no CPU
static bool parse_cpulist(const std::string &text, cpu_set_t *set) {
	CPU_ZERO(set);
	bool any = false;
	const char *p = text.c_str();
	while (*p) {
		char *end = nullptr;
		const long a = strtol(p, &end, 10);
		if (end == p) break;
		long b = a;
		p = end;
		if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1) break; p = end; }
		for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0) { CPU_SET((int)c, set); any = true; }
		if (*p == ',') ++p; else break;
	}
	return any;
}

// The GPU's CPUs this thread may actually run on: the list intersected with the thread's current affinity mask (a cpuset that
// overlaps the GPU's node in one or two CPUs would otherwise collect every hashing thread on those).  -> CPUs in *set
static int local_cpus_allowed(const std::string &local, cpu_set_t *set) {
	cpu_set_t want, have;
	if (!parse_cpulist(local, &want)) return 0;
	CPU_ZERO(&have);
	if (pthread_getaffinity_np(pthread_self(), sizeof have, &have) != 0) return 0;
	CPU_AND(set, &want, &have);
	return CPU_COUNT(set);
}

static void ingest_worker(smhv_ingest *q) {
	if (q->pin_workers) {
		cpu_set_t set;
		// (only when the GPU's side has a CPU per hashing thread for us: fewer, and the threads are better off wherever the scheduler puts them)
		if (local_cpus_allowed(q->local_cpus, &set) >= (int)std::max<size_t>(q->n_workers, 1)) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
	}
	for (;;) {
		uint32_t slot;
		{
			std::unique_lock<std::mutex> lk(q->mu);
			q->cv_job.wait(lk, [q] { return q->stop || !q->jobs.empty(); });
			if (q->jobs.empty()) return;                         // (stop, nothing left)
			slot = q->jobs.front();
			q->jobs.pop_front();
		}
		// one pass over the frame: a few rows are hashed, then what the pipeline reads of them is copied while it is in the cache
		const uint8_t *src = q->h_stage[slot];
		const Geom &g = q->g;
		uint8_t *roi_dst = q->h_pack[slot], *btn_dst = roi_dst + (size_t)g.rh * q->roi_row_bytes;
		const size_t pitch = (size_t)g.W * 4;
		uint32_t st = 0xFFFFFFFFu;
		constexpr uint32_t ROWS = 8;                             // 61 KB of a 1080p frame at a time
		for (uint32_t y0 = 0; y0 < g.H; y0 += ROWS) {
			const uint32_t y1 = std::min(g.H, y0 + ROWS);
			st = smh::crc32_host_update(st, src + (size_t)y0 * pitch, (size_t)(y1 - y0) * pitch);
			for (uint32_t y = y0; y < y1; ++y) {
				if (y >= g.ry && y < g.ry + g.rh) memcpy(roi_dst + (size_t)(y - g.ry) * q->roi_row_bytes, src + (size_t)y * pitch + (size_t)g.m_ax * 4, q->roi_row_bytes);
				if (y >= g.by && y < g.by + g.bh) memcpy(btn_dst + (size_t)(y - g.by) * q->btn_row_bytes, src + (size_t)y * pitch + (size_t)g.bx * 4, q->btn_row_bytes);
			}
		}
		const uint32_t crc = ~st;
		{
			std::lock_guard<std::mutex> lk(q->mu);
			q->h_crc[slot] = crc;
			q->crc_state[slot] = 2;
		}
		q->cv_done.notify_all();
	}
}

// crc = raw remainder ^ (init 0xFFFFFFFF carried over the whole message) ^ final xor
static uint32_t crc32_len_term(uint64_t n_dwords) { return crc32_mul(crc32_xpow(32u * n_dwords), 0xFFFFFFFFu) ^ 0xFFFFFFFFu; }

struct CrcPlan { uint32_t wgs, rounds, x_skip; std::vector<uint32_t> x_local, x_wg; };
static CrcPlan crc_plan(uint64_t n_dwords) {
	CrcPlan p;
	const uint64_t groups = (n_dwords + 3) / 4;               // 16-byte groups
	uint64_t wgs = (groups + SMH_CRC_BS - 1) / SMH_CRC_BS;
	if (wgs > 1024) wgs = 1024;                               // 4 workgroups per CU; larger messages take more rounds
	if (wgs == 0) wgs = 1;
	p.wgs = (uint32_t)wgs;
	const uint64_t G = wgs * SMH_CRC_BS;
	p.rounds = (uint32_t)((groups + G - 1) / G);
	if (p.rounds == 0) p.rounds = 1;
	p.x_skip = crc32_xpow(128u * (G - 1));
	p.x_local.resize(SMH_CRC_BS);
	for (uint32_t t = 0; t < SMH_CRC_BS; ++t) p.x_local[t] = crc32_xpow(128u * (uint64_t)(SMH_CRC_BS - 1 - t));
	p.x_wg.resize(wgs);
	const uint32_t step = crc32_xpow(128u * (uint64_t)SMH_CRC_BS);
	uint32_t acc = 0x80000000u;                               // x^0
	for (uint64_t g = wgs; g-- > 0;) { p.x_wg[g] = acc; acc = crc32_mul(acc, step); }
	return p;
}

extern "C" SMHV_API int smhv_crc32_device(smhv_ctx *c, const void *d_data, uint64_t nbytes, uint32_t *crc) {
	if (!c || !crc || (nbytes && !d_data) || (nbytes & 3u)) return fail(SMHV_E_INVALID, "crc32: null argument or length not a multiple of 4");
	CTX_OPEN(c);
	HIPCHK(hipSetDevice(c->device));
	if (nbytes == 0) { *crc = 0; return SMHV_OK; }
	const uint64_t nd = nbytes / 4;
	const CrcPlan p = crc_plan(nd);
	uint32_t *d = nullptr;
	HIPCHK(hipMalloc((void **)&d, sizeof(uint32_t) * (1 + SMH_CRC_BS + p.wgs)));
	hipStream_t s = c->s_main;
	uint32_t raw = 0;
	hipError_t e = hipMemsetAsync(d, 0, sizeof(uint32_t), s);
	if (e == hipSuccess) e = hipMemcpyAsync(d + 1, p.x_local.data(), sizeof(uint32_t) * SMH_CRC_BS, hipMemcpyHostToDevice, s);
	if (e == hipSuccess) e = hipMemcpyAsync(d + 1 + SMH_CRC_BS, p.x_wg.data(), sizeof(uint32_t) * p.wgs, hipMemcpyHostToDevice, s);
	if (e == hipSuccess) e = hipStreamSynchronize(s);         // the plan vectors are pageable host memory
	if (e == hipSuccess) e = launch_crc32(d_data, nd, p.wgs, p.rounds, p.x_skip, d + 1, d + 1 + SMH_CRC_BS, d, s);
	if (e == hipSuccess) e = hipMemcpyAsync(&raw, d, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
	if (e == hipSuccess) e = hipStreamSynchronize(s);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(SMHV_E_HIP, "crc32: %s", hipGetErrorString(e));
	*crc = raw ^ crc32_len_term(nd);
	return SMHV_OK;
}

extern "C" SMHV_API void smhv_ingest_destroy(smhv_ingest *q) {
	if (!q) return;
	if (!q->workers.empty()) {
		{ std::lock_guard<std::mutex> lk(q->mu); q->stop = true; q->jobs.clear(); }
		q->cv_job.notify_all();
		for (auto &t : q->workers) t.join();
	}
	if (q->ctx) (void)hipSetDevice(q->ctx->device);
	if (q->s) (void)hipStreamSynchronize(q->s);
	if (q->s_alt) (void)hipStreamSynchronize(q->s_alt);
	for (auto p : q->h_stage) if (p) (void)hipHostFree(p);
	for (auto p : q->d_stage) if (p) (void)hipFree(p);
	for (auto p : q->d_raw) if (p) (void)hipFree(p);
	for (auto p : q->h_pack) if (p) (void)hipHostFree(p);
	for (auto p : q->d_pack) if (p) (void)hipFree(p);
	for (auto ev : q->done) if (ev) (void)hipEventDestroy(ev);
	if (q->d_acc) (void)hipFree(q->d_acc);
	if (q->h_acc) (void)hipHostFree(q->h_acc);
	if (q->d_x_local) (void)hipFree(q->d_x_local);
	if (q->d_x_wg) (void)hipFree(q->d_x_wg);
	if (q->d_slab) (void)hipFree(q->d_slab);
	if (q->s) (void)hipStreamDestroy(q->s);
	if (q->s_alt) (void)hipStreamDestroy(q->s_alt);
	ctx_release(q->ctx);
	delete q;
}

static int ingest_setup(smhv_ingest *q) {
	HIPCHK(hipStreamCreateWithFlags(&q->s, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&q->s_alt, hipStreamNonBlocking));
	q->h_stage.assign(q->slots, nullptr); q->d_stage.assign(q->slots, nullptr); q->d_raw.assign(q->slots, nullptr); q->done.assign(q->slots, nullptr);
	for (uint32_t i = 0; i < q->slots; ++i) {
		HIPCHK(hipHostMalloc((void **)&q->h_stage[i], q->frame_bytes, hipHostMallocDefault));
		if (!q->roi) HIPCHK(hipMalloc((void **)&q->d_stage[i], q->frame_bytes));
		HIPCHK(hipEventCreateWithFlags(&q->done[i], hipEventDisableTiming));
	}
	HIPCHK(hipMalloc((void **)&q->d_acc, sizeof(uint32_t) * q->slots));
	HIPCHK(hipHostMalloc((void **)&q->h_acc, sizeof(uint32_t) * q->slots, hipHostMallocDefault));
	const uint64_t nd = q->frame_bytes / 4;
	const CrcPlan p = crc_plan(nd);
	q->wgs = p.wgs; q->rounds = p.rounds; q->x_skip = p.x_skip; q->len_term = crc32_len_term(nd);
	HIPCHK(hipMalloc((void **)&q->d_x_local, sizeof(uint32_t) * SMH_CRC_BS));
	HIPCHK(hipMalloc((void **)&q->d_x_wg, sizeof(uint32_t) * p.wgs));
	HIPCHK(hipMemcpy(q->d_x_local, p.x_local.data(), sizeof(uint32_t) * SMH_CRC_BS, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(q->d_x_wg, p.x_wg.data(), sizeof(uint32_t) * p.wgs, hipMemcpyHostToDevice));
	HIPCHK(hipMalloc((void **)&q->d_slab, q->frame_bytes * q->capacity));
	if (q->roi) {
		const Geom &g = q->g;
		// the columns the streaming pass loads: whole quads from the ROI's quad-aligned x on (clipped to the frame)
		const uint32_t roi_px = std::min<uint32_t>(g.m_quads * 4u, g.W - g.m_ax);
		q->roi_row_bytes = (size_t)roi_px * 4; q->btn_row_bytes = (size_t)g.bw * 4;
		q->pack_bytes = q->roi_row_bytes * g.rh + q->btn_row_bytes * g.bh;
		q->h_pack.assign(q->slots, nullptr); q->d_pack.assign(q->slots, nullptr); q->h_crc.assign(q->slots, 0u); q->crc_state.assign(q->slots, 0);
		for (uint32_t i = 0; i < q->slots; ++i) {
			HIPCHK(hipHostMalloc((void **)&q->h_pack[i], q->pack_bytes, hipHostMallocDefault));
			HIPCHK(hipMalloc((void **)&q->d_pack[i], q->pack_bytes));
		}
		// (frames outside the copied rectangles are never read by any kernel; zero them once so that the slab is deterministic)
		HIPCHK(hipMemset(q->d_slab, 0, q->frame_bytes * q->capacity));
		// one worker per staging slot, within half the cores the process may actually use: a container's CPU quota (cgroup
		// cpu.max) counts, not the host's thread count -- 32 hashing threads under a 16-core quota get throttled by the
		// scheduler in 100 ms periods, and the queue's rate with them (6.7-10.7 k frames/s from run to run)
		// (measured with the 512-bit CRC loop, a 16-core quota, the queue alone: 4 / 8 / 12 / 14 / 16 / 24 workers: 9.4 / 10.9 / 10.6 /
		// 8.5 / 9.4 / 9.4 k frames/s -- past eight the hashing is not what bounds the queue, PCIe and the producer's own calls are)
		const uint32_t nthreads = std::min<uint32_t>(q->slots, q->workers_opt ? q->workers_opt : std::max(2u, usable_cores() / 2u));
		q->n_workers = nthreads;
		for (uint32_t i = 0; i < nthreads; ++i) q->workers.emplace_back(ingest_worker, q);
	}
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_create(smhv_ctx *c, uint32_t w, uint32_t h, uint32_t slots, uint32_t capacity, smhv_ingest **out) {
	return smhv_ingest_create_ex(c, w, h, slots, capacity, 0u, out);
}

extern "C" SMHV_API int smhv_ingest_create_ex(smhv_ctx *c, uint32_t w, uint32_t h, uint32_t slots, uint32_t capacity, uint32_t flags, smhv_ingest **out) {
	if (!c || !out || w == 0 || h == 0 || slots < 2 || slots > 64 || capacity == 0 || (flags & ~(SMHV_INGEST_ROI_UPLOAD | SMHV_INGEST_NO_AFFINITY | 0xFF00u))) return fail(SMHV_E_INVALID, "ingest_create: bad arguments");
	*out = nullptr;
	CTX_OPEN(c);
	Geom g;
	int rc = compute_geom(w, h, &g);                          // same frame-size rules as load_frame
	if (rc) return rc;
	HIPCHK(hipSetDevice(c->device));
	smhv_ingest *q = new (std::nothrow) smhv_ingest();
	if (!q) return fail(SMHV_E_INVALID, "out of memory");
	c->refs.fetch_add(1, std::memory_order_relaxed);
	q->ctx = c; q->W = w; q->H = h; q->slots = slots; q->capacity = capacity; q->frame_bytes = (size_t)w * h * 4;
	q->roi = (flags & SMHV_INGEST_ROI_UPLOAD) != 0u; q->g = g;
	q->workers_opt = (flags >> 8) & 0xFFu;
	q->local_cpus = gpu_local_cpulist(c->device);
	q->pin_workers = !(flags & SMHV_INGEST_NO_AFFINITY) && !q->local_cpus.empty();
	// (the pinned staging buffers land next to the GPU whichever socket the creating thread sits on -- tools/numa_pinned_probe.py:
	// 45 GB/s read from the GPU's side and 56 GB/s H2D either way -- so nothing is rebound for the allocations)
	rc = ingest_setup(q);
	if (rc) { smhv_ingest_destroy(q); return rc; }
	*out = q;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_local_cpus(smhv_ingest *q, char *buf, size_t cap) {
	if (!q || !buf || cap == 0) return fail(SMHV_E_INVALID, "ingest_local_cpus: bad arguments");
	if (q->local_cpus.size() + 1 > cap) return fail(SMHV_E_INVALID, "ingest_local_cpus: the list needs %zu bytes", q->local_cpus.size() + 1);
	memcpy(buf, q->local_cpus.c_str(), q->local_cpus.size() + 1);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_bind_thread(smhv_ingest *q) {
	if (!q) return fail(SMHV_E_INVALID, "ingest_bind_thread: no queue");
	cpu_set_t set;
	if (local_cpus_allowed(q->local_cpus, &set) < 2) return SMHV_OK;    // one node, unknown, or (nearly) nothing of the GPU's side in this thread's mask: left alone
	const int e = pthread_setaffinity_np(pthread_self(), sizeof set, &set);
	if (e != 0) return fail(SMHV_E_STATE, "ingest_bind_thread: pthread_setaffinity_np: %s", strerror(e));
	return SMHV_OK;
}

// Resolve the oldest in-flight frame: wait for its upload + CRC, apply the reference's duplicate rule
// (capture.rs:44-47) and append it to the slab when it is new.  Returns INGEST_FULL (and leaves the frame queued, still
// intact in its staging slot) when the frame is new but the slab already holds `capacity` frames.
enum { INGEST_FULL = 1 };
static int ingest_resolve_one(smhv_ingest *q) {
	const uint32_t slot = (uint32_t)(q->tail % q->slots);
	if (q->roi) {
		uint32_t crc;
		{
			std::unique_lock<std::mutex> lk(q->mu);
			q->cv_done.wait(lk, [q, slot] { return q->crc_state[slot] == 2; });
			crc = q->h_crc[slot];
		}
		if (crc == q->last_crc) {                                // nothing was uploaded
			{ std::lock_guard<std::mutex> lk(q->mu); q->crc_state[slot] = 0; }
			q->tail++; q->n_dup++;
			return SMHV_OK;
		}
		if (q->count == q->capacity) return INGEST_FULL;
		q->last_crc = crc;
		const Geom &g = q->g;
		uint8_t *dst = q->d_slab + (size_t)q->count * q->frame_bytes;
		// (frames are independent appends to the slab: consecutive ones take two streams in turn, so that one frame's 3.3 MB cross PCIe
		// while the previous frame's are still on their way -- one stream of such copies reaches 25 GB/s of the link's 50)
		const hipStream_t st = (q->tail & 1ull) ? q->s_alt : q->s;
		HIPCHK(hipMemcpyAsync(q->d_pack[slot], q->h_pack[slot], q->pack_bytes, hipMemcpyHostToDevice, st));
		HIPCHK(launch_unpack_rows(q->d_pack[slot], dst, g.W, g.m_ax, g.ry, (uint32_t)(q->roi_row_bytes / 4), g.rh, g.bx, g.by, g.bw, g.bh, st));
		HIPCHK(hipEventRecord(q->done[slot], st));               // the slot's pack buffers are free again when this has passed
		{ std::lock_guard<std::mutex> lk(q->mu); q->crc_state[slot] = 0; }
		q->tail++; q->count++; q->n_new++;
		return SMHV_OK;
	}
	HIPCHK(wait_event(q->done[slot]));
	const uint32_t crc = q->h_acc[slot] ^ q->len_term;
	if (crc == q->last_crc) { q->tail++; q->n_dup++; return SMHV_OK; }
	if (q->count == q->capacity) return INGEST_FULL;
	q->last_crc = crc;
	HIPCHK(hipMemcpyAsync(q->d_slab + (size_t)q->count * q->frame_bytes, q->d_stage[slot], q->frame_bytes, hipMemcpyDeviceToDevice, q->s));
	q->tail++; q->count++; q->n_new++;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_acquire(smhv_ingest *q, uint8_t **host_bgra) {
	if (!q || !host_bgra) return fail(SMHV_E_INVALID, "ingest_acquire: null argument");
	if (q->acquired) return fail(SMHV_E_INVALID, "ingest_acquire: the previous buffer was not committed");
	HIPCHK(hipSetDevice(q->ctx->device));
	while (q->head - q->tail >= q->slots) {                   // every staging slot is in flight: retire the oldest
		int rc = ingest_resolve_one(q);
		if (rc == INGEST_FULL)                                // recoverable: nothing is lost, the queued frames go into the next slab
			return fail(SMHV_E_STATE, "ingest: the batch slab is full (%u frames) and every staging slot holds a queued frame; "
			                          "take the slab with smhv_ingest_batch, then smhv_ingest_reset", q->capacity);
		if (rc) return rc;
	}
	const uint32_t slot = (uint32_t)(q->head % q->slots);
	// the slot's previous device copy may still be the source of a slab append on q->s: stream order covers it
	if (q->roi) HIPCHK(wait_event(q->done[slot]));   // (its packed rows may still be on their way: the next frame's worker overwrites them)
	*host_bgra = q->h_stage[slot];
	q->acquired = true;
	return SMHV_OK;
}

static uint32_t pixel_layout_bytes(uint32_t layout) {
	switch (layout) {
	case SMHV_PIXELS_BGRA8: case SMHV_PIXELS_RGBA8: return 4u;
	case SMHV_PIXELS_RGB8: return 3u;
	case SMHV_PIXELS_LUMA_A8: return 2u;
	case SMHV_PIXELS_LUMA8: return 1u;
	default: return 0u;
	}
}

// The decode half of the hand-off (src/ui/debug.rs:169: `image::load_from_memory(..).into_bgra8()`): the host's codec leaves
// RGB8 / RGBA8 / L8 / LA8 pixels in the staging buffer, they are uploaded as they are (3 bytes per pixel for a JPEG or an
// opaque PNG / WebP instead of 4) and become BGRA on the device, in front of the CRC: the duplicate rule sees the same
// bytes the reference's capture thread would hash.
extern "C" SMHV_API int smhv_ingest_commit_pixels(smhv_ingest *q, uint32_t layout) {
	if (!q || !q->acquired) return fail(SMHV_E_INVALID, "ingest_commit: nothing acquired");
	const uint32_t bpp = pixel_layout_bytes(layout);
	if (!bpp) return fail(SMHV_E_INVALID, "ingest_commit_pixels: unknown pixel layout %u", layout);
	HIPCHK(hipSetDevice(q->ctx->device));
	const uint32_t slot = (uint32_t)(q->head % q->slots);
	if (q->roi) {
		if (layout != SMHV_PIXELS_BGRA8) return fail(SMHV_E_INVALID, "ingest_commit_pixels: a queue created with SMHV_INGEST_ROI_UPLOAD takes BGRA8 frames only");
		{
			std::lock_guard<std::mutex> lk(q->mu);
			q->crc_state[slot] = 1;
			q->jobs.push_back(slot);
		}
		q->cv_job.notify_one();
		q->head++;
		q->acquired = false;
		return SMHV_OK;
	}
	if (layout == SMHV_PIXELS_BGRA8) {
		HIPCHK(hipMemcpyAsync(q->d_stage[slot], q->h_stage[slot], q->frame_bytes, hipMemcpyHostToDevice, q->s));
	} else {
		if (!q->d_raw[slot]) HIPCHK(hipMalloc((void **)&q->d_raw[slot], q->frame_bytes));
		HIPCHK(hipMemcpyAsync(q->d_raw[slot], q->h_stage[slot], q->frame_bytes / 4 * bpp, hipMemcpyHostToDevice, q->s));
		HIPCHK(launch_to_bgra(q->d_raw[slot], q->d_stage[slot], (uint64_t)q->W * q->H, layout, q->s));
	}
	HIPCHK(hipMemsetAsync(q->d_acc + slot, 0, sizeof(uint32_t), q->s));
	HIPCHK(launch_crc32(q->d_stage[slot], q->frame_bytes / 4, q->wgs, q->rounds, q->x_skip, q->d_x_local, q->d_x_wg, q->d_acc + slot, q->s));
	HIPCHK(hipMemcpyAsync(q->h_acc + slot, q->d_acc + slot, sizeof(uint32_t), hipMemcpyDeviceToHost, q->s));
	HIPCHK(hipEventRecord(q->done[slot], q->s));
	q->head++;
	q->acquired = false;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_commit(smhv_ingest *q) { return smhv_ingest_commit_pixels(q, SMHV_PIXELS_BGRA8); }

extern "C" SMHV_API int smhv_ingest_push_pixels(smhv_ingest *q, const uint8_t *pixels, uint32_t layout) {
	if (!q || !pixels) return fail(SMHV_E_INVALID, "ingest_push: null argument");
	const uint32_t bpp = pixel_layout_bytes(layout);
	if (!bpp) return fail(SMHV_E_INVALID, "ingest_push_pixels: unknown pixel layout %u", layout);
	uint8_t *dst = nullptr;
	int rc = smhv_ingest_acquire(q, &dst);
	if (rc) return rc;
	memcpy(dst, pixels, q->frame_bytes / 4 * bpp);
	return smhv_ingest_commit_pixels(q, layout);
}

namespace smh { extern std::atomic<uint32_t> g_map_band_rows, g_map_band_major; }   // smh_stream.hip
extern "C" SMHV_API int smhv_debug_map_band_rows(uint32_t rows) {
	g_map_band_major.store((rows >> 31) ? 1u : ((rows >> 30) & 1u) ? 2u : 0u, std::memory_order_relaxed);
	rows &= 0x3FFFFFFFu;
	if (rows & 7u) return fail(SMHV_E_INVALID, "debug_map_band_rows: a multiple of 8 (0: the library's rule)");
	g_map_band_rows.store(rows, std::memory_order_relaxed);
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_debug_band_rows(uint32_t frame_w, uint32_t frame_h, uint32_t n, int fused, uint32_t *rows, uint32_t *bands, int *tiles) {
	Geom g;
	int rc = compute_geom(frame_w, frame_h, &g);
	if (rc) return rc;
	uint32_t rb = 0;
	map_band_rows(g.rh, n ? n : 1u, fused, &rb, tiles);
	if (rows) *rows = rb;
	if (bands) *bands = (g.rh + rb - 1) / rb;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_debug_ingest_feed(smhv_ingest *q, uint32_t n, uint32_t *counter) {
	if (!q || !counter) return fail(SMHV_E_INVALID, "ingest_feed: null argument");
	for (uint32_t i = 0; i < n; ++i) {
		uint8_t *buf = nullptr;
		int rc = smhv_ingest_acquire(q, &buf);
		if (rc) return rc;
		const uint32_t c = (*counter)++;
		buf[0] = (uint8_t)c; buf[1] = (uint8_t)(c >> 8); buf[2] = (uint8_t)(c >> 16); buf[3] = 255;
		rc = smhv_ingest_commit(q);
		if (rc) return rc;
	}
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_push(smhv_ingest *q, const uint8_t *bgra) { return smhv_ingest_push_pixels(q, bgra, SMHV_PIXELS_BGRA8); }

extern "C" SMHV_API int smhv_ingest_batch(smhv_ingest *q, const void **d_frames, uint32_t *n, uint32_t *last_crc) {
	if (!q || !d_frames || !n) return fail(SMHV_E_INVALID, "ingest_batch: null argument");
	if (q->acquired) return fail(SMHV_E_INVALID, "ingest_batch: a staging buffer is acquired but not committed");
	HIPCHK(hipSetDevice(q->ctx->device));
	while (q->tail < q->head) {                               // drain until everything is resolved or the slab is full
		int rc = ingest_resolve_one(q);
		if (rc == INGEST_FULL) break;                         // later frames stay queued for the next slab (after reset)
		if (rc) return rc;
	}
	HIPCHK(wait_stream(q->s));                                // slab appends done: any stream may read it now
	if (q->roi) HIPCHK(wait_stream(q->s_alt));
	*d_frames = q->d_slab; *n = q->count;
	if (last_crc) *last_crc = q->last_crc;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_reset(smhv_ingest *q) {
	if (!q) return fail(SMHV_E_INVALID, "ingest_reset: null argument");
	if (q->acquired) return fail(SMHV_E_INVALID, "ingest_reset: a staging buffer is acquired but not committed");
	// Frames still queued (committed after the slab filled up) are resolved into the fresh slab by the next
	// acquire / batch call.  last_crc is kept: the duplicate test continues across slabs.
	q->count = 0;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_ingest_counts(smhv_ingest *q, uint64_t *n_new, uint64_t *n_dup) {
	if (!q) return fail(SMHV_E_INVALID, "ingest_counts: null argument");
	if (n_new) *n_new = q->n_new;
	if (n_dup) *n_dup = q->n_dup;
	return SMHV_OK;
}
