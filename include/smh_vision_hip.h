/*
 * smh_vision_hip.h -- C ABI of libsmh_vision_hip.so, the MI355X (gfx950) back-end for the
 * squad-mortar-helper vision hot path.
 *
 * This is the drop-in boundary: one function per method of the reference's plugin trait
 * `vision-common::Vision` (reference vision-common/src/lib.rs:30-61), i.e. exactly the table the
 * reference resolves from its GPU plugin dylib (`{name}_init`, `{name}_shutdown`,
 * `{name}_{load_frame,thread_ctx,crop_to_map,...}`; vision-common/src/dylib.rs:15-27,125-150).
 * The reference's own table is `extern "Rust"` (unstable ABI), so a ~150-line Rust shim crate
 * forwards each trait method to the function below (INTEGRATION.md shows it).  Plain pointers and
 * sizes only; every function returns 0 on success or a negative SMHV_E_* code and never throws or
 * aborts across the boundary (reference: Result<T, anyhow::Error>, vision-gpu/src/lib.rs:148).
 *
 * Parity target is the reference's CPU back-end (vision-cpu/src/lib.rs), NOT its CUDA back-end
 * (the two differ: SURVEY.md Appendix A).
 */
#ifndef SMH_VISION_HIP_H
#define SMH_VISION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMHV_API __attribute__((visibility("default")))

/* error codes (negative).  smhv_last_error() returns the message of the calling thread's last failure. */
#define SMHV_OK 0
#define SMHV_E_INVALID (-1)  /* bad argument / call order (e.g. crop_to_map before load_frame)        */
#define SMHV_E_GEOMETRY (-2) /* frame size for which the reference's bounds arithmetic would panic    */
#define SMHV_E_HIP (-3)      /* HIP runtime error (message carries hipGetErrorString)                 */
#define SMHV_E_NO_DEVICE (-4)/* no usable gfx950 device -- the caller falls back to its CPU back-end
                                exactly as the reference does (src/vision/hardware.rs:73-76)          */
#define SMHV_E_STATE (-5)    /* map closed / stage output not available                               */

#define SMHV_MAX_LINES 32   /* find_lines::<32>, vision-common/src/lib.rs:58 */
#define SMHV_MAX_SCALES 3   /* src/vision/mod.rs:131 */

typedef struct smhv_ctx smhv_ctx;     /* one per device; ~ CudaInstance (vision-gpu/src/cuda.rs:15-94) */
typedef struct smhv_batch smhv_batch; /* resident frame batch + its output buffers                    */

/* == util::geometry::Line<f32> #[repr(C)] (util/src/geometry.rs:169-180): p0.x p0.y p1.x p1.y */
typedef struct { float x0, y0, x1, y1; } smhv_line;

/* log sink ~ the `&'static dyn log::Log` the reference passes to `{name}_init` (dylib.rs:79-83).
 * level: 1=error 2=warn 3=info 4=debug. May be NULL. */
typedef void (*smhv_log_fn)(int level, const char *msg);

/* reference debug::DebugView (vision-common/src/debug.rs:31-40) */
enum { SMHV_VIEW_NONE = 0, SMHV_VIEW_OCR_INPUT = 1, SMHV_VIEW_FIND_SCALES_INPUT = 2, SMHV_VIEW_LSD_PREPROCESS = 3,
       SMHV_VIEW_LSD_INPUT = 4, SMHV_VIEW_CROPPED_BRQ = 5 };

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* replaces smh_vision_gpu_init (dylib.rs:77-88 -> CudaInstance::init, vision-gpu/src/cuda.rs:31-94) */
SMHV_API int smhv_init(int device, smhv_log_fn log, smhv_ctx **out);
/* replaces smh_vision_gpu_shutdown (dylib.rs:90-97); idempotent, NULL-safe */
SMHV_API void smhv_shutdown(smhv_ctx *ctx);
/* replaces Vision::thread_ctx (vision-gpu/src/lib.rs:154-165): binds the device to the calling thread */
SMHV_API int smhv_thread_ctx(smhv_ctx *ctx);
/* The 3600 ray directions of find_longest_line are `((i as f32) / 10.0).to_radians()` through the PLATFORM libm's
 * cosf / sinf (vision-cpu/src/lib.rs:398-399), so the reference itself is not bit-stable across platforms.  The library
 * ships the glibc 2.35 values (csrc/ray_table.inc): line end points are bit-exact against a Linux/glibc build of
 * vision-cpu.  A host on another libm (Windows UCRT, musl, ...) passes its own f32::cos / f32::sin values here once
 * after smhv_init to get the same guarantee against its own build.  dx, dy: 3600 floats each.  The table is per DEVICE: the
 * call synchronises the device and rebuilds the derived offset tables of every open context on it; no other thread may be
 * launching line searches on that device while it runs. */
SMHV_API int smhv_set_ray_table(smhv_ctx *ctx, const float *dx, const float *dy);
/* thread-local message of the last failing call on this thread ("" if none) */
SMHV_API const char *smhv_last_error(void);

/* ---- screen-relative bounds (vision-common/src/screen.rs:4-66, consts/mod.rs:7-19) ----------- */
/* MAP_BOUNDS.into_absolute + "map fills remaining space" (vision-cpu/src/lib.rs:137-145) */
SMHV_API int smhv_map_bounds(uint32_t frame_w, uint32_t frame_h, uint32_t xywh[4]);
/* CLOSE_DEPLOYMENT_BUTTON_BOUNDS.into_absolute */
SMHV_API int smhv_button_bounds(uint32_t frame_w, uint32_t frame_h, uint32_t xywh[4]);

/* ---- per-frame trait surface ---------------------------------------------------------------- */
/* Vision::load_frame (vision-gpu/src/lib.rs:167-193).  bgra: tightly packed BGRA8, w*h*4 bytes, host
 * memory.  The bytes are copied; the caller keeps ownership (the Rust shim keeps the Arc<VisionFrame>
 * for get_cpu_frame itself).  (Re)allocates device buffers when the dimensions change. */
SMHV_API int smhv_load_frame(smhv_ctx *ctx, const uint8_t *bgra, uint32_t w, uint32_t h);
/* The sub-view case of load_frame (vision-gpu/src/lib.rs:175-179): the frame is the w x h rectangle at (x, y) of a
 * tightly packed parent_w x parent_h BGRA8 image (VisionFrame = OwnedSubImage, util/src/image.rs:238-262). */
SMHV_API int smhv_load_frame_view(smhv_ctx *ctx, const uint8_t *parent_bgra, uint32_t parent_w, uint32_t parent_h,
                                  uint32_t x, uint32_t y, uint32_t w, uint32_t h);
/* Same, but the frame already lives in device memory (zero-copy path for device-side producers). */
SMHV_API int smhv_load_frame_device(smhv_ctx *ctx, const void *d_bgra, uint32_t w, uint32_t h);

/* Vision::crop_to_map (vision-cpu/src/lib.rs:110-171).  *map_open = 0 reproduces Ok(None) (red button
 * fraction < 0.65): nothing else is written.  Otherwise roi = [x,y,w,h] and, if ui_rgba != NULL,
 * w*h*4 bytes of RGBA (grayscale: luma,luma,luma,255) are written to it. */
SMHV_API int smhv_crop_to_map(smhv_ctx *ctx, int grayscale, int *map_open, uint32_t roi[4], uint8_t *ui_rgba);
/* ... and the form that does not wait for the image: with ui_rgba == NULL crop_to_map returns as soon as the button test is known
 * (one host wait; the pass over the ROI, the marker mask and the minimap walk are enqueued with it), and the ui_map travels to
 * pinned host memory of the context on a stream of its own while the caller starts its two branches -- what the reference's
 * PinnedGpuImage is to its GPU back-end (vision-gpu/src/gpuimage.rs:117-166: copied when somebody looks).  smhv_ui_map waits for
 * that copy and hands out the pinned image: w x h RGBA8, tightly packed, readable until the SECOND crop_to_map after this
 * frame's (two buffers take turns).  SMHV_E_STATE when the map is closed. */
SMHV_API int smhv_ui_map(smhv_ctx *ctx, const uint8_t **rgba, uint32_t *w, uint32_t *h);
/* number of "Close Deployment" red pixels counted by the last crop_to_map (diagnostic) */
SMHV_API int smhv_red_pixels(smhv_ctx *ctx, uint32_t *count);

/* Vision::ocr_preprocess (vision-cpu/src/lib.rs:173-231): *out is a borrowed host pointer to
 * (w/2)*(h/2) bytes, valid until the next ocr_preprocess / load_frame on this context. */
SMHV_API int smhv_ocr_preprocess(smhv_ctx *ctx, const uint8_t **out, size_t *len);
/* Vision::find_scales_preprocess (vision-cpu/src/lib.rs:233-251): borrowed host image; rows above
 * scales_start_y keep whatever the previous call left there (as in the reference). */
SMHV_API int smhv_find_scales_preprocess(smhv_ctx *ctx, uint32_t scales_start_y, const uint8_t **out, uint32_t *w, uint32_t *h);

/* Vision::isolate_map_markers (vision-cpu/src/lib.rs:253-280) */
SMHV_API int smhv_isolate_map_markers(smhv_ctx *ctx);
/* Vision::mask_marker_lines (vision-cpu/src/lib.rs:357-375): threshold + L1 radius-1 dilation */
SMHV_API int smhv_mask_marker_lines(smhv_ctx *ctx);
/* Host copy of the LSD mask (w*h bytes, values {0,255}) == "detected marker pixel coords". */
SMHV_API int smhv_get_lsd_image(smhv_ctx *ctx, uint8_t *out, uint32_t *w, uint32_t *h);
/* Vision::find_longest_line (vision-cpu/src/lib.rs:387-449) on the context's LSD image */
SMHV_API int smhv_find_longest_line(smhv_ctx *ctx, float px, float py, float max_gap, smhv_line *line, float *len_sq);
/* Vision::find_marker_lines (vision-cpu/src/lib.rs:377-385 -> lsd::find_lines::<32>, lsd.rs:60-107) */
SMHV_API int smhv_find_marker_lines(smhv_ctx *ctx, uint32_t max_gap, smhv_line out[SMHV_MAX_LINES], uint32_t *n);
/* rounds (find_longest_line invocations) and mask samples of the last smhv_find_marker_lines; exact != 0 repeats the
 * scan with every ray cast (sample count == the reference's).  Diagnostic. */
SMHV_API int smhv_lsd_stats(smhv_ctx *ctx, uint32_t max_gap, int exact, uint32_t *rounds, uint64_t *ray_steps);
/* calc_meters_to_px_ratio (src/vision/mpx_ratio.rs:3-134) on the image of the last
 * find_scales_preprocess.  scales = n x {meters, x, y} (OCR label anchors, BRQ coordinates), n <= 3.
 * *has = 0 reproduces None.  bars (optional) = n x {left, y, right, found} (the scales_debug lines). */
SMHV_API int smhv_calc_meters_to_px_ratio(smhv_ctx *ctx, const uint32_t *scales, uint32_t n, double *ratio, int *has, uint32_t *bars);
/* find_minimap (src/vision/find_minimap.rs:47-146; host code in the reference, run on get_cpu_frame().view(roi)
 * right after crop_to_map, src/vision/mod.rs:85) on the resident frame: four directed walks from the ROI centre
 * over the "edginess" (max neighbour colour distance) of the BGRA frame.  rect = {left, right, top, bottom} in
 * ROI coordinates; *found = 0 reproduces None. */
SMHV_API int smhv_find_minimap(smhv_ctx *ctx, uint32_t rect[4], int *found);
/* Vision::get_debug_view (vision-cpu/src/lib.rs:451-460): RGBA copy; rgba may be NULL to query w,h. */
SMHV_API int smhv_get_debug_view(smhv_ctx *ctx, int which, uint8_t *rgba, uint32_t *w, uint32_t *h);
/* The per-call path's counterpart of the reference's Timeshares waterfall (vision-common/src/debug.rs:3-30; src/vision/mod.rs:54-66
 * wraps every trait call in one): host wall time of every call of the trait surface on this context, summed, and the number of
 * calls, since the context was created or last reset.  The two branches run on two threads: the time of a frame is
 * load_frame + crop_to_map + find_minimap + max(markers branch, scales branch), not the sum of everything. */
#define SMHV_T_LOAD_FRAME 0
#define SMHV_T_CROP_TO_MAP 1
#define SMHV_T_FIND_MINIMAP 2
#define SMHV_T_ISOLATE_MAP_MARKERS 3
#define SMHV_T_MASK_MARKER_LINES 4
#define SMHV_T_FIND_MARKER_LINES 5
#define SMHV_T_OCR_PREPROCESS 6
#define SMHV_T_FIND_SCALES_PREPROCESS 7
#define SMHV_T_CALC_METERS_TO_PX_RATIO 8
#define SMHV_T_GET_DEBUG_VIEW 9
#define SMHV_T_FIND_LONGEST_LINE 10
#define SMHV_T_UI_MAP 11
#define SMHV_TRAIT_CALLS 12
SMHV_API int smhv_trait_times(smhv_ctx *ctx, uint64_t ns[SMHV_TRAIT_CALLS], uint64_t calls[SMHV_TRAIT_CALLS], int reset);

/* ---- batched pipeline (BASELINE configs 2-5): frames resident in HBM -------------------------- */
#define SMHV_STAGE_MARKERS 0x1u /* button test + marker mask + dilation + LSD                  */
#define SMHV_STAGE_UI_MAP 0x2u  /* ui_map RGBA                                                   */
#define SMHV_STAGE_OCR 0x4u     /* ocr_preprocess                                                */
#define SMHV_STAGE_SCALES 0x8u  /* find_scales_preprocess + calc_meters_to_px_ratio (needs anchors) */
#define SMHV_STAGE_ALL 0xFu
#define SMHV_STAGE_MINIMAP 0x10u /* find_minimap (not a Vision trait method; the caller's next step, not in STAGE_ALL) */
#define SMHV_STAGE_EXACT_STATS 0x20u /* diagnostic: cast every ray of every visited pixel, so that `ray_steps` equals the
                                        reference's sample count.  Without it the LSD skips angular sectors that provably
                                        cannot hold an acceptable ray (lines, rounds and every other output are identical;
                                        ray_steps then counts only the samples actually taken). */

#define SMHV_STAGE_LSD_HELPERS 0x40u /* tuning (smhv_batch_run; a depth-1 pipeline sets it itself): workgroups of k_lsd that have finished their own frame help
                                       the frames still being searched (ray-cast candidates ahead of the owner; results are
                                       identical either way).  Shortens a single batch with a few heavy frames; only gets in
                                       the way when several batches are pipelined, so it is off by default. */

/* One record per frame (what a node-level gather moves between GPUs).  mpx/derived fields follow
 * src/ui/mod.rs:131-140 (length_px, meters in f64) and src/ui/markers.rs:98 (angle = atan2f). */
typedef struct {
	uint32_t map_open;              /* 0 = Ok(None): every other field is 0                        */
	uint32_t n_lines;
	smhv_line lines[SMHV_MAX_LINES];
	double mpx;                     /* meters per pixel, valid iff has_mpx                         */
	uint32_t has_mpx;
	uint32_t n_mask_px;             /* 255-pixels in the dilated marker mask                       */
	uint32_t red_pixels;            /* close-deployment button count                               */
	uint32_t rounds;                /* find_longest_line invocations (workload statistic)          */
	uint64_t ray_steps;             /* mask samples taken by the rays that were cast (== the reference's count
	                                   when SMHV_STAGE_EXACT_STATS is set)                          */
	double length_px[SMHV_MAX_LINES];
	double meters[SMHV_MAX_LINES];  /* length_px * mpx (0 when !has_mpx)                           */
	float angle[SMHV_MAX_LINES];
	uint32_t minimap[4];            /* find_minimap: {left, right, top, bottom} in map-ROI coordinates  */
	uint32_t has_minimap;           /* 1 iff SMHV_STAGE_MINIMAP ran and the map is open                 */
	uint32_t status;                /* SMHV_FRAME_OK, or why this frame has no valid marker lines (SMHV_FRAME_*) */
} smhv_frame_result;

/* smhv_frame_result.status.  The reference logs and drops a frame on any Err of a trait method
 * (src/vision/mod.rs:272-276); a frame whose status is not SMHV_FRAME_OK is to be dropped the same way.
 *   SMHV_FRAME_LSD_STUCK: the line search's watchdog gave the frame up (its waves made no progress for the spin budget --
 *   never observed outside the test that lowers the budget).  The record then has n_lines = 0 and rounds = 0xFFFFFFFF; every
 *   other stage output of the frame (ui_map, mask, ocr, scales, m/px) is valid.  smhv_batch_read_results,
 *   smhv_pipeline_wait(_all) and smhv_node_gather return SMHV_E_STATE when a frame of the run they cover has a non-zero
 *   status (the message names the first such frame and carries the watchdog's state dump); the records are still copied
 *   out, so the caller can drop exactly the frames whose status is set.  The condition is reported once, by the first
 *   of those calls that sees it. */
#define SMHV_FRAME_OK 0u
#define SMHV_FRAME_LSD_STUCK 1u

/* per-frame OCR anchors for SMHV_STAGE_SCALES: OCR (Tesseract) is outside this library */
typedef struct {
	uint32_t n;                     /* 0..3                                                        */
	uint32_t scales_start_y;        /* min(ocr.bottom), src/vision/mod.rs:182                      */
	uint32_t scales[SMHV_MAX_SCALES][3]; /* {meters, x, y}                                         */
} smhv_anchors;

typedef struct {
	uint32_t frame_w, frame_h;
	uint32_t roi[4], button[4];     /* map / button rects in frame coordinates                     */
	uint32_t brq_w, brq_h;
	/* device output layouts: row pitches in bytes, per-frame strides in bytes, and the byte offset
	 * of pixel (0,0) inside a frame's slab (rows are padded so 16-byte stores stay aligned)        */
	uint64_t ui_pitch, ui_stride, ui_offset;         /* RGBA8                                       */
	uint64_t mask_pitch, mask_stride, mask_offset;   /* u8 {0,255}                                  */
	uint64_t ocr_pitch, ocr_stride, ocr_offset;      /* u8                                          */
	uint64_t scales_pitch, scales_stride, scales_offset; /* u8 {0,255}                              */
	uint64_t bits_pitch_words, bits_stride, bits_xoff;   /* bit-packed mask: bit (x+xoff) of row y */
} smhv_batch_layout;

SMHV_API int smhv_batch_create(smhv_ctx *ctx, uint32_t frame_w, uint32_t frame_h, uint32_t max_frames, smhv_batch **out);
SMHV_API void smhv_batch_destroy(smhv_batch *b);
SMHV_API int smhv_batch_layout_get(smhv_batch *b, smhv_batch_layout *out);
/* Runs the selected stages over n resident frames (d_frames: n * frame_w*frame_h*4 bytes of BGRA8 in
 * device memory) on `stream` (a hipStream_t, NULL = default stream).  Asynchronous and non-blocking: every kernel and
 * copy is enqueued on `stream` and the call returns; results are in device memory when the stream reaches this point.
 * anchors: host array of n smhv_anchors or NULL (copied into pinned staging before the call returns). */
SMHV_API int smhv_batch_run(smhv_batch *b, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                            const smhv_anchors *anchors, void *stream);
/* device pointers of the batch outputs (valid for the life of the batch) */
SMHV_API int smhv_batch_device_ptrs(smhv_batch *b, void **d_results, void **d_ui, void **d_mask, void **d_ocr, void **d_scales, void **d_bits);
/* The marker mask a third time, as the streaming passes leave it for the line search (and for any host kernel that wants the
 * marker pixels without scanning a 1-4 % full image): TILE-MAJOR, 32 x 8 px tiles of the bit-packed rows -- tile (ty, wx) = rows
 * 8 ty .. 8 ty + 7 of word column wx (bits_pitch_words word columns, bit (x + bits_xoff) of a row = pixel x), eight consecutive
 * 32-bit words at d_tiled[frame * tile_rows * word_columns * 8 + (ty * word_columns + wx) * 8] -- written ONLY for tiles that hold a
 * set bit, and one occupancy byte per (tile row, group of eight word columns) saying which: bit j of
 * d_occ[frame * tile_rows * occ_pitch + ty * occ_pitch + g] = tile (ty, 8 g + j) is non-empty (every byte of an open frame's tile rows
 * is written).  Rows of the last tile row beyond the image are undefined.  A run writes them when its bands are whole tile rows --
 * the library takes such bands where they cost the streaming pass nothing or pay (ROIs up to 900 rows, i.e. frames up to 1080p, and
 * every size whose band count does not grow by it); smhv_batch_read_tile_mask reports per frame whether the last run did.
 * geometry[4] <- {tile_rows, word_columns, occ_pitch, bits_xoff}. */
SMHV_API int smhv_batch_tile_mask(smhv_batch *b, void **d_tiled, void **d_occ, uint32_t geometry[4]);
/* synchronising host copies of one frame's tile-major mask, occupancy bytes and bit-packed rows (any of them may be NULL);
 * *written <- 1 when the last run over that frame wrote the tile-major mask and the occupancy bytes, 0 when it wrote the bit rows only */
SMHV_API int smhv_batch_read_tile_mask(smhv_batch *b, uint32_t frame, uint32_t *tiled, uint8_t *occ, uint32_t *bits, int *written);
/* synchronising host copies (tightly packed) */
SMHV_API int smhv_batch_read_results(smhv_batch *b, uint32_t first, uint32_t n, smhv_frame_result *out);
SMHV_API int smhv_batch_read_image(smhv_batch *b, int which /* SMHV_VIEW_* or 100 = ui_map RGBA */, uint32_t frame, uint8_t *out);
/* Per-stage device time, AVERAGED over the timed smhv_batch_run calls since the last read (at most the
 * 64 most recent), measured with hipEvents on
 * the run's stream (the analogue of the reference's Timeshares, vision-common/src/debug.rs:3-30).
 * ms[0]=button ms[1]=map pass ms[2]=brq pass ms[3]=lsd ms[4]=scale ratio.  Synchronises. */
/* Kept for source compatibility; has no effect: the batched pipeline runs entirely on the stream given to
 * smhv_batch_run (the quadrant stages are fused into the streaming pass, the scale scan into the record kernel). */
SMHV_API int smhv_batch_set_scales_stream(smhv_batch *b, void *stream);
/* Make `stream` wait until the streaming pass (k_map_pass) of b's most recent smhv_batch_run has finished.  (What
 * smhv_pipeline_submit uses to start pipelined batches half a period apart; a host that pipelines smhv_batch objects by
 * hand can do the same.) */
SMHV_API int smhv_batch_wait_map_pass(smhv_batch *b, void *stream);
SMHV_API int smhv_batch_enable_timing(smhv_batch *b, int enable);
/* diagnostic: per-frame cooperation counters of the most recent line-segment launch, 4 words per frame:
 * {groups of the owner that had helpers attached, candidates it took from the helpers' cache, candidates cast by helpers,
 *  requests posted}.  Synchronises. */
SMHV_API int smhv_batch_lsd_coop_stats(smhv_batch *b, uint32_t first, uint32_t n, uint32_t *out);
SMHV_API int smhv_batch_stage_ms(smhv_batch *b, float ms[5]);

/* ---- pipeline: several batches in flight, scheduled by the library ----------------------------------------------
 * `depth` output buffer sets (smhv_batch objects) of max_frames frames.  The library owns every stream of the schedule (created
 * in a fixed order: the throughput does not depend on what streams the host created before or on how it interleaves its calls).
 *   submit : asynchronous.  Enqueues the stages for n resident frames on the next slot (round robin) and returns at
 *            once; it only waits when that slot's previous submission (`depth` submissions ago) is still running.
 *            after_stream (optional): a stream whose already enqueued work (e.g. the producer of d_frames) must finish
 *            first.  *slot receives the slot index.
 *   wait   : host waits for the slot's most recent submission (all of its frames' records are in device memory then).
 *   slot   : the slot's batch object (results, device pointers, images); with `stream` != NULL also a stream a consumer can use
 *            -- with the frame-granular search the call first WAITS (host) for the slot's submission: a submission's completion
 *            is a counter the search's waves count down, not a point on a stream.
 *   hold   : a consumer reads the slot's outputs on `stream` (work already enqueued there): the slot's next submission
 *            is ordered behind it.
 * depth: 1..32.  Two line-search schedules (smhv_pipeline_options::search pins one):
 *   batch-granular: one search launch per submission on the slot's own stream (k_lsd with helper workgroups at depth 1, k_lsd
 *     at depth 2 up to 1080p, k_lsd_tile -- eight waves per frame -- otherwise), staggered starts, and from depth 3 on an
 *     occupancy policy for the streaming pass and late helpers for heavy frames, both adapting to the workload.  Every slot's
 *     stream has a hardware queue of its own (up to 16 per pipeline, 20 for all live pipelines of the process on one device).
 *   frame-granular (depth >= 3, frame sizes up to ~4K): ONE long-lived search kernel per pipeline whose waves pull (slot,
 *     frame) items from a device-side ring -- one wave per frame, the reference's sequential scan, with the other waves of its
 *     workgroup casting a heavy frame's upcoming candidates -- write the frame's record and count it off against its
 *     submission; the submissions' streaming sides take two library-owned streams in turn.  A slot is done when its slowest
 *     frame is, nothing else waits for that frame.  The kernel closes by itself when nothing is outstanding (a device-wide
 *     synchronize by anybody still returns) and is launched again by the next submission.  A third of the wave-time per frame,
 *     but a frame is one wave's work from start to end: it needs ~3000 light frames in flight (three per resident wave: 12 x
 *     256 frames at 1080p, 526-538 k frames/s; more changes nothing).
 *     A frame that is still at work when most of its submission is done asks idle waves of OTHER workgroups as well (they copy its
 *     mask tiles from global memory and answer through a ring of 8/16-byte granules; smhv_pipeline_options::remote_*).
 *   SMHV_SEARCH_AUTO (the default): below depth 6 batch-granular.  From depth 6 on the pipeline has both and MEASURES which is
 *     faster on the workload it is given -- a window of 8 x depth submissions in each, after warm-ups, ~24 x depth submissions in
 *     all; again every 16384 submissions and when the submissions change shape for good (frames per submission by more than 25 %,
 *     stages or gap threshold, for `depth` submissions in a row; at most four such re-measurements between two periodic ones) --
 *     keeping the faster one (both write byte-identical records).  Measured at depth 12: the synthetic 256 x 1080p scene 540 k
 *     frames/s on the frame-granular search (430 k batch-granular); the reference's own 1440p screenshots in batches of 128 (0-372
 *     search rounds per frame) 300 k batch-granular, 243 k frame-granular (210 k without the help across workgroups); at depth 20
 *     (24 GB of output slots) 290-300 k on either.
 *   Completion: only smhv_pipeline_wait / smhv_pipeline_wait_all / smhv_pipeline_slot(.., &stream) guarantee a submission's
 *     records.  A frame-granular submission's completion is not a point on any stream: a device-wide synchronize returns when the
 *     search kernel has closed, which it also does after 20 ms without progress (e.g. behind a slow producer on `after_stream`) --
 *     the library's waits launch it again, a bare hipDeviceSynchronize does not.
 *   smhv_debug_skip_line_search and stage_ms[3..4] of smhv_batch_stage_ms apply to batch-granular submissions only (a frame-granular
 *     submission's search and records are the service's: smhv_debug_pipeline_stats; both in smh_vision_hip_debug.h). */
typedef struct smhv_pipeline smhv_pipeline;
SMHV_API int smhv_pipeline_create(smhv_ctx *ctx, uint32_t frame_w, uint32_t frame_h, uint32_t max_frames, uint32_t depth, smhv_pipeline **out);
/* The same with explicit choices: zero-initialise, set `size` = sizeof(smhv_pipeline_options), change what you need (every 0 is
 * the library's default; there are no environment variables). */
#define SMHV_SEARCH_AUTO 0u             /* batch-granular below depth 6; from 6 on whichever of the two the pipeline measures faster on its workload */
#define SMHV_SEARCH_BATCH 1u
#define SMHV_SEARCH_FRAME 2u            /* SMHV_E_INVALID when depth < 3 or the frame's mask tiles do not fit the LDS beside the streaming pass (8K) */
#define SMHV_PIPE_NO_TEAM_HELP 1u       /* flags, diagnostics (A/B): frame-granular search without waves helping the heavy frames of their workgroup */
#define SMHV_PIPE_NO_STREAM_PRIORITY 2u /*   ... without wave priority for the streaming pass */
#define SMHV_PIPE_NO_PROLOGUE 4u        /*   ... button test and anchor upload on the streaming streams instead of a stream of their own */
#define SMHV_PIPE_NO_REMOTE_HELP 8u     /*   ... a heavy frame is helped by the waves of its own workgroup only, not by idle waves of other workgroups */
#define SMHV_PIPE_WALK_BIT_ROWS 32u     /*   ... the service builds a frame's tile store by walking the bit rows' bounding box (rounds 2-5) instead of from the pass's tile-major mask */
#define SMHV_PIPE_THREE_LOAD_SETS 64u   /*   ... the streaming pass of a frame-granular pipeline with three register sets of loads in flight (128 registers: two workgroups per CU beside the service; round 5's form up to 1080p) instead of two (112: three) */
#define SMHV_PIPE_HELP_FIRST 16u        /*   ... frames ask other workgroups, and waves answer, even while frames are waiting for a wave (default: only then not) */
typedef struct {
	uint32_t size;
	uint32_t search;                    /* SMHV_SEARCH_* */
	uint32_t streams;                   /* frame-granular: streaming streams the submissions take in turn (0 = 2; 1..8) */
	uint32_t idle_close_us;             /* frame-granular: the search kernel closes after this long without work when nothing is outstanding (0 = 45) */
	uint32_t occupancy_policy;          /* batch-granular, depth >= 3: 0 = adaptive (on unless the workload is search-bound), 1 = always on, 2 = off */
	uint32_t late_helpers;              /* batch-granular: workgroups that have finished their frame help one still at work: 0 = when the
	                                       workload is search-bound, 1 = always, 2 = never */
	uint32_t service_workgroups;        /* frame-granular, diagnostic: workgroups of the search kernel (0 = one per CU) */
	uint32_t flags;                     /* SMHV_PIPE_* */
	uint32_t remote_after;              /* frame-granular: search rounds after which a frame asks the waves of OTHER workgroups for help (0 = 24) */
	uint32_t remote_tickets;            /*   ... and how many of them it asks for (0 = 3; at most 12 attach) */
	uint32_t remote_last;               /*   ... once at most 1 / remote_last of its submission's frames are still at work (0 = 6) */
	uint32_t room_for_others;           /* frame-granular: 1 = the search kernel leaves an eighth of the CUs without a workgroup of its own, so that kernels of
	                                       OTHER owners (RCCL's: 21 KB of LDS and 280 VGPRs per workgroup; torch's) always find a CU to run on beside the pipeline.
	                                       A search workgroup holds most of its CU's LDS for as long as the pipeline is busy: with one on EVERY CU (the default
	                                       up to 1080p: the fastest when the pipeline has the GPU to itself, +7 %) such a kernel waits until the pipeline runs
	                                       dry -- and holds up the queues behind it meanwhile (measured: seconds, and a quarter of the pipeline's rate).
	                                       smhv_node_create sets it (its gather is an RCCL kernel); set it when the process launches anything else beside a
	                                       pipeline.  0 = the library's choice (off; on for smhv_node), 1 = on, 2 = off */
} smhv_pipeline_options;
SMHV_API int smhv_pipeline_create_ex(smhv_ctx *ctx, uint32_t frame_w, uint32_t frame_h, uint32_t max_frames, uint32_t depth,
                                     const smhv_pipeline_options *options, smhv_pipeline **out);
SMHV_API void smhv_pipeline_destroy(smhv_pipeline *p);
SMHV_API int smhv_pipeline_submit(smhv_pipeline *p, const void *d_frames, uint32_t n, uint32_t stages, int grayscale, uint32_t max_gap,
                                  const smhv_anchors *anchors, void *after_stream, uint32_t *slot);
SMHV_API int smhv_pipeline_wait(smhv_pipeline *p, uint32_t slot);
SMHV_API int smhv_pipeline_wait_all(smhv_pipeline *p);
SMHV_API int smhv_pipeline_slot(smhv_pipeline *p, uint32_t slot, smhv_batch **batch, void **stream);
SMHV_API int smhv_pipeline_hold(smhv_pipeline *p, uint32_t slot, void *stream);

/* ---- node: every GPU of a machine from ONE process -------------------------------------------
 * Frames are independent: a global batch is block-sharded over the devices (smhv_shard_range), each device runs the
 * single-GPU pipeline on its resident shard, and the only exchange is one ncclGather (RCCL over xGMI, rccl.h:745) of the
 * per-frame result records to devices[0].  RCCL is dlopen'ed by smhv_node_create (SMHV_E_NO_DEVICE if it is absent).
 *   run    : asynchronous; d_frames[i] = n[i] resident frames on devices[i] (n[i] <= max_frames_per_device, may be 0),
 *            anchors[i] (optional) the shard's anchors.
 *   gather : all records of the most recent run in device order (sum of n[i]) into `out`; synchronises.
 *   ctx    : the per-device context / pipeline (for uploads, images, device pointers).
 *   depth  : slots of every device's pipeline; 0 = 12 (what bench.py runs one GPU with).  The pipelines are created with
 *            smhv_pipeline_options::room_for_others: the gather's RCCL kernel has to find a CU beside the search kernel. */
typedef struct smhv_node smhv_node;
SMHV_API void smhv_shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi);
SMHV_API int smhv_node_create(const int *devices, uint32_t n_devices, uint32_t frame_w, uint32_t frame_h, uint32_t max_frames_per_device,
                              uint32_t depth, smhv_log_fn log, smhv_node **out);
SMHV_API void smhv_node_destroy(smhv_node *node);
SMHV_API int smhv_node_ctx(smhv_node *node, uint32_t i, smhv_ctx **ctx, smhv_pipeline **pipe);
SMHV_API int smhv_node_run(smhv_node *node, const void *const *d_frames, const uint32_t *n, uint32_t stages, int grayscale, uint32_t max_gap,
                           const smhv_anchors *const *anchors);
SMHV_API int smhv_node_gather(smhv_node *node, smhv_frame_result *out, uint32_t *n_total);

/* ---- ingest queue: the capture hand-off in front of load_frame (src/capture.rs:33-63) --------------------------
 * The reference's capture thread hashes each captured BGRA frame with crc32fast::hash (CRC-32/IEEE) and passes it
 * on only if the CRC differs from the previous capture's (capture.rs:44-47, `last_frame_crc32` starts at 0).  Here
 * the capture source fills pinned staging buffers (acquire -> write -> commit), every frame goes to HBM with an
 * asynchronous copy on the queue's own stream, its CRC-32 is computed on the device, and frames that are not
 * duplicates of the last accepted frame are appended, in order, to a device slab of `capacity` frames that
 * smhv_batch_run (or smhv_load_frame_device) consumes.  One producer thread per queue. */
typedef struct smhv_ingest smhv_ingest;
SMHV_API int smhv_ingest_create(smhv_ctx *ctx, uint32_t frame_w, uint32_t frame_h, uint32_t slots, uint32_t capacity, smhv_ingest **out);
/* The same with flags.  SMHV_INGEST_ROI_UPLOAD: the CRC-32 of every committed frame is computed on the HOST (worker threads of the
 * queue, carry-less-multiply folding: the whole frame, as the reference hashes it) and only what the pipeline reads -- the map
 * ROI's rows and the button's rows, 39 % of a 1080p frame -- travels over PCIe, and nothing at all for a duplicate.  The slab
 * frames then hold exactly those two rectangles (zero elsewhere), which is all smhv_batch_run / the pipelines read.  BGRA8
 * commits only. */
#define SMHV_INGEST_ROI_UPLOAD 1u
#define SMHV_INGEST_WORKERS(n) (((n) & 0xFFu) << 8)   /* diagnostic, with SMHV_INGEST_ROI_UPLOAD: hashing threads (0 = the library's choice: one per staging slot, at most half the cores the process may use -- the cgroup CPU quota counts -- and at least two) */
#define SMHV_INGEST_NO_AFFINITY 2u   /* the hashing threads are left to the scheduler (default: on a multi-socket host they run on the CPUs next to the GPU) */
SMHV_API int smhv_ingest_create_ex(smhv_ctx *ctx, uint32_t frame_w, uint32_t frame_h, uint32_t slots, uint32_t capacity, uint32_t flags, smhv_ingest **out);
SMHV_API void smhv_ingest_destroy(smhv_ingest *q);
/* The CPUs next to the queue's GPU (the NUMA node of its PCI device, from sysfs) as a Linux cpulist, e.g. "64-127,192-255"; "" on a
   single-node host or where sysfs does not say.  The queue's hashing threads run there.  The staging buffers are pinned next to
   the GPU, so the thread that FILLS them (the capture thread: `src/capture.rs`) does well to run there too --
   smhv_ingest_bind_thread binds the calling thread to those of them its affinity mask already allows (SMHV_OK and nothing done when
   the list is empty or fewer than two of its CPUs are allowed; the binding stays until the caller changes it).  The hashing threads
   are bound the same way, and only when the allowed CPUs on the GPU's side are at least as many as the threads.  Measured on a
   two-socket MI355X host, 1080p frames, the queue alone: 11.8 k frames/s from the GPU's socket, 7.3 k from the other, 9.4-11.5 k
   when the scheduler chooses. */
SMHV_API int smhv_ingest_local_cpus(smhv_ingest *q, char *buf, size_t cap);
SMHV_API int smhv_ingest_bind_thread(smhv_ingest *q);
/* next pinned staging buffer (frame_w * frame_h * 4 bytes); blocks only when all `slots` uploads are in flight */
SMHV_API int smhv_ingest_acquire(smhv_ingest *q, uint8_t **host_bgra);
/* start upload + CRC of the acquired buffer; returns at once */
SMHV_API int smhv_ingest_commit(smhv_ingest *q);
/* acquire + memcpy + commit for frames that live in ordinary host memory */
SMHV_API int smhv_ingest_push(smhv_ingest *q, const uint8_t *bgra);
/* Frames that come out of an image decoder instead of the screen capture (src/ui/debug.rs:169:
 * `image::load_from_memory(..).into_bgra8()`): the staging buffer holds frame_w * frame_h pixels in the decoder's layout,
 * they are uploaded as they are and converted to BGRA8 on the device exactly as image 0.23's into_bgra8 does for 8-bit
 * images (RGB: alpha = 255; L / LA: b = g = r = l), BEFORE the CRC, so the duplicate rule sees the reference's bytes. */
#define SMHV_PIXELS_BGRA8   0u
#define SMHV_PIXELS_RGBA8   1u
#define SMHV_PIXELS_RGB8    2u
#define SMHV_PIXELS_LUMA8   3u
#define SMHV_PIXELS_LUMA_A8 4u
SMHV_API int smhv_ingest_commit_pixels(smhv_ingest *q, uint32_t layout);
SMHV_API int smhv_ingest_push_pixels(smhv_ingest *q, const uint8_t *pixels, uint32_t layout);
/* wait for everything committed (or until the slab is full); *d_frames = slab of *n <= capacity accepted frames, valid
 * until smhv_ingest_reset; *last_crc (optional) = CRC-32 of the last accepted frame.  Frames committed after the slab
 * filled up are not lost: they stay queued in their staging slots (at most `slots` of them -- smhv_ingest_acquire
 * returns SMHV_E_STATE when the slab is full and no slot is free) and go into the next slab after smhv_ingest_reset. */
SMHV_API int smhv_ingest_batch(smhv_ingest *q, const void **d_frames, uint32_t *n, uint32_t *last_crc);
/* start a new slab (the consumer of the previous one must have finished with it); the duplicate test keeps comparing
 * with the last accepted frame; frames still queued are resolved into the new slab by the next acquire / batch */
SMHV_API int smhv_ingest_reset(smhv_ingest *q);
SMHV_API int smhv_ingest_counts(smhv_ingest *q, uint64_t *n_new, uint64_t *n_dup);
/* CRC-32/IEEE of nbytes of HOST memory (any length, any alignment; PCLMULQDQ folding where the CPU has it); needs no device */
SMHV_API uint32_t smhv_crc32_host(const void *data, uint64_t nbytes);
/* CRC-32/IEEE of nbytes (multiple of 4) of device memory; == crc32fast::hash / zlib crc32 of the same bytes */
SMHV_API int smhv_crc32_device(smhv_ctx *ctx, const void *d_data, uint64_t nbytes, uint32_t *crc);

#ifdef __cplusplus
}
#endif
#endif
