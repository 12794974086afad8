/* smh_vision_hip_debug.h -- diagnostics, calibration kernels and test switches of libsmh_vision_hip.so.
 *
 * Nothing here belongs to the drop-in boundary (include/smh_vision_hip.h): no host needs these to run the path.  They exist
 * for the tests (tests/), the benchmark (bench.py) and the experiment tools (tools/): process-wide switches that force a code
 * path, counters of the long-lived search kernel, calibration kernels.  Same library, same C ABI rules (plain pointers and
 * sizes, int status + smhv_last_error()). */
#ifndef SMH_VISION_HIP_DEBUG_H
#define SMH_VISION_HIP_DEBUG_H

#include "smh_vision_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic (process-wide): batched find_lines launches have two kernels -- the task-based k_lsd_tile (on == 0; the default:
 * waves of a frame's workgroup claim 64-ray units of the oldest candidate in flight, candidates retire in order through a
 * reorder buffer; the mask sits in LDS as a sparse store of 32 x 8 px tiles) and the workgroup-synchronous k_lsd (on != 0;
 * always used for Vision::find_longest_line and for batches run with SMHV_STAGE_LSD_HELPERS).  Both produce the reference's
 * results bit for bit; the tests run every fuzz scene through both (and through the frame-granular search of a deep pipeline). */
SMHV_API int smhv_debug_lsd_classic(int on);
/* diagnostic (process-wide): k_lsd_tile keeps at most `cap` non-empty mask tiles of a frame in LDS (0 = as many as fit: 1023 up
 * to 1440p, 541 at 4K; a marker scene has 40-260); a frame with more is searched on the mask in global memory (slow).  The
 * tests lower the cap to run frames through that path. */
SMHV_API int smhv_debug_lsd_tile_cap(uint32_t cap);
/* diagnostic (process-wide): threads per workgroup of every k_lsd_tile launch: 128..1024 (multiples of 64), 0 restores the
 * library's choice (1024 for a batch that runs alone, 512 inside pipelines). */
SMHV_API int smhv_debug_lsd_threads(uint32_t threads);
/* calibration: the fused streaming pass's memory traffic without its arithmetic -- every quad of the map ROI of n resident frames
 * read once (16-byte loads out of the full-width frame rows), ui_map / mask / ocr / scales rows of the batch written with the
 * pass's own store widths and pitches (their contents are garbage afterwards).  Asynchronous on `stream`.  rows_in_flight: loads
 * a thread issues before it stores (0 = 4, 4, 8, 12; the pass itself: 12).  Its rate is what the memory system gives this
 * access pattern; bench.py reports the best variant beside the pass (roofline_isolated.pattern_copy_GBps). */
SMHV_API int smhv_debug_pattern_copy(smhv_batch *b, const void *d_frames, uint32_t n, uint32_t rows_in_flight, void *stream);
/* diagnostic (process-wide): batched runs launch everything but the line search, so that the streaming
 * pass can be timed back to back with itself (bench.py, roofline_isolated.back_to_back).  The records of such a run hold no
 * valid lines. */
SMHV_API int smhv_debug_skip_line_search(int on);
/* diagnostic (process-wide): idle polls (about 0.25 us each) a wave of k_lsd_tile may spend without progress before the
 * watchdog gives its frame up (SMHV_FRAME_LSD_STUCK).  0 restores the default (4,000,000: about a second).  The tests lower
 * it to 1 to force the error path. */
SMHV_API int smhv_debug_lsd_spin_limit(uint32_t polls);
/* diagnostic (process-wide): pipelines created while this is on behave as on a platform whose device cannot perform atomics
 * on mapped host memory (no PCIe atomics: smhv_pipeline_create probes for them) -- SMHV_SEARCH_AUTO keeps the batch-granular
 * search, an explicit SMHV_SEARCH_FRAME is SMHV_E_INVALID.  The tests use it to walk that path on a machine that has them. */
SMHV_API int smhv_debug_no_host_atomics(int on);
/* co-residency probe: launches, asynchronously on `stream`, `workgroups` (1..1024) 256-thread workgroups of a kernel with the
 * footprint of a collective's kernel -- 21 KB of LDS and 280 VGPRs, what RCCL's kernels take on gfx950 -- that does next to
 * nothing.  A host that runs kernels of its own beside a pipeline (RCCL, torch) can measure with it how soon they get onto the
 * chip (tests/test_gpu_configs.py, bench.py --side-probe). */
SMHV_API int smhv_debug_side_kernel(smhv_ctx *ctx, uint32_t workgroups, void *stream);

/* diagnostic: the frame-granular line search of a pipeline of depth >= 3 (synchronises the device).  out[0] = 1 when the
 * pipeline has one, [1] launches of the search kernel so far, [2] frames it searched, [3] waves that came and went, [4] cycles
 * those waves spent on frames, [5] cycles they were resident, [6] waves per launch, [7] submissions completed, [8..11] the cycles of [4] by phase: cache invalidation after the claim,
 * tile store + search, record (scale ratio + derived outputs), write-back + counting the frame off; [12] cycles the waves spent
 * casting candidates for other waves' frames (not part of [4]); [13..15] the search the pipeline is on and the two measured rates;
 * [16] help requests frames opened to other workgroups, [17] helpers that attached to one, [18] candidates they cast,
 * [19] help tickets nobody has taken; [20..22] in ticks of the 100 MHz timer, summed: publication -> a wave takes the frame (over frames), publication -> last frame
 * counted off (over submissions), how long that last frame was at work; [23..29] the helpers of other workgroups' frames: polls of a
 * request's ring, polls that found nothing to take, claims lost, exits because nothing came / because the request closed, cycles attached,
 * cycles of those spent casting; [30..31] 0. */
SMHV_API int smhv_debug_pipeline_stats(smhv_pipeline *p, uint64_t out[32]);
/* diagnostic, does NOT synchronise the device (usable from another thread while a wait is stuck): [0] submissions counted,
 * [1] epoch of the search launch alive (0: none), [2] launches, [3] last sequence number handed out, [4..9] the ring's
 * counters (available, head, reserved, closing epoch, submissions completed, waves at work), [10..13] slots 0-3:
 * sequence number of the latest submission << 32 | of the last one completed. */
SMHV_API int smhv_debug_pipeline_peek(smhv_pipeline *p, uint64_t out[16]);

/* exhaustive colour-predicate check support: writes 2^24/32 words, bit (r<<16|g<<8|b) = device
 * is_any_map_marker_color(r,g,b) (vision-common/src/markers/mod.rs:40-54) */
SMHV_API int smhv_debug_marker_table(smhv_ctx *ctx, uint32_t *bits);

/* diagnostic: the same CRC by ONE of the library's three host loops -- level 2: VPCLMULQDQ (512-bit folding, 2048 bits per step),
 * 1: PCLMULQDQ (128-bit lanes), 0: slicing-by-8 tables; a loop the machine lacks falls back to the next lower one.  Returns the
 * machine's level (what smhv_crc32_host and the ingest workers use); level < 0 (data may be NULL) only reports it. */
SMHV_API int smhv_debug_crc32_host_level(const void *data, uint64_t nbytes, int level, uint32_t *crc);
/* host logic only (no device needed): the band height the streaming pass takes for a launch over n frames of this frame size (fused = 1:
 * the fused map + quadrant pass of runs with the OCR / scales stages; 2: the same inside a frame-granular pipeline, beside the search
 * service; 0: the plain map pass), the number of bands per frame, and whether such a launch also writes the tile-major mask (bands of
 * whole tile rows: smhv_batch_tile_mask).  The three-set form and the grid-stride form of the fused pass keep 58-row bands whatever
 * this says. */
SMHV_API int smhv_debug_band_rows(uint32_t frame_w, uint32_t frame_h, uint32_t n, int fused, uint32_t *rows, uint32_t *bands, int *tiles);
/* Rows per band of the streaming launches that write the tile-major mask (a multiple of 8, at most 56; 0 = the library's rule:
 * smhv_debug_band_rows reports what a launch takes).  Bits 31 / 30 of the argument: the fused pass's work items in band-major order
 * always / never (neither: the rule -- band-major where the launch runs alone, frame-major beside the search service).
 * Process-wide; for measuring band heights and work-item orders against each other. */
SMHV_API int smhv_debug_map_band_rows(uint32_t rows);
/* benchmark driver: a NATIVE capture loop for the ingest queue (the reference's capture thread is native code, src/capture.rs) --
 * n times: smhv_ingest_acquire, stamp the 24-bit value (*counter)++ into pixel (0, 0) of the staging buffer (whose other
 * pixels keep what they last held; (0, 0) lies outside every region the path reads, so every frame hashes differently and
 * computes the same), smhv_ingest_commit.  What a Python loop spends per frame on three ctypes calls (~90 us: 11 k frames/s)
 * is the interpreter's, not the queue's. */
SMHV_API int smhv_debug_ingest_feed(smhv_ingest *q, uint32_t n, uint32_t *counter);

#ifdef __cplusplus
}
#endif

#endif /* SMH_VISION_HIP_DEBUG_H */
