// smh_kernels.h -- launch interface between the host runtime (smh_runtime.cpp) and the gfx950
// kernels (smh_stream.hip, smh_lsd.hip, smh_misc.hip).  Internal; the public boundary is include/smh_vision_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smh_vision_hip.h"
#include "../../include/smh_vision_hip_debug.h"

namespace smh {

// Geometry + device buffer layout of one frame size.  All buffers are per-batch slabs indexed by
// frame; rows are padded so that the 16-byte-per-lane accesses of the streaming passes are
// aligned: a "quad" is 4 horizontally adjacent pixels whose frame x is a multiple of 4, and the
// padded output rows start at the ROI's quad-aligned x (ROI pixel 0 sits `xoff` pixels in).
struct Geom {
	uint32_t W, H;                 // frame
	uint32_t rx, ry, rw, rh;       // map ROI (reference "cropped_map")
	uint32_t bx, by, bw, bh;       // close-deployment button
	uint32_t qx, qy, qw, qh;       // bottom-right quadrant in FRAME coordinates / size
	// map pass
	uint32_t m_ax, m_xoff, m_quads, m_block;   // aligned x, rx - ax, quads per row, threads per block
	// brq pass
	uint32_t q_ax, q_xoff, q_quads, q_block;
	// layouts (bytes unless noted)
	uint64_t frame_bytes;
	uint64_t ui_pitch, ui_stride;
	uint64_t mask_pitch, mask_stride;
	uint32_t bits_pitch_w; uint64_t bits_stride_w;  // words
	uint64_t ocr_pitch, ocr_stride;                // also used for the scales image
};

// per-frame scratch written by k_button / k_map_pass, read by k_lsd
struct FrameAux {
	uint32_t open;        // button test passed
	uint32_t red;         // red pixel count
	uint32_t n_mask_px;   // popcount of the dilated mask
	uint32_t y_min, y_max, w_min, w_max;  // bounding box of set bits: rows, and words of the bit-packed rows
	uint32_t tiles;       // 1: the pass that wrote this frame's mask also wrote it tile-major with occupancy bytes (bands of whole tile rows)
};

// ---- cooperation between the workgroups of one k_lsd launch (smh_lsd.hip) ---------------------------------------
// A frame's line scan is sequential, but casting the rays of a candidate pixel is a pure function of (mask, pixel):
// workgroups that have finished their own frame ("helpers") load the mask of a frame that is still being searched and
// ray-cast candidates its owner has posted ahead of its own position; the results travel back through a per-frame
// hash table keyed by the pixel.  The owner never waits for a helper and a missing result only means it casts itself,
// so the outputs are the sequential algorithm's bit for bit.
struct LsdCoop {             // one 64-byte line per frame, zeroed before every launch
	uint32_t req_tail;       // requests posted by the owner
	uint32_t req_head;       // requests claimed by helpers (the owner raises it to req_tail to cancel what is pending)
	uint32_t done;           // the owner has finished the frame
	uint32_t helpers;        // helpers attached
	uint32_t remaining;      // surviving candidate pixels the owner still has in front of it (what helpers go by)
	uint32_t mode1;          // 1 + mask residency mode of the owner's kernel (helpers of another mode's kernel keep off)
	uint32_t stat_groups, stat_hits, stat_casts;   // diagnostics: helped groups of the owner, cache hits, candidates cast by helpers
	uint32_t pad[7];
};
struct LsdCtl {              // per launch, zeroed with the LsdCoop array; [residency mode]
	uint32_t started[3];     // owner workgroups dispatched
	uint32_t finished[3];    // owner workgroups that have left their frame
	uint32_t pad[10];
};
struct LsdCacheEntry {       // 32 bytes
	unsigned long long tag;  // epoch << 32 | busy << 31 | y << 12 | x ; epoch identifies the launch (stale entries = empty)
	unsigned long long best; // max over rays of (len^2 bits << 32 | ray index)
	float ex, ey;            // end point of that ray
	uint32_t steps, pad;     // mask samples of all rays of the candidate
};
#define SMH_LSD_REQ_CAP 64u          // request ring entries per frame (seq << 24 | y << 12 | x)
#define SMH_LSD_CACHE_SLOTS 512u     // hash slots per frame
struct LsdCoopBufs {
	LsdCtl *ctl;             // null: no cooperation (Vision::find_longest_line, or the buffers are absent)
	LsdCoop *coop;
	uint32_t *req;           // n x SMH_LSD_REQ_CAP
	LsdCacheEntry *cache;    // n x SMH_LSD_CACHE_SLOTS
	uint32_t epoch;
};

// Error mailbox of a batch: pinned host memory mapped into the device, written by the device only when a frame fails
// (SMHV_FRAME_*), read by the host after it has synchronised with the run (smhv_batch_read_results, smhv_pipeline_wait,
// smhv_node_gather).  Sticky until reported.
struct BatchError {
	uint32_t count;          // frames whose status is not SMHV_FRAME_OK since the last report
	uint32_t frame, status;  // the first of them
	uint32_t info[8];        // SMHV_FRAME_LSD_STUCK: head, tail, disp_e, disp_end, n_lines, rounds, spec, state of the head slot
	uint32_t pad[5];
};

// ---- late helpers of k_lsd_tile (round 3) ---------------------------------------------------------------------------------
// A workgroup that has written its own frame's record looks once over the launch's frames, takes one that is still at work and
// has asked for help (FarmFrame::want), builds the same tile store and casts whole CANDIDATES its owner posts to it: the owner's
// reorder buffer holds local and remote candidates alike and retires them in order, so lines, rounds and sample counts are the
// sequential scan's whichever workgroup cast a candidate (casting is a pure function of (mask, pixel)).  Every word below is ONE
// 8-byte agent-scope atomic (a granule is never torn, needs no fence); the three payload words of a result are stored, drained
// (s_waitcnt vmcnt(0)), then the tagged word.
#define SMH_FARM_RING 4u
#define SMH_REC_ON 0x80000000u
#define SMH_LSD_LATE_HELP 1u
struct FarmEntry {                  // 64 bytes
	unsigned long long post;        // owner -> helper: epoch16 << 48 | (k + 1)24 << 24 | py12 << 12 | px12   (k = posts so far)
	unsigned long long best;        // helper -> owner: max over rays of (len^2 bits << 32 | ray index)
	unsigned long long end;         //   ey bits << 32 | ex bits
	unsigned long long start;       //   pty bits << 32 | ptx bits
	unsigned long long done;        //   tag32 << 32 | steps, tag = epoch16 << 16 | (k + 1)16: written last
	unsigned long long pad[3];
};
struct FarmFrame {                  // one per frame of the launch
	unsigned long long attached;    // helper -> owner: epoch when the helper has built its tile store and listens
	unsigned long long owner_done;  // owner -> helper: epoch when the frame is finished
	// late helpers (SMH_LSD_LATE_HELP): a workgroup that has finished its own frame helps one that is still at work
	unsigned long long want;        // owner -> anybody: epoch16 << 48 | survivor-list entries still to visit (refreshed now and then)
	unsigned long long claimed;     // finished workgroup -> the others: epoch when one of them has taken this frame (compare-and-swap)
	unsigned long long pad[4];
	FarmEntry ring[SMH_FARM_RING];
};

struct Buffers {
	BatchError *err;         // device address of the batch's mailbox (null: none)
	FarmFrame *farm;         // late-helper exchange of k_lsd_tile, one entry per frame (null: none)
	// k_lsd_tile writes the frame's record itself (smh_record.inc: scale ratio + derived marker outputs) when SMH_REC_ON is set:
	// rec_stages = SMH_REC_ON | the run's stage mask (SMHV_STAGE_SCALES cleared when the run has no anchors), rec_bars = the
	// scale-bar debug slab or null
	uint32_t rec_stages;
	uint32_t *rec_bars;
	uint32_t lsd_flags, lsd_late_kc; // SMH_LSD_LATE_HELP: `farm` holds one entry per frame and finished workgroups help frames still at work
	                                 // lsd_late_kc x 1024 cycles after they began
	const uint8_t *frames;   // n * frame_bytes
	uint8_t *ui, *mask, *ocr, *scales;
	uint32_t *bits;
	// the same mask tile-major, and which tiles hold a set bit (written by the streaming passes, read by the search service's
	// tile-store builder: tiled_* / occ_* below)
	uint32_t *tiled;
	uint8_t *occ;
	FrameAux *aux;
	smhv_frame_result *results;
	const smhv_anchors *anchors;   // device copy, may be null
	// k_lsd sector culling table for this max_gap (null: cast every ray), SMH_CULL_TAB_WORDS words:
	//   [SMH_CULL_CELLS words] cell (row oy + R) * 4 + j: mask of the annulus pixels among offsets ox = -R + 32 j + [0, 32)
	//   [(2R+1)^2 bytes, row-major] unit range of each annulus pixel: first_unit | (n_units - 1) << 6, 0xFF = every unit
	const uint32_t *cull_tab;
	// x/y offsets of every ray after 32 j additions of its direction, j = 0..SMH_RAY_OFF_BATCHES (float2 [3600][SMH_RAY_OFF_BATCHES + 1];
	// built on the device with the reference's own repeated f32 addition): lets several lanes walk different 32-sample
	// batches of ONE long ray at the same time (k_lsd_wave).  null: long rays are walked batch after batch.
	const float *ray_off;
	LsdCoopBufs co;
};
// ---- the mask as the streaming passes leave it for the line search (round 6) -------------------------------------------------
// A marker mask is 1-4 % non-empty, and what the search keeps in LDS is its non-empty 32 x 8 px tiles.  Finding them in the
// row-major bit rows meant walking the bounding box of the set bits: (tile rows x tile columns) x 16 strided dword loads, ~68 KB
// and ~19 dependent round trips per 1080p frame, from one wave, while the streaming pass saturates the memory system.  The
// pass knows which tiles are empty when it writes them, so it also leaves
//   tiled  the bit rows tile-major: tile (ty, wx) = rows 8 ty .. 8 ty + 7 of word column wx of the bit-packed rows (Geom::bits_pitch_w
//          word columns; bit 0 of word column 0 is the ROI's quad-aligned x, m_xoff bits left of pixel 0), 8 consecutive words.
//          Only tiles with a set bit are written (by the wave that owns the word column in that band).
//   occ    one byte per (tile row, wave of the pass): bit j = tile (ty, 8 wave + j) holds a set bit.  Every band of an open frame
//          writes the bytes of its tile rows (zeros too), so nothing stale is ever read.
// The builder reads the occupancy bytes of the bounding box's tile rows (one coalesced load) and then exactly the non-empty
// tiles (32 contiguous bytes each, all in flight together): 2-3 round trips and 4-9 KB.  Bands have to be a multiple of 8 rows tall
// for this (24 rows up to 1080p, where short bands are also the faster ones; 56 instead of 58 above, 20 bands instead of 19 at 1440p): a
// launch takes them where they are free or pay (band_rows_for, smh_stream.hip) and says so per frame in FrameAux::tiles; otherwise the search walks the bit rows as before.
__host__ __device__ inline uint32_t tiled_rows(const Geom &g) { return (g.rh + 7u) >> 3; }
__host__ __device__ inline uint64_t tiled_stride_w(const Geom &g) { return (uint64_t)tiled_rows(g) * g.bits_pitch_w * 8u; }
__host__ __device__ inline uint32_t occ_pitch(const Geom &g) { return ((g.m_block >> 6) + 3u) & ~3u; }
__host__ __device__ inline uint64_t occ_stride(const Geom &g) { return (uint64_t)tiled_rows(g) * occ_pitch(g); }

#define SMH_RAY_OFF_BATCHES 168u     // 5376 steps: longer than the diagonal of the largest supported ROI (4096 x 3300)

// Sector culling table as k_build_sector_table writes it: (2R+1)^2 entries, entry (oy+R)*(2R+1) + (ox+R) = bit mask of the
// 64-ray units that can sample the pixel at offset (ox, oy) from floor(start point) at a step in [50 - T, 50].  The host
// condenses it into the per-word cells of Buffers::cull_tab.
#define SMH_SECTOR_R 52
#define SMH_SECTOR_DIM (2 * SMH_SECTOR_R + 1)
#define SMH_SECTOR_ENTRIES (SMH_SECTOR_DIM * SMH_SECTOR_DIM)
#define SMH_CULL_CELLS (SMH_SECTOR_DIM * 4)
#define SMH_CULL_TAB_WORDS (SMH_CULL_CELLS + (SMH_SECTOR_ENTRIES + 3) / 4)

// ---- the frame-granular line search of a pipeline (k_lsd_service, smh_service.inc; round 4) -------------------------------
// A batch-granular search launch lasts as long as its slowest frame, and the slot cannot be resubmitted before.  A pipeline of
// depth >= 3 therefore runs ONE long-lived search kernel per device: its waves pull (slot, frame) items from a ring in device
// memory, search the frame (one wave per frame: the reference's sequential scan as it stands, smh_lsd_seq.inc), write the
// frame's record and count it off against its submission; the wave that finishes a submission's last frame stores the
// submission's sequence number into host-mapped memory, which is what smhv_pipeline_wait polls.  The streaming side of a
// submission ends with k_svc_publish, which writes the slot's descriptor and the items.
//   SvcSlot   per pipeline slot, device memory: what the waves need to know about the slot's current submission
//   SvcCtl    per pipeline, device memory: the ring and its counters
//   SvcHost   per pipeline, mapped host memory: life-cycle handshake and completion flags
// Life cycle: the kernel is launched by the submission that finds it not alive and closes ITSELF when every submitted
// batch is complete and nothing has arrived for `idle_short` cycles -- atomically with respect to the host's next submission
// (compare-and-swap on SvcHost::state, which names the launch that is alive), so a device-wide synchronize by anybody still
// returns.
#define SVC_MAX_SLOTS 32u
struct SvcSlot {
	Buffers b;                          // the submission's buffers, record stages, anchors, ... (as the batch kernels get them)
	uint32_t n;                         // frames of the submission
	uint32_t seq;                       // its sequence number (never 0)
	uint32_t done;                      // frames finished so far
	uint32_t pad;
	unsigned long long t_pub;           // s_memrealtime (100 MHz) when the items were published (diagnostics: where a submission's time goes)
};
// ---- help across workgroups (round 5) -----------------------------------------------------------------------------------------
// The help desk of smh_lsd_seq.inc lives in the owner's LDS: only the waves of its workgroup (three or four) can serve it, and a
// frame with several hundred rounds still takes milliseconds -- long enough to block its submission's slot.  A frame that has
// been at work for SvcParams::remote_after rounds therefore also asks the waves of OTHER workgroups: it writes its tile store
// (the mask as it holds it in LDS: the non-empty tiles + the 16-bit index) into its own block of SvcParams::remote_store,
// opens its SvcRemote and puts `tickets` into the service's help ring; a wave between two frames of its own that pops a
// ticket attaches (compare-and-swap on SvcRemote::state), copies the store into its own LDS block and casts the candidates the
// owner posts in SvcRemote::ring -- exactly what a helper of the owner's workgroup does through LDS, with 8-byte agent-scope
// granules in global memory instead.  The owner matches results to pixels and visits the scan's pixels in the scan's order:
// lines, rounds and sample counts are the sequential reference's whoever cast what (a cast is a pure function of mask and pixel).
#define SVC_REMOTE_RING 12u             // posts a frame may have out to helpers of other workgroups
#define SVC_REMOTE_OPEN 0x80000000ull
struct SvcRemoteEntry {                 // 64 bytes
	unsigned long long post;            // owner -> helpers: k32 << 32 | py16 << 16 | px16 (k = 1, 2, ...: the request's posts in order; 0 = none); one 8-byte agent-scope granule
	unsigned long long claim;           // the highest k anybody has taken (atomic max: a helper to cast it, or the owner taking it back)
	// helper -> owner: the result as two 16-byte granules, each written by ONE write-through store and read by ONE 16-byte load,
	// each carrying the post's number -- a half is valid when its k is the one waited for, whatever the order the two arrive in
	// (one round trip for the owner instead of tag-then-payload):
	uint32_t res_a[4];                  //   k, mask samples, len^2 bits of the longest exactly evaluated ray (0: none), its ray index
	uint32_t res_b[4];                  //   k, ex bits, ey bits (that ray's end point), start point: (2 pty)16 << 16 | (2 ptx)16 (get_centre yields multiples of 0.5)
	unsigned long long pad[2];
};
struct SvcRemote {                      // one per wave of the service launch (owner = workgroup * waves + wave)
	unsigned long long state;           // request number32 << 32 | SVC_REMOTE_OPEN | helpers attached
	unsigned long long info;            // non-empty tiles of the frame's store (what a helper has to copy), written before the request opens
	unsigned long long pad[6];
	SvcRemoteEntry ring[SVC_REMOTE_RING];
};
#define SVC_HELP_RING 256u              // tickets in flight (a power of two)
struct SvcCtl {
	// (avail and closing are what an idle wave looks at: one 8-byte load.  Every idle wave of the chip polls this one address,
	// i.e. one memory channel: smh_service.inc spaces the polls out -- at 2 us per wave the streaming pass beside them took 1.6 ms
	// instead of 0.9: a pass is as slow as its slowest channel)
	int32_t avail;                      // items published and not yet claimed (semaphore; transiently negative)
	uint32_t closing;                   // epoch of the service launch that has closed
	int32_t help_avail;                 // help tickets nobody has taken (semaphore): the second word of an idle wave's look
	uint32_t help_head, help_reserve;   // help tickets handed out / reserved by owners
	uint32_t stat_remote;               // diagnostics: candidates cast for frames of other workgroups
	uint32_t head;                      // tickets handed out
	uint32_t reserve;                   // ring entries reserved by publishers
	uint32_t completed;                 // submissions finished (counts like SvcHost::state >> 1)
	uint32_t busy;                      // waves working on a frame
	uint32_t stat_items, stat_waves;    // diagnostics (smhv_debug_pipeline_stats): frames searched; waves that have come and gone
	unsigned long long stat_busy, stat_life;   // cycles spent on frames / between a wave's first poll and its exit, summed over those waves
	unsigned long long stat_phase[4];   // of stat_busy: acquire (cache invalidation), tile store + search, record (scale ratio + derived outputs), release + count
	unsigned long long stat_help;       // cycles spent casting candidates for other waves' frames (not in stat_busy)
	uint32_t stat_requests, stat_attached;   // help requests opened across workgroups; helpers that attached to one
	// where a submission's time goes, in ticks of s_memrealtime (100 MHz), summed: publication -> a wave takes the frame (per frame);
	// publication -> the submission's last frame is counted off (per submission); how long that last frame itself was at work
	unsigned long long stat_t_wait, stat_t_sub, stat_t_last;
	// helpers of other workgroups' frames: polls of a request's ring, polls that found nothing to take, claims lost to somebody quicker, exits because
	// nothing came / the request closed, cycles attached
	uint32_t stat_h_polls, stat_h_empty, stat_h_lost, stat_h_idle_exit, stat_h_closed_exit, stat_h_pad;
	unsigned long long stat_h_cycles, stat_h_cast_cycles;
	unsigned long long help_ring[SVC_HELP_RING];   // ticket gen32 << 32 | request number16 << 16 | owner16
	// (the item ring is an allocation of its own: gen32 << 32 | slot << 24 | frame, gen = (ticket >> log2 cap) + 1)
};
struct SvcHost {
	// submissions so far << 32 | epoch of the service launch that is alive (0: none).  The host adds submissions and -- only
	// when it finds the low half 0 -- puts a new launch's epoch there; only waves of THAT launch clear it again (64-bit
	// compare-and-swap over PCIe), and a wave that finds another launch's epoch (or none) there leaves: no launch can outlive
	// its own entry, whatever the interleaving.
	unsigned long long state;
	uint32_t launches;                  // service launches so far (diagnostic)
	uint32_t pad0;
	uint32_t done_seq[SVC_MAX_SLOTS];   // per slot: sequence number of its last completed submission
	uint32_t pad[12];
};
struct SvcParams {
	SvcCtl *ctl;
	unsigned long long *ring;
	SvcSlot *slots;
	SvcHost *host;                      // device address of the mapped host block
	const uint32_t *cull_tab;           // the sector table every submission of this launch uses (null: every ray is cast)
	const float *ray_off;               // Buffers::ray_off of the context (the same for every slot)
	float max_gap;
	uint32_t tile_cap, list_cap;        // per-wave tile store and list sizes
	uint32_t part_words;                // LDS words per wave (WShared + tile store + lists + window)
	uint32_t ring_log2;
	uint32_t epoch;                     // of this launch (> 0)
	uint32_t idle_short, idle_long;     // in units of 1024 cycles
	uint32_t flags;                     // experiments (SMH_SVC_FLAGS; results may be WRONG): 1 = no cache invalidation per item, 2 = no write-back per frame, 4 = s_setprio 3, 8 = no helping among the waves of a workgroup
	                                    // 32 = the tile store is built by walking the bit rows' bounding box (rounds 2-5) instead of from the pass's tile-major mask (A/B)
	// help across workgroups (null / 0: none)
	SvcRemote *remote;                  // one per wave of the launch
	uint32_t *remote_store;             // owner o's tile store: remote_store + o * remote_store_words
	uint32_t remote_store_words;
	uint32_t remote_after;              // rounds a frame works (with its workgroup's help) before it asks the other workgroups
	uint32_t remote_tickets;            // helpers it asks for
	uint32_t remote_last_div;           // ... once it is among the last 1 / remote_last_div of its submission's frames still at work
	uint32_t compact;                   // the waves' tile stores sit behind the compact index (LSD_MODE_TILEC): decided by svc_waves_for for the frame size
	// (flags & 16: frames ask, and waves take tickets, whether or not frames are waiting for a wave -- A/B)
};
// waves per service workgroup and LDS per workgroup for this frame size (0 waves: the frame size does not fit -> no service)
// (*compact <- whether the tile stores use the compact index: where that lets one more wave fit)
uint32_t svc_waves_for(const Geom &g, uint32_t tile_limit, uint32_t *part_words, uint32_t *tile_cap, uint32_t *list_cap, uint32_t *lds_bytes, uint32_t *compact = nullptr);
uint32_t svc_store_words_for(const Geom &g, uint32_t tile_cap, uint32_t compact);   // words of a frame's tile store (tiles + index), rounded up to whole 16-byte quads
hipError_t svc_probe_host_atomics(SvcHost *h, SvcHost *d_h, bool *ok);   // pipeline creation: do device-side system-scope atomics reach mapped host memory?
hipError_t launch_svc_publish(SvcCtl *ctl, unsigned long long *ring, SvcSlot *slots, uint32_t slot, const Buffers &b, uint32_t n, uint32_t seq, uint32_t ring_log2, hipStream_t s);
hipError_t launch_lsd_service(const Geom &g, const SvcParams &p, uint32_t workgroups, uint32_t waves, uint32_t lds_bytes, hipStream_t s);

enum : uint32_t { MAP_UI = 1u, MAP_MASK = 2u, MAP_PRIO = 0x100u, MAP_BAND_MAJOR = 0x200u };   // MAP_BAND_MAJOR: the order of the work items (launch_map_brq_pass)
enum : uint32_t { BRQ_OCR = 1u, BRQ_SCALES = 2u };

hipError_t launch_button(const Geom &g, const Buffers &b, uint32_t n, int force_open, hipStream_t s);
// tiles_wanted: the batch path (the mask also tile-major where the bands allow it: band_rows_for); the per-call path leaves it
hipError_t launch_map_pass(const Geom &g, const Buffers &b, uint32_t n, uint32_t flags, int grayscale, hipStream_t s, bool tiles_wanted = false, bool overlapped = false);
// map pass + quadrant pass in one (the quadrant pixels are read once); flags: MAP_*, qflags: BRQ_*
// Occupancy policy of a pipelined batch (smhv_pipeline_create).  A streaming workgroup beyond the two per
// CU that saturate HBM only waits in the memory queues -- while holding wave slots and registers the other batches' line
// searches need -- so the pipeline (a) makes every streaming workgroup reserve enough LDS that a third does not fit on a
// CU and a line-search workgroup still does, and (b) caps the streaming grid (workgroups walk the (frame, band) items with
// a grid stride).  All zero: no policy (a batch that runs alone).
struct LaunchTuning {
	uint32_t map_lds_total;   // LDS a streaming workgroup occupies, static + dynamic, in bytes (0: what it needs)
	uint32_t map_grid_cap;    // streaming workgroups per launch (0: one per item)
	uint32_t lsd_tile_limit;  // k_lsd_tile keeps at most this many mask tiles in LDS (0: what fits the kernel's own budget)
	uint32_t map_prio;        // != 0: the streaming waves run at wave priority 3 -- ahead of the search service's waves on their SIMD,
	                          // which have slack (measured: 470 k -> 516 k frames/s at depth 12; beside the batch-granular search it cost 1-8 %)
	uint32_t map_deep;        // != 0: three register sets of loads in flight per wave instead of two (launch_map_brq_pass)
	uint32_t map_overlapped;  // != 0: the launch overlaps other kernels of its pipeline (the search service, or the searches and passes of a batch-granular
	                          // pipeline of depth >= 3): 56-row bands up to 1080p and frame-major work items, where a launch that runs ALONE (plain
	                          // runs, the per-call path, pipelines of depth 1 and 2) takes 24-row bands in band-major order (band_rows_for, k_map_brq_pass)
};
// LDS of one workgroup of the fused streaming pass without a reservation / of k_lsd_tile with `tile_cap` tiles (static + dynamic)
uint32_t map_brq_lds_bytes(const Geom &g);
uint32_t lsd_tile_lds_bytes(const Geom &g, uint32_t tile_limit);
hipError_t launch_map_brq_pass(const Geom &g, const Buffers &b, uint32_t n, uint32_t flags, uint32_t qflags, int grayscale, uint32_t fixed_start_y, int use_anchor_start, hipStream_t s,
                               const LaunchTuning *tune = nullptr);
// the fused streaming pass's loads and stores without its arithmetic (calibration: smhv_debug_pattern_copy)
hipError_t launch_pattern_copy(const Geom &g, const Buffers &b, uint32_t n, uint32_t rows_in_flight, hipStream_t s);
void map_band_rows(uint32_t rh, uint32_t n, int fused, uint32_t *rows, int *tiles);
hipError_t launch_brq_pass(const Geom &g, const Buffers &b, uint32_t n, uint32_t flags, uint32_t fixed_start_y, int use_anchor_start, hipStream_t s);
// k_lsd is three kernels, one per mask residency mode, each over all frames (a workgroup whose frame needs another mode
// exits at once).  When the whole ROI fits the LDS window (<= 1080p) every frame is a ROWS frame and only that kernel is
// launched; otherwise the three run concurrently on s, fk->s1 and fk->s2 (fork / join with the events), or back to back
// on s when fk is null.
struct LsdFork { hipStream_t s1, s2; hipEvent_t fork, join1, join2; };
bool lsd_rows_only(const Geom &g);   // every frame of this size is a ROWS frame: find_lines launches one kernel
// mode 0: find_lines (whole frame); mode 1: one find_longest_line round from (px,py), result in results[f].lines[0], len^2 in length_px[0]
// b.co (if present) is zeroed on `s` before the launch; extra_helpers: additional workgroups that only help (small batches).
// find_lines runs on k_lsd_tile unless the buffers carry the helper scheme (b.co), prefer_classic is set (the schedule knows
// better: smhv_pipeline at depth 2 on frames up to 1080p) or the process-wide diagnostic switch is.
// tile_bs: threads per workgroup of k_lsd_tile (0: 512; a batch that runs alone takes 1024)
// record_fused (optional) <- whether the launch writes the frames' records itself (b.rec_stages has SMH_REC_ON and the kernel
// picked is k_lsd_tile); otherwise the caller launches the record kernel behind it
hipError_t launch_lsd(const Geom &g, const Buffers &b, uint32_t n, float max_gap, int mode, float px, float py, hipStream_t s, const LsdFork *fk, uint32_t tile_bs = 0,
                      bool prefer_classic = false, uint32_t tile_limit = 0, bool *record_fused = nullptr);
size_t lsd_coop_ctl_bytes(uint32_t n);      // LsdCtl + n LsdCoop (one allocation, zeroed per launch)
hipError_t launch_scale_ratio(const Geom &g, const Buffers &b, uint32_t n, uint32_t *d_bars, hipStream_t s);
hipError_t launch_find_minimap(const Geom &g, const Buffers &b, uint32_t n, hipStream_t s);
hipError_t launch_finalize(const Geom &g, const Buffers &b, uint32_t n, uint32_t stages, hipStream_t s);
hipError_t launch_scales_finalize(const Geom &g, const Buffers &b, uint32_t n, uint32_t stages, uint32_t *d_bars, hipStream_t s);   // scale ratio + finalize
// which: SMHV_VIEW_*; isolated: LSDPreprocess shows the marker-isolated crop (after isolate_map_markers)
hipError_t launch_debug_view(const Geom &g, const Buffers &b, uint32_t frame, int which, int isolated, uint8_t *d_rgba, hipStream_t s);
hipError_t launch_marker_table(uint32_t *d_bits, hipStream_t s);
hipError_t launch_pack_rows(const void *d_src, uint32_t src_pitch_bytes, void *d_dst, uint32_t row_bytes, uint32_t rows, hipStream_t s);   // pitched rows -> a tight buffer (dword granularity)
hipError_t launch_side_probe(uint32_t *d_out, uint32_t workgroups, uint32_t spin, hipStream_t s);   // a 21 KB-LDS / 280-VGPR kernel that does nothing (co-residency probe)
hipError_t launch_build_sector_table(unsigned long long *d_tab, uint32_t T, hipStream_t s);
hipError_t launch_build_ray_offsets(float *d_off, hipStream_t s);   // 3600 x (SMH_RAY_OFF_BATCHES + 1) float2
size_t lsd_lds_bytes();
// diagnostic: find_lines on the workgroup-synchronous k_lsd for every frame instead of k_lsd_tile 
void lsd_set_classic(bool on);
// diagnostic: cap the tile store of k_lsd_tile (0 = what fits), to exercise the path of frames with more tiles than that
void lsd_set_tile_cap(uint32_t cap);
// diagnostic: threads per workgroup of every k_lsd_tile launch (0 = the caller's choice; 128 .. 1024)
void lsd_set_threads(uint32_t threads);
// diagnostic: watchdog budget of k_lsd_tile in idle polls (0 = default)
void lsd_set_spin_limit(uint32_t polls);
// overwrite the 3600 ray directions of the current device's code object (synchronous)
hipError_t set_ray_table(const float *dx, const float *dy);
// CRC-32 of n_dwords 32-bit words at d_msg, xor-ed into *d_acc (zero it first) WITHOUT the init / final-xor terms:
//   crc = *d_acc ^ crc32_mul(crc32_xpow(32 * n_dwords), 0xFFFFFFFF) ^ 0xFFFFFFFF.
// wgs workgroups of SMH_CRC_BS threads, rounds * wgs * SMH_CRC_BS * 4 >= n_dwords; x_skip = x^(128 (G - 1)),
// d_x_local[t] = x^(128 (SMH_CRC_BS - 1 - t)), d_x_wg[g] = x^(128 SMH_CRC_BS (wgs - 1 - g)), G = wgs * SMH_CRC_BS.
#define SMH_CRC_BS 1024
// decoder layouts -> BGRA8 (n_px pixels; layout = SMHV_PIXELS_*)
hipError_t launch_to_bgra(const void *d_src, void *d_bgra, uint64_t n_px, uint32_t layout, hipStream_t s);
hipError_t launch_unpack_rows(const void *d_pack, void *d_frame, uint32_t pitch_px, uint32_t roi_x, uint32_t roi_y, uint32_t roi_w, uint32_t roi_h, uint32_t btn_x,
                              uint32_t btn_y, uint32_t btn_w, uint32_t btn_h, hipStream_t s);
hipError_t launch_crc32(const void *d_msg, uint64_t n_dwords, uint32_t wgs, uint32_t rounds, uint32_t x_skip, const uint32_t *d_x_local,
                        const uint32_t *d_x_wg, uint32_t *d_acc, hipStream_t s);
uint32_t crc32_xpow(uint64_t n);                 // x^n mod P (reflected representation, x^0 = 0x80000000)
uint32_t crc32_mul(uint32_t a, uint32_t b);      // a * b mod P

}  // namespace smh
