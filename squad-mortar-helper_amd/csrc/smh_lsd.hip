// smh_lsd.hip -- k_lsd: lsd::find_lines::<32> incl. find_longest_line, one workgroup per frame (gfx950, wave64):
// speculative candidate groups, sector culling, batched ray walking, mask window resident in LDS
// (vision-common/src/lsd.rs:5-107, vision-cpu/src/lib.rs:387-449); k_build_sector_table.
//
// Build with -ffp-contract=off and correctly rounded f32 division: several results are truncated to integers
// right at a threshold, so the reference's scalar f32 operation order (no FMA contraction, IEEE divide) is
// part of the contract.  Semantics follow the reference's CPU back-end (vision-cpu/src/lib.rs) bit for bit;
// structure does not follow its CUDA file at all (SURVEY.md Appendix A lists how that differs).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "smh_device.h"
#include "smh_proximity.h"
#include "smh_record.inc"

namespace smh {

// ------------------------------------------------------------------------------------------------
// k_lsd: lsd::find_lines::<32> (vision-common/src/lsd.rs:60-107) with find_longest_line
// (vision-cpu/src/lib.rs:387-449) inlined; one 1024-thread workgroup per frame.
//
// The reference scans the mask in raster order and, per surviving white pixel, casts 3600 rays;
// on its CUDA back-end that is one kernel launch + sync per pixel.  Here the whole scan runs
// on-device against a window of the bit-packed mask held in LDS:
//   * candidates: the non-zero mask words are compacted in raster order; each thread owns one word
//     and keeps its "survivor" bits (white pixels not within sqrt(50) px of an accepted line);
//     new lines only ever clear bits, so the proximity tests are done once per (pixel, line).
//   * speculation: the next LSD_C survivors are ray-cast TOGETHER (a candidate's ray result does not
//     depend on the lines accepted before it, only whether it is visited does), then resolved in
//     raster order; a candidate invalidated by a line accepted earlier in the same group is dropped.
//   * rays: phase A walks every ray for its first 32 samples with no divergence (positions by
//     repeated f32 addition as in the reference, 32 LDS reads in flight, the gap state machine runs
//     on the 32-bit whiteness mask); almost all rays end there.  Survivors (rays running along a
//     marker line) are compacted into an LDS queue and finished in phase B with full lanes.
// Results are bit-identical to the sequential reference, including the max-len^2 / highest-angle
// tie rule and the sample counts.
// ------------------------------------------------------------------------------------------------
struct RayDir { uint32_t dx, dy; };
// Not const: smhv_set_ray_table lets a host whose libm is not glibc overwrite it with its own f32::cos / f32::sin values
// (the reference's ray directions come from the platform libm, vision-cpu/src/lib.rs:398-399).
__device__ RayDir g_ray_table[SMH_LSD_RAYS] = {
#include "ray_table.inc"
};

#ifndef LSD_BS
#define LSD_BS 1024
#endif
#define LSD_NW (LSD_BS / 64)
#ifndef LSD_C
#define LSD_C 8u                                   // candidates ray-cast per group
#endif
#ifndef LSD_C_RESET
#define LSD_C_RESET 2u
#endif
#define LSD_LOOKAHEAD 24u                          // candidates beyond its own group an owner posts for its helpers
#define LSD_GROUPS ((SMH_LSD_RAYS + 63) / 64)      // 64-ray units per candidate (57)
#define LSD_GROUPS_HOST 57
#define LSD_UNITS (LSD_C * LSD_GROUPS)
#define LSD_LIST_CAP 1024u
#define LSD_QCAP 1024u
#define LSD_QPT (LSD_QCAP / LSD_BS)                   // queue entries per thread in phase B
#define LSD_CACHE_MARGIN 72                         // rows around a candidate kept in the GLOBAL mode's LDS row cache (pass 1 walks 64 samples)
#define LSD_A_BATCHES 2u                            // 32-sample batches walked in phase A before a ray is queued
#define LSD_WIN_WORDS_CAP 27400u                   // 1080p whole ROI in ROWS mode = 824 rows x 33 words + 4 = 27196 words;
// with the candidate list, the ray queue, LsdShared and the sector culling table (12.7 KB) a workgroup takes 146 KB.
// That and the 96-VGPR cap on the kernels (amdgpu_waves_per_eu(5, 5): 4 waves x 96 of a SIMD's 512 registers, 48 bytes
// of scratch) leave room for ONE streaming workgroup of the other pipelined step (k_map_pass: 10.5 KB of LDS, one
// 96-VGPR wave per SIMD) on the same CU: k_lsd keeps its VALU busy 14 % of the time, so the HBM passes run underneath
// it: +12 % frames/s at pipeline depth 2 (-3 % at depth 1, the spills).  (Before the sector culling, when this
// kernel was VALU-bound, the same co-residency cost 2 %.)
#define LSD_ROWS_PITCH(gp) ((gp) | 1u)                 // LDS row pitch (words) of LSD_MODE_ROWS
#define LSD_DYN_LDS_BYTES ((LSD_WIN_WORDS_CAP + LSD_LIST_CAP + 2u * LSD_QCAP) * 4u)

// Window of the bit-packed mask.  In LDS it is the bounding box of the set bits plus a one-word /
// one-row border of zeros, and coordinates are clamped into it, so any read outside the box yields
// 0 without a branch.  In the global-memory fallback (box larger than LDS) it is the whole mask.
typedef __attribute__((address_space(3))) uint32_t LdsWord;
typedef __attribute__((address_space(3))) char LdsByte;
typedef __attribute__((address_space(3))) unsigned short LdsU16;

struct Win {
	const uint32_t *p;
	uint32_t pitch4;                  // row pitch in BYTES
	int y_lo, xbias;
	uint32_t rows_hi, cols_hi;
	uint32_t w, h;
	float wf, hf;
	// LSD_MODE_ROWS only: whole (bit-realigned: bit x of a row = pixel x) rows of the window in LDS
	const char *rows0;                // address of bit 0 of image row 0 (may lie before the buffer)
	float ylo_f, yhi_f;               // the two zero border rows, as floats
	// LSD_MODE_GLOBAL only: LDS copy of mask rows [c_y0, c_y0 + c_rows) in the global layout (same word columns);
	// every read tries it first.  The rows around the candidates being cast are kept resident (lsd_frame), so only
	// the long rays of phase B ever fall through to global memory.  c_rows == 0: no cache.
	const LdsWord *c_p;               // explicitly an LDS pointer: its reads must stay ds_read, apart from the global ones
	uint32_t c_y0, c_rows, c_pitch4;
	// LSD_MODE_TILE only: the mask as 32 x 8 px tiles.  t_idx points at the entry of tile (0, 0) of a 16-bit table with
	// t_pitch entries per tile row (two columns of padding on either side, one row above and below): 0 = an empty tile,
	// k = tile k of t_tiles (8 words, one per row; tile 0 is all zeros).
	const LdsU16 *t_idx;
	const LdsWord *t_tiles;
	uint32_t t_pitch;
	bool tiled;
	// LSD_MODE_TILEC only (the search service above 1080p): the same tiles behind a COMPACT index -- per tile row and group of 64
	// tile columns one 16-byte entry {occupancy mask (64 bits), number of the non-empty tiles before the group's first, 0}: tile
	// (row, column c of the group) is entry.base + popcount(mask below c) + 1 when bit c is set, the empty tile 0 otherwise
	// (tiles are numbered in raster order).  2.2 KB for a 1440p ROI where the 16-bit table takes 12.8 KB -- a quarter of what a
	// wave of the service may hold there -- for a handful of integer instructions per look-up.  tc_rows points at the entry of
	// (tile row 0, group 0); tile row -1 and the row behind the last are all-empty entries; column index = word column + 2 (the
	// same two padding columns as t_idx).
	const LdsWord *tc_rows;
	uint32_t tc_groups;
	bool compact;
	// LSD_MODE_WIN2 only (k_lsd_tile): the candidate's own window of the mask -- whole words, LSD_W2_PITCH words per row,
	// zero outside the image -- addressed directly: w2 is the byte address of (word 0, row 0) of the IMAGE, so the word of
	// pixel (x, y) sits at w2 + y * 4 LSD_W2_PITCH + 4 (x >> 5) for every (x, y) inside the window.
	const LdsByte *w2;
};
#define LSD_W2_PITCH 6u

// Mask residency modes of k_lsd, chosen per frame from the bounding box of the set bits:
//   ROWS   whole rows [y_min-1, y_max+1] in LDS, realigned so that bit x == pixel x.  The sample needs
//          no horizontal clamp: a batch never strays more than 33 px from the image (see ray_batch), and
//          the two words in front of / behind the block absorb that; out-of-window rows are clamped (in
//          the float domain, one v_med3_f32) onto the zero border rows.
//   XWIN   bounding box (+ zero border) in LDS, coordinates clamped into it (narrow, tall boxes)
//   GLOBAL the whole bit-packed mask in global memory (box larger than LDS: 1440p, 4K), with a sliding cache of as
//          many whole rows as LDS holds around the candidates in flight (they come in raster order)
//   TILE   (k_lsd_tile only) the non-empty 32 x 8 px tiles of the mask plus a 16-bit index over the whole ROI: a marker
//          mask is a few thin lines, 1-4 % of its tiles hold a set bit, so a frame of ANY size takes 20-50 KB of LDS
//          instead of 110 KB (1080p) or not fitting at all (1440p, 4K), and two frames share a CU.  One more dependent
//          LDS read per sample.
//   WIN2   (k_lsd_tile only) the first 64 steps of every ray of a candidate stay within 67 px of its pixel: the candidate's
//          neighbourhood is copied once (from the tile store) into a small window with a fixed pitch, and those steps --
//          most of all samples -- read it with one LDS access and no index look-up.
//   TILEC  (k_lsd_service above 1080p) TILE with the compact index (Win::tc_rows)
enum { LSD_MODE_ROWS = 0, LSD_MODE_XWIN = 1, LSD_MODE_GLOBAL = 2, LSD_MODE_TILE = 3, LSD_MODE_WIN2 = 4, LSD_MODE_TILEC = 5 };

// One sample.  Coordinates below/left of the window wrap to huge unsigned values and clamp to the far
// (zero) border just like coordinates beyond it, so each axis costs one v_min_u32.  24-bit multiply:
// rows * pitch4 < 2^24 for any supported frame.
template <bool CACHED = false>
__device__ __forceinline__ uint32_t win_raw(const Win &m, int xi, int yi) {   // bit 0 = the pixel, upper bits garbage
	const uint32_t ry = min((uint32_t)(yi - m.y_lo), m.rows_hi);
	const int X = xi + m.xbias;
	const uint32_t rc = min((uint32_t)(X >> 5), m.cols_hi);
	uint32_t word;
	const uint32_t cr = ry - m.c_y0;                       // GLOBAL: y_lo = 0, ry is the image row
	if (CACHED && cr < m.c_rows) word = *(const LdsWord *)((const LdsByte *)m.c_p + __umul24(cr, m.c_pitch4) + (rc << 2));
	else word = *(const uint32_t *)((const char *)m.p + __umul24(ry, m.pitch4) + (rc << 2));
	return word >> ((uint32_t)X & 31u);
}
// LSD_MODE_TILE: word `wq` (pixels 32 wq .. 32 wq + 31) of image row yi; -2 <= wq < t_pitch - 2, -8 <= yi < 8 (tile rows + 1)
__device__ __forceinline__ uint32_t tile_word(const Win &m, int wq, int yi) {
	const uint32_t t = m.t_idx[__mul24(yi >> 3, (int)m.t_pitch) + wq];
	return m.t_tiles[(t << 3) + ((uint32_t)yi & 7u)];
}
// LSD_MODE_TILEC: the tile number of (word column wq, tile row ty) through the compact index; -2 <= wq < t_pitch - 2, -1 <= ty <= rows
typedef uint32_t lds_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t tilec_number(const Win &m, int wq, int ty) {
	const uint32_t col = (uint32_t)(wq + 2);
	const lds_u32x4 e = *(const __attribute__((address_space(3))) lds_u32x4 *)(m.tc_rows + (__mul24(ty, (int)m.tc_groups) + (int)(col >> 6)) * 4);
	const uint32_t c = col & 63u;
	// bits below c of the 64-bit mask, as two halves
	const uint32_t lo_m = c >= 32u ? 0xFFFFFFFFu : ((1u << c) - 1u), hi_m = c > 32u ? ((1u << (c - 32u)) - 1u) : 0u;
	const uint32_t bit = ((c >= 32u ? e.y >> (c - 32u) : e.x >> c) & 1u);
	const uint32_t before = (uint32_t)__builtin_popcount(e.x & lo_m) + (uint32_t)__builtin_popcount(e.y & hi_m);
	return bit ? e.z + before + 1u : 0u;
}
__device__ __forceinline__ uint32_t tilec_word(const Win &m, int wq, int yi) {
	return m.t_tiles[(tilec_number(m, wq, yi >> 3) << 3) + ((uint32_t)yi & 7u)];
}
__device__ __forceinline__ uint32_t tilec_raw(const Win &m, float x, float y) {
	const int yi = (int)__builtin_amdgcn_fmed3f(y, -1.0f, m.hf);
	const int xi = (int)x;
	return tilec_word(m, xi >> 5, yi) >> ((uint32_t)xi & 31u);
}
// (eight samples of a ray, as tile_raw8: the eight index entries in one round trip, the eight tile words in a second)
__device__ __forceinline__ void tilec_raw8(const Win &m, float xs, float ys, float dx, float dy, float &xo, float &yo, float &x, float &y, uint32_t &Wm) {
	int yi[8], xi[8];
	uint32_t t[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		x = xo + xs; y = yo + ys;
		yi[j] = (int)__builtin_amdgcn_fmed3f(y, -1.0f, m.hf);
		xi[j] = (int)x;
		xo += dx; yo += dy;
	}
#pragma unroll
	for (int j = 0; j < 8; ++j) t[j] = tilec_number(m, xi[j] >> 5, yi[j] >> 3);
#pragma unroll
	for (int j = 0; j < 8; ++j) t[j] = m.t_tiles[(t[j] << 3) + ((uint32_t)yi[j] & 7u)];
#pragma unroll
	for (int j = 0; j < 8; ++j) Wm = __builtin_amdgcn_alignbit(t[j] >> ((uint32_t)xi[j] & 31u), Wm, 1);
}
// LSD_MODE_TILE sample straight from the float position (bit 0 = the pixel): rows clamped onto the zero rows -1 and h in
// the float domain; x needs no clamp (a batch never strays more than 33 px from the image: two padding columns)
__device__ __forceinline__ uint32_t tile_raw(const Win &m, float x, float y) {
	const int yi = (int)__builtin_amdgcn_fmed3f(y, -1.0f, m.hf);
	const int xi = (int)x;
	return tile_word(m, xi >> 5, yi) >> ((uint32_t)xi & 31u);
}
// Eight consecutive LSD_MODE_TILE samples of a ray: the eight index entries, then the eight tile words, then the bits -- two
// LDS round trips for the group.  Written sample by sample (tile_raw in an unrolled loop) the compiler waits for every
// sample's index entry before it issues that sample's word read: sixteen dependent round trips per group, and a long ray
// is nothing but such groups.  Same additions in the same order as the scalar form (x = xo + xs, then xo += dx).
__device__ __forceinline__ void tile_raw8(const Win &m, float xs, float ys, float dx, float dy, float &xo, float &yo, float &x, float &y, uint32_t &Wm) {
	int yi[8], xi[8];
	uint32_t t[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		x = xo + xs; y = yo + ys;
		yi[j] = (int)__builtin_amdgcn_fmed3f(y, -1.0f, m.hf);
		xi[j] = (int)x;
		xo += dx; yo += dy;
	}
#pragma unroll
	for (int j = 0; j < 8; ++j) t[j] = m.t_idx[__mul24(yi[j] >> 3, (int)m.t_pitch) + (xi[j] >> 5)];
#pragma unroll
	for (int j = 0; j < 8; ++j) t[j] = m.t_tiles[(t[j] << 3) + ((uint32_t)yi[j] & 7u)];
#pragma unroll
	for (int j = 0; j < 8; ++j) Wm = __builtin_amdgcn_alignbit(t[j] >> ((uint32_t)xi[j] & 31u), Wm, 1);
}
// LSD_MODE_WIN2 sample straight from the float position (bit 0 = the pixel).  No clamp: the caller guarantees the position
// lies inside the window (rows and columns outside the image are zero there); (int) of a position in (-1, 0) is 0 -- an
// out-of-image sample either way, masked by the caller like every sample behind the image's edge.
__device__ __forceinline__ uint32_t win2_raw(const Win &m, float x, float y) {
	const int yi = (int)y, xi = (int)x;
	const uint32_t word = *(const LdsWord *)(m.w2 + __mul24(yi, (int)(4u * LSD_W2_PITCH)) + ((xi >> 3) & ~3));
	return word >> ((uint32_t)xi & 31u);
}
__device__ __forceinline__ uint32_t win2_word(const Win &m, int wq, int yi) { return *(const LdsWord *)(m.w2 + __mul24(yi, (int)(4u * LSD_W2_PITCH)) + (wq << 2)); }
__device__ __forceinline__ uint32_t win2_bit(const Win &m, int xi, int yi) { return (win2_word(m, xi >> 5, yi) >> ((uint32_t)xi & 31u)) & 1u; }

// (not on the hot path: the cache test is a run-time one here)
__device__ __forceinline__ uint32_t win_bit(const Win &m, int xi, int yi) {
	if (m.tiled) return ((m.compact ? tilec_word(m, xi >> 5, yi) : tile_word(m, xi >> 5, yi)) >> ((uint32_t)xi & 31u)) & 1u;      // callers pass in-image coordinates
	return (m.c_rows ? win_raw<true>(m, xi, yi) : win_raw<false>(m, xi, yi)) & 1u;
}

// The 32 mask bits of word `wq` (in the view's own bit coordinate B = x + xbias) of image row yi; 0 outside the
// window / image.  Works for all three residency modes (ROWS: xbias = 0).
__device__ __forceinline__ uint32_t win_word(const Win &m, int wq, int yi) {
	if (m.tiled) return ((uint32_t)yi < m.h && (uint32_t)(wq + 2) < m.t_pitch) ? (m.compact ? tilec_word(m, wq, yi) : tile_word(m, wq, yi)) : 0u;
	const uint32_t ry = (uint32_t)(yi - m.y_lo), rc = (uint32_t)wq;
	uint32_t v = 0;
	if (ry <= m.rows_hi && rc <= m.cols_hi) {
		const uint32_t cr = ry - m.c_y0;
		if (cr < m.c_rows) v = *(const LdsWord *)((const LdsByte *)m.c_p + __umul24(cr, m.c_pitch4) + (rc << 2));
		else v = *(const uint32_t *)((const char *)m.p + __umul24(ry, m.pitch4) + (rc << 2));
	}
	return v;
}

// LSD_MODE_ROWS sample straight from the float position (bit 0 = the pixel).
__device__ __forceinline__ uint32_t win_raw_rows(const Win &m, float x, float y) {
	const int yi = (int)__builtin_amdgcn_fmed3f(y, m.ylo_f, m.yhi_f);
	const int xi = (int)x;
	const char *row = m.rows0 + __mul24(yi, (int)m.pitch4);
	const uint32_t word = *(const uint32_t *)(row + ((xi >> 5) << 2));
	return word >> ((uint32_t)xi & 31u);
}

__device__ __forceinline__ bool in_image(const Win &m, float x, float y) { return x >= 0.0f && y >= 0.0f && x < m.wf && y < m.hf; }

// find_line_in_image closure, vision-cpu/src/lib.rs:388-432, literally.  Used for max_gap values the
// batched walker does not cover (<= 0, NaN, huge) -- never in production (max_gap = 15).
__device__ void cast_ray_literal(const Win &m, float xs, float ys, float max_gap, float dx, float dy, float &xe, float &ye, uint32_t &steps) {
	float x = xs, y = ys, xo = 0.0f, yo = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
	while (in_image(m, x, y)) {
		++steps;
		if (win_bit(m, (int)x, (int)y)) {
			g0 = 0.0f; g1 = 0.0f; g2 = 0.0f;
		} else if (g0 >= max_gap) {
			x = g1; y = g2;
			break;
		} else if (g0 == 0.0f) {
			g0 = 1.0f; g1 = x; g2 = y;
		} else {
			g0 += 1.0f;
		}
		xo += dx; yo += dy;
		x = xo + xs; y = yo + ys;
	}
	xe = xs; ye = ys;
	const uint32_t xi = f2u(x), yi = f2u(y);
	if (xi < m.w && yi < m.h && win_bit(m, (int)xi, (int)yi) == 0u) { xe = x - dx; ye = y - dy; }
}

// State of a ray between 32-sample batches.  The reference's gap tuple (count, saved_x, saved_y) is
// kept as: g = consecutive non-white samples so far, and where that run started as (offsets at the
// start of ITS batch, index in that batch, first step of that batch) -- the saved position is
// re-derived by replaying the additions only for rays that can still win (see ray_engine).
struct RayState {
	float bxo, byo;        // x_offset / y_offset at the start of the current batch
	uint32_t k0, g;        // samples taken before this batch; current non-white run length
	float gxo, gyo;        // offsets at the start of the batch in which the current run began
	uint32_t gj, gk0;      // ... its index in that batch, and that batch's first step
	uint32_t nexit;        // status 2: number of in-image samples of the last batch
};
enum { RAY_CONTINUE = 0, RAY_ABORTED = 1, RAY_LEFT_IMAGE = 2 };

// One batch of up to 32 samples.  T = smallest integer >= max_gap (>= 1): the reference aborts at a
// non-white sample when the run count before it is >= max_gap, i.e. at the (T+1)-th consecutive
// non-white sample, and restores the position where that run started.
//   RAY_ABORTED    : the run started at global step s.gk0 + s.gj; samples taken = *steps
//   RAY_LEFT_IMAGE : the first out-of-image position is step s.k0 + s.nexit
// G: samples whose LDS reads are in flight together (8, 16 or 32).
template <int MODE, int G = 8>
__device__ __forceinline__ int ray_batch(const Win &m, float xs, float ys, float dx, float dy, uint32_t T, RayState &s, uint32_t &steps) {
	static_assert((G == 8 || G == 16 || G == 32) && ((MODE != LSD_MODE_TILE && MODE != LSD_MODE_TILEC) || G == 8), "group size");
	float xo = s.bxo, yo = s.byo, x = 0.0f, y = 0.0f;
	uint32_t Wm = 0;
	// 4 x 8 samples: eight LDS reads in flight per wave keep the register footprint small enough for
	// 16 waves per CU; the whiteness bits are shifted in from the top (sample j ends up in bit j).  (G = 16 / 32: the
	// one-wave-per-frame scan of the search service, a single wave on its SIMD beside the streaming pass, where an LDS round
	// trip takes several hundred cycles and there is nobody else to hide it behind.)
	uint32_t taken = 0;
#pragma unroll 1
	for (int jj = 0; jj < 32 / G; ++jj) {
		if constexpr (MODE == LSD_MODE_TILE) tile_raw8(m, xs, ys, dx, dy, xo, yo, x, y, Wm);
		else if constexpr (MODE == LSD_MODE_TILEC) tilec_raw8(m, xs, ys, dx, dy, xo, yo, x, y, Wm);
		else
#pragma unroll
		for (int j = 0; j < G; ++j) {
			x = xo + xs; y = yo + ys;                   // x = x_offset + x_start
			// shifts in bit 0 of its first operand
			if (MODE == LSD_MODE_ROWS) Wm = __builtin_amdgcn_alignbit(win_raw_rows(m, x, y), Wm, 1);
			else if (MODE == LSD_MODE_WIN2) Wm = __builtin_amdgcn_alignbit(win2_raw(m, x, y), Wm, 1);
			else if (MODE == LSD_MODE_TILE) Wm = __builtin_amdgcn_alignbit(tile_raw(m, x, y), Wm, 1);
			else Wm = __builtin_amdgcn_alignbit(win_raw<MODE == LSD_MODE_GLOBAL>(m, (int)x, (int)y), Wm, 1);
			xo += dx; yo += dy;                         // x_offset += dx
		}
		taken += (uint32_t)G;
		// Early out: when the most recent T+1 samples of EVERY lane of the wave are non-white, every
		// ray of the wave has aborted inside this batch (most rays die T+1 samples after leaving the
		// blob they start in), so the remaining samples of the batch cannot matter.
		if (taken > T && taken < 32u && __all((Wm >> (31u - T)) == 0u)) break;
	}
	Wm >>= 32u - taken;                                 // sample j -> bit j
	// x(j) and y(j) are monotonic in j, so the in-image samples are a prefix: test the last one.
	uint32_t n = taken;
	if (!in_image(m, x, y)) {
		float rx = s.bxo, ry = s.byo;
#pragma unroll 1
		for (n = 0; n < taken; ++n) {
			if (!in_image(m, rx + xs, ry + ys)) break;
			rx += dx; ry += dy;
		}
	}
	const uint32_t valid = n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u);
	const uint32_t Wv = Wm & valid, Z = ~Wm & valid;
	const uint32_t g = s.g;
	if (T <= 31u) {
		// Branch-free gap state machine: prepend the g carried-in non-white samples, then find the first
		// run of T+1 consecutive non-white samples by AND-ing shifted copies (run-length doubling).
		const unsigned long long Z64 = ((unsigned long long)Z << g) | ((1ull << g) - 1ull);
		unsigned long long R = Z64;
		const uint32_t L = T + 1u;
		uint32_t have = 1u;
		while (have * 2u <= L) { R &= R >> have; have *= 2u; }
		if (L > have) R &= R >> (L - have);
		if (R) {
			const uint32_t p = (uint32_t)__builtin_ctzll(R);    // start of that run (a run start, or 0 = carried in)
			if (p >= g) { s.gxo = s.bxo; s.gyo = s.byo; s.gj = p - g; s.gk0 = s.k0; }
			steps += s.k0 + (p + T - g) + 1u;
			return RAY_ABORTED;
		}
		if (n < taken) { steps += s.k0 + n; s.nexit = n; return RAY_LEFT_IMAGE; }
		// no abort in a full batch with T <= 31 implies at least one white sample (and taken == 32: the
		// early out above is taken only when every lane aborts)
		const uint32_t msb = 31u - (uint32_t)__builtin_clz(Wv);
		s.g = 31u - msb;
		if (s.g) { s.gxo = s.bxo; s.gyo = s.byo; s.gj = msb + 1u; s.gk0 = s.k0; }
	} else {
		uint32_t rem = valid, gg = g;
		while (rem) {
			const uint32_t pos = (uint32_t)__builtin_ctz(rem);
			if ((Wv >> pos) & 1u) {                         // white run: the gap state resets
				gg = 0;
				const uint32_t z = Z & rem;
				if (!z) break;
				rem &= 0xFFFFFFFFu << __builtin_ctz(z);
			} else {                                        // non-white run [pos, end)
				const uint32_t wr = Wv & rem;
				const uint32_t end = wr ? (uint32_t)__builtin_ctz(wr) : n;
				const uint32_t len = end - pos;
				if (gg == 0) { s.gxo = s.bxo; s.gyo = s.byo; s.gj = pos; s.gk0 = s.k0; }
				if (gg + len > T) {                         // "gap didn't close, abort"
					steps += s.k0 + pos + (T - gg) + 1u;
					return RAY_ABORTED;
				}
				gg += len;
				if (end >= 32u) break;
				rem &= 0xFFFFFFFFu << end;
			}
		}
		if (n < 32u) { steps += s.k0 + n; s.nexit = n; return RAY_LEFT_IMAGE; }
		s.g = gg;
	}
	s.bxo = xo; s.byo = yo; s.k0 += 32u;
	return RAY_CONTINUE;
}

// Exact end point of a finished ray (vision-cpu/src/lib.rs:415-429), by replaying the additions.
__device__ __forceinline__ void ray_endpoint(const Win &m, int status, const RayState &s, float xs, float ys, float dx, float dy, float &xe, float &ye) {
	if (status == RAY_ABORTED) {                            // restore the gap start; that pixel is in the image and 0
		float gx = s.gxo, gy = s.gyo;
		for (uint32_t k = 0; k < s.gj; ++k) { gx += dx; gy += dy; }
		xe = (gx + xs) - dx; ye = (gy + ys) - dy;
	} else {                                                // walked out of the image
		float rx = s.bxo, ry = s.byo;
		for (uint32_t k = 0; k < s.nexit; ++k) { rx += dx; ry += dy; }
		const float px = rx + xs, py = ry + ys;
		xe = xs; ye = ys;
		const uint32_t xi = f2u(px), yi = f2u(py);          // get_pixel_checked(x as u32, y as u32) == Some(0)
		if (xi < m.w && yi < m.h && win_bit(m, (int)xi, (int)yi) == 0u) { xe = px - dx; ye = py - dy; }
	}
}

// get_centre, vision-common/src/lsd.rs:5-44 (coordinates clamped like the oracle).
__device__ __forceinline__ uint32_t white_at(const Win &m, float fx, float fy) {
	const uint32_t xi = min(f2u(fx), m.w - 1u), yi = min(f2u(fy), m.h - 1u);
	return win_bit(m, (int)xi, (int)yi);
}
__device__ void get_centre(const Win &m, float px, float py, float &ox, float &oy) {
	float left = px;
	while (left > 0.0f && fabsf(left - px) < SMH_LSD_CENTRE_REACH && white_at(m, left, py)) left -= 1.0f;
	float right = px;
	while (right < (float)(m.w - 1u) && fabsf(right - px) < SMH_LSD_CENTRE_REACH && white_at(m, right, py)) right += 1.0f;
	float up = py;
	while (up > 0.0f && fabsf(up - py) < SMH_LSD_CENTRE_REACH && white_at(m, px, up)) up -= 1.0f;
	float down = py;
	while (down < (float)(m.h - 1u) && fabsf(down - py) < SMH_LSD_CENTRE_REACH && white_at(m, px, down)) down += 1.0f;
	ox = (left + right) / 2.0f;
	oy = (up + down) / 2.0f;
}

// near_line (lsd.rs:47-58,84-89) and its cheap word-level classifier live in smh_proximity.h (shared with a host test)

// Diagnostic build only (-DSMH_LSD_PROFILE): thread 0 accumulates s_memtime deltas per phase and stores
// them in the tail of the record's `meters` array (never read by product code in that build).
#ifdef SMH_LSD_PROFILE
#define PROF_DECL unsigned long long prof_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_last = __builtin_amdgcn_s_memtime();
#define PROF_MARK(i) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof_t[i] += _n - prof_last; prof_last = _n; } while (0)
#define PROF_STORE(res) do { if (threadIdx.x == 0) for (int _i = 0; _i < 12; ++_i) ((unsigned long long *)(res)->meters)[20 + _i] = prof_t[_i]; } while (0)
#define PROF_ARG , prof_t
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_STORE(res)
#define PROF_ARG
#endif

struct LsdShared {
	unsigned long long cand_best[LSD_C];   // max over rays of (len^2 bits << 32 | ray index): ties -> highest angle
	unsigned long long unit_key[LSD_UNITS];
	float unit_end[LSD_UNITS][2];
	uint32_t unit_kmax[LSD_UNITS];
	float cand_pt[LSD_C][2];
	float cand_end[LSD_C][2];
	uint32_t cand_steps[LSD_C];
	uint32_t cand_kmax[LSD_C];
	uint32_t cand_key[LSD_C];
	uint32_t scan[LSD_NW];
	uint32_t qtail, segnext, unit_next;
	uint32_t cand_px[LSD_C][2];              // the candidates' pixels (start of get_centre)
	uint32_t cand_wkey[LSD_C];               // their raster keys (word index << 5 | bit, + 1)
	uint32_t cached;                         // candidates whose result came out of the helpers' cache
	uint32_t post[LSD_LOOKAHEAD];            // look-ahead candidates to post (valid << 31 | pixel)
	uint32_t req_tail, posted_hi, posted_new, helpers_seen, head_seen;
	uint32_t h_n;                            // helper: claimed requests / flags
	unsigned long long pick;
	uint32_t nlive;                          // live units of the group, listed (in no particular order) in ulist
	unsigned short ulist[LSD_UNITS];
	unsigned long long live[LSD_C];          // units (64-ray sectors) of each candidate that have to be cast
	float lines[SMH_LSD_MAX_LINES][4];
	// prox_line() of each accepted line (smh_proximity.h): slope of the signed distance along x, unit direction
	float prox_a[SMH_LSD_MAX_LINES];
	double prox_d[SMH_LSD_MAX_LINES][2];
#ifdef SMH_LSD_PROFILE
	uint32_t exp_far, exp_units;
#endif
};

__device__ __forceinline__ ProxLine shared_prox_line(const LsdShared &sh, uint32_t l) {
	ProxLine L;
	L.x0 = sh.lines[l][0]; L.y0 = sh.lines[l][1]; L.x1 = sh.lines[l][2]; L.y1 = sh.lines[l][3];
	L.a = sh.prox_a[l]; L.dxl = sh.prox_d[l][0]; L.dyl = sh.prox_d[l][1];
	L.degenerate = L.dxl == 0.0 && L.dyl == 0.0;
	return L;
}

// find_longest_line for nc candidates at once (start points in sh.cand_pt).  On return (after a
// barrier) sh.cand_best / cand_end / cand_steps hold, per candidate, the winning key, its end point
// and the number of mask samples all 3600 rays took.
//
// Only the max-len^2 ray matters.  A ray that aborted with its gap starting at step K ends K-1 unit
// steps from the start (to within accumulated f32 rounding < 0.25 for the longest possible ray), so
// pass 1 only records K per ray and the largest K per 64-ray unit; the exact end point (a replay of
// the additions) is computed in pass 2, and only for the rays of units whose K comes within 2 of the
// candidate's largest K.  Every ray that could win or tie is therefore still evaluated exactly, with
// the reference's arithmetic; rays that left the image (which may legitimately end with length 0)
// and the long rays of phase B are always evaluated exactly.
template <int MODE>
__device__ void ray_engine(const Win &m, LsdShared &sh, uint32_t *queue, uint32_t nc, float max_gap, bool always_exact, unsigned long long *prof_t = nullptr) {
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef SMH_LSD_PROFILE
	unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif
	const bool fast_gap = max_gap > 0.0f && max_gap <= 60000.0f;
	const uint32_t T = fast_gap ? (uint32_t)ceilf(max_gap) : 0u;
	const uint32_t nunits = nc * LSD_GROUPS, nlive = sh.nlive;

	// ---- pass 1: first LSD_A_BATCHES x 32 samples of every ray of the live units; unit = (candidate, 64 consecutive angles).
	// Units cost one or two batches depending on the scene, so waves pull them from a shared counter
	// (the next unit and its ray directions are fetched while the current one is walked).
	uint32_t pos = wave;                                   // first unit is static; sh.unit_next starts at LSD_NW
	uint32_t pn = 0;
	if (lane == 0) pn = atomicAdd(&sh.unit_next, 1u);
	pn = (uint32_t)__builtin_amdgcn_readfirstlane((int)pn);
	uint32_t u = pos < nlive ? sh.ulist[pos] : 0u;
	RayDir nd = g_ray_table[min((u % LSD_GROUPS) * 64u + lane, (uint32_t)SMH_LSD_RAYS - 1u)];
	for (; pos < nlive; ) {
		const uint32_t c = u / LSD_GROUPS, i = (u - c * LSD_GROUPS) * 64u + lane;
		const bool valid = i < SMH_LSD_RAYS;
		const float dx = __uint_as_float(nd.dx), dy = __uint_as_float(nd.dy);
		const uint32_t ucur = u;
		pos = pn;
		if (pos < nlive) {
			u = sh.ulist[pos];
			nd = g_ray_table[min((u % LSD_GROUPS) * 64u + lane, (uint32_t)SMH_LSD_RAYS - 1u)];
			if (lane == 0) pn = atomicAdd(&sh.unit_next, 1u);
			pn = (uint32_t)__builtin_amdgcn_readfirstlane((int)pn);
		}
		const float xs = sh.cand_pt[c][0], ys = sh.cand_pt[c][1];
		// the batched walker needs a start inside the image (always true for find_lines candidates)
		const bool fast = fast_gap && in_image(m, xs, ys);
		float xe = xs, ye = ys;
		uint32_t steps = 0;
		int status = RAY_ABORTED;                          // invalid lanes: "finished", K = 0
		bool exact = false;                                // end point computed in this pass
		RayState s = {0.0f, 0.0f, 0u, 0u, 0.0f, 0.0f, 0u, 0u, 0u};
		if (valid) {
			if (fast) {
				status = RAY_CONTINUE;
#pragma unroll 1
				for (uint32_t bi = 0; bi < LSD_A_BATCHES && status == RAY_CONTINUE; ++bi) status = ray_batch<MODE>(m, xs, ys, dx, dy, T, s, steps);
			} else {
				cast_ray_literal(m, xs, ys, max_gap, dx, dy, xe, ye, steps);
				exact = true;
			}
		}
#ifdef SMH_LSD_PROFILE
		if (prof_t) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof_t[0] += _n - prof_last; prof_last = _n; }
#endif
		const uint64_t sv = __ballot(status == RAY_CONTINUE);
		if (sv) {
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(&sh.qtail, (uint32_t)__popcll(sv));
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
			const uint32_t slot = base + (uint32_t)__popcll(sv & ((1ull << lane) - 1ull));
			if (status == RAY_CONTINUE && slot >= LSD_QCAP) {      // queue full: finish this ray here, exactly
				while (status == RAY_CONTINUE) status = ray_batch<MODE>(m, xs, ys, dx, dy, T, s, steps);
				ray_endpoint(m, status, s, xs, ys, dx, dy, xe, ye);
				exact = true;
			} else if (status == RAY_CONTINUE) {
				queue[2u * slot] = (c << 16) | i;
				queue[2u * slot + 1u] = s.g | (s.gj << 16) | ((s.gk0 >> 5) << 24);
			}
		}
		const bool fin = valid && status != RAY_CONTINUE;
		// find_lines (always_exact == false): a ray whose gap starts at step K >= 50 gets its exact end point right here, and
		// the second pass below is not needed -- a ray with K <= 49 is shorter than 50 px (it ends K - 1 unit steps from the
		// start, +-0.25), so it never wins in a candidate find_lines keeps (len^2 > 2500, lsd.rs:94), and for one it rejects
		// nobody asks which ray was the longest.  Vision::find_longest_line still takes the second pass (exact shorter rays).
		if (fin && !exact && (status == RAY_LEFT_IMAGE || (!always_exact && s.gk0 + s.gj > SMH_LSD_REJECT_K))) { ray_endpoint(m, status, s, xs, ys, dx, dy, xe, ye); exact = true; }
		// exactly evaluated rays (rare) compete right away
		if (__any(fin && exact)) {
			unsigned long long key = 0;
			if (fin && exact) {
				const float ddx = xs - xe, ddy = ys - ye;       // p0.distance_sqr(&p1)
				key = ((unsigned long long)__float_as_uint(ddx * ddx + ddy * ddy) << 32) | i;
			}
			const unsigned long long wkey = wave_max64(key);
			if (lane == 0) atomicMax(&sh.cand_best[c], wkey);
			if (key != 0ull && key == wkey) { sh.unit_key[ucur] = key; sh.unit_end[ucur][0] = xe; sh.unit_end[ucur][1] = ye; }
		}
		const uint32_t Kw = wave_max32_dpp((fin && !exact) ? s.gk0 + s.gj : 0u);
		const uint32_t wsteps = wave_sum32_dpp(steps);
		if (lane == 0) { sh.unit_kmax[ucur] = Kw; atomicMax(&sh.cand_kmax[c], Kw); atomicAdd(&sh.cand_steps[c], wsteps); }
#ifdef SMH_LSD_PROFILE
		if (prof_t) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof_t[1] += _n - prof_last; prof_last = _n; }
		{   // experiment: how many rays of the live units get anywhere near the acceptance length?
			const uint32_t far = (uint32_t)__popcll(__ballot(valid && (status == RAY_CONTINUE || s.gk0 + s.gj >= 35u || steps >= 36u)));
			if (lane == 0) { atomicAdd(&sh.exp_far, far); atomicAdd(&sh.exp_units, 1u); }
		}
#endif
	}
	__syncthreads();
	PROF_MARK(3);

	// ---- phase B: the few long rays, packed 64 to a wave; always exact ----
	const uint32_t Q = min(sh.qtail, LSD_QCAP);
#ifdef SMH_LSD_PROFILE
	if (prof_t) { prof_t[6] += sh.exp_far; prof_t[7] += sh.exp_units; }
#endif
	unsigned long long bkey[LSD_QPT];
	float bxe[LSD_QPT], bye[LSD_QPT];
	uint32_t bc[LSD_QPT];
#pragma unroll
	for (int r = 0; r < (int)LSD_QPT; ++r) { bkey[r] = 0ull; bxe[r] = 0.0f; bye[r] = 0.0f; bc[r] = 0u; }
#pragma unroll
	for (int r = 0; r < (int)LSD_QPT; ++r) {
		const uint32_t e = tid + (uint32_t)r * LSD_BS;
		if (e < Q) {
			const uint32_t id = queue[2u * e], st = queue[2u * e + 1u];
			const uint32_t c = id >> 16, i = id & 0xFFFFu;
			const float dx = __uint_as_float(g_ray_table[i].dx), dy = __uint_as_float(g_ray_table[i].dy);
			const float xs = sh.cand_pt[c][0], ys = sh.cand_pt[c][1];
			RayState s;
			s.g = st & 0xFFFFu; s.gj = (st >> 16) & 31u; s.gk0 = (st >> 24) << 5; s.k0 = 32u * LSD_A_BATCHES; s.nexit = 0u;
			float xo = 0.0f, yo = 0.0f;
			s.gxo = 0.0f; s.gyo = 0.0f;
			for (uint32_t k = 0; k < 32u * LSD_A_BATCHES; ++k) {      // replay the additions of pass 1
				if (k == s.gk0) { s.gxo = xo; s.gyo = yo; }
				xo += dx; yo += dy;
			}
			s.bxo = xo; s.byo = yo;
			float xe = xs, ye = ys;
			uint32_t steps = 0;
			int status = RAY_CONTINUE;
			while (status == RAY_CONTINUE) status = ray_batch<MODE>(m, xs, ys, dx, dy, T, s, steps);
			ray_endpoint(m, status, s, xs, ys, dx, dy, xe, ye);
			const float ddx = xs - xe, ddy = ys - ye;
			const unsigned long long key = ((unsigned long long)__float_as_uint(ddx * ddx + ddy * ddy) << 32) | i;
			bkey[r] = key; bxe[r] = xe; bye[r] = ye; bc[r] = c;
			atomicMax(&sh.cand_best[c], key);
			atomicAdd(&sh.cand_steps[c], steps);
			if (status == RAY_ABORTED) atomicMax(&sh.cand_kmax[c], s.gk0 + s.gj);   // raises the bar for pass 2
		}
	}
	__syncthreads();
	PROF_MARK(4);

	// ---- pass 2 (Vision::find_longest_line only): exact end points for the units that can still hold the winner ----
	if (fast_gap && always_exact) {
		for (uint32_t p2 = wave; p2 < nlive; p2 += LSD_NW) {
			const uint32_t u2 = sh.ulist[p2], c = u2 / LSD_GROUPS;
			const uint32_t kbar = sh.cand_kmax[c];
			if (sh.unit_kmax[u2] + 2u < kbar) continue;        // wave-uniform
			// find_lines only keeps a candidate whose best ray has len^2 > 2500 (lsd.rs:94).  With every
			// aborted ray's gap starting at step <= 49 its length is < 48.3 < 50, so the candidate is rejected
			// whatever the exact end points are (rays evaluated exactly in pass 1 / phase B already compete in
			// cand_best): no need for them.  Vision::find_longest_line (mode 1) always gets the exact answer.
			if (!always_exact && kbar <= SMH_LSD_REJECT_K) continue;
			if (!in_image(m, sh.cand_pt[c][0], sh.cand_pt[c][1])) continue;   // walked literally (exactly) in pass 1
			const uint32_t i = (u2 - c * LSD_GROUPS) * 64u + lane;
			const bool valid = i < SMH_LSD_RAYS;
			const RayDir d = g_ray_table[min(i, (uint32_t)SMH_LSD_RAYS - 1u)];
			const float dx = __uint_as_float(d.dx), dy = __uint_as_float(d.dy);
			const float xs = sh.cand_pt[c][0], ys = sh.cand_pt[c][1];
			float xe = xs, ye = ys;
			uint32_t steps = 0;
			int status = RAY_CONTINUE;
			RayState s = {0.0f, 0.0f, 0u, 0u, 0.0f, 0.0f, 0u, 0u, 0u};
			if (valid) {
#pragma unroll 1
				for (uint32_t bi = 0; bi < LSD_A_BATCHES && status == RAY_CONTINUE; ++bi) status = ray_batch<MODE>(m, xs, ys, dx, dy, T, s, steps);
			}
			unsigned long long key = 0;
			if (valid && status == RAY_ABORTED && s.gk0 + s.gj + 2u >= kbar) {
				ray_endpoint(m, status, s, xs, ys, dx, dy, xe, ye);
				const float ddx = xs - xe, ddy = ys - ye;
				key = ((unsigned long long)__float_as_uint(ddx * ddx + ddy * ddy) << 32) | i;
			}
			const unsigned long long wkey = wave_max64(key);
			// a unit may already hold an exact (left-the-image) winner from pass 1: keep the larger
			if (key != 0ull && key == wkey && key > sh.unit_key[u2]) { sh.unit_key[u2] = key; sh.unit_end[u2][0] = xe; sh.unit_end[u2][1] = ye; }
			if (lane == 0) atomicMax(&sh.cand_best[c], wkey);
		}
	}
	__syncthreads();
#ifdef SMH_LSD_PROFILE
	if (prof_t) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof_t[2] += _n - prof_last; prof_last = _n; }
#endif
	// ---- the winning ray of each candidate publishes its end point (keys are unique per ray) ----
#pragma unroll
	for (int r = 0; r < (int)LSD_QPT; ++r)
		if (bkey[r] != 0ull && bkey[r] == sh.cand_best[bc[r]]) { sh.cand_end[bc[r]][0] = bxe[r]; sh.cand_end[bc[r]][1] = bye[r]; }
	if (tid < nunits) {
		const uint32_t c = tid / LSD_GROUPS;
		if (sh.unit_key[tid] != 0ull && sh.unit_key[tid] == sh.cand_best[c]) { sh.cand_end[c][0] = sh.unit_end[tid][0]; sh.cand_end[c][1] = sh.unit_end[tid][1]; }
	}
	__syncthreads();
}

// ------------------------------------------------------------------------------------------------
// cooperation primitives (LsdCoop / request ring / result cache; smh_kernels.h explains the scheme)
// ------------------------------------------------------------------------------------------------
#define LSD_MAX_HELPERS 4u            // helpers per frame
#define LSD_HELP_MIN_REMAINING 24u    // a frame with fewer surviving candidates is not worth loading its mask
#define LSD_CACHE_PROBES 6
#define LSD_HELPER_IDLE_POLLS 12u     // polls (~4 us each) without a request before a helper leaves its frame / the kernel

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_acquire(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_release(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t pix_pack(uint32_t x, uint32_t y) { return x | (y << 12); }     // ROI sides are below 4096
__device__ __forceinline__ uint32_t cache_slot0(uint32_t pix) { return (pix * 2654435761u) >> 23; }   // 9 bits = SMH_LSD_CACHE_SLOTS

// -> true and the payload when the result for `pix` of this launch is present and complete
__device__ bool cache_lookup(const LsdCacheEntry *tab, uint32_t epoch, uint32_t pix, unsigned long long &best, float &ex, float &ey, uint32_t &steps) {
	static_assert(SMH_LSD_CACHE_SLOTS == 512u, "cache_slot0 yields 9 bits");
	const unsigned long long want = ((unsigned long long)epoch << 32) | pix;
	const uint32_t h = cache_slot0(pix);
	for (int p = 0; p < LSD_CACHE_PROBES; ++p) {
		const LsdCacheEntry *e = tab + ((h + (uint32_t)p) & (SMH_LSD_CACHE_SLOTS - 1u));
		const unsigned long long tag = __hip_atomic_load(&e->tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
		if (tag == want) { best = e->best; ex = e->ex; ey = e->ey; steps = e->steps; return true; }
		if ((uint32_t)(tag >> 32) != epoch) return false;           // a slot of another launch is an empty slot: end of the probe sequence
	}
	return false;
}

__device__ void cache_insert(LsdCacheEntry *tab, uint32_t epoch, uint32_t pix, unsigned long long best, float ex, float ey, uint32_t steps) {
	const unsigned long long ready = ((unsigned long long)epoch << 32) | pix, busy = ready | 0x80000000ull;
	const uint32_t h = cache_slot0(pix);
	for (int p = 0; p < LSD_CACHE_PROBES; ++p) {
		LsdCacheEntry *e = tab + ((h + (uint32_t)p) & (SMH_LSD_CACHE_SLOTS - 1u));
		unsigned long long tag = __hip_atomic_load(&e->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		while ((uint32_t)(tag >> 32) != epoch) {                    // free: claim it, fill it, publish it
			if (__hip_atomic_compare_exchange_strong(&e->tag, &tag, busy, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				e->best = best; e->ex = ex; e->ey = ey; e->steps = steps;
				__hip_atomic_store(&e->tag, ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
		}
		if ((tag & ~0x80000000ull) == ready) return;                // another helper has (or is writing) this pixel
	}
}

// One thread: claims up to LSD_C pending requests of a frame.  Entries carry the lap number of the ring so that one the
// owner has meanwhile overwritten is recognised and dropped (a request is only ever a hint).
__device__ uint32_t ring_claim(LsdCoop *co, const uint32_t *ring, uint32_t *pix_out) {
	uint32_t head = ld_relaxed(&co->req_head);
	const uint32_t tail = ld_acquire(&co->req_tail);
	uint32_t n = 0;
	while ((int)(tail - head) > 0) {
		n = min(tail - head, LSD_C);
		if (__hip_atomic_compare_exchange_strong(&co->req_head, &head, head + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
		n = 0;
	}
	uint32_t k = 0;
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t v = ld_relaxed(&ring[(head + i) % SMH_LSD_REQ_CAP]);
		if ((v >> 24) == (((head + i) / SMH_LSD_REQ_CAP) & 0xFFu)) pix_out[k++] = v & 0xFFFFFFu;
	}
	return k;
}

// ------------------------------------------------------------------------------------------------
// A frame's mask as a workgroup sees it: the sampling window `m` (three residency modes) and the compaction domain
// (wrows x wwords words; word (r, c) holds pixels x = xorg + 32 c + [0,32) of image row wy0 + r).
// ------------------------------------------------------------------------------------------------
struct FrameView {
	Win m;
	const uint32_t *gbits;
	uint32_t wy0, wrows, wwords;
	int xorg;
	uint32_t c_pitch, c_cap_rows;      // GLOBAL: geometry of the LDS row cache
};

template <int MODE, bool WAVE = false>   // WAVE (LSD_MODE_GLOBAL only): called by one wave on its own (k_lsd_service): nothing in LDS to wait for
__device__ __forceinline__ void frame_setup(const Geom &g, const Buffers &b, uint32_t f, const FrameAux &aux, uint32_t *smem, FrameView &v) {
	static_assert(!WAVE || MODE == LSD_MODE_GLOBAL, "only the global-memory view is built without the workgroup");
	const uint32_t tid = threadIdx.x;
	const uint32_t *gbits = b.bits + (size_t)f * g.bits_stride_w;
	Win &m = v.m;
	v.gbits = gbits;
	m.w = g.rw; m.h = g.rh; m.wf = (float)g.rw; m.hf = (float)g.rh;
	m.rows0 = nullptr; m.ylo_f = 0.0f; m.yhi_f = 0.0f;
	m.c_p = (const LdsWord *)smem; m.c_y0 = 0u; m.c_rows = 0u; m.c_pitch4 = 0u;
	m.t_idx = nullptr; m.t_tiles = nullptr; m.t_pitch = 0u; m.tiled = false; m.w2 = nullptr;
	m.tc_rows = nullptr; m.tc_groups = 0u; m.compact = false;
	v.c_pitch = g.bits_pitch_w | 1u; v.c_cap_rows = min(g.rh, LSD_WIN_WORDS_CAP / v.c_pitch);
	if (MODE == LSD_MODE_ROWS) {
		const uint32_t wy0 = aux.y_min, wrows = aux.y_max - aux.y_min + 1u;
		v.wy0 = wy0; v.wrows = wrows; v.wwords = LSD_ROWS_PITCH(g.bits_pitch_w); v.xorg = 0;
		const uint32_t pitch = v.wwords, gp = g.bits_pitch_w;          // odd LDS pitch: consecutive rows fall on different banks
		// [2 pad words][row y_min-1 = zeros][rows y_min..y_max, shifted right by xoff bits][row y_max+1 = zeros][2 pad words]
		const uint32_t total = (wrows + 2u) * pitch + 4u;
		for (uint32_t idx = tid; idx < total; idx += blockDim.x) {
			uint32_t val = 0;
			if (idx >= 2u + pitch && idx < 2u + (wrows + 1u) * pitch) {
				const uint32_t k = idx - 2u - pitch, r = k / pitch, c = k - r * pitch;
				if (c < gp) {
					const uint32_t *src = gbits + (size_t)(wy0 + r) * gp + c;
					const uint32_t lo = src[0], hi = (c + 1u < gp) ? src[1] : 0u;
					val = __builtin_amdgcn_alignbit(hi, lo, g.m_xoff);       // bit x of the row = pixel x
				}
			}
			smem[idx] = val;
		}
		m.p = smem + 2; m.pitch4 = pitch * 4u;
		m.y_lo = (int)wy0 - 1; m.rows_hi = wrows + 1u;
		m.xbias = 0; m.cols_hi = pitch - 1u;
		m.rows0 = (const char *)(smem + 2) - (ptrdiff_t)((int)wy0 - 1) * (ptrdiff_t)(pitch * 4u);
		m.ylo_f = (float)((int)wy0 - 1); m.yhi_f = (float)(wy0 + wrows);
	} else if (MODE == LSD_MODE_XWIN) {
		const uint32_t ww0 = aux.w_min, wy0 = aux.y_min, wrows = aux.y_max - aux.y_min + 1u, wwords = aux.w_max - aux.w_min + 1u;
		v.wy0 = wy0; v.wrows = wrows; v.wwords = wwords;
		v.xorg = (int)(ww0 * 32u) - (int)g.m_xoff;
		const uint32_t pitch = (wwords + 2u) | 1u;                     // odd pitch: rows spread over the LDS banks
		m.p = smem; m.pitch4 = pitch * 4u;
		m.y_lo = (int)wy0 - 1; m.rows_hi = wrows + 1u;
		m.xbias = (int)g.m_xoff - 32 * ((int)ww0 - 1); m.cols_hi = wwords + 1u;
		const uint32_t total = (wrows + 2u) * pitch;
		for (uint32_t idx = tid; idx < total; idx += blockDim.x) {
			const uint32_t r = idx / pitch, c = idx - r * pitch;
			uint32_t val = 0;
			if (r >= 1u && r <= wrows && c >= 1u && c <= wwords) val = gbits[(size_t)(wy0 + r - 1u) * g.bits_pitch_w + ww0 + c - 1u];
			smem[idx] = val;
		}
	} else {
		v.wy0 = 0; v.wrows = g.rh; v.wwords = g.bits_pitch_w; v.xorg = -(int)g.m_xoff;
		m.p = gbits; m.pitch4 = g.bits_pitch_w * 4u;
		m.y_lo = 0; m.rows_hi = g.rh - 1u;
		m.xbias = (int)g.m_xoff; m.cols_hi = g.bits_pitch_w - 1u;
	}
	if (!WAVE) __syncthreads();
}

// GLOBAL: (re)load the row cache so that it covers rows [lo, hi] (a uniform decision).  Rows outside it are read from
// global memory, so the coverage only ever matters for speed.
template <int MODE>
__device__ __forceinline__ void cache_cover(const Geom &g, FrameView &v, uint32_t *smem, int lo, int hi) {
	if (MODE != LSD_MODE_GLOBAL) return;
	Win &m = v.m;
	lo = max(lo, 0); hi = min(hi, (int)g.rh - 1);
	if (m.c_rows && lo >= (int)m.c_y0 && hi < (int)(m.c_y0 + m.c_rows)) return;
	__syncthreads();                                   // nobody still reads the old rows
	const uint32_t y0 = (uint32_t)min(lo, (int)(g.rh - v.c_cap_rows));
	for (uint32_t idx = threadIdx.x; idx < v.c_cap_rows * v.c_pitch; idx += LSD_BS) {
		const uint32_t r = idx / v.c_pitch, c = idx - r * v.c_pitch;
		smem[idx] = c < g.bits_pitch_w ? v.gbits[(size_t)(y0 + r) * g.bits_pitch_w + c] : 0u;
	}
	m.c_y0 = y0; m.c_rows = v.c_cap_rows; m.c_pitch4 = v.c_pitch * 4u;
	__syncthreads();
}

// Reset the per-group accumulators for nc candidates (followed by a barrier at the caller).
__device__ __forceinline__ void group_reset(LsdShared &sh, uint32_t nc, bool cull) {
	const uint32_t tid = threadIdx.x;
	if (tid < nc * LSD_GROUPS) { sh.unit_key[tid] = 0ull; sh.unit_kmax[tid] = 0u; }
	if (tid < LSD_C) { sh.cand_best[tid] = 0ull; sh.cand_steps[tid] = 0u; sh.cand_kmax[tid] = 0u; sh.live[tid] = cull ? 0ull : ~0ull; }
	if (tid == 0) { sh.qtail = 0u; sh.unit_next = LSD_NW; sh.nlive = 0u; sh.cached = 0u;
#ifdef SMH_LSD_PROFILE
		sh.exp_far = 0u; sh.exp_units = 0u;
#endif
	}
}

// find_longest_line for the nc pixels in sh.cand_px (their accumulators reset, barrier passed): get_centre of each,
// sector culling, unit list and the ray engine for the candidates whose bit in sh.cached is clear.  On return (after a
// barrier) sh.cand_pt holds every start point and sh.cand_best / cand_end / cand_steps the results of the cast ones.
template <int MODE>
__device__ __forceinline__ void cast_group(const Win &m, LsdShared &sh, uint32_t *queue, const uint32_t *cull_tab, uint32_t nc, bool cull, float max_gap,
                                           unsigned long long *prof_t = nullptr) {
	const uint32_t tid = threadIdx.x;
#ifdef SMH_LSD_PROFILE
	unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif
	if (tid < nc * 32u) {
		// get_centre (lsd.rs:5-44) of the nc integer pixel positions, 32 lanes per candidate: lane (dir, k)
		// evaluates the k-th loop condition of direction dir (left, right, up, down); the walk length is the
		// number of leading true conditions (k = 5 always fails: |offset| < 5.0).  px - k is exact in f32.
		const uint32_t c = tid >> 5, l = tid & 31u, dir = l >> 3, k = l & 7u;
		const float cx = (float)sh.cand_px[c][0], cy = (float)sh.cand_px[c][1];
		const float fk = (float)k;
		bool cond = k < 5u;
		if (dir == 0u) cond = cond && (cx - fk > 0.0f) && white_at(m, cx - fk, cy);
		else if (dir == 1u) cond = cond && (cx + fk < (float)(m.w - 1u)) && white_at(m, cx + fk, cy);
		else if (dir == 2u) cond = cond && (cy - fk > 0.0f) && white_at(m, cx, cy - fk);
		else cond = cond && (cy + fk < (float)(m.h - 1u)) && white_at(m, cx, cy + fk);
		const uint32_t bits = (uint32_t)(__ballot(cond) >> (tid & 32u));
		if (l == 0u) {
			const float nl = (float)__builtin_ctz(~(bits & 0xFFu)), nr = (float)__builtin_ctz(~((bits >> 8) & 0xFFu));
			const float nu = (float)__builtin_ctz(~((bits >> 16) & 0xFFu)), nd = (float)__builtin_ctz(~((bits >> 24) & 0xFFu));
			sh.cand_pt[c][0] = ((cx - nl) + (cx + nr)) / 2.0f;
			sh.cand_pt[c][1] = ((cy - nu) + (cy + nd)) / 2.0f;
		}
	}
	__syncthreads();
#ifdef SMH_LSD_PROFILE
	if (prof_t) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof_t[10] += _n - prof_last; prof_last = _n; }
#endif
	const uint32_t cached = sh.cached;
	if (cull) {
		// ---- sector culling: which 64-ray units can see a white pixel at a distance in [50 - T, 50]? ----
		// One thread per (candidate, row, 32-pixel word) cell of the (2R+1)-row neighbourhood: the mask word,
		// funnel-shifted so that bit 0 is offset -R from floor(start point), is ANDed with the annulus mask of
		// the cell; every white annulus pixel left contributes its unit range (a byte code per pixel).
		for (uint32_t id = tid; id < nc * SMH_CULL_CELLS; id += LSD_BS) {
			const uint32_t c = id / SMH_CULL_CELLS, cell = id - c * SMH_CULL_CELLS;
			if ((cached >> c) & 1u) continue;
			const int fx = (int)floorf(sh.cand_pt[c][0]), fy = (int)floorf(sh.cand_pt[c][1]);
			const int yi = fy + (int)(cell >> 2) - SMH_SECTOR_R;
			if ((uint32_t)yi >= m.h) continue;
			const int b0 = fx - SMH_SECTOR_R + (int)((cell & 3u) << 5) + m.xbias;   // view bit coordinate of the cell's bit 0
			const uint32_t lo = win_word(m, b0 >> 5, yi), hi = win_word(m, (b0 >> 5) + 1, yi);
			uint32_t hit = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)b0 & 31u) & cull_tab[cell];
			if (hit) {
				const uint8_t *codes = (const uint8_t *)(cull_tab + SMH_CULL_CELLS) + (cell >> 2) * SMH_SECTOR_DIM + ((cell & 3u) << 5);
				unsigned long long acc = 0ull;
				while (hit) {                           // white annulus pixels of this word: their unit ranges
					const uint32_t code = codes[__builtin_ctz(hit)];
					hit &= hit - 1u;
					acc |= code == 0xFFu ? ~0ull : ((2ull << (code >> 6)) - 1ull) << (code & 63u);   // first <= 56, n <= 4
				}
				atomicOr(&sh.live[c], (acc | (acc >> LSD_GROUPS)) & ((1ull << LSD_GROUPS) - 1ull));   // units wrap at 57
			}
		}
		__syncthreads();
	}
	if (tid < nc * LSD_GROUPS) {                    // list the units that have to be cast
		const uint32_t c = tid / LSD_GROUPS;
		if (!((cached >> c) & 1u) && ((sh.live[c] >> (tid - c * LSD_GROUPS)) & 1ull)) sh.ulist[atomicAdd(&sh.nlive, 1u)] = (unsigned short)tid;
	}
	__syncthreads();
#ifdef SMH_LSD_PROFILE
	if (prof_t) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); prof_t[11] += _n - prof_last; prof_last = _n; }
#endif
	if (sh.nlive != 0u) ray_engine<MODE>(m, sh, queue, nc, max_gap, false PROF_ARG);   // else every cast candidate is rejected: cand_best = 0
}

// ------------------------------------------------------------------------------------------------
// lsd::find_lines::<32> (vision-common/src/lsd.rs:60-107) of frame f by its owner workgroup; mode 1: one
// Vision::find_longest_line round.  `co` != nullptr: helpers may be attached (posting + result cache).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ void lsd_frame(const Geom &g, const Buffers &b, uint32_t f, float max_gap, int mode, float spx, float spy, const FrameAux &aux,
                          uint32_t *smem, LsdShared &sh, uint32_t *cull_tab, LsdCoop *co) {
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	smhv_frame_result *res = &b.results[f];
	PROF_DECL
	FrameView v;
	frame_setup<MODE>(g, b, f, aux, smem, v);
	Win &m = v.m;
	uint32_t *list = smem + LSD_WIN_WORDS_CAP, *queue = list + LSD_LIST_CAP;
	const uint32_t wy0 = v.wy0, wrows = v.wrows, wwords = v.wwords;
	const int xorg = v.xorg;
	const uint32_t WT = wrows * wwords;   // word index wi -> row wi / wwords, column wi % wwords
	auto dom_word = [&](uint32_t wi) -> uint32_t {
		const uint32_t r = wi / wwords, c = wi - r * wwords;
		if (MODE == LSD_MODE_ROWS) return m.p[(r + 1u) * wwords + c];
		if (MODE == LSD_MODE_XWIN) return m.p[(r + 1u) * (m.pitch4 >> 2) + c + 1u];
		return m.p[wi];
	};

	if (mode == 1) {   // Vision::find_longest_line on an arbitrary point
		if (tid < LSD_GROUPS) { sh.unit_key[tid] = 0ull; sh.unit_kmax[tid] = 0u; sh.ulist[tid] = (unsigned short)tid; }
		cache_cover<MODE>(g, v, smem, (int)spy - (int)(v.c_cap_rows / 2u), (int)spy - (int)(v.c_cap_rows / 2u) + (int)v.c_cap_rows - 1);
		if (tid == 0) { sh.live[0] = ~0ull; sh.cand_best[0] = 0ull; sh.cand_steps[0] = 0u; sh.cand_kmax[0] = 0u; sh.qtail = 0u; sh.unit_next = LSD_NW; sh.nlive = LSD_GROUPS; sh.cand_pt[0][0] = spx; sh.cand_pt[0][1] = spy; }
		__syncthreads();
		ray_engine<MODE>(m, sh, queue, 1u, max_gap, true PROF_ARG);
		if (tid == 0) {
			res->lines[0].x0 = spx; res->lines[0].y0 = spy; res->lines[0].x1 = sh.cand_end[0][0]; res->lines[0].y1 = sh.cand_end[0][1];
			res->length_px[0] = (double)__uint_as_float((uint32_t)(sh.cand_best[0] >> 32));
			res->n_lines = 1; res->rounds = 1; res->ray_steps = sh.cand_steps[0]; res->status = SMHV_FRAME_OK;
		}
		return;
	}

	// sector culling needs the table for this max_gap (absent in exact-statistics mode) and a gap threshold below 50
	const bool cull = b.cull_tab != nullptr && max_gap > 0.0f && max_gap <= 49.0f;
	const uint32_t *ring = co ? b.co.req + (size_t)f * SMH_LSD_REQ_CAP : nullptr;
	const LsdCacheEntry *ctab = co ? b.co.cache + (size_t)f * SMH_LSD_CACHE_SLOTS : nullptr;
	if (tid == 0) { sh.req_tail = 0u; sh.posted_hi = 0u; sh.posted_new = 0u; sh.helpers_seen = 0u; sh.head_seen = 0u; if (co) st_relaxed(&co->mode1, (uint32_t)MODE + 1u); }
	if (tid < LSD_LOOKAHEAD) sh.post[tid] = 0u;
	uint32_t rounds = 0, n_lines = 0;
	unsigned long long steps = 0ull;
	uint32_t seg_start = 0, cmax = 1u;
	bool done = false;
	PROF_MARK(8);   // window load
	while (!done) {
		// ---- ordered compaction of the non-zero mask words in [seg_start, WT) into `list` ----
		const uint32_t range = WT - seg_start;
		const uint32_t per = (range + LSD_BS - 1u) / LSD_BS;
		const uint32_t my0 = min(seg_start + tid * per, WT), my1 = min(my0 + per, WT);
		uint32_t cnt = 0;
		for (uint32_t wi = my0; wi < my1; ++wi) cnt += dom_word(wi) != 0u ? 1u : 0u;
		uint32_t incl = cnt;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
		if (lane == 63) sh.scan[wave] = incl;
		if (tid == 0) sh.segnext = WT;
		__syncthreads();
		uint32_t wprefix = 0, total = 0;
#pragma unroll
		for (int k = 0; k < LSD_NW; ++k) { const uint32_t s = sh.scan[k]; if ((uint32_t)k < wave) wprefix += s; total += s; }
		uint32_t o = wprefix + incl - cnt;
		for (uint32_t wi = my0; wi < my1; ++wi)
			if (dom_word(wi) != 0u) {
				if (o < LSD_LIST_CAP) list[o] = wi;
				else if (o == LSD_LIST_CAP) sh.segnext = wi;
				++o;
			}
		__syncthreads();
		const uint32_t Tn = min(total, LSD_LIST_CAP);
		const uint32_t segnext = sh.segnext;
		PROF_MARK(9);   // compaction

		for (uint32_t cbase = 0; cbase < Tn && !done; cbase += LSD_BS) {
			const uint32_t e = cbase + tid;
			uint32_t surv = 0, wkey0 = 0;
			float py = 0.0f, px0 = 0.0f;
			if (e < Tn) {
				const uint32_t wi = list[e];
				surv = dom_word(wi);
				const uint32_t r = wi / wwords, c = wi - r * wwords;
				py = (float)(wy0 + r);
				px0 = (float)(xorg + (int)(c * 32u));
				wkey0 = (wi << 5) + 1u;                                // raster key of the word's bit 0 (> 0, grows along the scan)
				for (uint32_t l = 0; l < n_lines && surv; ++l) surv = prox_filter_word(surv, px0, py, shared_prox_line(sh, l));
			}
			while (true) {
				// the coop line of this frame is fetched under the scan: are helpers attached, how far have they claimed?
				if (co && tid == 0) {
					sh.helpers_seen = ld_relaxed(&co->helpers); sh.head_seen = ld_relaxed(&co->req_head);
					sh.posted_hi = max(sh.posted_hi, sh.posted_new);       // what the previous group posted
				}
				// ---- the next (up to) LSD_C surviving white pixels in raster order ----
				const uint32_t pc = __popc(surv);
				uint32_t pin = pc;
				for (int o2 = 1; o2 < 64; o2 <<= 1) { const uint32_t t = __shfl_up(pin, o2); if (lane >= (uint32_t)o2) pin += t; }
				if (lane == 63) sh.scan[wave] = pin;
				__syncthreads();
				uint32_t wp = 0, tot = 0;
#pragma unroll
				for (int k = 0; k < LSD_NW; ++k) { const uint32_t s = sh.scan[k]; if ((uint32_t)k < wave) wp += s; tot += s; }
				if (tot == 0u) break;
				const bool helped = co != nullptr && sh.helpers_seen != 0u;
				const uint32_t posted_hi = sh.posted_hi;
				uint32_t nc = min(tot, cmax);
				uint32_t rank = wp + pin - pc;
				while (surv && rank < cmax) {
					const uint32_t bit = __builtin_ctz(surv);
					surv &= surv - 1u;
					sh.cand_key[rank] = (tid << 5) | bit;
					sh.cand_wkey[rank] = wkey0 + bit;
					sh.cand_px[rank][0] = (uint32_t)((int)px0 + (int)bit); sh.cand_px[rank][1] = (uint32_t)py;
					++rank;
				}
				if (helped) {
					// the LSD_LOOKAHEAD survivors behind this group, as far as they have not been posted yet, go to the helpers
					uint32_t s2 = surv, r2 = rank;
					while (s2 && r2 < cmax + LSD_LOOKAHEAD) {
						const uint32_t bit = __builtin_ctz(s2);
						s2 &= s2 - 1u;
						if (r2 >= cmax && wkey0 + bit > posted_hi) {
							sh.post[r2 - cmax] = 0x80000000u | pix_pack((uint32_t)((int)px0 + (int)bit), (uint32_t)py);
							atomicMax(&sh.posted_new, wkey0 + bit);
						}
						++r2;
					}
				}
				if (co && tid == 0) st_relaxed(&co->remaining, tot);
				group_reset(sh, nc, cull);
				__syncthreads();
				if (MODE == LSD_MODE_GLOBAL) {
					// Keep the mask rows every candidate of this group can reach in pass 1 (64 samples), the culling scan and
					// get_centre inside the LDS row cache; candidates (in raster order) that would not fit are handed back.
					const int lo = max((int)sh.cand_px[0][1] - LSD_CACHE_MARGIN, 0);
					uint32_t nce = 1;
					while (nce < nc && (int)sh.cand_px[nce][1] + LSD_CACHE_MARGIN - lo < (int)v.c_cap_rows) ++nce;
					for (uint32_t c = nce; c < nc; ++c) {
						const uint32_t key = sh.cand_key[c];
						if ((key >> 5) == tid) surv |= 1u << (key & 31u);
					}
					nc = nce;
					cache_cover<MODE>(g, v, smem, lo, (int)sh.cand_px[nc - 1u][1] + LSD_CACHE_MARGIN);
				}
				if (helped) {
					if (wave == 0) {
						// post the look-ahead (wave 0 owns the ring), in raster order, as far as the ring has room
						const uint32_t pv = lane < LSD_LOOKAHEAD ? sh.post[lane] : 0u;
						if (lane < LSD_LOOKAHEAD) sh.post[lane] = 0u;
						const unsigned long long valid = __ballot((pv >> 31) != 0u);
						const uint32_t tail = sh.req_tail, used = tail - min(sh.head_seen, tail);
						const uint32_t room = SMH_LSD_REQ_CAP - min(used, SMH_LSD_REQ_CAP);
						const uint32_t pos = (uint32_t)__popcll(valid & ((1ull << lane) - 1ull));
						const uint32_t np = min((uint32_t)__popcll(valid), room);
						if ((pv >> 31) && pos < np) {
							const uint32_t idx = tail + pos;
							st_relaxed(const_cast<uint32_t *>(ring) + idx % SMH_LSD_REQ_CAP, (((idx / SMH_LSD_REQ_CAP) & 0xFFu) << 24) | (pv & 0xFFFFFFu));
						}
						if (lane == 0 && np) { st_release(&co->req_tail, tail + np); sh.req_tail = tail + np; }
						if (lane == 0) __hip_atomic_fetch_add(&co->stat_groups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					} else if (tid >= 64u && tid < 64u + nc) {
						// has a helper already cast this candidate?  (keys are per launch; a hit is the result this workgroup
						// would compute itself: ray casting is a pure function of the mask and the pixel)
						const uint32_t c = tid - 64u;
						unsigned long long best; float ex, ey; uint32_t st;
						if (cache_lookup(ctab, b.co.epoch, pix_pack(sh.cand_px[c][0], sh.cand_px[c][1]), best, ex, ey, st)) {
							sh.cand_best[c] = best; sh.cand_end[c][0] = ex; sh.cand_end[c][1] = ey; sh.cand_steps[c] = st;
							atomicOr(&sh.cached, 1u << c);
							__hip_atomic_fetch_add(&co->stat_hits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
					}
					__syncthreads();
				}
				PROF_MARK(10);  // chunk filter + candidate selection
				cast_group<MODE>(m, sh, queue, cull_tab, nc, cull, max_gap PROF_ARG);
#ifdef SMH_LSD_PROFILE
				prof_last = __builtin_amdgcn_s_memtime();
#endif

				// ---- resolve in raster order (every thread computes the same thing) ----
				const uint32_t first_new = n_lines;
				for (uint32_t c = 0; c < nc; ++c) {
					const float cx = (float)sh.cand_px[c][0], cy = (float)sh.cand_px[c][1];
					bool skip = false;
					for (uint32_t l = first_new; l < n_lines; ++l)
						skip = skip || near_line(cx, cy, sh.lines[l][0], sh.lines[l][1], sh.lines[l][2], sh.lines[l][3]);
					if (skip) continue;                     // the sequential scan would not have visited it
					++rounds;
					steps += sh.cand_steps[c];
					const float len = __uint_as_float((uint32_t)(sh.cand_best[c] >> 32));
					if (len > SMH_LSD_ACCEPT_LEN_SQ) {
						float ex, ey;
						get_centre(m, sh.cand_end[c][0], sh.cand_end[c][1], ex, ey);
						// every thread stores the same values (no barrier needed for its own later reads)
						sh.lines[n_lines][0] = sh.cand_pt[c][0]; sh.lines[n_lines][1] = sh.cand_pt[c][1];
						sh.lines[n_lines][2] = ex; sh.lines[n_lines][3] = ey;
						const ProxLine pl = prox_line(sh.cand_pt[c][0], sh.cand_pt[c][1], ex, ey);
						sh.prox_a[n_lines] = pl.a; sh.prox_d[n_lines][0] = pl.dxl; sh.prox_d[n_lines][1] = pl.dyl;
						++n_lines;
						if (n_lines == SMH_LSD_MAX_LINES) { done = true; break; }
					}
				}
				if (done) break;
				if (helped && n_lines != first_new && tid == 0) {
					// a new line: what is pending in the ring was chosen without it -- drop it and post afresh from here on
					__hip_atomic_fetch_max(&co->req_head, sh.req_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					sh.posted_hi = sh.cand_wkey[nc - 1u]; sh.posted_new = 0u;
				}
				// Speculation width: a line accepted inside a group invalidates the later candidates of that
				// group near it (wasted ray casts), and acceptances cluster (first pixels of a marker), so the
				// width restarts at 1 after an acceptance and doubles after every acceptance-free group.
				cmax = (n_lines != first_new) ? LSD_C_RESET : min(cmax * 2u, LSD_C);
				PROF_MARK(5);   // resolve
				for (uint32_t l = first_new; l < n_lines && surv; ++l) surv = prox_filter_word(surv, px0, py, shared_prox_line(sh, l));
			}
			__syncthreads();
		}
		if (segnext >= WT) break;
		seg_start = segnext;
		__syncthreads();
	}
	__syncthreads();
	if (tid < n_lines) {
		res->lines[tid].x0 = sh.lines[tid][0]; res->lines[tid].y0 = sh.lines[tid][1];
		res->lines[tid].x1 = sh.lines[tid][2]; res->lines[tid].y1 = sh.lines[tid][3];
	}
	if (tid == 0) { res->n_lines = n_lines; res->rounds = rounds; res->ray_steps = steps; res->status = SMHV_FRAME_OK; }
	PROF_STORE(res);
}

// ------------------------------------------------------------------------------------------------
// Helper: a workgroup with nothing (left) to do of its own attaches to the frame with the most surviving candidates
// per helper, loads its mask and ray-casts the candidates the owner has posted; results go into the frame's cache.
// It leaves when every owner of the launch has finished, or when nothing has needed it for a while.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ void lsd_help(const Geom &g, const Buffers &b, uint32_t n_frames, float max_gap, uint32_t *smem, LsdShared &sh, uint32_t *cull_tab, bool have_cull_tab) {
	const uint32_t tid = threadIdx.x;
	const bool cull = b.cull_tab != nullptr && max_gap > 0.0f && max_gap <= 49.0f;
	if (cull && !have_cull_tab) for (uint32_t i = tid; i < SMH_CULL_TAB_WORDS; i += LSD_BS) cull_tab[i] = b.cull_tab[i];
	uint32_t *queue = smem + LSD_WIN_WORDS_CAP + LSD_LIST_CAP;
	uint32_t idle = 0;
	while (true) {
		// ---- pick: most remaining candidates per attached helper ----
		unsigned long long key = 0ull;
		for (uint32_t f = tid; f < n_frames; f += LSD_BS) {
			const LsdCoop *c = &b.co.coop[f];
			const uint32_t rem = ld_relaxed(&c->remaining), hl = ld_relaxed(&c->helpers), dn = ld_relaxed(&c->done);
			// (the mode is published before `remaining`: a frame of another residency kernel would not fit this one's window)
			if (!dn && hl < LSD_MAX_HELPERS && rem >= LSD_HELP_MIN_REMAINING && ld_relaxed(&c->mode1) == (uint32_t)MODE + 1u) {
				const unsigned long long k = ((unsigned long long)(rem / (hl + 1u)) << 32) | f;
				key = k > key ? k : key;
			}
		}
		if (tid == 0) sh.pick = 0ull;
		__syncthreads();
		key = wave_max64(key);
		if ((tid & 63u) == 0 && key) atomicMax(&sh.pick, key);
		__syncthreads();
		const unsigned long long pick = sh.pick;
		if (pick == 0ull) {
			if (tid == 0) sh.h_n = ld_relaxed(&b.co.ctl->finished[MODE]);      // one reader: the decision must be the same in every wave
			__syncthreads();
			const uint32_t fin = sh.h_n;
			__syncthreads();
			if (fin >= n_frames || ++idle > LSD_HELPER_IDLE_POLLS) return;
			__builtin_amdgcn_s_sleep(127);
			continue;
		}
		const uint32_t f = (uint32_t)pick;
		LsdCoop *co = &b.co.coop[f];
		if (tid == 0) {
			const uint32_t prev = __hip_atomic_fetch_add(&co->helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			sh.h_n = prev < LSD_MAX_HELPERS ? 1u : 0u;
			if (prev >= LSD_MAX_HELPERS) __hip_atomic_fetch_sub(&co->helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__syncthreads();
		if (sh.h_n == 0u) { __syncthreads(); continue; }
		__syncthreads();
		const FrameAux aux = b.aux[f];
		FrameView v;
		frame_setup<MODE>(g, b, f, aux, smem, v);
		const uint32_t *ring = b.co.req + (size_t)f * SMH_LSD_REQ_CAP;
		LsdCacheEntry *ctab = b.co.cache + (size_t)f * SMH_LSD_CACHE_SLOTS;
		uint32_t misses = 0;
		while (true) {
			if (tid == 0) {
				uint32_t px[LSD_C];
				uint32_t n = 0;
				if (ld_relaxed(&co->done)) n = 0xFFFFFFFFu;
				else {
					const uint32_t got = ring_claim(co, ring, px);
					for (uint32_t i = 0; i < got; ++i) {
						const uint32_t x = px[i] & 0xFFFu, y = px[i] >> 12;
						unsigned long long best; float ex, ey; uint32_t st;
						if (x < v.m.w && y < v.m.h && !cache_lookup(ctab, b.co.epoch, px[i], best, ex, ey, st)) { sh.cand_px[n][0] = x; sh.cand_px[n][1] = y; ++n; }
					}
				}
				sh.h_n = n;
			}
			__syncthreads();
			const uint32_t nc = sh.h_n;
			if (nc == 0xFFFFFFFFu) break;                      // the owner has finished
			if (nc == 0u) {
				__syncthreads();
				if (++misses > LSD_HELPER_IDLE_POLLS) break;
				__builtin_amdgcn_s_sleep(127);
				continue;
			}
			misses = 0; idle = 0;
			group_reset(sh, nc, cull);
			__syncthreads();
			if (MODE == LSD_MODE_GLOBAL) {
				int lo = (int)sh.cand_px[0][1], hi = lo;
				for (uint32_t c = 1; c < nc; ++c) { lo = min(lo, (int)sh.cand_px[c][1]); hi = max(hi, (int)sh.cand_px[c][1]); }
				cache_cover<MODE>(g, v, smem, lo - LSD_CACHE_MARGIN, min(hi + LSD_CACHE_MARGIN, lo - LSD_CACHE_MARGIN + (int)v.c_cap_rows - 1));
			}
			cast_group<MODE>(v.m, sh, queue, cull_tab, nc, cull, max_gap);
			if (tid == 0) __hip_atomic_fetch_add(&co->stat_casts, nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (tid < nc) cache_insert(ctab, b.co.epoch, pix_pack(sh.cand_px[tid][0], sh.cand_px[tid][1]), sh.cand_best[tid], sh.cand_end[tid][0], sh.cand_end[tid][1], sh.cand_steps[tid]);
			__syncthreads();
		}
		if (tid == 0) __hip_atomic_fetch_sub(&co->helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
	}
}

// One kernel per mask residency mode, each over all frames: a workgroup whose frame needs another mode exits at
// once (launch_lsd runs them side by side, or only the ROWS one when the frame size guarantees it).  Split this way
// the common ROWS kernel carries no call to the rarely used variants: 113 VGPRs and no scratch, where a kernel
// holding all three needed 128 and spilled at its call sites.
__device__ __forceinline__ int lsd_mode_for(const Geom &g, const FrameAux &aux, uint32_t cap = LSD_WIN_WORDS_CAP) {
	int lmode = LSD_MODE_GLOBAL;                           // also the empty-mask single-round case
	if (aux.n_mask_px != 0) {
		const uint32_t wrows = aux.y_max - aux.y_min + 1u, wwords = aux.w_max - aux.w_min + 1u;
		if ((wrows + 2u) * LSD_ROWS_PITCH(g.bits_pitch_w) + 4u <= cap) lmode = LSD_MODE_ROWS;
		else if ((wrows + 2u) * ((wwords + 2u) | 1u) <= cap) lmode = LSD_MODE_XWIN;
	}
	return lmode;
}

// grid = n_frames owner workgroups (+ extra workgroups that only help, for batches smaller than the chip)
// COOP: the instantiation with the helper machinery (SMHV_STAGE_LSD_HELPERS); the plain one carries none of it (its
// second copy of the ray engine costs the register allocator of the hot loops dearly: 40 -> 676 bytes of scratch in the
// global-memory mode)
template <int MODE, bool COOP>
__global__ void __launch_bounds__(LSD_BS) __attribute__((amdgpu_waves_per_eu(5, 5))) k_lsd(Geom g, Buffers b, float max_gap, int mode, float spx, float spy, uint32_t n_frames) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	__shared__ LsdShared sh;
	__shared__ uint32_t cull_tab[SMH_CULL_TAB_WORDS];
	const uint32_t f = blockIdx.x;
	const bool coop = COOP && mode == 0 && b.co.ctl != nullptr;
	bool have_tab = false;
	if (f < n_frames) {
		if (coop && threadIdx.x == 0) __hip_atomic_fetch_add(&b.co.ctl->started[MODE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const FrameAux aux = b.aux[f];
		bool mine = true;
		if (mode == 0) {
			// Record ownership is exclusive: the three mode kernels may run concurrently on forked streams, so a frame's record is
			// touched only by the kernel that owns the frame -- lsd_frame always stores n_lines / rounds / ray_steps at its end --
			// and frames nobody searches (map closed, empty mask) are zeroed by the ROWS kernel alone.
			if (!aux.open || aux.n_mask_px == 0) {
				if (MODE == LSD_MODE_ROWS && threadIdx.x == 0) { b.results[f].n_lines = 0; b.results[f].rounds = 0; b.results[f].ray_steps = 0; b.results[f].status = SMHV_FRAME_OK; }
				mine = false;
			}
		}
		// the kernel that owns a frame's record also completes it (scale ratio + derived outputs, smh_record.inc) when the run asks
		// for that: frames nobody searches belong to the ROWS kernel, the others to the kernel of their mode
		bool record = mode == 0 && (b.rec_stages & SMH_REC_ON) != 0u && !mine && MODE == LSD_MODE_ROWS;
		if (mine && lsd_mode_for(g, aux) != MODE) mine = false;
		if (mine) {
			const bool cull = mode == 0 && b.cull_tab != nullptr && max_gap > 0.0f && max_gap <= 49.0f;
			if (cull) { for (uint32_t i = threadIdx.x; i < SMH_CULL_TAB_WORDS; i += LSD_BS) cull_tab[i] = b.cull_tab[i]; have_tab = true; }   // visible after the barrier in frame_setup
			lsd_frame<MODE>(g, b, f, max_gap, mode, spx, spy, aux, smem, sh, cull_tab, (COOP && coop) ? &b.co.coop[f] : nullptr);
			record = mode == 0 && (b.rec_stages & SMH_REC_ON) != 0u;
		}
		if (record) {
			struct Head { Geom g; Buffers b; };
			const Head *ka = (const Head *)__builtin_amdgcn_kernarg_segment_ptr();   // (the kernel's own parameter list)
			__syncthreads();
			frame_record_tail(&ka->g, &ka->b, f);
		}
		if (!COOP || !coop) return;
		__syncthreads();
		if (threadIdx.x == 0) {
			if (mine) { st_relaxed(&b.co.coop[f].remaining, 0u); st_release(&b.co.coop[f].done, 1u); }
			__hip_atomic_fetch_add(&b.co.ctl->finished[MODE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// Helping holds this CU: only when every owner workgroup of the launch has been dispatched (they are dispatched in
			// index order), and never for a workgroup that had no frame of its own (it would only sit on the CU another mode's
			// kernel is waiting for).
			sh.h_n = (mine && ld_relaxed(&b.co.ctl->started[MODE]) >= n_frames) ? 1u : 0u;
		}
		__syncthreads();
		if (sh.h_n == 0u) return;
		__syncthreads();
	} else if (!coop) {
		return;
	}
	if (COOP) lsd_help<MODE>(g, b, n_frames, max_gap, smem, sh, cull_tab, have_tab);
}

#include "smh_lsd_wave.inc"
#include "smh_lsd_seq.inc"
#include "smh_service.inc"

// ------------------------------------------------------------------------------------------------
// Sector culling for find_lines (k_lsd).  lsd.rs:94 keeps a candidate only if its longest ray has
// len^2 > 2500, i.e. the ray's fatal gap starts at step K >= 51 (an aborted ray ends K-1 unit steps from its
// start).  Such a ray has a WHITE sample at some step in [50 - T, 50]: T+1 consecutive non-white samples
// there would have aborted it earlier (T = ceil(max_gap)); the same holds for a ray that leaves the image
// after more than 50 steps.  So for a candidate only the angular sectors that can see a white pixel at a
// distance in that range have to be ray-cast at all: the longest ray, if it is acceptable, lies in one of
// them, and if none is acceptable the candidate is rejected whatever the other rays do.
// The table maps a pixel offset (relative to floor(start point)) to the 64-ray units (6.4 degree sectors)
// that could sample it at such a step, conservatively: the sample may sit anywhere in the pixel (0.71 px),
// the start point anywhere in its pixel (0.71 px), plus 0.05 px for accumulated f32 rounding.
// ------------------------------------------------------------------------------------------------
__global__ void k_build_sector_table(unsigned long long *tab, uint32_t T) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= SMH_SECTOR_ENTRIES) return;
	const int oy = (int)(i / SMH_SECTOR_DIM) - SMH_SECTOR_R, ox = (int)(i % SMH_SECTOR_DIM) - SMH_SECTOR_R;
	const double rho = 0.7072 + 0.7072 + 0.05;
	const double kmin = T >= 50u ? 0.0 : (double)(50u - T), kmax = 50.0;
	const double r0 = sqrt((double)(ox * ox + oy * oy));
	unsigned long long m = 0ull;
	if (r0 >= kmin - rho && r0 <= kmax + rho) {
		if (r0 <= rho) {
			m = ~0ull;                                    // the start pixel's neighbourhood: every direction
		} else {
			const double PI = 3.14159265358979323846;
			double phi = atan2((double)oy, (double)ox) * 180.0 / PI;   // ray i points at (cos, sin)(i/10 degrees), y down
			if (phi < 0.0) phi += 360.0;
			const double delta = asin(rho / r0 > 1.0 ? 1.0 : rho / r0) * 180.0 / PI + 0.2;   // + two ray steps
			for (int u = 0; u < LSD_GROUPS_HOST; ++u) {
				const double a0 = 6.4 * u, a1 = 6.4 * u + 6.3;   // rays 64u .. 64u+63
				// intersects [phi - delta, phi + delta] modulo 360?
				bool hit = false;
				for (int wrap = -1; wrap <= 1 && !hit; ++wrap) {
					const double lo = phi - delta + 360.0 * wrap, hi = phi + delta + 360.0 * wrap;
					hit = !(hi < a0 || lo > a1);
				}
				if (hit) m |= 1ull << u;
			}
		}
	}
	tab[i] = m;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
size_t lsd_lds_bytes() { return LSD_DYN_LDS_BYTES; }

static std::atomic<bool> &lsd_classic_flag() {
	// default: k_lsd_tile; smhv_debug_lsd_classic(1) selects the workgroup-synchronous k_lsd
	static std::atomic<bool> flag{false};
	return flag;
}
void lsd_set_classic(bool on) { lsd_classic_flag().store(on, std::memory_order_relaxed); }
static std::atomic<uint32_t> g_tile_cap_override{0};
void lsd_set_tile_cap(uint32_t cap) { g_tile_cap_override.store(cap, std::memory_order_relaxed); }
static std::atomic<uint32_t> g_bs_override{0};
void lsd_set_threads(uint32_t threads) { g_bs_override.store(threads, std::memory_order_relaxed); }
static std::atomic<uint32_t> g_spin_limit{W_SPIN_LIMIT_DEFAULT};
void lsd_set_spin_limit(uint32_t polls) { g_spin_limit.store(polls ? polls : W_SPIN_LIMIT_DEFAULT, std::memory_order_relaxed); }

bool lsd_rows_only(const Geom &g) { return (g.rh + 2u) * LSD_ROWS_PITCH(g.bits_pitch_w) + 4u <= LSD_WIN_WORDS_CAP; }

static uint32_t lsd_tile_static_lds() {
	static const uint32_t v = [] { hipFuncAttributes a; return hipFuncGetAttributes(&a, (const void *)k_lsd_tile) == hipSuccess ? (uint32_t)a.sharedSizeBytes : 17408u; }();
	return v;
}
static uint32_t lsd_tile_cap_of(const Geom &g, uint32_t tile_limit) {
	const uint32_t cap_o = g_tile_cap_override.load(std::memory_order_relaxed);   // process-wide diagnostic, wins
	uint32_t cap = tile_cap_for(g);
	if (tile_limit) cap = std::min(cap, tile_limit);
	if (cap_o) cap = std::min(cap_o, tile_cap_for(g));
	return cap;
}
uint32_t lsd_tile_lds_bytes(const Geom &g, uint32_t tile_limit) {
	return lsd_tile_static_lds() + (tile_mask_words(g.rw, g.rh, lsd_tile_cap_of(g, tile_limit)) + 2u * tile_list_cap_for(g) + W_NWIN * W_WIN_STRIDE) * 4u;
}

hipError_t launch_lsd(const Geom &g, const Buffers &b, uint32_t n, float max_gap, int mode, float px, float py, hipStream_t s, const LsdFork *fk, uint32_t tile_bs, bool prefer_classic,
                      uint32_t tile_limit, bool *record_fused) {
	if (record_fused) *record_fused = false;
	const unsigned lds_full = LSD_DYN_LDS_BYTES;
	// more than 64 KB of dynamic LDS has to be allowed per function and per device
	static std::atomic<uint64_t> attr_devices{0};
	int dev = 0;
	hipError_t e0 = hipGetDevice(&dev);
	if (e0 != hipSuccess) return e0;
	if (dev >= 64 || !((attr_devices.load(std::memory_order_acquire) >> dev) & 1ull)) {
		hipError_t e = hipSuccess;
		const void *fns[] = {(const void *)k_lsd<LSD_MODE_ROWS, false>, (const void *)k_lsd<LSD_MODE_XWIN, false>, (const void *)k_lsd<LSD_MODE_GLOBAL, false>,
		                     (const void *)k_lsd<LSD_MODE_ROWS, true>, (const void *)k_lsd<LSD_MODE_XWIN, true>, (const void *)k_lsd<LSD_MODE_GLOBAL, true>};
		for (const void *fn : fns) if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_lsd_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSD_TILE_DYN_LDS_MAX);
		if (e != hipSuccess) return e;
		if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_release);
	}
	// every non-empty frame of this size is a ROWS frame (lsd_mode_for): the other two kernels would only exit
	const bool rows_only = mode == 0 && lsd_rows_only(g);
	const bool coop = mode == 0 && b.co.ctl != nullptr;
	// find_lines has two implementations: the task-based k_lsd_tile (default: any frame size, two frames per CU) and the
	// workgroup-synchronous k_lsd (with helper workgroups when the batch asks for them, for Vision::find_longest_line, and
	// on request: smhv_debug_lsd_classic).
	if (mode == 0 && !coop && !prefer_classic && !lsd_classic_flag().load(std::memory_order_relaxed)) {
		const uint32_t bs_o = g_bs_override.load(std::memory_order_relaxed);
		const uint32_t bs = std::max<uint32_t>(128u, bs_o ? bs_o : (tile_bs ? std::min<uint32_t>(tile_bs, LSD_TILE_BS) : 512u));
		const uint32_t cap = lsd_tile_cap_of(g, tile_limit);
		const unsigned t_lds = std::min<unsigned>((tile_mask_words(g.rw, g.rh, cap) + 2u * tile_list_cap_for(g) + W_NWIN * W_WIN_STRIDE) * 4u, LSD_TILE_DYN_LDS_MAX);
		if (record_fused) *record_fused = (b.rec_stages & SMH_REC_ON) != 0u;
		hipLaunchKernelGGL(k_lsd_tile, dim3(n), dim3(bs), t_lds, s, g, b, max_gap, cap, g_spin_limit.load(std::memory_order_relaxed), tile_list_cap_for(g), n);
		return hipGetLastError();
	}
	if (coop) {
		hipError_t e = hipMemsetAsync(b.co.ctl, 0, lsd_coop_ctl_bytes(n), s);
		if (e != hipSuccess) return e;
	}
	// Batches smaller than the chip get workgroups that only help (they are dispatched after the owners and leave as soon
	// as nothing needs them); with three concurrent mode kernels only finished owners help.
	uint32_t extra = 0;
	if (coop && rows_only) {
		static std::atomic<int> n_cus{0};
		int cus = n_cus.load(std::memory_order_relaxed);
		if (cus == 0) {
			if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
			n_cus.store(cus, std::memory_order_relaxed);
		}
		if (n < (uint32_t)cus) extra = std::min((uint32_t)cus - n, 4u * n);
	}
	hipStream_t s1 = s, s2 = s;
	if (!rows_only && fk) {
		hipError_t e = hipEventRecord(fk->fork, s);
		if (e == hipSuccess) e = hipStreamWaitEvent(fk->s1, fk->fork, 0);
		if (e == hipSuccess) e = hipStreamWaitEvent(fk->s2, fk->fork, 0);
		if (e != hipSuccess) return e;
		s1 = fk->s1; s2 = fk->s2;
	}
	if (record_fused && mode == 0) *record_fused = (b.rec_stages & SMH_REC_ON) != 0u;   // (every frame's record has exactly one owner among the kernels below)
	if (coop) hipLaunchKernelGGL((k_lsd<LSD_MODE_ROWS, true>), dim3(n + extra), dim3(LSD_BS), lds_full, s, g, b, max_gap, mode, px, py, n);
	else hipLaunchKernelGGL((k_lsd<LSD_MODE_ROWS, false>), dim3(n), dim3(LSD_BS), lds_full, s, g, b, max_gap, mode, px, py, n);
	if (!rows_only) {
		if (coop) hipLaunchKernelGGL((k_lsd<LSD_MODE_XWIN, true>), dim3(n), dim3(LSD_BS), lds_full, s1, g, b, max_gap, mode, px, py, n);
		else hipLaunchKernelGGL((k_lsd<LSD_MODE_XWIN, false>), dim3(n), dim3(LSD_BS), lds_full, s1, g, b, max_gap, mode, px, py, n);
		if (coop) hipLaunchKernelGGL((k_lsd<LSD_MODE_GLOBAL, true>), dim3(n), dim3(LSD_BS), lds_full, s2, g, b, max_gap, mode, px, py, n);
		else hipLaunchKernelGGL((k_lsd<LSD_MODE_GLOBAL, false>), dim3(n), dim3(LSD_BS), lds_full, s2, g, b, max_gap, mode, px, py, n);
		if (fk) {
			hipError_t e = hipEventRecord(fk->join1, s1);
			if (e == hipSuccess) e = hipEventRecord(fk->join2, s2);
			if (e == hipSuccess) e = hipStreamWaitEvent(s, fk->join1, 0);
			if (e == hipSuccess) e = hipStreamWaitEvent(s, fk->join2, 0);
			if (e != hipSuccess) return e;
		}
	}
	return hipGetLastError();
}

// ---- the frame-granular search service (smh_service.inc) ----
static uint32_t lsd_service_static_lds() {
	static const uint32_t v = [] { hipFuncAttributes a; return hipFuncGetAttributes(&a, (const void *)k_lsd_service) == hipSuccess ? (uint32_t)a.sharedSizeBytes : SMH_CULL_TAB_WORDS * 4u; }();
	return v;
}
// Waves per service workgroup (one workgroup per CU) for this frame size: as many as fit beside two workgroups of the
// streaming pass, at most SVC_MAX_WAVES -- one per SIMD, which leaves two 128-register wave slots per SIMD to the streaming
// pass (a service wave allocates 136-144 registers).  0: not even one wave fits (8K frames: the tile index alone is 106 KB) -> the pipeline keeps its batch-granular search.
uint32_t svc_waves_for(const Geom &g, uint32_t tile_limit, uint32_t *part_words, uint32_t *tile_cap, uint32_t *list_cap, uint32_t *lds_bytes, uint32_t *compact) {
	const uint32_t cap = lsd_tile_cap_of(g, tile_limit);
	const uint32_t lds_cu = 160u * 1024u, beside = 2u * ((map_brq_lds_bytes(g) + 1023u) & ~1023u) + 1024u;
	// The 16-bit tile table, a look-up per sample of a long ray, where SVC_MAX_WAVES waves fit with it (up to 1080p); above, the
	// compact index (a tenth of the table's LDS at 1440p: 2.2 KB against 12.8) and half the list segment -- the scan goes on
	// segment by segment -- where that buys another wave.
	uint32_t best_w = 0, best_part = 0, best_lc = 0, best_c = 0;
	for (uint32_t c = 0; c < 2u; ++c) {
		const uint32_t lc = c ? W_LIST_CAP_MAX / 2u : tile_list_cap_for(g);
		const uint32_t part = (SVC_WS_WORDS + (c ? tilec_mask_words(g.rw, g.rh, cap) : tile_mask_words(g.rw, g.rh, cap)) + 2u * lc + W_WIN_STRIDE + 3u) & ~3u;
		uint32_t w = SVC_MAX_WAVES;
		while (w > 0u && lsd_service_static_lds() + w * part * 4u + beside > lds_cu) --w;
		if (w > best_w) { best_w = w; best_part = part; best_lc = lc; best_c = c; }
	}
	if (part_words) *part_words = best_part;
	if (tile_cap) *tile_cap = cap;
	if (list_cap) *list_cap = best_lc;
	if (lds_bytes) *lds_bytes = best_w * best_part * 4u;
	if (compact) *compact = best_c;
	return best_w;
}

uint32_t svc_store_words_for(const Geom &g, uint32_t tile_cap, uint32_t compact) { return ((compact ? tilec_mask_words(g.rw, g.rh, tile_cap) : tile_mask_words(g.rw, g.rh, tile_cap)) + 7u) & ~3u; }

// -> *ok: a device-side 64-bit system-scope compare-and-swap on the mapped host block `h` (device address d_h) took effect
// and the host sees it (synchronous; pipeline creation)
hipError_t svc_probe_host_atomics(SvcHost *h, SvcHost *d_h, bool *ok) {
	*ok = false;
	uint32_t *d_ok = nullptr;
	hipError_t e = hipMalloc((void **)&d_ok, sizeof(uint32_t));
	if (e != hipSuccess) return e;
	const unsigned long long expect = 0x0123456789ABCDEFull, desired = 0xFEDCBA9876543210ull;
	unsigned long long *w = (unsigned long long *)&h->pad[0];
	__atomic_store_n(w, expect, __ATOMIC_RELEASE);
	__atomic_store_n(&h->pad[2], 0u, __ATOMIC_RELEASE);
	e = hipMemset(d_ok, 0, sizeof(uint32_t));
	uint32_t dev_ok = 0;
	if (e == hipSuccess) { hipLaunchKernelGGL(k_svc_probe, dim3(1), dim3(1), 0, nullptr, d_h, expect, desired, d_ok); e = hipGetLastError(); }
	if (e == hipSuccess) e = hipMemcpy(&dev_ok, d_ok, sizeof dev_ok, hipMemcpyDeviceToHost);   // (synchronises with the probe)
	(void)hipFree(d_ok);
	if (e != hipSuccess) return e;
	*ok = dev_ok == 1u && __atomic_load_n(w, __ATOMIC_ACQUIRE) == desired && __atomic_load_n(&h->pad[2], __ATOMIC_ACQUIRE) == 0x600Du;
	__atomic_store_n(w, 0ull, __ATOMIC_RELEASE);
	__atomic_store_n(&h->pad[2], 0u, __ATOMIC_RELEASE);
	return hipSuccess;
}

hipError_t launch_svc_publish(SvcCtl *ctl, unsigned long long *ring, SvcSlot *slots, uint32_t slot, const Buffers &b, uint32_t n, uint32_t seq, uint32_t ring_log2, hipStream_t s) {
	hipLaunchKernelGGL(k_svc_publish, dim3(1), dim3(256), 0, s, ctl, ring, slots, slot, b, n, seq, ring_log2);
	return hipGetLastError();
}

hipError_t launch_lsd_service(const Geom &g, const SvcParams &p, uint32_t workgroups, uint32_t waves, uint32_t lds_bytes, hipStream_t s) {
	static std::atomic<uint64_t> attr_devices{0};
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev >= 64 || !((attr_devices.load(std::memory_order_acquire) >> dev) & 1ull)) {
		e = hipFuncSetAttribute((const void *)k_lsd_service, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)lsd_service_static_lds());
		if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_lsd_service_c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)lsd_service_static_lds());
		if (e != hipSuccess) return e;
		if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_release);
	}
	if (p.compact) hipLaunchKernelGGL(k_lsd_service_c, dim3(workgroups), dim3(64u * waves), lds_bytes, s, g, p);
	else hipLaunchKernelGGL(k_lsd_service, dim3(workgroups), dim3(64u * waves), lds_bytes, s, g, p);
	return hipGetLastError();
}

size_t lsd_coop_ctl_bytes(uint32_t n) { return sizeof(LsdCtl) + sizeof(LsdCoop) * (size_t)n; }

hipError_t set_ray_table(const float *dx, const float *dy) {
	static RayDir host[SMH_LSD_RAYS];
	for (int i = 0; i < SMH_LSD_RAYS; ++i) { memcpy(&host[i].dx, &dx[i], 4); memcpy(&host[i].dy, &dy[i], 4); }
	return hipMemcpyToSymbol(HIP_SYMBOL(g_ray_table), host, sizeof host, 0, hipMemcpyHostToDevice);
}

// offsets of ray i at the start of batch j: exactly what `x_offset += dx` (vision-cpu/src/lib.rs:411-413) has accumulated after 32 j steps
__global__ void k_build_ray_offsets(float *off) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= SMH_LSD_RAYS) return;
	const float dx = __uint_as_float(g_ray_table[i].dx), dy = __uint_as_float(g_ray_table[i].dy);
	float xo = 0.0f, yo = 0.0f;
	float2 *o = (float2 *)off + (size_t)i * (SMH_RAY_OFF_BATCHES + 1u);
	for (uint32_t j = 0; j <= SMH_RAY_OFF_BATCHES; ++j) {
		o[j] = make_float2(xo, yo);
		for (int k = 0; k < 32; ++k) { xo += dx; yo += dy; }
	}
}

hipError_t launch_build_ray_offsets(float *d_off, hipStream_t s) {
	hipLaunchKernelGGL(k_build_ray_offsets, dim3((SMH_LSD_RAYS + 63) / 64), dim3(64), 0, s, d_off);
	return hipGetLastError();
}

hipError_t launch_build_sector_table(unsigned long long *d_tab, uint32_t T, hipStream_t s) {
	static_assert(LSD_GROUPS == LSD_GROUPS_HOST && LSD_GROUPS <= 64, "sector masks are 64-bit");
	hipLaunchKernelGGL(k_build_sector_table, dim3((SMH_SECTOR_ENTRIES + 255) / 256), dim3(256), 0, s, d_tab, T);
	return hipGetLastError();
}

}  // namespace smh
