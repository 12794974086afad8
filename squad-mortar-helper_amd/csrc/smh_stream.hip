// smh_stream.hip -- the HBM-bound streaming kernels of the vision hot path (gfx950, wave64).
//   k_button      red "Close Deployment" pixel count -> map_open           (lib.rs:116-133)
//   k_map_pass    one streaming pass over the map ROI: ui_map RGBA + marker colour predicate +
//                 L1 radius-1 dilation -> u8 mask + bit-packed mask + bbox   (lib.rs:137-171,253-280,357-375)
//   k_brq_pass    bottom-right quadrant: ocr_preprocess + find_scales_preprocess (lib.rs:173-251)
//
// Build with -ffp-contract=off and correctly rounded f32 division: several results are truncated to integers
// right at a threshold, so the reference's scalar f32 operation order (no FMA contraction, IEEE divide) is
// part of the contract.  Semantics follow the reference's CPU back-end (vision-cpu/src/lib.rs) bit for bit;
// structure does not follow its CUDA file at all (SURVEY.md Appendix A lists how that differs).
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "smh_device.h"

namespace smh {

// ------------------------------------------------------------------------------------------------
// Tile outputs of one wave of a band (smh_kernels.h, "the mask as the streaming passes leave it for the line search").
// rows_any: the lane's dilated column masks OR-ed together, bit 0 = the band's first row.  Writes the wave's occupancy bytes
// of the band's `ntr` tile rows and returns, in the lanes 8 j that store the bit-packed words, which of the band's (up to
// seven) tiles of their word column hold a set bit (bit t = tile row t of the band); 0 in every other lane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tile_occupancy(uint64_t rows_any, uint32_t lane, uint32_t ntr, uint8_t *occp, uint32_t pitch) {
	uint32_t alo = (uint32_t)rows_any, ahi = (uint32_t)(rows_any >> 32);
	alo |= SMH_DPP(alo, 0x101); ahi |= SMH_DPP(ahi, 0x101);         // row_shl:1 -- lane l reads lane l + 1 (0 beyond the 16-lane row)
	alo |= SMH_DPP(alo, 0x102); ahi |= SMH_DPP(ahi, 0x102);
	alo |= SMH_DPP(alo, 0x104); ahi |= SMH_DPP(ahi, 0x104);         // lane 8 j: the rows set anywhere in its eight quads = its word column
	uint32_t occ = 0;
#pragma unroll
	for (int t = 0; t < 4; ++t) occ |= (((alo >> (8 * t)) & 255u) ? 1u : 0u) << t;
#pragma unroll
	for (int t = 0; t < 3; ++t) occ |= (((ahi >> (8 * t)) & 255u) ? 1u : 0u) << (4 + t);
	if (lane & 7u) occ = 0u;
	uint32_t mine = 0;
#pragma unroll
	for (int t = 0; t < 7; ++t) {
		const uint64_t bal = __ballot((occ >> t) & 1u) & 0x0101010101010101ull;     // lanes 0, 8, .., 56
		const uint32_t byte = (uint32_t)((bal * 0x0102040810204080ull) >> 56);      // lane 8 j -> bit j
		if (lane == (uint32_t)t) mine = byte;
	}
	if (lane < ntr) occp[(size_t)lane * pitch] = (uint8_t)mine;
	return occ;
}

// ------------------------------------------------------------------------------------------------
// k_button: one workgroup per frame.  Also resets the per-frame scratch for the later passes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_button(Geom g, Buffers b, int force_open) {
	const uint32_t f = blockIdx.x;
	const uint8_t *fp = b.frames + (size_t)f * g.frame_bytes;
	__shared__ uint32_t s_cnt;
	if (threadIdx.x == 0) s_cnt = 0;
	__syncthreads();
	uint32_t cnt = 0;
	const uint32_t npx = g.bw * g.bh;
	// eight pixels per thread and round, loaded together: the kernel is a chain of load latencies (10 k pixels per frame), and
	// in a pipeline it sits at the head of every batch's chain
	constexpr uint32_t U = 8;
	for (uint32_t i0 = threadIdx.x; i0 < npx; i0 += blockDim.x * U) {
		uint32_t p[U];
#pragma unroll
		for (uint32_t k = 0; k < U; ++k) {
			const uint32_t i = min(i0 + k * blockDim.x, npx - 1u);
			const uint32_t y = i / g.bw, x = i - y * g.bw;
			p[k] = *(const uint32_t *)(fp + ((size_t)(g.by + y) * g.W + g.bx + x) * 4);   // B | G<<8 | R<<16 | A<<24
		}
#pragma unroll
		for (uint32_t k = 0; k < U; ++k) {
			const uint32_t bb = p[k] & 255u, gg = (p[k] >> 8) & 255u, rr = (p[k] >> 16) & 255u;
			const bool red = absdiff(SMH_BUTTON_R, rr) <= SMH_BUTTON_TOLERANCE && absdiff(SMH_BUTTON_G, gg) <= SMH_BUTTON_TOLERANCE &&
			                 absdiff(SMH_BUTTON_B, bb) <= SMH_BUTTON_TOLERANCE;
			cnt += (red && i0 + k * blockDim.x < npx) ? 1u : 0u;
		}
	}
	cnt = wave_sum32(cnt);
	if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, cnt);
	__syncthreads();
	if (threadIdx.x == 0) {
		const uint32_t red = s_cnt;
		// `red_pixels as f32 / (w * h) as f32 < 0.65` => Ok(None)   (vision-cpu/src/lib.rs:130-133)
		const float ratio = (float)red / (float)npx;
		FrameAux a;
		a.open = (force_open || !(ratio < SMH_BUTTON_RED_PIXEL_THRESHOLD)) ? 1u : 0u;
		a.red = red; a.n_mask_px = 0;
		a.y_min = 0xFFFFFFFFu; a.y_max = 0; a.w_min = 0xFFFFFFFFu; a.w_max = 0; a.tiles = 0;
		b.aux[f] = a;
	}
}

// ------------------------------------------------------------------------------------------------
// k_map_pass
//
// grid = (row bands, frames); block = one thread per quad (4 pixels, 16-byte BGRA load) across the
// whole ROI width, so the waves of a workgroup sit side by side on the same rows.
// Each thread marches down its quad column over the band's rows (+1 halo row above and below) and
// keeps the marker predicate as four 64-bit column masks (bit = row).  In that form
//   vertical dilation   = p | p<<1 | p>>1            (all rows of the band at once)
//   horizontal dilation = neighbouring column masks  (own registers, lane+-1 via DPP shuffles,
//                                                      wave edges via 16 B of LDS per wave)
// so the 3x3-cross dilation of the reference (imageproc dilate_mut(L1,1)) costs a dozen
// instructions per band instead of a second pass over an intermediate image.  ui_map is written
// straight from the loaded registers; the frame is read exactly once (+2 halo rows per band).
// ------------------------------------------------------------------------------------------------
// (rows per band: the column masks hold 1 + 62 + 1; band_rows_for takes 56 -- whole tile rows of the line search: the pass then
// writes the mask tile-major as well, smh_kernels.h -- where that does not cost the launch a band)
#define MAP_RB_MAX 62

template <bool GRAY, bool TILES>
__global__ void __launch_bounds__(1024) k_map_pass(Geom g, Buffers b, uint32_t flags, uint32_t RB) {
	const uint32_t f = blockIdx.y;
	if (!b.aux[f].open) return;
	const uint32_t q = threadIdx.x, lane = q & 63u, wave = q >> 6, nwave = blockDim.x >> 6;
	const int r0 = (int)(blockIdx.x * RB);
	const int r1 = min(r0 + (int)RB, (int)g.rh);
	const bool qact = q < g.m_quads;
	uint32_t vmask = 0;
#pragma unroll
	for (int c = 0; c < 4; ++c)
		if ((uint32_t)(4 * q + c - g.m_xoff) < g.rw) vmask |= 1u << c;
	if (!qact) vmask = 0;

	const uint8_t *fp = b.frames + (size_t)f * g.frame_bytes + ((size_t)g.ry * g.W + g.m_ax + 4 * q) * 4;
	uint8_t *uip = b.ui + (size_t)f * g.ui_stride + (size_t)q * 16;
	const size_t row_bytes = (size_t)g.W * 4;

	uint64_t P[4] = {0, 0, 0, 0};
	const int rs = max(r0 - 1, 0), re = min(r1, (int)g.rh - 1);
	const bool do_ui = (flags & MAP_UI) != 0, do_mask = (flags & MAP_MASK) != 0;

	// Software pipeline: the loads of the next four rows are issued before the current four are
	// processed, so they fly under the compute and the stores (vmcnt is in-order: a load issued after
	// the stores would also wait for them).  Inactive lanes re-read quad 0 (no divergent load).
	const uint8_t *lp = qact ? fp : fp - (size_t)q * 16;
	uint4 nx[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) nx[k] = *(const uint4 *)(lp + (size_t)min(rs + k, re) * row_bytes);
	// per-wave hit lists, sized by the launch (640 bytes per wave: 2.5 KB at 1080p, so that several of these workgroups fit
	// into the LDS a k_lsd workgroup of the other pipelined step leaves free on its CU)
	extern __shared__ __attribute__((aligned(16))) uint32_t s_hits[];
	for (int r = rs; r <= re; r += 4) {
		uint4 px[4];
		uint32_t prehits = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) px[k] = nx[k];
		if (r + 4 <= re) {
#pragma unroll
			for (int k = 0; k < 4; ++k) nx[k] = *(const uint4 *)(lp + (size_t)min(r + 4 + k, re) * row_bytes);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int row = r + k;
			if (row > re) break;
			const uint32_t pv[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
			if (do_ui && row >= r0 && row < r1 && qact) {
				uint4 o;
				uint32_t ov[4];
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const uint32_t p = pv[c], bb = p & 255u, gg = (p >> 8) & 255u, rr8 = (p >> 16) & 255u;
					if (GRAY) ov[c] = luma8(rr8, gg, bb) * 0x00010101u | 0xFF000000u;   // Bgra::to_luma -> (l,l,l,255)
					else ov[c] = rr8 | (gg << 8) | (bb << 16) | 0xFF000000u;           // (r,g,b,255)
				}
				o.x = ov[0]; o.y = ov[1]; o.z = ov[2]; o.w = ov[3];
				*(uint4 *)(uip + (size_t)row * g.ui_pitch) = o;
			}
			if (do_mask) {
				// branch-free integer pre-filter; hits of the four rows are collected (bit 4k+c)
				uint32_t pre = 0;
#pragma unroll
				for (int c = 0; c < 4; ++c) pre |= marker_prefilter(pv[c]) ? (1u << c) : 0u;
				prehits |= (pre & vmask) << (4 * k);
			}
		}
		// ---- exact f32 HSV test for the pre-filter hits of this wave, one hit per lane ----
		// A marker line crosses most rows of a band but only a few pixels of each, so testing hits where
		// they sit would run the (long, divergent) exact test several times per row for two or three
		// active lanes.  Instead the hit pixels of the whole wave and of all four rows are compacted into
		// a 64-entry LDS list, every lane tests one of them, and the verdicts are scattered back with
		// LDS atomic ORs.  (This path used to be 40 % of the kernel's time.)
		if (do_mask && __any(prehits != 0u)) {
			uint32_t *hpx = s_hits + wave * 160u;
			uint32_t *hres = hpx + 64;
			unsigned short *hid = (unsigned short *)(hres + 64);
			const uint32_t cnt = (uint32_t)__popc(prehits);
			uint32_t incl = cnt;
			for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
			const uint32_t total = __shfl(incl, 63);
			const uint32_t off = incl - cnt;
			hres[lane] = 0u;
			for (uint32_t base = 0; base < total; base += 64u) {
				uint32_t o = off - base;                           // may wrap: compared unsigned below
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const uint32_t pk[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
					for (int c = 0; c < 4; ++c)
						if ((prehits >> (4 * k + c)) & 1u) {
							if (o < 64u) { hpx[o] = pk[c]; hid[o] = (unsigned short)((lane << 4) | (uint32_t)(4 * k + c)); }
							++o;
						}
				}
				__builtin_amdgcn_wave_barrier();
				const uint32_t e = base + lane;
				if (e < total) {
					const uint32_t p = hpx[lane];
					if (marker_exact((p >> 16) & 255u, (p >> 8) & 255u, p & 255u)) {
						const uint32_t id = hid[lane];
						atomicOr(&hres[id >> 4], 1u << (id & 15u));
					}
				}
				__builtin_amdgcn_wave_barrier();
			}
			const uint32_t res = hres[lane];                       // bit 4k+c: pixel c of row r+k is a marker colour
			if (res) {
				const int sh = r - (r0 - 1);
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					uint32_t y = (res >> c) & 0x1111u;                 // rows k = 0..3 at bits 0,4,8,12
					y = (y | (y >> 3) | (y >> 6) | (y >> 9)) & 0xFu;   // -> bits 0..3
					P[c] |= (uint64_t)y << sh;
				}
			}
		}
	}
	if (!do_mask) return;

	// ---- dilation on the column masks ----
	__shared__ uint64_t s_edge_first[16], s_edge_last[16];
	if (lane == 0) s_edge_first[wave] = P[0];
	if (lane == 63) s_edge_last[wave] = P[3];
	__syncthreads();
	uint64_t left = __shfl_up(P[3], 1), right = __shfl_down(P[0], 1);
	if (lane == 0) left = wave > 0 ? s_edge_last[wave - 1] : 0ull;
	if (lane == 63) right = wave + 1 < nwave ? s_edge_first[wave + 1] : 0ull;
	const int nrows = r1 - r0;
	const uint64_t rowmask = ((nrows >= 63 ? ~0ull : ((1ull << nrows) - 1ull)) << 1);   // bits 1..nrows
	uint64_t D[4];
#define SMH_VERT(p) ((p) | ((p) << 1) | ((p) >> 1))
	D[0] = SMH_VERT(P[0]) | left | P[1];
	D[1] = SMH_VERT(P[1]) | P[0] | P[2];
	D[2] = SMH_VERT(P[2]) | P[1] | P[3];
	D[3] = SMH_VERT(P[3]) | P[2] | right;
#undef SMH_VERT
#pragma unroll
	for (int c = 0; c < 4; ++c) D[c] = ((vmask >> c) & 1u) ? (D[c] & rowmask) : 0ull;

	// ---- outputs: u8 mask rows and bit-packed rows ----
	const uint32_t quads_padded = (g.m_quads + 15u) & ~15u;
	const uint64_t any = D[0] | D[1] | D[2] | D[3];
	const uint64_t lanes_set = __ballot(any != 0ull);
	// the tile-major mask and its occupancy bytes (TILES: bands of whole tile rows, launch_map_pass; a launch without them runs the
	// instantiation that has no trace of this)
	uint32_t occ7 = 0;
	if constexpr (TILES) {
		const uint32_t ntr = ((uint32_t)nrows + 7u) >> 3, op = occ_pitch(g);
		uint8_t *occp = b.occ + (size_t)f * occ_stride(g) + (size_t)((uint32_t)r0 >> 3) * op + wave;
		if (lanes_set) occ7 = tile_occupancy(any >> 1, lane, ntr, occp, op);
		else if (lane < ntr) occp[(size_t)lane * op] = 0;
		if (blockIdx.x == 0 && q == 0) b.aux[f].tiles = 1u;           // (k_button cleared it: this frame's tile-major mask is being written)
	}
	if (q < quads_padded) {
		uint8_t *mp = b.mask + (size_t)f * g.mask_stride + (size_t)q * 4;
		uint32_t *bp = b.bits + (size_t)f * g.bits_stride_w + (q >> 3);
		uint32_t *tp = nullptr;
		if constexpr (TILES) tp = b.tiled + (size_t)f * tiled_stride_w(g) + ((size_t)((uint32_t)r0 >> 3) * g.bits_pitch_w + (q >> 3)) * 8u;
		for (int row = r0; row < r1; ++row) {
			const int bit = row - r0 + 1;
			const uint32_t nib = (uint32_t)((D[0] >> bit) & 1ull) | ((uint32_t)((D[1] >> bit) & 1ull) << 1) |
			                     ((uint32_t)((D[2] >> bit) & 1ull) << 2) | ((uint32_t)((D[3] >> bit) & 1ull) << 3);
			*(uint32_t *)(mp + (size_t)row * g.mask_pitch) = ((nib * 0x00204081u) & 0x01010101u) * 0xFFu;
			// gather 8 lanes' nibbles into one dword of the bit-packed row (lane l supplies bits 4(l%8)..)
			uint32_t v = nib;
			v |= __shfl_down(v, 1) << 4;
			v |= __shfl_down(v, 2) << 8;
			v |= __shfl_down(v, 4) << 16;
			if ((lane & 7u) == 0) bp[(size_t)row * g.bits_pitch_w] = v;
			if constexpr (TILES) {
				const uint32_t i = (uint32_t)(row - r0);
				if ((occ7 >> (i >> 3)) & 1u) tp[(size_t)(i >> 3) * g.bits_pitch_w * 8u + (i & 7u)] = v;    // (occ7 is 0 outside the lanes 8 j)
			}
		}
	}
	// ---- bounding box + population count of the set bits (drives the LDS window of k_lsd) ----
	if (lanes_set) {
		const uint64_t rows_set = wave_or64(any);
		const uint32_t cnt = wave_sum32(__popcll(D[0]) + __popcll(D[1]) + __popcll(D[2]) + __popcll(D[3]));
		if (lane == 0) {
			FrameAux *a = &b.aux[f];
			atomicMin(&a->y_min, (uint32_t)(r0 - 1 + __builtin_ctzll(rows_set)));
			atomicMax(&a->y_max, (uint32_t)(r0 - 1 + 63 - __builtin_clzll(rows_set)));
			atomicMin(&a->w_min, (wave * 64u + (uint32_t)__builtin_ctzll(lanes_set)) >> 3);
			atomicMax(&a->w_max, (wave * 64u + 63u - (uint32_t)__builtin_clzll(lanes_set)) >> 3);
			atomicAdd(&a->n_mask_px, cnt);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// k_brq_pass: ocr_preprocess (lib.rs:173-231) + find_scales_preprocess (lib.rs:233-251) over the
// bottom-right quadrant, same column-mask technique with a 3-row halo.
//   monochromaticy = sum over ordered pairs |ci-cj| = 4*(max-min)
//     "<= 3"  <=> r == g == b            "<= 48" <=> max-min <= 12
//   keep(x,y) = W(x,y) || (E(x,y) && exists W in [x-3, min(x+3, w-3)] x [y-3, min(y+3, h-3)])
//     W = r==g==b && all >= 200,  E = max-min <= 12 && all >= 130
// ------------------------------------------------------------------------------------------------
#define BRQ_RB 58

__global__ void __launch_bounds__(1024) k_brq_pass(Geom g, Buffers b, uint32_t flags, uint32_t fixed_start_y, int use_anchor_start) {
	const uint32_t f = blockIdx.y;
	if (!b.aux[f].open) return;
	uint32_t start_y = fixed_start_y;
	bool do_scales = (flags & BRQ_SCALES) != 0;
	if (use_anchor_start) {
		const smhv_anchors an = b.anchors[f];
		start_y = an.scales_start_y;
		// src/vision/mod.rs:196-198: no labels => the scales branch returns before find_scales_preprocess
		if (an.n == 0 || start_y > g.qh) do_scales = false;
	}
	const bool do_ocr = (flags & BRQ_OCR) != 0;
	const uint32_t q = threadIdx.x, lane = q & 63u, wave = q >> 6, nwave = blockDim.x >> 6;
	const int r0 = (int)(blockIdx.x * BRQ_RB);
	const int r1 = min(r0 + BRQ_RB, (int)g.qh);
	const bool qact = q < g.q_quads;
	uint32_t vmask = 0, wmask = 0;   // valid pixel / pixel allowed as a "white neighbour" (x <= w-3)
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const uint32_t x = 4 * q + c - g.q_xoff;
		if (x < g.qw) vmask |= 1u << c;
		if (x + SMH_OCR_DILATE_RADIUS <= g.qw) wmask |= 1u << c;   // x <= w - 3
	}
	if (!qact) { vmask = 0; wmask = 0; }
	wmask &= vmask;

	const uint8_t *fp = b.frames + (size_t)f * g.frame_bytes + ((size_t)g.qy * g.W + g.q_ax + 4 * q) * 4;
	const size_t row_bytes = (size_t)g.W * 4;
	uint8_t *op = b.ocr + (size_t)f * g.ocr_stride + (size_t)q * 4;
	uint8_t *sp = b.scales + (size_t)f * g.ocr_stride + (size_t)q * 4;

	uint64_t Wb[4] = {0, 0, 0, 0}, Eb[4] = {0, 0, 0, 0};
	const int rs = max(r0 - 3, 0), re = min(r1 + 2, (int)g.qh - 1);
	// software pipeline as in k_map_pass: the loads of the next four rows fly under the work on the current four
	// (inactive lanes re-read quad 0 instead of diverging; their results are masked by vmask)
	const uint8_t *lp = qact ? fp : fp - (size_t)q * 16;
	uint4 nx[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) nx[k] = *(const uint4 *)(lp + (size_t)min(rs + k, re) * row_bytes);
	for (int r = rs; r <= re; r += 4) {
		uint4 px[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) px[k] = nx[k];
		if (r + 4 <= re) {
#pragma unroll
			for (int k = 0; k < 4; ++k) nx[k] = *(const uint4 *)(lp + (size_t)min(r + 4 + k, re) * row_bytes);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int row = r + k;
			if (row > re) break;
			const int bit = row - (r0 - 3);
			const uint32_t pv[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
			const bool out_row = row >= r0 && row < r1;
			const bool nb_row = (uint32_t)row + SMH_OCR_DILATE_RADIUS <= g.qh;   // y <= h - 3
			uint32_t ocr_w = 0, sc_w = 0;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const uint32_t p = pv[c], bb = p & 255u, gg = (p >> 8) & 255u, rr8 = (p >> 16) & 255u;
				const uint32_t mx = max(rr8, max(gg, bb)), mn = min(rr8, min(gg, bb));
				const bool w = (mx == mn) && mn >= SMH_OCR_BRIGHTNESS_THRESHOLD;
				const bool e = (4u * (mx - mn) <= SMH_OCR_SIMILARITY_EDGE_THRESHOLD) && mn >= SMH_OCR_BRIGHTNESS_EDGE_THRESHOLD && !w;
				const bool valid = (vmask >> c) & 1u;
				Wb[c] |= (uint64_t)((w && nb_row && ((wmask >> c) & 1u)) ? 1u : 0u) << bit;
				Eb[c] |= (uint64_t)((e && valid && out_row) ? 1u : 0u) << bit;
				const uint32_t l = luma8(rr8, gg, bb);
				ocr_w |= ((w && valid) ? (255u - l) : 255u) << (8 * c);
				sc_w |= (l != 0u ? 255u : 0u) << (8 * c);
			}
			if (out_row && qact) {
				if (do_ocr) *(uint32_t *)(op + (size_t)row * g.ocr_pitch) = ocr_w;
				if (do_scales && (uint32_t)row >= start_y) *(uint32_t *)(sp + (size_t)row * g.ocr_pitch) = sc_w;
			}
		}
	}
	if (!do_ocr) return;

	// ---- 7x7 "white neighbour" dilation on the column masks ----
	uint64_t V[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const uint64_t w = Wb[c];
		V[c] = w | (w << 1) | (w << 2) | (w << 3) | (w >> 1) | (w >> 2) | (w >> 3);
	}
	__shared__ uint64_t s_first[16][4], s_last[16][4];
	if (lane == 0) { s_first[wave][0] = V[0]; s_first[wave][1] = V[1]; s_first[wave][2] = V[2]; s_first[wave][3] = V[3]; }
	if (lane == 63) { s_last[wave][0] = V[0]; s_last[wave][1] = V[1]; s_last[wave][2] = V[2]; s_last[wave][3] = V[3]; }
	__syncthreads();
	uint64_t X[12];   // columns -4..7 relative to this quad
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		uint64_t l = __shfl_up(V[c], 1), r = __shfl_down(V[c], 1);
		if (lane == 0) l = wave > 0 ? s_last[wave - 1][c] : 0ull;
		if (lane == 63) r = wave + 1 < nwave ? s_first[wave + 1][c] : 0ull;
		X[c] = l; X[4 + c] = V[c]; X[8 + c] = r;
	}
	bool any_patch = false;
	uint64_t K[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		uint64_t d = 0;
#pragma unroll
		for (int k = -3; k <= 3; ++k) d |= X[4 + c + k];
		K[c] = Eb[c] & d;
		any_patch = any_patch || K[c] != 0ull;
	}
	// Pixels kept only because of a white neighbour are rare (anti-aliased glyph edges): re-read
	// just those pixels for their luma and patch the byte written above (same thread => ordered).
	if (any_patch) {
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			uint64_t k = K[c];
			while (k) {
				const int bit = __builtin_ctzll(k);
				k &= k - 1;
				const int row = r0 - 3 + bit;
				const uint32_t p = *(const uint32_t *)(fp + (size_t)row * row_bytes + 4 * c);
				const uint32_t l = luma8((p >> 16) & 255u, (p >> 8) & 255u, p & 255u);
				op[(size_t)row * g.ocr_pitch + c] = (uint8_t)(255u - l);
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// k_map_brq_pass: k_map_pass and k_brq_pass in ONE pass over the map ROI (the batched pipeline with the OCR / scales
// stages selected).  The bottom-right quadrant is part of the ROI and both kernels walk the same quad grid (quads are
// aligned to frame x % 4 == 0), so a thread whose quad column crosses the quadrant evaluates ocr_preprocess /
// find_scales_preprocess on the pixels it has already loaded for ui_map and the marker mask: the quadrant is not read a
// second time (k_brq_pass re-read 2 x 126 MB per 256 frames) and one launch and the branch stream go away.
// Column masks are indexed from row r0 - 3 (the OCR neighbourhood reaches 3 rows up; the marker dilation 1), so a band
// holds at most 58 output rows: 3 + 58 + 3 = 64 bits.  band_rows_for: whole tile rows of the line search -- the pass then writes
// the mask tile-major as well (smh_kernels.h) -- for ROIs up to 900 rows (1080p), where the search service sits on every CU and the
// cheaper tile-store build buys the pass a third workgroup per CU, and 24 of them: with four-wave workgroups (ROIs up to 1024 px
// wide) a launch alone is 5-11 % shorter in 24-row bands than in 56-row ones in spite of reading 6 extra rows per 24 instead of per
// 56 (one box, 256 frames, 56 / 48 / 40 / 32 / 24 / 16 rows: 1080p 0.466 / 0.470 / 0.450 / 0.444 / 0.441 / 0.478 ms, 1600 x 900 0.376 /
// 0.363 / 0.361 / 0.353 / 0.346 / 0.369, 720p 0.263 / 0.263 / 0.250 / 0.244 / 0.233 / 0.241; 64 frames of 1080p 0.153 -> 0.142; the plain
// pass k_map_pass 0.437 -> 0.410), and inside the frame-granular pipeline the band height is immaterial (548-554 k frames/s at every
// height from 24 to 56 in runs of 600 submissions; in runs of 8,000, three interleaved rounds, 24 rows 548-552 k, 32 rows 550-558 k, 56 rows
// 556-558 k: profiles/r06_sweep_band_rows.txt -- so the launches of a frame-granular pipeline, beside_service, keep 56).  Taller ROIs: 56 where that costs no band, else 58 without the tile-major
// mask (1440p, six-wave workgroups: 19 bands instead of 20; alone 58 = 32 rows = 0.420 ms, 24 rows 0.439; in the pipeline 58-row bands
// and the walk over the bit rows 284-287 k frames/s, 56 rows 271-273 k, 32 rows 254-258 k, 24 rows 240-246 k).
// ------------------------------------------------------------------------------------------------
#define MAPQ_RB_MAX 58
// rows per band of a launch over n frames of an ROI rh rows tall, for column masks that hold rb_cap rows.  A multiple of 8 means
// "this launch writes the tile-major mask" (the kernels tell the search through FrameAux::tiles).
// Fewer frames than fill the chip: shorter bands.
std::atomic<uint32_t> g_map_band_major{0};  // diagnostic (smhv_debug_map_band_rows, bits 31 / 30 of its argument): 1 = band-major order always, 2 = never; 0: the rule
std::atomic<uint32_t> g_map_band_rows{0};   // diagnostic (smhv_debug_map_band_rows): rows per band of the launches that write the tile-major mask; 0: the rule
static inline uint32_t band_rows_for(uint32_t rh, uint32_t n, uint32_t rb_cap, bool tiles_wanted = true, bool beside_service = false) {
	const uint32_t rb8 = rb_cap & ~7u, forced = g_map_band_rows.load(std::memory_order_relaxed);
	uint32_t RB = rb_cap;
	if (tiles_wanted && forced) RB = std::min(rb8, forced);
	else if (tiles_wanted && rh <= 900u) RB = beside_service ? rb8 : 24u;
	else if (tiles_wanted && (rh + rb8 - 1) / rb8 == (rh + rb_cap - 1) / rb_cap) RB = rb8;
	const uint64_t fill = RB == 24u ? 768u : 512u;            // (work items that fill the chip: measured with the band heights they go with)
	while (RB > 8 && (uint64_t)((rh + RB - 1) / RB) * n < fill) RB = (RB & 7u) ? (RB + 1) / 2 : (RB > 32 ? 32 : RB > 16 ? 16 : 8);
	return RB;
}

// ---- small pieces of the fused pass ----
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// A 16-byte load the compiler does not track (uniform row base + 32-bit lane offset) and the wait that releases a set of four
// of them: the streaming loop of k_map_brq_pass places its own waits (see there).  Nothing may read a destination before
// SMH_WAIT_SET has named it; tools/check_untracked_loads.py checks the compiled code for that.
// -DSMH_TRACKED_LOADS: the fallback the Makefile builds when the checker rejects the compiled code (a compiler that moved,
// copied or spilled a destination inside its load's window): ordinary loads the compiler tracks and waits for itself; the
// same pipeline shape, its conservative waits: the same 0.46 ms alone, 1.06 instead of 0.87 ms per launch inside the depth-4
// pipeline (416 k against 457 k frames/s).
#ifdef SMH_TRACKED_LOADS
#define SMH_LD128(dst, voff, sbase) (dst) = *(const u32x4 *)((sbase) + (voff))
#else
#define SMH_LD128(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2 ; smh-load" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#endif
// stores in the same addressing form (a per-lane 64-bit pointer per output would cost the loop six registers it does not have)
#define SMH_ST128(voff, data, sbase) asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff), "v"(data), "s"(sbase) : "memory")
#define SMH_ST32(voff, data, sbase) asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(data), "s"(sbase) : "memory")
#ifdef SMH_TRACKED_LOADS
#define SMH_WAIT_SET(n, X) ((void)0)
#else
#define SMH_WAIT_SET(n, X) asm volatile("s_waitcnt vmcnt(" #n ") ; smh-release" : "+v"((X)[0]), "+v"((X)[1]), "+v"((X)[2]), "+v"((X)[3]) : : "memory")
#endif
// image 0.23.14 rgb_to_luma of a BGRA dword, without luma8()'s clamp: the f32 sum is at most 255.0 (r = g = b = 255; the
// sum is monotonic in every channel), so the truncation cannot exceed 255.
__device__ __forceinline__ uint32_t luma_bgra(uint32_t p) {
	const float l = SMH_LUMA_R * (float)((p >> 16) & 255u) + SMH_LUMA_G * (float)((p >> 8) & 255u) + SMH_LUMA_B * (float)(p & 255u);
	return (uint32_t)l;
}
// Necessary condition of marker_prefilter for four pixels at once: some colour channel is >= 178 (bit 7 set and the low
// seven bits >= 50; bytes are tested in place, the alpha byte is masked out).  Terrain darker than that -- most of a map
// -- skips the per-pixel test altogether.
__device__ __forceinline__ uint32_t bright4(const uint32_t pv[4]) {
	uint32_t acc = 0;
#pragma unroll
	for (int c = 0; c < 4; ++c) acc |= ((pv[c] & 0x007F7F7Fu) + 0x004E4E4Eu) & pv[c];
	return acc & 0x00808080u;
}
// bytes of 0x00 / 0x01 -> bytes of 0x00 / 0xFF: as a v_perm_b32 selector, 0x0C yields the constant 0x00 and 0x0D the constant
// 0xFF (the compiler turns (x << 8) - x back into a full-width multiply by 255, four times the issue cost)
__device__ __forceinline__ uint32_t ones_to_bytes(uint32_t ones) { return __builtin_amdgcn_perm(0u, 0u, ones + 0x0C0C0C0Cu); }
// 4 flag bits -> 4 bytes of 0x00 / 0xFF
__device__ __forceinline__ uint32_t nib_to_bytes(uint32_t nib) {
	return ones_to_bytes((nib * 0x00204081u) & 0x01010101u);
}
// rows k = 0..3 of a group as nibbles (bit 4k + c = pixel column c of row k) -> bits 0..3 of column c
__device__ __forceinline__ uint32_t rows_of_col(uint32_t nibbles, int c) {
	const uint32_t y = (nibbles >> c) & 0x1111u;
	return (y | (y >> 3) | (y >> 6) | (y >> 9)) & 0xFu;
}

// One work item = one band of RB rows of one frame.  A real call: inlined into the kernel's grid-stride loop the item spills
// (the loop's own state on top of 123 registers), and a spilled load destination would be spilled before its data has
// arrived.  The callee reads the launch parameters from the kernel argument segment (scalar loads), so nothing but the
// item's coordinates crosses the call.
#ifndef SMH_MAP_ITEM_INLINE
#define SMH_MAP_ITEM_INLINE __forceinline__
#endif
struct MapKernelArgs { Geom g; Buffers b; uint32_t flags, qflags, RB, fixed_start_y; int use_anchor_start; uint32_t nbands, items; };
#ifdef __HIP_DEVICE_COMPILE__
typedef const __attribute__((address_space(4))) MapKernelArgs *MapKernelArgsPtr;   // constant address space: scalar loads
#else
typedef const MapKernelArgs *MapKernelArgsPtr;                                     // (host pass: the body is only parsed)
#endif
// TILES: this launch also writes the mask tile-major with occupancy bytes (bands of whole tile rows; launch_map_brq_pass).  A
// launch that does not runs the instantiation without a trace of it: the same code as before round 6 (measured: the extra
// epilogue code alone cost the 1440p pipeline 2.4 % and the three-set form 6 %, whether or not it was executed).
template <bool GRAY, int SETS, bool TILES>
__device__ SMH_MAP_ITEM_INLINE void map_brq_item(MapKernelArgsPtr ka, uint32_t f, uint32_t band0) {
	constexpr int GR = 4;                                       // rows per group (the hand-placed waits count four loads per set)
	const Geom g = ka->g;
	const Buffers b = ka->b;
	const uint32_t flags = ka->flags, qflags = ka->qflags, RB = ka->RB, fixed_start_y = ka->fixed_start_y;
	const int use_anchor_start = ka->use_anchor_start;
	if (!b.aux[f].open) return;
	const uint32_t band = band0;
	constexpr bool live = true;
	uint32_t start_y = fixed_start_y;
	bool do_scales = (qflags & BRQ_SCALES) != 0;
	if (use_anchor_start && do_scales) {
		const smhv_anchors an = b.anchors[f];
		start_y = an.scales_start_y;
		// src/vision/mod.rs:196-198: no labels => the scales branch returns before find_scales_preprocess
		if (an.n == 0 || start_y > g.qh) do_scales = false;
	}
	const bool do_ocr = (qflags & BRQ_OCR) != 0;
	const uint32_t q = threadIdx.x, lane = q & 63u, wave = q >> 6;   // quad, lane and wave within the band
	const uint32_t nwave = blockDim.x >> 6, gwave = wave;
	const int r0 = live ? (int)(band * RB) : 0;
	const int r1 = live ? min(r0 + (int)RB, (int)g.rh) : 0;
	const bool qact = q < g.m_quads && live;
	uint32_t vmask = 0;
#pragma unroll
	for (int c = 0; c < 4; ++c)
		if ((uint32_t)(4 * q + c - g.m_xoff) < g.rw) vmask |= 1u << c;
	if (!qact) vmask = 0;
	// ---- the quadrant as this thread sees it: quad index in the quadrant's own (padded) rows, valid / neighbour pixels ----
	const int qy0 = (int)(g.qy - g.ry);                          // first quadrant row, ROI coordinates
	const int qq = (int)q - (int)((g.q_ax - g.m_ax) >> 2);       // quad index in the quadrant's output rows
	const bool in_q = qq >= 0 && (uint32_t)qq < g.q_quads && qact;
	uint32_t qv = 0, qw_ = 0;                                    // valid pixel / pixel allowed as a "white neighbour" (x <= w - 3)
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const uint32_t x = (uint32_t)(4 * qq + c) - g.q_xoff;
		if (in_q && x < g.qw) qv |= 1u << c;
		if (in_q && x + SMH_OCR_DILATE_RADIUS <= g.qw) qw_ |= 1u << c;
	}
	qw_ &= qv;
	const int qr0 = max(r0 - qy0, 0), qr1 = min(r1 - qy0, (int)g.qh);     // quadrant rows this band writes: [qr0, qr1)
	const bool band_q = qr1 > qr0 && (do_ocr || do_scales);               // uniform: the band touches the quadrant

	// addresses: a uniform 64-bit base per row plus a 32-bit lane offset (inactive lanes re-read quad 0: no divergent load)
	const uint8_t *fbase = b.frames + (size_t)f * g.frame_bytes + ((size_t)g.ry * g.W + g.m_ax) * 4;
	const uint32_t loff = qact ? q * 16u : 0u;
	const uint8_t *fp = fbase + (size_t)q * 16;                  // (patch loop below)
	uint8_t *uibase = b.ui + (size_t)f * g.ui_stride;
	uint8_t *op = b.ocr + (size_t)f * g.ocr_stride;
	uint8_t *sp = b.scales + (size_t)f * g.ocr_stride;
	const uint32_t qoff = (uint32_t)(in_q ? qq : 0) * 4u;
	const size_t row_bytes = (size_t)g.W * 4;

	uint64_t P[4] = {0, 0, 0, 0}, Wb[4] = {0, 0, 0, 0}, Eb[4] = {0, 0, 0, 0};
	extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
	auto or_w = [&](int c, uint64_t v) { Wb[c] |= v; };
	auto or_e = [&](int c, uint64_t v) { Eb[c] |= v; };
	auto get_w = [&](int c) -> uint64_t { return Wb[c]; };
	auto get_e = [&](int c) -> uint64_t { return Eb[c]; };
	// rows walked by the streaming loop: r0-1 .. r1, as in k_map_pass (the marker dilation's halo)
	const int rs = max(r0 - 1, 0), re = min(r1, (int)g.rh - 1);
	const bool do_ui = (flags & MAP_UI) != 0, do_mask = (flags & MAP_MASK) != 0;
	const int base = r0 - 3;                                     // row of bit 0 of every column mask
	const bool my_q = band_q && in_q;                            // this thread has quadrant pixels in this band
	// (uniform per WAVE: at 1080p the first of a band's four waves has no quadrant column at all, and its share of the white /
	// edge classification -- 70 instructions per row and quad in the lower half of the ROI -- would be thrown away)
	const bool wave_q = band_q && __any(in_q);

	// ---- the OCR neighbourhood reaches 3 rows up and 2 down: the four rows beyond the loop's halo contribute their "white"
	// bits only, and only the threads of the quadrant's columns look at them
	if (my_q && do_ocr && live) {
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int row = k < 2 ? r0 - 3 + k : r1 + k - 1;          // r0-3, r0-2, r1+1, r1+2
			const int qrow = row - qy0;
			if (qrow < 0 || qrow >= (int)g.qh || row < 0 || row >= (int)g.rh) continue;
			if ((uint32_t)qrow + SMH_OCR_DILATE_RADIUS > g.qh) continue;             // y <= h - 3
			const uint4 v = *(const uint4 *)(fp + (size_t)row * row_bytes);
			const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const uint32_t p = pv[c], bb = p & 255u, gg = (p >> 8) & 255u, rr8 = (p >> 16) & 255u;
				const uint32_t mx = max(rr8, max(gg, bb)), mn = min(rr8, min(gg, bb));
				const bool w = (mx == mn) && mn >= SMH_OCR_BRIGHTNESS_THRESHOLD && ((qw_ >> c) & 1u);
				or_w(c, (uint64_t)(w ? 1u : 0u) << (row - base));
			}
		}
	}

	uint32_t *s_hits = s_dyn;
	// One group = four rows r .. r+3 held in px.  Everything that depends on the row only (band membership, quadrant
	// membership) is a uniform branch; lanes outside the ROI / the quadrant compute along and are masked at the end.
	// Branch hints for the blocks of the streaming loop that few waves enter (a marker colour's pre-filter hit, the exact test, a
	// white / edge pixel of the quadrant): where the compiler puts those blocks decides 1-2 % of the PIPELINE's rate -- the loop is
	// 4 k instructions beside a search kernel of its own size on the same instruction cache -- and in both directions: one box,
	// three interleaved rounds, with / without the hints: 256 x 1080p 547-551 against 539-541 k frames/s, 128 x 1440p 281 against
	// 286-287 k.  So only the instantiation that the frame sizes up to 1080p launch (TILES) carries them; the others are round 5's
	// code to the instruction.
	auto rare = [](bool c) -> bool { if constexpr (TILES) return __builtin_expect(c, 0); else return c; };
	auto group = [&](const u32x4 (&px)[GR], int r) {
		uint32_t prehits = 0, wrows = 0, erows = 0;                // bit 4k + c: pixel column c of row r + k
#pragma unroll
		for (int k = 0; k < GR; ++k) {
			const int row = r + k;
			if (row > re) break;
			const uint32_t pv[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
			const bool out_row = row >= r0 && row < r1;
			const int qrow = row - qy0;                              // quadrant row
			const bool q_row = wave_q && qrow >= 0 && qrow < (int)g.qh;   // uniform
			uint32_t wnib = 0, enib = 0;                             // white / edge pixels of this row (quadrant rows only)
			if (q_row) {
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const uint32_t p = pv[c], bb = p & 255u, gg = (p >> 8) & 255u, rr8 = (p >> 16) & 255u;
					const uint32_t mx = max(rr8, max(gg, bb)), mn = min(rr8, min(gg, bb));
					const bool w = (mx == mn) && mn >= SMH_OCR_BRIGHTNESS_THRESHOLD;
					const bool e = (4u * (mx - mn) <= SMH_OCR_SIMILARITY_EDGE_THRESHOLD) && mn >= SMH_OCR_BRIGHTNESS_EDGE_THRESHOLD && !w;
					wnib |= (w ? 1u : 0u) << c;
					enib |= (e ? 1u : 0u) << c;
				}
				const bool nb_row = (uint32_t)qrow + SMH_OCR_DILATE_RADIUS <= g.qh;   // y <= h - 3
				if (nb_row) wrows |= (wnib & qw_) << (4 * k);
				if (out_row) erows |= (enib & qv) << (4 * k);           // (a band row inside the quadrant is a row this band writes)
			}
			if (out_row) {
				uint32_t lum[4] = {0, 0, 0, 0};
				if (GRAY || q_row) {
#pragma unroll
					for (int c = 0; c < 4; ++c) lum[c] = luma_bgra(pv[c]);
				}
				if (do_ui && qact) {
					u32x4 o;
					uint32_t ov[4];
#pragma unroll
					for (int c = 0; c < 4; ++c) {
						const uint32_t p = pv[c], bb = p & 255u, gg = (p >> 8) & 255u, rr8 = (p >> 16) & 255u;
						if (GRAY) ov[c] = __builtin_amdgcn_perm(0u, lum[c], 0x0D000000u);       // Bgra::to_luma -> (l,l,l,255): bytes 0..2 = l, byte 3 = 0xFF
						else ov[c] = rr8 | (gg << 8) | (bb << 16) | 0xFF000000u;               // (r,g,b,255)
					}
					o.x = ov[0]; o.y = ov[1]; o.z = ov[2]; o.w = ov[3];
					SMH_ST128(loff, o, uibase + (size_t)row * g.ui_pitch);                   // (an active quad's lane offset is q * 16)
				}
				if (q_row && in_q) {
					// ocr_preprocess keeps a white pixel as 255 - luma and blanks the rest (edge pixels are patched at the end);
					// find_scales_preprocess is luma != 0.  Four pixels at a time on the packed luma bytes.
					const uint32_t l4 = lum[0] | (lum[1] << 8) | (lum[2] << 16) | (lum[3] << 24);
					if (do_ocr) SMH_ST32(qoff, ~(l4 & nib_to_bytes(wnib & qv)), op + (size_t)qrow * g.ocr_pitch);
					if (do_scales && (uint32_t)qrow >= start_y) {
						const uint32_t nz = ((((l4 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | l4) >> 7) & 0x01010101u;     // bytes != 0
						SMH_ST32(qoff, ones_to_bytes(nz), sp + (size_t)qrow * g.ocr_pitch);
					}
				}
			}
			if (rare(do_mask && __any(bright4(pv) != 0u))) {
				uint32_t pre = 0;
#pragma unroll
				for (int c = 0; c < 4; ++c) pre |= marker_prefilter(pv[c]) ? (1u << c) : 0u;
				prehits |= (pre & vmask) << (4 * k);
			}
		}
		if (rare(wave_q && __any((wrows | erows) != 0u))) {                // the group's four rows into the column masks
			const int sh = r - base;
#pragma unroll
			for (int c = 0; c < 4; ++c) { or_w(c, (uint64_t)rows_of_col(wrows, c) << sh); or_e(c, (uint64_t)rows_of_col(erows, c) << sh); }
		}
		// ---- exact f32 HSV test for the pre-filter hits of this wave, one hit per lane (see k_map_pass) ----
		if (rare(do_mask && __any(prehits != 0u))) {
			uint32_t *hpx = s_hits + gwave * 160u;
			uint32_t *hres = hpx + 64;
			unsigned short *hid = (unsigned short *)(hres + 64);
			const uint32_t cnt = (uint32_t)__popc(prehits);
			uint32_t incl = cnt;
			for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
			const uint32_t total = __shfl(incl, 63);
			const uint32_t off = incl - cnt;
			hres[lane] = 0u;
			for (uint32_t hb = 0; hb < total; hb += 64u) {
				uint32_t o = off - hb;                             // may wrap: compared unsigned below
#pragma unroll
				for (int k = 0; k < GR; ++k) {
					const uint32_t pk[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
					for (int c = 0; c < 4; ++c)
						if ((prehits >> (4 * k + c)) & 1u) {
							if (o < 64u) { hpx[o] = pk[c]; hid[o] = (unsigned short)((lane << 4) | (uint32_t)(4 * k + c)); }
							++o;
						}
				}
				__builtin_amdgcn_wave_barrier();
				const uint32_t e = hb + lane;
				if (e < total) {
					const uint32_t p = hpx[lane];
					if (marker_exact((p >> 16) & 255u, (p >> 8) & 255u, p & 255u)) {
						const uint32_t id = hid[lane];
						atomicOr(&hres[id >> 4], 1u << (id & 15u));
					}
				}
				__builtin_amdgcn_wave_barrier();
			}
			const uint32_t res = hres[lane];                       // bit 4k+c: pixel c of row r+k is a marker colour
			if (res) {
				const int sh = r - base;
#pragma unroll
				for (int c = 0; c < 4; ++c) P[c] |= (uint64_t)rows_of_col(res, c) << sh;
			}
		}
	};

	// Software pipeline, three register sets, prefetch distance two groups: the loads of set n are issued before group n - 2
	// is processed, so two groups' worth of loads (8 KB per wave) are in flight while a group is processed, and the wait for
	// set n is "at most the eight younger loads outstanding".  vmcnt counts stores too and a store's acknowledgement can
	// overtake an older load (measured: crediting the stores issued since, which an in-order counter would allow, hands out
	// stale pixels), so the wait also covers the stores of the two groups before -- two groups old by then, and cheaper
	// than what the compiler makes of tracked loads here: its waits assume the fewest stores any path could have issued,
	// none, and every group then waits for the stores of the group right before it.  (Same-box A/B: no difference in time to
	// the compiler-scheduled version -- the memory system sets the pace, DESIGN.md -- but fewer registers and no spills.)
	// A set is always four loads (the count behind a wait has to be static; making the loads of the last sets conditional
	// makes the compiler copy registers that still have a load in flight -- tools/check_untracked_loads.py caught it): the
	// loads of rows beyond the band's last go, all lanes, to one 16-byte location that is always in the L2 (the frame's aux
	// record) instead of re-reading pixel rows, which by then have been evicted by the pass's own stores (measured: 13 % more
	// bytes fetched).  Their data is never looked at.
	{
		const uint8_t *dummy = (const uint8_t *)&b.aux[f];
		auto load4 = [&](u32x4 (&dst)[GR], int r) {
#pragma unroll
			for (int k = 0; k < GR; ++k) {
				const bool inside = r + k <= re;
				const uint8_t *rowp = inside ? fbase + (size_t)(r + k) * row_bytes : dummy;
				SMH_LD128(dst[k], inside ? loff : 0u, rowp);
			}
		};
		if constexpr (SETS == 2) {
			// two register sets, prefetch distance one group (4 KB per wave in flight while a group is processed): what frames above
			// 1080p run -- one box, three interleaved rounds, 128 x 1440p at depth 12: 289 k frames/s against 279 k with three
			// sets; 256 x 1080p: 538 against 560 k (DESIGN.md A.-1).  "At most the four younger loads outstanding" releases a set.
			u32x4 S0[GR], S1[GR];
			load4(S0, rs);
			for (int r = rs; ; r += 8) {
				load4(S1, r + 4);
				SMH_WAIT_SET(4, S0);
				group(S0, r);
				if (r + 4 > re) break;
				load4(S0, r + 8);
				SMH_WAIT_SET(4, S1);
				group(S1, r + 4);
				if (r + 8 > re) break;
			}
			asm volatile("s_waitcnt vmcnt(0) ; smh-drain" : "+v"(S0[0]), "+v"(S0[1]), "+v"(S0[2]), "+v"(S0[3]), "+v"(S1[0]), "+v"(S1[1]), "+v"(S1[2]), "+v"(S1[3]) : : "memory");
		} else {
			u32x4 S0[GR], S1[GR], S2[GR];
			load4(S0, rs);
			load4(S1, rs + 4);
			for (int r = rs; ; r += 12) {
				load4(S2, r + 8);
				SMH_WAIT_SET(8, S0);
				group(S0, r);
				if (r + 4 > re) break;
				load4(S0, r + 12);
				SMH_WAIT_SET(8, S1);
				group(S1, r + 4);
				if (r + 8 > re) break;
				load4(S1, r + 16);
				SMH_WAIT_SET(8, S2);
				group(S2, r + 8);
				if (r + 12 > re) break;
			}
			// the clamped loads of the sets nobody consumed are still in flight: their registers must not be reused before they land
			asm volatile("s_waitcnt vmcnt(0) ; smh-drain" : "+v"(S0[0]), "+v"(S0[1]), "+v"(S0[2]), "+v"(S0[3]), "+v"(S1[0]), "+v"(S1[1]), "+v"(S1[2]), "+v"(S1[3]),
			             "+v"(S2[0]), "+v"(S2[1]), "+v"(S2[2]), "+v"(S2[3]) : : "memory");
		}
	}

	// ---- lane / wave neighbours of the column masks: marker dilation (P) and the 7-row-dilated white masks (V) ----
	uint64_t V[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const uint64_t w = get_w(c);
		V[c] = w | (w << 1) | (w << 2) | (w << 3) | (w >> 1) | (w >> 2) | (w >> 3);
	}
	__shared__ uint64_t s_edge_first[16], s_edge_last[16], s_first[16][4], s_last[16][4];
	// (gwave: the wave's place in the workgroup; a band's waves are consecutive, `wave` counts within the band)
	if (lane == 0) { s_edge_first[gwave] = P[0]; s_first[gwave][0] = V[0]; s_first[gwave][1] = V[1]; s_first[gwave][2] = V[2]; s_first[gwave][3] = V[3]; }
	if (lane == 63) { s_edge_last[gwave] = P[3]; s_last[gwave][0] = V[0]; s_last[gwave][1] = V[1]; s_last[gwave][2] = V[2]; s_last[gwave][3] = V[3]; }
	__syncthreads();
	if (do_mask) {
		uint64_t left = __shfl_up(P[3], 1), right = __shfl_down(P[0], 1);
		if (lane == 0) left = wave > 0 ? s_edge_last[gwave - 1] : 0ull;
		if (lane == 63) right = wave + 1 < nwave ? s_edge_first[gwave + 1] : 0ull;
		const int nrows = r1 - r0;
		const uint64_t rowmask = ((nrows >= 61 ? ~0ull : ((1ull << nrows) - 1ull)) << 3);   // bits 3..3+nrows-1
		uint64_t D[4];
#define SMH_VERT(p) ((p) | ((p) << 1) | ((p) >> 1))
		D[0] = SMH_VERT(P[0]) | left | P[1];
		D[1] = SMH_VERT(P[1]) | P[0] | P[2];
		D[2] = SMH_VERT(P[2]) | P[1] | P[3];
		D[3] = SMH_VERT(P[3]) | P[2] | right;
#undef SMH_VERT
#pragma unroll
		for (int c = 0; c < 4; ++c) D[c] = ((vmask >> c) & 1u) ? (D[c] & rowmask) : 0ull;
		const uint32_t quads_padded = (g.m_quads + 15u) & ~15u;
		const uint64_t any = D[0] | D[1] | D[2] | D[3];
		const uint64_t lanes_set = __ballot(any != 0ull);
		// the tile-major mask and its occupancy bytes for the search service's tile-store builder (smh_kernels.h; bands of whole
		// tile rows: launch_map_brq_pass).  A wave without a marker pixel -- most of them -- stores its (up to seven) zero bytes.
		// (their two pointers are fetched HERE, through a copy of the argument pointer the compiler cannot see through: held in
		// scalar registers across the streaming loop they cost it 84 more spill reloads per iteration of the three-set form)
		uint32_t occ7 = 0;
		uint32_t *tiled_out = nullptr;
		if constexpr (TILES) {
			MapKernelArgsPtr kb = ka;
			asm volatile("" : "+s"(kb));
			tiled_out = kb->b.tiled;
			uint8_t *const occ_out = kb->b.occ;
			const uint32_t ntr = ((uint32_t)nrows + 7u) >> 3, opitch = occ_pitch(g);
			uint8_t *occp = occ_out + (size_t)f * occ_stride(g) + (size_t)((uint32_t)r0 >> 3) * opitch + wave;
			if (lanes_set) occ7 = tile_occupancy(any >> 3, lane, ntr, occp, opitch);
			else if (lane < ntr) occp[(size_t)lane * opitch] = 0;
			if (band == 0u && q == 0u) b.aux[f].tiles = 1u;             // (k_button cleared it: this frame's tile-major mask is being written)
		}
		if (q < quads_padded && !lanes_set) {
			// no marker pixel in this wave's 256 columns of the band (most of a map): rows of zeros, nothing to extract or gather
			uint8_t *mbase = b.mask + (size_t)f * g.mask_stride;
			uint32_t *bbase = b.bits + (size_t)f * g.bits_stride_w;
			const uint32_t moff = q * 4u, boff = (q >> 3) * 4u;
			const bool bit_lane = (lane & 7u) == 0;
			for (int row = r0; row < r1; ++row) {
				*(uint32_t *)(mbase + (size_t)row * g.mask_pitch + moff) = 0u;
				if (bit_lane) *(uint32_t *)((uint8_t *)bbase + (size_t)row * g.bits_pitch_w * 4u + boff) = 0u;
			}
		} else if (q < quads_padded) {
			// u8 mask rows and bit-packed rows.  The column masks are walked one 32-row half at a time (32-bit bit-field
			// extracts on a uniform bit index); eight lanes' nibbles meet in one dword through three DPP row shifts
			// (lane l supplies bits 4 (l % 8) ..): no LDS round trip, nothing to wait for.
			uint8_t *mbase = b.mask + (size_t)f * g.mask_stride;
			uint32_t *bbase = b.bits + (size_t)f * g.bits_stride_w;
			const uint32_t moff = q * 4u, boff = (q >> 3) * 4u;
			const bool bit_lane = (lane & 7u) == 0;
			uint32_t *tp = nullptr;
			uint32_t tpitch = 0;
			if constexpr (TILES) { tp = tiled_out + (size_t)f * tiled_stride_w(g) + ((size_t)((uint32_t)r0 >> 3) * g.bits_pitch_w + (q >> 3)) * 8u; tpitch = g.bits_pitch_w * 8u; }
#pragma unroll
			for (int half = 0; half < 2; ++half) {
				const uint32_t d0 = (uint32_t)(D[0] >> (32 * half)), d1 = (uint32_t)(D[1] >> (32 * half)), d2 = (uint32_t)(D[2] >> (32 * half)), d3 = (uint32_t)(D[3] >> (32 * half));
				const int row_lo = max(r0, base + 32 * half), row_hi = min(r1, base + 32 * half + 32);
				for (int row = row_lo; row < row_hi; ++row) {
					const uint32_t bit = (uint32_t)(row - base) & 31u;
					const uint32_t nib = ((d0 >> bit) & 1u) | (((d1 >> bit) & 1u) << 1) | (((d2 >> bit) & 1u) << 2) | (((d3 >> bit) & 1u) << 3);
					*(uint32_t *)(mbase + (size_t)row * g.mask_pitch + moff) = nib_to_bytes(nib);
					uint32_t v = nib;
					v |= SMH_DPP(v, 0x101) << 4;                    // row_shl:1 -- lane l reads lane l + 1 (0 beyond the 16-lane row)
					v |= SMH_DPP(v, 0x102) << 8;
					v |= SMH_DPP(v, 0x104) << 16;
					if (bit_lane) *(uint32_t *)((uint8_t *)bbase + (size_t)row * g.bits_pitch_w * 4u + boff) = v;
					if constexpr (TILES) {
						const uint32_t i = (uint32_t)(row - r0);
						if ((occ7 >> (i >> 3)) & 1u) tp[(size_t)(i >> 3) * tpitch + (i & 7u)] = v;   // (occ7 is 0 outside the lanes 8 j)
					}
				}
			}
		}
		if (lanes_set) {
			const uint64_t rows_set = wave_or64(any);
			const uint32_t cnt = wave_sum32(__popcll(D[0]) + __popcll(D[1]) + __popcll(D[2]) + __popcll(D[3]));
			if (lane == 0) {
				FrameAux *a = &b.aux[f];
				atomicMin(&a->y_min, (uint32_t)(base + __builtin_ctzll(rows_set)));
				atomicMax(&a->y_max, (uint32_t)(base + 63 - __builtin_clzll(rows_set)));
				atomicMin(&a->w_min, (wave * 64u + (uint32_t)__builtin_ctzll(lanes_set)) >> 3);
				atomicMax(&a->w_max, (wave * 64u + 63u - (uint32_t)__builtin_clzll(lanes_set)) >> 3);
				atomicAdd(&a->n_mask_px, cnt);
			}
		}
	}
	if (!(wave_q && do_ocr)) return;
	// ---- ocr_preprocess: edge pixels with a white pixel in their 7x7 neighbourhood (see k_brq_pass) ----
	uint64_t X[12];   // columns -4..7 relative to this quad
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		uint64_t l = __shfl_up(V[c], 1), r = __shfl_down(V[c], 1);
		if (lane == 0) l = wave > 0 ? s_last[gwave - 1][c] : 0ull;
		if (lane == 63) r = wave + 1 < nwave ? s_first[gwave + 1][c] : 0ull;
		X[c] = l; X[4 + c] = V[c]; X[8 + c] = r;
	}
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		uint64_t d = 0;
#pragma unroll
		for (int k = -3; k <= 3; ++k) d |= X[4 + c + k];
		uint64_t kk = get_e(c) & d;
		// rare (anti-aliased glyph edges): re-read just those pixels for their luma and patch the byte written above
		while (kk) {
			const int bit = __builtin_ctzll(kk);
			kk &= kk - 1;
			const int row = base + bit;
			const uint32_t p = *(const uint32_t *)(fp + (size_t)row * row_bytes + 4 * c);
			const uint32_t l = luma_bgra(p);
			op[(size_t)(row - qy0) * g.ocr_pitch + qoff + c] = (uint8_t)(255u - l);
		}
	}
}

// LOOP: the grid is capped (launch_map_brq_pass) and a workgroup walks the (frame, band) items with a grid stride.  Beyond
// the number of workgroups that saturates HBM, more resident streaming workgroups only wait on each other in the memory
// queues while holding wave slots and registers the other batches' line searches need.  The loop's
// state costs the kernel 76 bytes of scratch per lane (spilled in the prologue / epilogue of an item, never in the streaming
// loop: tools/check_untracked_loads.py); a launch with one workgroup per item (a batch that runs alone) takes the variant
// without the loop and without the spills.
template <bool GRAY, bool LOOP, int SETS, bool TILES>
__global__ void __launch_bounds__(1024) k_map_brq_pass(Geom g, Buffers b, uint32_t flags, uint32_t qflags, uint32_t RB, uint32_t fixed_start_y, int use_anchor_start,
                                                       uint32_t nbands, uint32_t items) {
#ifdef SMH_MAP_FAT
	asm volatile("; register footprint experiment" ::: SMH_MAP_FAT);   // e.g. -DSMH_MAP_FAT='"v255"': the wave is allocated that many VGPRs
#endif
	static_assert(sizeof(MapKernelArgs) == sizeof(Geom) + sizeof(Buffers) + 7 * 4 + 4 || sizeof(MapKernelArgs) == sizeof(Geom) + sizeof(Buffers) + 7 * 4, "the kernel's parameter list");
	MapKernelArgsPtr ka = (MapKernelArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
	if (flags & MAP_PRIO) __builtin_amdgcn_s_setprio(3);        // ahead of the search service's waves on the same SIMD (which have slack)
	if (LOOP) {
		for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
			const uint32_t f = item / nbands, band = item - f * nbands;
			map_brq_item<GRAY, SETS, TILES>(ka, f, band);
			__syncthreads();                                   // the next item reuses the LDS exchange arrays
		}
	} else {
		// grid = (bands, frames); workgroups start in the order of their linear id.  MAP_BAND_MAJOR (a launch that runs alone): that order
		// takes the same band of consecutive frames -- rows 8 MB apart -- instead of consecutive bands of one frame: the launch is 2.5-11 %
		// shorter (one box, frame-major / band-major: 256 x 1080p 0.473 / 0.452 ms, 1024 x 1080p 1.645 / 1.469, 64 x 1080p 0.142 / 0.135,
		// 1600 x 900 0.337 / 0.328, 720p 0.234 / 0.225, 128 x 1440p 0.418 / 0.407), while every XCD taking a contiguous run of (frame, band)
		// items -- the halo rows of neighbouring bands in one L2 -- is 2-18 % LONGER (0.470 -> 0.559 ms in 24-row bands): what this access
		// pattern wants is its requests spread over the memory system, not reuse.  Beside the search service the frame-major order stays
		// (1080p level, 1440p 291 -> 281 k frames/s band-major: profiles/r06_sweep_band_major.txt).
		uint32_t f = blockIdx.y, band = blockIdx.x;
		if (flags & MAP_BAND_MAJOR) {
			const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
			band = id / gridDim.y; f = id - band * gridDim.y;
		}
		map_brq_item<GRAY, SETS, TILES>(ka, f, band);
	}
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
hipError_t launch_button(const Geom &g, const Buffers &b, uint32_t n, int force_open, hipStream_t s) {
	hipLaunchKernelGGL(k_button, dim3(n), dim3(256), 0, s, g, b, force_open);
	return hipGetLastError();
}

// the order of a streaming launch's work items: band-major where it runs alone (k_map_brq_pass), frame-major where it overlaps other kernels
// of its pipeline; smhv_debug_map_band_rows can force either
static uint32_t work_item_order(bool overlapped) {
	const uint32_t forced = g_map_band_major.load(std::memory_order_relaxed);   // diagnostic: 1 = band-major always, 2 = never
	return (forced == 1u || (forced == 0u && !overlapped)) ? (uint32_t)MAP_BAND_MAJOR : 0u;
}

// (the plain pass keeps its frame-major order: band-major measured level for it, 256 x 1080p alone 0.411 against 0.409 ms)
hipError_t launch_map_pass(const Geom &g, const Buffers &b, uint32_t n, uint32_t flags, int grayscale, hipStream_t s, bool tiles_wanted, bool overlapped) {
	// Few frames: shorter bands so a single frame still spreads over the chip.
	const uint32_t RB = band_rows_for(g.rh, n, MAP_RB_MAX, tiles_wanted && b.tiled != nullptr, overlapped);
	const dim3 grid((g.rh + RB - 1) / RB, n);
	const unsigned lds = (g.m_block / 64u) * 640u;             // 64 x (pixel, verdict, id) per wave
	const bool tiles = tiles_wanted && b.tiled != nullptr && (RB & 7u) == 0u;
	if (tiles) {
		if (grayscale) hipLaunchKernelGGL((k_map_pass<true, true>), grid, dim3(g.m_block), lds, s, g, b, flags, RB);
		else hipLaunchKernelGGL((k_map_pass<false, true>), grid, dim3(g.m_block), lds, s, g, b, flags, RB);
	} else {
		if (grayscale) hipLaunchKernelGGL((k_map_pass<true, false>), grid, dim3(g.m_block), lds, s, g, b, flags, RB);
		else hipLaunchKernelGGL((k_map_pass<false, false>), grid, dim3(g.m_block), lds, s, g, b, flags, RB);
	}
	return hipGetLastError();
}

static uint32_t map_brq_static_lds() {
	static const uint32_t v = [] { hipFuncAttributes a; return hipFuncGetAttributes(&a, (const void *)k_map_brq_pass<true, true, 2, false>) == hipSuccess ? (uint32_t)a.sharedSizeBytes : 2048u; }();
	return v;
}
uint32_t map_brq_lds_bytes(const Geom &g) { return map_brq_static_lds() + (g.m_block / 64u) * 640u; }

hipError_t launch_map_brq_pass(const Geom &g, const Buffers &b, uint32_t n, uint32_t flags, uint32_t qflags, int grayscale, uint32_t fixed_start_y, int use_anchor_start, hipStream_t s,
                               const LaunchTuning *tune) {
	// (the grid-stride form and the three-set form never write the tile-major mask -- see below -- and keep the 58-row bands they had)
	const uint32_t cap0 = tune ? tune->map_grid_cap : 0u;
	const uint32_t RB0 = band_rows_for(g.rh, n, MAPQ_RB_MAX, false);
	const bool no_tiles = b.tiled == nullptr || (tune && tune->map_deep) || (cap0 && cap0 < ((g.rh + RB0 - 1) / RB0) * n);
	const uint32_t RB = no_tiles ? RB0 : band_rows_for(g.rh, n, MAPQ_RB_MAX, true, tune && tune->map_overlapped);
	const uint32_t nbands = (g.rh + RB - 1) / RB, items = nbands * n;
	unsigned lds = (g.m_block / 64u) * 640u;                 // 64 x (pixel, verdict, id) per wave
	if (tune && tune->map_lds_total > map_brq_lds_bytes(g)) lds = tune->map_lds_total - map_brq_static_lds();
	if (lds > 65536u) {                                      // more than 64 KB of dynamic LDS has to be allowed per function (and per device)
		static std::atomic<uint64_t> attr_devices{0};
		int dev = 0;
		hipError_t e = hipGetDevice(&dev);
		if (e != hipSuccess) return e;
		if (dev >= 64 || !((attr_devices.load(std::memory_order_acquire) >> dev) & 1ull)) {
			const void *fns[] = {(const void *)k_map_brq_pass<true, true, 2, false>, (const void *)k_map_brq_pass<false, true, 2, false>, (const void *)k_map_brq_pass<true, false, 2, false>,
			                     (const void *)k_map_brq_pass<false, false, 2, false>, (const void *)k_map_brq_pass<true, false, 2, true>, (const void *)k_map_brq_pass<false, false, 2, true>,
			                     (const void *)k_map_brq_pass<true, false, 3, false>, (const void *)k_map_brq_pass<false, false, 3, false>};
			for (const void *fn : fns) if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
			if (e != hipSuccess) return e;
			if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_release);
		}
	}
	if (tune && tune->map_prio) flags |= MAP_PRIO;
	flags |= work_item_order(tune && tune->map_overlapped);
	const uint32_t cap = tune ? tune->map_grid_cap : 0u;
	// Loads in flight per wave: two register sets of four rows (one group ahead: 107 registers, three workgroups per CU beside a
	// search-service workgroup), or three (two groups ahead: 123 registers, two workgroups) where the launch asks for it
	// (LaunchTuning::map_deep: SMHV_PIPE_THREE_LOAD_SETS, round 5's choice for frame-granular pipelines up to 1080p; since round 6
	// two sets are ahead there too).  A launch alone takes the same time either way.
	const bool deep = tune && tune->map_deep && !(cap && cap < items);
	// The tile-major mask: only launches with one workgroup per item and two load sets, whose bands are whole tile rows
	// (band_rows_for) -- the frame-granular pipelines and plain runs up to 1080p; the grid-stride form (batch-granular pipelines:
	// k_lsd_tile walks the bit rows) and the three-set form (SMHV_PIPE_THREE_LOAD_SETS: round 5's streaming pass as it was) never.
	const bool tiles = !no_tiles && (RB & 7u) == 0u && !(cap && cap < items) && !deep;
#define SMH_LAUNCH_MAPQ(GRAYV, LOOPV, SETSV, TILESV, GRID) \
	hipLaunchKernelGGL((k_map_brq_pass<GRAYV, LOOPV, SETSV, TILESV>), GRID, dim3(g.m_block), lds, s, g, b, flags, qflags, RB, fixed_start_y, use_anchor_start, nbands, items)
	if (cap && cap < items) {
		if (grayscale) SMH_LAUNCH_MAPQ(true, true, 2, false, dim3(cap)); else SMH_LAUNCH_MAPQ(false, true, 2, false, dim3(cap));
	} else if (deep) {
		if (grayscale) SMH_LAUNCH_MAPQ(true, false, 3, false, dim3(nbands, n)); else SMH_LAUNCH_MAPQ(false, false, 3, false, dim3(nbands, n));
	} else if (tiles) {
		if (grayscale) SMH_LAUNCH_MAPQ(true, false, 2, true, dim3(nbands, n)); else SMH_LAUNCH_MAPQ(false, false, 2, true, dim3(nbands, n));
	} else {
		if (grayscale) SMH_LAUNCH_MAPQ(true, false, 2, false, dim3(nbands, n)); else SMH_LAUNCH_MAPQ(false, false, 2, false, dim3(nbands, n));
	}
#undef SMH_LAUNCH_MAPQ
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_pattern_copy: the streaming pass's memory traffic without its arithmetic -- the calibration of what the memory system
// gives THIS access pattern (smhv_debug_pattern_copy; bench.py: roofline_isolated.pattern_copy_GBps).  Same decomposition as
// k_map_brq_pass (one workgroup per (frame, band of MAPQ_RB_MAX rows), one thread per quad across the ROI width): every ROI
// quad is read once with a 16-byte load out of the 1920-px-pitch frame (986 of 1920 pixels of a row at 1080p), 16 bytes go to
// the ui_map row, 4 to the mask row, and inside the bottom-right quadrant 4 each to the ocr and scales rows: per 256 1080p
// frames 0.83 GB read and 1.14 GB written (the pass reads 0.95 GB: two halo rows per band on top).  What is stored is a
// cheap function of what was loaded, so that no load can be dropped.  NR = rows a thread has in flight (the pass: three sets of
// four = 12); the bench line quotes the best of 4 / 8 / 12.
// ------------------------------------------------------------------------------------------------
template <int NR>
__global__ void __launch_bounds__(1024) k_pattern_copy(Geom g, Buffers b, uint32_t RB, uint32_t band_major) {
	uint32_t f = blockIdx.y, band = blockIdx.x;                // (the pass's work-item order: k_map_brq_pass)
	if (band_major) {
		const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
		band = id / gridDim.y; f = id - band * gridDim.y;
	}
	const uint32_t q = threadIdx.x;
	const int r0 = (int)(band * RB), r1 = min(r0 + (int)RB, (int)g.rh);
	if (q >= g.m_quads) return;
	const uint8_t *fp = b.frames + (size_t)f * g.frame_bytes + ((size_t)g.ry * g.W + g.m_ax) * 4 + (size_t)q * 16;
	uint8_t *up = b.ui + (size_t)f * g.ui_stride + (size_t)q * 16;
	uint8_t *mp = b.mask + (size_t)f * g.mask_stride + (size_t)q * 4;
	const int qy0 = (int)(g.qy - g.ry);
	const int qq = (int)q - (int)((g.q_ax - g.m_ax) >> 2);
	const bool in_q = qq >= 0 && (uint32_t)qq < g.q_quads;
	uint8_t *op = b.ocr + (size_t)f * g.ocr_stride + (size_t)(in_q ? qq : 0) * 4;
	uint8_t *sp = b.scales + (size_t)f * g.ocr_stride + (size_t)(in_q ? qq : 0) * 4;
	const size_t row_bytes = (size_t)g.W * 4;
	for (int r = r0; r < r1; r += NR) {
		uint4 v[NR];
#pragma unroll
		for (int k = 0; k < NR; ++k) v[k] = *(const uint4 *)(fp + (size_t)min(r + k, r1 - 1) * row_bytes);
#pragma unroll
		for (int k = 0; k < NR; ++k) {
			const int row = r + k;
			if (row >= r1) break;
			*(uint4 *)(up + (size_t)row * g.ui_pitch) = v[k];
			const uint32_t m = v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
			*(uint32_t *)(mp + (size_t)row * g.mask_pitch) = m;
			const int qrow = row - qy0;
			if (in_q && qrow >= 0 && qrow < (int)g.qh) {
				*(uint32_t *)(op + (size_t)qrow * g.ocr_pitch) = ~m;
				*(uint32_t *)(sp + (size_t)qrow * g.ocr_pitch) = m + 1u;
			}
		}
	}
}

hipError_t launch_pattern_copy(const Geom &g, const Buffers &b, uint32_t n, uint32_t rows_in_flight, hipStream_t s) {
	const uint32_t RB = band_rows_for(g.rh, n, MAPQ_RB_MAX);   // (the pass's own band height for this launch)
	const dim3 grid((g.rh + RB - 1) / RB, n), block(g.m_block);
	const uint32_t bm = g_map_band_major.load(std::memory_order_relaxed) == 2u ? 0u : 1u;   // (the order of a pass that runs alone)
	switch (rows_in_flight) {
	case 0: case 4: hipLaunchKernelGGL(k_pattern_copy<4>, grid, block, 0, s, g, b, RB, bm); break;
	case 8: hipLaunchKernelGGL(k_pattern_copy<8>, grid, block, 0, s, g, b, RB, bm); break;
	case 12: hipLaunchKernelGGL(k_pattern_copy<12>, grid, block, 0, s, g, b, RB, bm); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// (host logic, for the tests: rows per band the fused / the plain pass takes for a launch over n frames of an ROI rh rows tall, and
// whether such a launch writes the tile-major mask)
void map_band_rows(uint32_t rh, uint32_t n, int fused, uint32_t *rows, int *tiles) {
	const uint32_t RB = band_rows_for(rh, n, fused ? MAPQ_RB_MAX : MAP_RB_MAX, true, fused == 2);   // (2: the fused pass of a frame-granular pipeline)
	if (rows) *rows = RB;
	if (tiles) *tiles = (RB & 7u) == 0u ? 1 : 0;
}

hipError_t launch_brq_pass(const Geom &g, const Buffers &b, uint32_t n, uint32_t flags, uint32_t fixed_start_y, int use_anchor_start, hipStream_t s) {
	const dim3 grid((g.qh + BRQ_RB - 1) / BRQ_RB, n);
	hipLaunchKernelGGL(k_brq_pass, grid, dim3(g.q_block), 0, s, g, b, flags, fixed_start_y, use_anchor_start);
	return hipGetLastError();
}

}  // namespace smh
