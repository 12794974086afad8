// smh_misc.hip -- the small kernels around the hot path (gfx950, wave64).
//   k_scale_ratio  calc_meters_to_px_ratio / find_scale_width                 (src/vision/mpx_ratio.rs:3-134)
//   k_find_minimap find_minimap (the caller's next step)                      (src/vision/find_minimap.rs)
//   k_finalize     derived marker outputs                                     (src/ui/mod.rs:131-140, markers.rs:98)
//   k_debug_view   DebugView images                                           (vision-cpu/src/lib.rs:451-460)
//   k_marker_table exhaustive colour-predicate table (test support)
//   k_crc32        CRC-32 of a frame in HBM for the capture hand-off          (src/capture.rs:44-47)
//
// Build with -ffp-contract=off and correctly rounded f32 division: several results are truncated to integers
// right at a threshold, so the reference's scalar f32 operation order (no FMA contraction, IEEE divide) is
// part of the contract.  Semantics follow the reference's CPU back-end (vision-cpu/src/lib.rs) bit for bit;
// structure does not follow its CUDA file at all (SURVEY.md Appendix A lists how that differs).
#include "smh_device.h"
#include "smh_record.inc"

namespace smh {

__global__ void __launch_bounds__(64 * SMHV_MAX_SCALES) k_scale_ratio(Geom g, Buffers b, uint32_t *bars) { scale_ratio_body(g, b, blockIdx.x, bars); }

// ------------------------------------------------------------------------------------------------
// k_find_minimap: src/vision/find_minimap.rs (the caller's step right after crop_to_map).
// One wave per direction (Left, Right, Up, Down), four waves per frame.  The reference walks pixel by
// pixel from the ROI centre and, at every pixel whose "edginess" is <= 0.01, tries a perpendicular run of
// min_line_length equally flat pixels.  Here 64 steps of the main walk are tested at once (ballot, handled
// in walk order) and the perpendicular run is tested 64 pixels per step; the result is the reference's.
// edginess = max over the 8 neighbours of |dB|+|dG|+|dR|, as f32 / 765.0 <= 0.01  <=>  that max <= 7
// (7/765 = 0.00915, 8/765 = 0.01046).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool flat_pixel(const uint8_t *roi0, uint32_t W, uint32_t x, uint32_t y) {
	const uint32_t *p = (const uint32_t *)(roi0 + ((size_t)y * W + x) * 4);
	const int Wi = (int)W;
	const uint32_t c = p[0] & 0x00FFFFFFu;
	uint32_t mx = 0;
#pragma unroll
	for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
		for (int dx = -1; dx <= 1; ++dx)
			if (dx != 0 || dy != 0) mx = max(mx, (uint32_t)__builtin_amdgcn_sad_u8(c, p[dy * Wi + dx] & 0x00FFFFFFu, 0u));
	return (float)mx / 765.0f <= 0.01f;
}

__device__ uint32_t find_edge(const uint8_t *roi0, uint32_t W, uint32_t w, uint32_t h, uint32_t x0, uint32_t y0, int dir) {
	const uint32_t lane = threadIdx.x & 63u;
	const bool vertical = dir < 2;                         // Up, Down move y; Left, Right move x
	uint32_t c_max = vertical ? h : w, oc_max = vertical ? w : h;
	const int cod = (dir == 0 || dir == 2) ? -1 : 1;
	const uint32_t c0 = vertical ? y0 : x0, oc0 = vertical ? x0 : y0;
	const uint32_t d = oc_max > oc0 ? oc_max - oc0 : oc0 - oc_max;
	const uint32_t mll = d / 2u - 1u;                      // min_line_length (wraps like release Rust; >= 0 for dims >= 3)
	c_max -= 3u; oc_max -= 3u;
	uint32_t cbase = c0;
	for (;;) {
		const uint32_t cc = (uint32_t)((int32_t)cbase + cod * (int32_t)(lane + 1u));
		const int st = cc > c_max ? 1 : (cc < 3u ? 2 : 0);     // order of the reference's two tests
		bool low = false;
		if (st == 0) low = flat_pixel(roi0, W, vertical ? oc0 : cc, vertical ? cc : oc0);
		const uint64_t term = __ballot(st != 0);
		uint64_t lows = __ballot(low);
		for (;;) {
			const uint64_t both = term | lows;
			if (!both) break;
			const uint32_t first = (uint32_t)__builtin_ctzll(both);
			const uint32_t fc = (uint32_t)((int32_t)cbase + cod * (int32_t)(first + 1u));
			if ((term >> first) & 1ull) return fc > c_max ? c_max + 2u : 0u;
			// a flat pixel: "try and find a straight line of pixels that are also under the edginess threshold"
			bool ok = true;
			for (uint32_t kb = 0; kb < mll && ok; kb += 64u) {
				const uint32_t k = kb + lane + 1u;
				bool good = true;
				if (k <= mll) {
					const uint32_t oc = (uint32_t)((int32_t)oc0 - cod * (int32_t)k);
					good = !(oc < 3u || oc > oc_max) && flat_pixel(roi0, W, vertical ? oc : fc, vertical ? fc : oc);
				}
				ok = __all(good);
			}
			if (ok) return (uint32_t)((int32_t)fc - cod);
			lows &= ~(1ull << first);
		}
		cbase = (uint32_t)((int32_t)cbase + cod * 64);
	}
}

__global__ void __launch_bounds__(256) k_find_minimap(Geom g, Buffers b) {
	const uint32_t f = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	smhv_frame_result *res = &b.results[f];
	if (!b.aux[f].open) { if (threadIdx.x < 4) res->minimap[threadIdx.x] = 0; if (threadIdx.x == 0) res->has_minimap = 0; return; }
	const uint8_t *roi0 = b.frames + (size_t)f * g.frame_bytes + ((size_t)g.ry * g.W + g.rx) * 4;
	// rect = {left, right, top, bottom}; reference direction order: Left, Right, Up, Down
	const int dir = wave == 0 ? 2 : (wave == 1 ? 3 : (wave == 2 ? 0 : 1));
	const uint32_t v = find_edge(roi0, g.W, g.rw, g.rh, g.rw / 2u, g.rh / 2u, dir);
	if (lane == 0) res->minimap[wave] = v;
	if (threadIdx.x == 0) res->has_minimap = 1;
}

__global__ void __launch_bounds__(64) k_finalize(Geom g, Buffers b, uint32_t stages) { finalize_body(g, b, blockIdx.x, stages); }

// calc_meters_to_px_ratio + the record in one launch (the batched pipeline: one stream, no branch to join)
__global__ void __launch_bounds__(64 * SMHV_MAX_SCALES) k_scales_finalize(Geom g, Buffers b, uint32_t stages, uint32_t *bars) {
	scale_ratio_body(g, b, blockIdx.x, bars);
	__syncthreads();                                       // thread 0's has_mpx / mpx are visible to the block
	if (threadIdx.x < 64) finalize_body(g, b, blockIdx.x, stages);
}

// ------------------------------------------------------------------------------------------------
// debug views (vision-cpu/src/lib.rs:451-460) and the exhaustive colour table
// ------------------------------------------------------------------------------------------------
__global__ void k_debug_view(Geom g, Buffers b, uint32_t f, int which, int isolated, uint8_t *out) {
	const bool brq = which == SMHV_VIEW_OCR_INPUT || which == SMHV_VIEW_FIND_SCALES_INPUT || which == SMHV_VIEW_CROPPED_BRQ;
	const uint32_t w = brq ? g.qw : g.rw, h = brq ? g.qh : g.rh;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= w * h) return;
	const uint32_t y = i / w, x = i - y * w;
	const uint8_t *fp = b.frames + (size_t)f * g.frame_bytes;
	uint32_t o;
	if (which == SMHV_VIEW_OCR_INPUT || which == SMHV_VIEW_FIND_SCALES_INPUT) {
		const uint8_t *img = (which == SMHV_VIEW_OCR_INPUT ? b.ocr : b.scales) + (size_t)f * g.ocr_stride + g.q_xoff;
		o = (uint32_t)img[(size_t)y * g.ocr_pitch + x] * 0x00010101u | 0xFF000000u;
	} else if (which == SMHV_VIEW_LSD_INPUT) {
		o = (uint32_t)b.mask[(size_t)f * g.mask_stride + (size_t)y * g.mask_pitch + g.m_xoff + x] * 0x00010101u | 0xFF000000u;
	} else {
		const uint32_t fx = brq ? g.qx + x : g.rx + x, fy = brq ? g.qy + y : g.ry + y;
		const uint32_t p = *(const uint32_t *)(fp + ((size_t)fy * g.W + fx) * 4);
		uint32_t bb = p & 255u, gg = (p >> 8) & 255u, rr = (p >> 16) & 255u;
		if (which == SMHV_VIEW_LSD_PREPROCESS && isolated && !is_marker(rr, gg, bb)) { rr = 0; gg = 0; bb = 0; }
		o = rr | (gg << 8) | (bb << 16) | 0xFF000000u;
	}
	((uint32_t *)out)[i] = o;
}

__global__ void k_marker_table(uint32_t *bits) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= (1u << 24) / 32u) return;
	uint32_t acc = 0;
	for (uint32_t k = 0; k < 32; ++k) {
		const uint32_t c = w * 32u + k;
		if (is_marker((c >> 16) & 255u, (c >> 8) & 255u, c & 255u)) acc |= 1u << k;
	}
	bits[w] = acc;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
// Co-residency probe (smhv_debug_side_kernel): a kernel with the footprint of a collective's kernel -- RCCL's rcclGenericKernel on
// gfx950 takes 19.7-21.2 KB of LDS and 261-280 VGPRs per 256-thread workgroup -- that does next to nothing.  What the tests and
// bench.py measure with it is whether such a kernel gets onto the chip at all beside a saturated pipeline, and how soon.
__global__ void __launch_bounds__(256) k_side_probe(uint32_t *out, uint32_t spin) {
	__shared__ uint32_t lds[21u * 256u];                        // 21 KB
	asm volatile("; the register footprint of the kernel this one stands in for" ::: "v255", "a23");   // 256 + 24 of the unified file
	for (uint32_t i = threadIdx.x; i < 21u * 256u; i += blockDim.x) lds[i] = i ^ blockIdx.x;
	__syncthreads();
	uint32_t acc = lds[(threadIdx.x * 37u) % (21u * 256u)];
	for (uint32_t i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;
	if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
hipError_t launch_side_probe(uint32_t *d_out, uint32_t workgroups, uint32_t spin, hipStream_t s) {
	hipLaunchKernelGGL(k_side_probe, dim3(workgroups), dim3(256), 0, s, d_out, spin);
	return hipGetLastError();
}

hipError_t launch_scale_ratio(const Geom &g, const Buffers &b, uint32_t n, uint32_t *d_bars, hipStream_t s) {
	hipLaunchKernelGGL(k_scale_ratio, dim3(n), dim3(64 * SMHV_MAX_SCALES), 0, s, g, b, d_bars);
	return hipGetLastError();
}

hipError_t launch_find_minimap(const Geom &g, const Buffers &b, uint32_t n, hipStream_t s) {
	hipLaunchKernelGGL(k_find_minimap, dim3(n), dim3(256), 0, s, g, b);
	return hipGetLastError();
}

hipError_t launch_finalize(const Geom &g, const Buffers &b, uint32_t n, uint32_t stages, hipStream_t s) {
	hipLaunchKernelGGL(k_finalize, dim3(n), dim3(64), 0, s, g, b, stages);
	return hipGetLastError();
}

hipError_t launch_scales_finalize(const Geom &g, const Buffers &b, uint32_t n, uint32_t stages, uint32_t *d_bars, hipStream_t s) {
	hipLaunchKernelGGL(k_scales_finalize, dim3(n), dim3(64 * SMHV_MAX_SCALES), 0, s, g, b, stages, d_bars);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_pack_rows: `rows` rows of `row_w` dwords each, `src_pitch_w` dwords apart, into a tight buffer -- the per-call path's ui_map on
// its way to the host: a tight image leaves the device as ONE contiguous copy per row block (hipMemcpyAsync), which the runtime
// always runs at the link's rate; the pitched copy it replaces (hipMemcpy2DAsync) took 0.2 ms for 3.2 MB in a fresh process and 0.6 ms
// in one that had created and destroyed a pipeline (tools/trait_before_after_r06.py).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_rows(const uint32_t *__restrict__ src, uint32_t src_pitch_w, uint32_t *__restrict__ dst, uint32_t row_w, uint32_t rows) {
	const uint32_t r = blockIdx.y;
	const uint32_t *s = src + (size_t)r * src_pitch_w;
	uint32_t *d = dst + (size_t)r * row_w;
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < row_w; x += gridDim.x * blockDim.x) d[x] = s[x];
}
hipError_t launch_pack_rows(const void *d_src, uint32_t src_pitch_bytes, void *d_dst, uint32_t row_bytes, uint32_t rows, hipStream_t s) {
	if ((src_pitch_bytes | row_bytes) & 3u) return hipErrorInvalidValue;
	const uint32_t row_w = row_bytes / 4u;
	hipLaunchKernelGGL(k_pack_rows, dim3((row_w + 255u) / 256u, rows), dim3(256), 0, s, (const uint32_t *)d_src, src_pitch_bytes / 4u, (uint32_t *)d_dst, row_w, rows);
	return hipGetLastError();
}

hipError_t launch_debug_view(const Geom &g, const Buffers &b, uint32_t frame, int which, int isolated, uint8_t *d_rgba, hipStream_t s) {
	const bool brq = which == SMHV_VIEW_OCR_INPUT || which == SMHV_VIEW_FIND_SCALES_INPUT || which == SMHV_VIEW_CROPPED_BRQ;
	const uint32_t npx = brq ? g.qw * g.qh : g.rw * g.rh;
	hipLaunchKernelGGL(k_debug_view, dim3((npx + 255) / 256), dim3(256), 0, s, g, b, frame, which, isolated, d_rgba);
	return hipGetLastError();
}

hipError_t launch_marker_table(uint32_t *d_bits, hipStream_t s) {
	hipLaunchKernelGGL(k_marker_table, dim3(((1u << 24) / 32u + 255) / 256), dim3(256), 0, s, d_bits);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_crc32: CRC-32 (IEEE 802.3, reflected polynomial 0xEDB88320) of a frame in HBM -- the value the
// reference's capture thread computes with crc32fast::hash to drop duplicate captures
// (src/capture.rs:44-47).  CRC without its init / final xor is linear over GF(2):
//     R(A || B) = R(A) * x^(8|B|) mod P  xor  R(B)
// so the frame is cut into 16-byte groups dealt round-robin to every thread of the grid; a thread
// folds its groups with the usual slice-by-4 table step (tables in LDS) and a multiplication by
// x^(128 (G - 1)) between rounds (G = threads in the grid), then aligns its remainder to the end of
// the message with one multiplication by x^(128 (G - 1 - T)) and all remainders are xor-ed together
// (DPP within the wave, LDS across waves, one atomicXor per workgroup).  Leading zero padding does
// not change R, so the message is right-aligned in the last round; the init / final-xor terms
// depend on the length only and are applied by the host (smh_runtime.cpp: crc32_finish).
// HBM-bound: 1 byte read per byte; ~30 VALU + 16 LDS lookups per 16 bytes.
// ------------------------------------------------------------------------------------------------
#define CRC_POLY 0xEDB88320u
#define CRC_BS 1024

// a * b mod P in the reflected representation (x^0 = 0x80000000)
__host__ __device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
	uint32_t p = 0;
	for (int i = 0; i < 32; ++i) {
		p ^= (a & 0x80000000u) ? b : 0u;
		a <<= 1;
		b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
	}
	return p;
}

// n_dwords: message length in 32-bit words; rounds * gridDim.x * CRC_BS * 4 >= n_dwords.
// x_skip = x^(128 (G - 1)); x_local[t] = x^(128 (CRC_BS - 1 - t)); x_wg[g] = x^(128 CRC_BS (gridDim.x - 1 - g)).
__global__ void __launch_bounds__(CRC_BS) k_crc32(const uint32_t *msg, uint64_t n_dwords, uint32_t rounds, uint32_t x_skip,
                                                 const uint32_t *x_local, const uint32_t *x_wg, uint32_t *acc) {
	__shared__ uint32_t tab[4][256];
	__shared__ uint32_t wsum[CRC_BS / 64];
	const uint32_t tid = threadIdx.x;
	if (tid < 256u) {
		uint32_t c = tid;
		for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
		tab[0][tid] = c;
	}
	__syncthreads();
	if (tid < 256u) {
		uint32_t c = tab[0][tid];
		for (int k = 1; k < 4; ++k) { c = (c >> 8) ^ tab[0][c & 255u]; tab[k][tid] = c; }
	}
	__syncthreads();
	const uint64_t G = (uint64_t)gridDim.x * CRC_BS, T = (uint64_t)blockIdx.x * CRC_BS + tid;
	const uint64_t pad = (uint64_t)rounds * G * 4u - n_dwords;        // virtual leading zero words
	uint32_t v = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		const uint64_t vi = ((uint64_t)r * G + T) * 4u;                 // virtual index of this thread's group
		uint32_t d[4] = {0u, 0u, 0u, 0u};
		if (vi >= pad && ((vi - pad) & 3u) == 0u && (((uintptr_t)msg) & 15u) == 0u) {
			const uint4 q = *(const uint4 *)(msg + (vi - pad));
			d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
		} else {
#pragma unroll
			for (int j = 0; j < 4; ++j) if (vi + j >= pad) d[j] = msg[vi + j - pad];
		}
		if (r) v = gf2_mulmod(v, x_skip);
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t c = v ^ d[j];
			v = tab[3][c & 255u] ^ tab[2][(c >> 8) & 255u] ^ tab[1][(c >> 16) & 255u] ^ tab[0][c >> 24];
		}
	}
	v = gf2_mulmod(v, x_local[tid]);
	v = wave_xor32_dpp(v);
	if ((tid & 63u) == 0u) wsum[tid >> 6] = v;
	__syncthreads();
	if (tid == 0) {
		uint32_t w = 0;
		for (int k = 0; k < CRC_BS / 64; ++k) w ^= wsum[k];
		atomicXor(acc, gf2_mulmod(w, x_wg[blockIdx.x]));
	}
}

uint32_t crc32_xpow(uint64_t n) {                             // x^n mod P
	uint32_t r = 0x80000000u, b = 0x40000000u;                // x^0, x^1
	for (; n; n >>= 1) { if (n & 1u) r = gf2_mulmod(r, b); b = gf2_mulmod(b, b); }
	return r;
}
uint32_t crc32_mul(uint32_t a, uint32_t b) { return gf2_mulmod(a, b); }

hipError_t launch_crc32(const void *d_msg, uint64_t n_dwords, uint32_t wgs, uint32_t rounds, uint32_t x_skip, const uint32_t *d_x_local,
                        const uint32_t *d_x_wg, uint32_t *d_acc, hipStream_t s) {
	hipLaunchKernelGGL(k_crc32, dim3(wgs), dim3(CRC_BS), 0, s, (const uint32_t *)d_msg, n_dwords, rounds, x_skip, d_x_local, d_x_wg, d_acc);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_to_bgra: a decoder's 8-bit pixels -> BGRA8 (image 0.23 `DynamicImage::into_bgra8`, src/ui/debug.rs:169).  Streaming: one
// thread converts four pixels (4 / 8 / 12 / 16 source bytes as dword loads, one 16-byte store); bytes in, bytes out, no
// arithmetic but byte permutes.  BPP = source bytes per pixel: 1 L, 2 LA, 3 RGB, 4 RGBA.
// ------------------------------------------------------------------------------------------------
template <int BPP>
__global__ void __launch_bounds__(256) k_to_bgra(const uint32_t *__restrict__ src, uint4 *__restrict__ dst, uint64_t n_groups, uint64_t n_px) {
	for (uint64_t gidx = (uint64_t)blockIdx.x * 256u + threadIdx.x; gidx < n_groups; gidx += (uint64_t)gridDim.x * 256u) {
		uint32_t w[4] = {0u, 0u, 0u, 0u}, px[4];
		const uint64_t n_src_dwords = (n_px * BPP + 3u) / 4u;
#pragma unroll
		for (int k = 0; k < BPP; ++k) { const uint64_t i = gidx * BPP + k; if (i < n_src_dwords) w[k] = src[i]; }
		if (BPP == 4) {                                          // r g b a -> b g r a
#pragma unroll
			for (int k = 0; k < 4; ++k) px[k] = __builtin_amdgcn_perm(w[k], w[k], 0x03000102u);
		} else if (BPP == 3) {                                     // 12 bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
			const uint32_t ff = 0xFFFFFFFFu;
			px[0] = __builtin_amdgcn_perm(ff, w[0], 0x04000102u);                                  // b0 g0 r0 ff
			px[1] = __builtin_amdgcn_perm(ff, __builtin_amdgcn_perm(w[1], w[0], 0x00050403u), 0x04000102u);   // r1 g1 b1 . -> b1 g1 r1 ff
			px[2] = __builtin_amdgcn_perm(ff, __builtin_amdgcn_perm(w[2], w[1], 0x00040302u), 0x04000102u);   // r2 g2 b2 .
			px[3] = __builtin_amdgcn_perm(ff, w[2], 0x04010203u);                                  // b3 g3 r3 ff
		} else if (BPP == 2) {                                     // l0 a0 l1 a1 | l2 a2 l3 a3
			px[0] = __builtin_amdgcn_perm(w[0], w[0], 0x01000000u);
			px[1] = __builtin_amdgcn_perm(w[0], w[0], 0x03020202u);
			px[2] = __builtin_amdgcn_perm(w[1], w[1], 0x01000000u);
			px[3] = __builtin_amdgcn_perm(w[1], w[1], 0x03020202u);
		} else {                                                   // l0 l1 l2 l3
			const uint32_t ff = 0xFFFFFFFFu;
			px[0] = __builtin_amdgcn_perm(ff, w[0], 0x04000000u);
			px[1] = __builtin_amdgcn_perm(ff, w[0], 0x04010101u);
			px[2] = __builtin_amdgcn_perm(ff, w[0], 0x04020202u);
			px[3] = __builtin_amdgcn_perm(ff, w[0], 0x04030303u);
		}
		const uint64_t p0 = gidx * 4u;
		if (p0 + 4u <= n_px) dst[gidx] = make_uint4(px[0], px[1], px[2], px[3]);
		else for (int k = 0; k < 4; ++k) if (p0 + k < n_px) ((uint32_t *)dst)[p0 + k] = px[k];
	}
}

hipError_t launch_to_bgra(const void *d_src, void *d_bgra, uint64_t n_px, uint32_t layout, hipStream_t s) {
	const uint64_t groups = (n_px + 3u) / 4u;
	const uint32_t wgs = (uint32_t)std::min<uint64_t>((groups + 255u) / 256u, 2048u);
	const uint32_t *src = (const uint32_t *)d_src;
	uint4 *dst = (uint4 *)d_bgra;
	switch (layout) {
	case 1u: hipLaunchKernelGGL(k_to_bgra<4>, dim3(wgs), dim3(256), 0, s, src, dst, groups, n_px); break;
	case 2u: hipLaunchKernelGGL(k_to_bgra<3>, dim3(wgs), dim3(256), 0, s, src, dst, groups, n_px); break;
	case 3u: hipLaunchKernelGGL(k_to_bgra<1>, dim3(wgs), dim3(256), 0, s, src, dst, groups, n_px); break;
	case 4u: hipLaunchKernelGGL(k_to_bgra<2>, dim3(wgs), dim3(256), 0, s, src, dst, groups, n_px); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_unpack_rows: the ingest queue's region-of-interest upload (smh_runtime.cpp): the packed rows of the map ROI followed by
// the packed rows of the button rectangle go to their places in a frame of the slab.  One workgroup per row, dword copies
// (the button's x is any pixel), 3.2 MB per 1080p frame.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unpack_rows(const uint32_t *__restrict__ pack, uint32_t *__restrict__ frame, uint32_t pitch_px, uint32_t roi_x, uint32_t roi_y,
                                                     uint32_t roi_w, uint32_t roi_h, uint32_t btn_x, uint32_t btn_y, uint32_t btn_w) {
	const uint32_t r = blockIdx.x;
	const bool roi = r < roi_h;
	const uint32_t w = roi ? roi_w : btn_w;
	const uint32_t *src = roi ? pack + (size_t)r * roi_w : pack + (size_t)roi_h * roi_w + (size_t)(r - roi_h) * btn_w;
	uint32_t *dst = roi ? frame + (size_t)(roi_y + r) * pitch_px + roi_x : frame + (size_t)(btn_y + r - roi_h) * pitch_px + btn_x;
	for (uint32_t i = threadIdx.x; i < w; i += 256u) dst[i] = src[i];
}

hipError_t launch_unpack_rows(const void *d_pack, void *d_frame, uint32_t pitch_px, uint32_t roi_x, uint32_t roi_y, uint32_t roi_w, uint32_t roi_h, uint32_t btn_x,
                              uint32_t btn_y, uint32_t btn_w, uint32_t btn_h, hipStream_t s) {
	hipLaunchKernelGGL(k_unpack_rows, dim3(roi_h + btn_h), dim3(256), 0, s, (const uint32_t *)d_pack, (uint32_t *)d_frame, pitch_px, roi_x, roi_y, roi_w, roi_h, btn_x, btn_y,
	                   btn_w);
	return hipGetLastError();
}

}  // namespace smh
