// smh_proximity.h -- the proximity filter of lsd::find_lines (vision-common/src/lsd.rs:47-58,84-89): a white pixel is
// skipped when it lies within sqrt(50) px of the INFINITE line through an accepted segment.  Shared by the device code
// (smh_lsd.hip) and by a host test program (tests/proximity_check.cpp compiles this header with g++), hence SMH_HD.
#pragma once
#include <math.h>
#include <stdint.h>

#include "smh_consts.h"

#if defined(__HIPCC__)
#define SMH_HD __host__ __device__ __forceinline__
#else
#define SMH_HD static inline
#endif

namespace smh {

// lsd.rs:47-58 + the `< 50.0` test of lsd.rs:84-89, in the reference's f32 operation order (u is not clamped).
SMH_HD bool near_line(float x, float y, float x0, float y0, float x1, float y1) {
	const float dx = x1 - x0, dy = y1 - y0;
	float nx = x0, ny = y0;
	if (!(dx == 0.0f && dy == 0.0f)) {
		const float u = ((x - x0) * dx + (y - y0) * dy) / (dx * dx + dy * dy);
		nx = x0 + u * dx; ny = y0 + u * dy;
	}
	const float ex = x - nx, ey = y - ny;
	return ex * ex + ey * ey < SMH_LSD_PROXIMITY_SQ;
}

// Cheap classifier in front of near_line for the pixels (px0 + bit, py), bit = 0..31, of one mask word.
// The signed distance of a pixel from the infinite line is linear in x:  s(bit) = s0 + a * bit.  The cross product
// behind s0 is formed in f64 (exact for f32 inputs of this size), so |s - true distance| stays below 1e-3 for segments
// longer than 1 px, while near_line's own f32 evaluation of the squared distance is within 0.02 of the true value for
// coordinates below 4096 (|ex|, |ey| <= 8 with absolute errors of a few 1e-4).  sqrt(50) = 7.0711:
//   |s| <  SMH_PROX_SURE_NEAR (6.9)   => near_line is true   (distance^2 <= 47.7)
//   |s| >  SMH_PROX_SURE_FAR  (7.25)  => near_line is false  (distance^2 >= 52.5)
// and only the pixels in the 0.35 px wide ring between them take the exact test.  tests/test_proximity.py checks both
// implications by brute force (random and adversarial segments, 4K coordinate range).
#define SMH_PROX_SURE_NEAR 6.9f
#define SMH_PROX_SURE_FAR 7.25f

struct ProxLine {
	float x0, y0, x1, y1;
	float a;          // d s / d x  = dy / len
	double dxl, dyl;  // dx / len, dy / len
	bool degenerate;  // zero-length segment: distance to the point (x0, y0); every pixel takes the exact test
};

SMH_HD ProxLine prox_line(float x0, float y0, float x1, float y1) {
	ProxLine L;
	L.x0 = x0; L.y0 = y0; L.x1 = x1; L.y1 = y1;
	const double dx = (double)x1 - (double)x0, dy = (double)y1 - (double)y0;
	const double len2 = dx * dx + dy * dy;
	L.degenerate = !(len2 >= 1.0);
	const double inv = L.degenerate ? 0.0 : 1.0 / sqrt(len2);
	L.dxl = dx * inv; L.dyl = dy * inv;
	L.a = (float)L.dyl;
	return L;
}

// Clears from `surv` (bit b = pixel (px0 + b, py)) every pixel near_line puts within sqrt(50) of the line.
SMH_HD uint32_t prox_filter_word(uint32_t surv, float px0, float py, const ProxLine &L) {
	if (L.degenerate) {
		uint32_t s = surv;
		while (s) {
			const uint32_t bit = (uint32_t)__builtin_ctz(s);
			s &= s - 1u;
			if (near_line(px0 + (float)bit, py, L.x0, L.y0, L.x1, L.y1)) surv &= ~(1u << bit);
		}
		return surv;
	}
	const float s0 = (float)(((double)px0 - (double)L.x0) * L.dyl - ((double)py - (double)L.y0) * L.dxl);
	// the whole word on one side, far away: |s| is smallest at one of its two ends when it has no zero inside
	const float s31 = s0 + 31.0f * L.a;
	if ((s0 > SMH_PROX_SURE_FAR + 0.01f && s31 > SMH_PROX_SURE_FAR + 0.01f) || (s0 < -SMH_PROX_SURE_FAR - 0.01f && s31 < -SMH_PROX_SURE_FAR - 0.01f)) return surv;
	uint32_t s = surv;
	while (s) {
		const uint32_t bit = (uint32_t)__builtin_ctz(s);
		s &= s - 1u;
		const float d = fabsf(s0 + (float)bit * L.a);
		if (d > SMH_PROX_SURE_FAR) continue;
		if (d < SMH_PROX_SURE_NEAR || near_line(px0 + (float)bit, py, L.x0, L.y0, L.x1, L.y1)) surv &= ~(1u << bit);
	}
	return surv;
}

}  // namespace smh
