// smh_device.h -- device helpers shared by the kernel translation units (smh_stream.hip, smh_lsd.hip,
// smh_misc.hip): Rust cast emulation, image-0.23.14 luma, the exact marker colour predicate and its integer
// pre-filter, wave-wide reductions.  Internal.
#pragma once
#include "smh_kernels.h"
#include "smh_consts.h"

namespace smh {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// Rust `f32 as u32`: truncate toward zero, saturate, NaN -> 0.
__device__ __forceinline__ uint32_t f2u(float v) {
	return (v >= 0.0f) ? ((v >= 4294967296.0f) ? 0xFFFFFFFFu : (uint32_t)v) : 0u;
}

// image 0.23.14 rgb_to_luma: (0.2126 r + 0.7152 g) + 0.0722 b in f32, truncated to u8.
__device__ __forceinline__ uint32_t luma8(uint32_t r, uint32_t g, uint32_t b) {
	float l = SMH_LUMA_R * (float)r + SMH_LUMA_G * (float)g + SMH_LUMA_B * (float)b;
	uint32_t u = (uint32_t)l;   // l is in [0, 255.0001]
	return u > 255u ? 255u : u;
}

__device__ __forceinline__ uint32_t absdiff(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

// util/src/image.rs:159-187 hsv() + vision-common/src/markers/mod.rs:17-19,40-54, evaluated exactly
// as the reference does (f32, same operation order).  Two identities remove the fmodf calls:
//   ((g-b)/delta) % 6.0   : |(g-b)/delta| <= 1 < 6, so fmodf returns its argument unchanged;
//   modulo(h, 360.0)      : h is in [-60, 300], so fmodf(h,360) == h and only the `+ 360` applies.
// tests/test_gpu_parity.py checks the device predicate against the oracle on all 2^24 colours.
static __device__ bool marker_exact(uint32_t r8, uint32_t g8, uint32_t b8) {
	const float r = (float)r8 / 255.0f, g = (float)g8 / 255.0f, b = (float)b8 / 255.0f;
	const float mx = fmaxf(r, fmaxf(g, b));
	const float mn = fminf(r, fminf(g, b));
	const float delta = mx - mn;
	float h;
	if (mx == mn) h = 0.0f;
	else if (mx == r) h = 60.0f * ((g - b) / delta);
	else if (mx == g) h = 60.0f * (((b - r) / delta) + 2.0f);
	else h = 60.0f * (((r - g) / delta) + 4.0f);
	if (h < 0.0f) h = h + 360.0f;
	const float sf = (100.0f * delta) / mx;     // NaN when mx == 0 -> 0
	const float vf = 100.0f * mx;
	const uint32_t hu = f2u(h);
	uint32_t su = f2u(sf); su = su > 255u ? 255u : su;
	uint32_t vu = f2u(vf); vu = vu > 255u ? 255u : vu;
	if (su < SMH_HSV_MIN_SAT) return false;
	bool any = false;
#define SMH_TEAM(MH, MS, MV)                                                                              \
	any = any || (absdiff(MH, hu) <= SMH_HSV_HUE_TOLERANCE &&                                             \
	              (absdiff(MS, su) <= SMH_HSV_SAT_TOLERANCE ||                                            \
	               (uint32_t)abs((int)su - ((int)(MS) - SMH_PLAYER_DIR_ARC_SAT)) <= SMH_HSV_SAT_TOLERANCE) && \
	              absdiff(MV, vu) <= SMH_HSV_VIB_TOLERANCE)
	SMH_TEAM(SMH_ALPHA_H, SMH_ALPHA_S, SMH_ALPHA_V);
	SMH_TEAM(SMH_BRAVO_H, SMH_BRAVO_S, SMH_BRAVO_V);
	SMH_TEAM(SMH_CHARLIE_H, SMH_CHARLIE_S, SMH_CHARLIE_V);
#undef SMH_TEAM
	return any;
}

// Cheap integer necessary condition in front of the exact float path (most map terrain fails it,
// so whole waves skip the divisions):  s >= 35 needs 100*d/m >= 34.99 (the f32 result is within
// 1e-4 of the rational), and every team window needs v >= 70, i.e. max channel >= 178.
__device__ __forceinline__ bool marker_prefilter(uint32_t bgra) {
	const uint32_t b8 = bgra & 255u, g8 = (bgra >> 8) & 255u, r8 = (bgra >> 16) & 255u;
	const uint32_t m = max(r8, max(g8, b8)), n = min(r8, min(g8, b8)), d = m - n;
	return (uint32_t)(m >= 178u) & (uint32_t)(d * 10000u >= 3499u * m);
}
__device__ __forceinline__ bool is_marker(uint32_t r8, uint32_t g8, uint32_t b8) {
	bool res = false;
	if (marker_prefilter(b8 | (g8 << 8) | (r8 << 16))) res = marker_exact(r8, g8, b8);
	return res;
}

__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {
	for (int o = 32; o; o >>= 1) v |= __shfl_xor(v, o);
	return v;
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
	for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
	return v;
}
__device__ __forceinline__ uint64_t wave_max64(uint64_t v) {
	for (int o = 32; o; o >>= 1) { uint64_t t = __shfl_xor(v, o); v = t > v ? t : v; }
	return v;
}

// Wave-wide reductions on DPP lane permutes (no LDS round trips): butterflies inside each 16-lane row,
// then the four row results are combined on the scalar unit.  All 64 lanes must be active.
#define SMH_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ uint32_t wave_max32_dpp(uint32_t v) {
	v = max(v, SMH_DPP(v, 0xB1));    // quad_perm [1,0,3,2]
	v = max(v, SMH_DPP(v, 0x4E));    // quad_perm [2,3,0,1]
	v = max(v, SMH_DPP(v, 0x141));   // row_half_mirror
	v = max(v, SMH_DPP(v, 0x140));   // row_mirror
	return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
	           max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ uint32_t wave_or32_dpp(uint32_t v) {
	v |= SMH_DPP(v, 0xB1);
	v |= SMH_DPP(v, 0x4E);
	v |= SMH_DPP(v, 0x141);
	v |= SMH_DPP(v, 0x140);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
	       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ uint32_t wave_sum32_dpp(uint32_t v) {
	v += SMH_DPP(v, 0xB1);
	v += SMH_DPP(v, 0x4E);
	v += SMH_DPP(v, 0x141);
	v += SMH_DPP(v, 0x140);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
	       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ uint32_t wave_xor32_dpp(uint32_t v) {
	v ^= SMH_DPP(v, 0xB1);
	v ^= SMH_DPP(v, 0x4E);
	v ^= SMH_DPP(v, 0x141);
	v ^= SMH_DPP(v, 0x140);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^
	       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

}  // namespace smh
