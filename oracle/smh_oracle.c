/*
 * smh_oracle.c -- CPU restatement of the reference's vision-cpu back-end.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it; the product (libsmh_vision_hip.so) never
 * links, loads or falls back to anything in oracle/.
 *
 * Parity status: PARITY UNPINNED by the reference -- it ships NO asserting tests / golden vectors for this path
 * (SURVEY.md section 4), and its Rust sources cannot be compiled here (no rustc/cargo).
 * The restatement is therefore pinned by (a) the expected ROI geometry / workload counts /
 * line lists recorded by an independent numpy probe in SURVEY.md Appendix B, and (b) the
 * committed goldens under tests/golden/ produced by this file on the reference's own
 * sample screenshots (the PNGs in vision-common/samples).  Where it restates third-party crates
 * that are not vendored under /root/reference, the crate and pinned version are named.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  All float
 * arithmetic is scalar IEEE f32/f64 in the reference's operation order; Rust `as` casts are
 * emulated (truncate, saturate, NaN -> 0).
 *
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---- constants: vision-common/src/consts/consts.toml:1-63 ------------------------------- */
static const int16_t CLOSE_DEPLOYMENT_BUTTON_COLOR[3] = {217, 67, 49};
#define CLOSE_DEPLOYMENT_BUTTON_TOLERANCE 25
#define CLOSE_DEPLOYMENT_BUTTON_RED_PIXEL_THRESHOLD 0.65f
#define OCR_PREPROCESS_BRIGHTNESS_THRESHOLD 200
#define OCR_PREPROCESS_MONOCHROMATICY_THRESHOLD 3
#define OCR_PREPROCESS_BRIGHTNESS_EDGE_THRESHOLD 130
#define OCR_PREPROCESS_SIMILARITY_EDGE_THRESHOLD 48
#define OCR_PREPROCESS_DILATE_RADIUS 3u
static const uint16_t MARKER_HSV[3][3] = {{105, 100, 100}, {285, 46, 85}, {158, 60, 91}};
#define FIND_MARKER_HSV_HUE_TOLERANCE 15
#define FIND_MARKER_HSV_SAT_TOLERANCE 15
#define FIND_MARKER_HSV_VIB_TOLERANCE 15
#define FIND_MARKER_HSV_MIN_SAT 35
#define FIND_MARKER_PLAYER_DIR_ARC_SAT 50

/* ---- Rust `as` casts (float -> unsigned): truncate toward zero, saturate, NaN -> 0 -------- */
static inline uint32_t f32_as_u32(float v) {
	if (!(v == v) || v <= 0.0f) return 0u;
	if (v >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)v;
}
static inline uint16_t f32_as_u16(float v) {
	if (!(v == v) || v <= 0.0f) return 0u;
	if (v >= 65535.0f) return 65535u;
	return (uint16_t)v;
}
static inline uint8_t f32_as_u8(float v) {
	if (!(v == v) || v <= 0.0f) return 0u;
	if (v >= 255.0f) return 255u;
	return (uint8_t)v;
}
static inline uint32_t f64_round_as_u32(double v) {
	/* f64::round = half away from zero == C round(); then `as u32` */
	double r = round(v);
	if (!(r == r) || r <= 0.0) return 0u;
	if (r >= 4294967296.0) return 0xFFFFFFFFu;
	return (uint32_t)r;
}

/* ---- screen-relative bounds: vision-common/src/screen.rs:4-66, consts/mod.rs:7-19 -------- */
/* Every RelativeBound in the two constants is ScreenH(frac). */
static uint32_t screen_h(double frac, uint32_t H) { return f64_round_as_u32(frac * (double)H); }

/* MAP_BOUNDS.into_absolute, then "map fills remaining space" (vision-cpu/src/lib.rs:137-145).
 * Returns 0 when the u32 arithmetic of the reference would underflow / the crop would panic
 * (util/src/image.rs:71-77: `x + w >= width || y + h >= height`). */
ORC_API int orc_map_bounds(uint32_t W, uint32_t H, uint32_t out[4]) {
	uint32_t w = screen_h(0.864930556, H), h = screen_h(0.761078559, H);
	uint32_t x = screen_h(0.018522135, H);                 /* Left(...) */
	uint32_t yb = screen_h(0.07421875, H);                 /* Bottom(...) */
	if ((uint64_t)yb + h > H) return 0;
	uint32_t y = H - yb - h;
	if (w > W) return 0;
	uint32_t w2 = W - w;                                   /* let w = frame.width() - w; */
	if ((uint64_t)x + w2 > W) return 0;
	uint32_t x2 = W - x - w2;                              /* let x = frame.width() - x - w; */
	out[0] = x2; out[1] = y; out[2] = w2; out[3] = h;
	if ((uint64_t)x2 + w2 >= W || (uint64_t)y + h >= H) return 0; /* par_crop_into panic */
	if (w2 < 8 || h < 8) return 0;
	return 1;
}

/* CLOSE_DEPLOYMENT_BUTTON_BOUNDS.into_absolute (consts/mod.rs:14-19). */
ORC_API int orc_button_bounds(uint32_t W, uint32_t H, uint32_t out[4]) {
	uint32_t w = screen_h(0.236132813, H), h = screen_h(0.038205295, H);
	uint32_t xr = screen_h(0.0078125, H), yb = screen_h(0.0078125, H);
	if ((uint64_t)xr + w > W || (uint64_t)yb + h > H) return 0;
	out[0] = W - xr - w; out[1] = H - yb - h; out[2] = w; out[3] = h;
	return (w > 0 && h > 0) ? 1 : 0;
}

/* ---- luma: image 0.23.14 (Cargo.lock:1487) color.rs rgb_to_luma/bgr_to_luma -------------- */
/* l = 0.2126*r + 0.7152*g + 0.0722*b in f32, left-to-right, NumCast -> u8 (truncation).
 * Mirrored constant-for-constant by the reference's CUDA side (vision-gpu/cuda/cuda.cu:23-26).
 * Call sites: vision-cpu/src/lib.rs:154,224,242. */
static inline uint8_t luma8(uint8_t r, uint8_t g, uint8_t b) {
	float l = 0.2126f * (float)r + 0.7152f * (float)g + 0.0722f * (float)b;
	return f32_as_u8(l);
}
ORC_API uint8_t orc_luma8(uint8_t r, uint8_t g, uint8_t b) { return luma8(r, g, b); }

/* ---- hsv: util/src/image.rs:159-187 ---------------------------------------------------- */
static inline float rust_modulo(float a, float b) {
	float r = fmodf(a, b);                                  /* Rust f32 `%` == fmodf */
	return (r < 0.0f) ? r + b : r;
}
ORC_API void orc_hsv(uint8_t r8, uint8_t g8, uint8_t b8, uint16_t *ho, uint8_t *so, uint8_t *vo) {
	float r = (float)r8 / 255.0f, g = (float)g8 / 255.0f, b = (float)b8 / 255.0f;
	float max = fmaxf(r, fmaxf(g, b));
	float min = fminf(r, fminf(g, b));
	float delta = max - min;
	float h;
	if (max == min) h = 0.0f;
	else if (max == r) h = 60.0f * fmodf((g - b) / delta, 6.0f);
	else if (max == g) h = 60.0f * (((b - r) / delta) + 2.0f);
	else h = 60.0f * (((r - g) / delta) + 4.0f);
	float s = 100.0f * delta / max;                         /* (100*delta)/max; NaN when max==0 */
	float v = 100.0f * max;
	*ho = f32_as_u16(rust_modulo(h, 360.0f));
	*so = f32_as_u8(s);
	*vo = f32_as_u8(v);
}

/* ---- marker colour predicate: vision-common/src/markers/mod.rs:17-19,40-54 --------------- */
static inline int abs_diff_i(int a, int b) { return a > b ? a - b : b - a; }
static inline int saturation_ok(uint8_t s, uint8_t ms) {
	if (abs_diff_i(ms, s) <= FIND_MARKER_HSV_SAT_TOLERANCE) return 1;
	int16_t t = (int16_t)((int16_t)s - ((int16_t)ms - FIND_MARKER_PLAYER_DIR_ARC_SAT));
	uint8_t a = (uint8_t)(t < 0 ? -t : t);                  /* (i16).abs() as u8 */
	return a <= FIND_MARKER_HSV_SAT_TOLERANCE;
}
ORC_API int orc_is_any_map_marker_color(uint8_t r, uint8_t g, uint8_t b) {
	uint16_t h; uint8_t s, v;
	orc_hsv(r, g, b, &h, &s, &v);
	if (s < FIND_MARKER_HSV_MIN_SAT) return 0;
	for (int t = 0; t < 3; ++t) {
		uint16_t mh = MARKER_HSV[t][0]; uint8_t ms = (uint8_t)MARKER_HSV[t][1], mv = (uint8_t)MARKER_HSV[t][2];
		if (abs_diff_i(mh, h) <= FIND_MARKER_HSV_HUE_TOLERANCE && saturation_ok(s, ms) &&
		    abs_diff_i(mv, v) <= FIND_MARKER_HSV_VIB_TOLERANCE)
			return 1;
	}
	return 0;
}

/* Bit-packed table of the predicate over all 2^24 colours: bit index = r<<16 | g<<8 | b.
 * Used by the exhaustive device-vs-oracle colour test. */
ORC_API void orc_marker_table(uint32_t *bits /* 2^24/32 words */) {
#pragma omp parallel for schedule(static)
	for (int64_t w = 0; w < (1 << 24) / 32; ++w) {
		uint32_t acc = 0;
		for (uint32_t k = 0; k < 32; ++k) {
			uint32_t c = (uint32_t)w * 32u + k;
			if (orc_is_any_map_marker_color((uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c)) acc |= 1u << k;
		}
		bits[w] = acc;
	}
}

/* ---- crop_to_map: vision-cpu/src/lib.rs:110-171 ------------------------------------------ */
/* Red "Close Deployment" button pixel count (lib.rs:116-133). Frame is BGRA8, stride 4*W. */
ORC_API uint32_t orc_button_red_pixels(const uint8_t *bgra, uint32_t W, uint32_t H) {
	uint32_t r[4];
	if (!orc_button_bounds(W, H, r)) return 0;
	uint32_t count = 0;
	for (uint32_t y = r[1]; y < r[1] + r[3]; ++y)
		for (uint32_t x = r[0]; x < r[0] + r[2]; ++x) {
			const uint8_t *p = bgra + ((size_t)y * W + x) * 4;
			const uint8_t rgb[3] = {p[2], p[1], p[0]};      /* pixel.to_rgb() */
			int ok = 1;
			for (int i = 0; i < 3; ++i) {
				int16_t d = (int16_t)(CLOSE_DEPLOYMENT_BUTTON_COLOR[i] - (int16_t)rgb[i]);
				uint16_t a = (uint16_t)(d < 0 ? -d : d);
				if (a > CLOSE_DEPLOYMENT_BUTTON_TOLERANCE) { ok = 0; break; }
			}
			count += (uint32_t)ok;
		}
	return count;
}

/* Returns: 1 = Some((ui_map,[x,y,w,h])), 0 = Ok(None) (map closed), -1 = geometry invalid.
 * ui_rgba: w*h*4; map_rgb: w*h*3; brq_rgb: (w/2)*(h/2)*3. Any output pointer may be NULL. */
ORC_API int orc_crop_to_map(const uint8_t *bgra, uint32_t W, uint32_t H, int grayscale, uint8_t *ui_rgba,
                            uint8_t *map_rgb, uint8_t *brq_rgb, uint32_t roi[4]) {
	uint32_t bb[4], mb[4];
	if (!orc_button_bounds(W, H, bb) || !orc_map_bounds(W, H, mb)) return -1;
	uint32_t red = orc_button_red_pixels(bgra, W, H);
	float ratio = (float)red / (float)(bb[2] * bb[3]);     /* red_pixels as f32 / (w*h) as f32 */
	if (ratio < CLOSE_DEPLOYMENT_BUTTON_RED_PIXEL_THRESHOLD) return 0;
	uint32_t x = mb[0], y = mb[1], w = mb[2], h = mb[3];
	uint32_t brq_w = w / 2, brq_h = h / 2;
	roi[0] = x; roi[1] = y; roi[2] = w; roi[3] = h;
	for (uint32_t j = 0; j < h; ++j)
		for (uint32_t i = 0; i < w; ++i) {
			const uint8_t *p = bgra + ((size_t)(y + j) * W + (x + i)) * 4;
			if (ui_rgba) {
				uint8_t *o = ui_rgba + ((size_t)j * w + i) * 4;
				if (grayscale) {
					uint8_t l = luma8(p[2], p[1], p[0]);       /* Bgra::to_luma */
					o[0] = l; o[1] = l; o[2] = l; o[3] = 255;
				} else {
					o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = 255;
				}
			}
			if (map_rgb) {                                   /* Bgra -> Rgb: util/src/image.rs:293-298 */
				uint8_t *o = map_rgb + ((size_t)j * w + i) * 3;
				o[0] = p[2]; o[1] = p[1]; o[2] = p[0];
			}
		}
	if (brq_rgb)
		for (uint32_t j = 0; j < brq_h; ++j)
			for (uint32_t i = 0; i < brq_w; ++i) {
				const uint8_t *p = bgra + ((size_t)(y + brq_h + j) * W + (x + brq_w + i)) * 4;
				uint8_t *o = brq_rgb + ((size_t)j * brq_w + i) * 3;
				o[0] = p[2]; o[1] = p[1]; o[2] = p[0];
			}
	return 1;
}

/* ---- ocr_preprocess: vision-cpu/src/lib.rs:39-53,173-231 --------------------------------- */
static inline uint8_t ocr_brightness_all_ge(const uint8_t *p, uint8_t thr) { return p[0] >= thr && p[1] >= thr && p[2] >= thr; }
static inline uint16_t ocr_monochromaticy(const uint8_t *p) {
	uint16_t diff = 0;
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b) diff += (uint16_t)abs_diff_i(p[a], p[b]);
	return diff;
}
ORC_API int orc_ocr_preprocess(const uint8_t *brq_rgb, uint32_t w, uint32_t h, uint8_t *out) {
	const uint32_t R = OCR_PREPROCESS_DILATE_RADIUS;
	if (w < R || h < R) return -1;
#pragma omp parallel for schedule(static)
	for (int64_t yy0 = 0; yy0 < (int64_t)h; ++yy0) {
		uint32_t y = (uint32_t)yy0;
		for (uint32_t x = 0; x < w; ++x) {
			const uint8_t *pixel = brq_rgb + ((size_t)y * w + x) * 3;
			int keep = 0;
			uint16_t diff = ocr_monochromaticy(pixel);
			if (diff <= OCR_PREPROCESS_MONOCHROMATICY_THRESHOLD && ocr_brightness_all_ge(pixel, OCR_PREPROCESS_BRIGHTNESS_THRESHOLD)) {
				keep = 1;
			} else if (diff <= OCR_PREPROCESS_SIMILARITY_EDGE_THRESHOLD &&
			           ocr_brightness_all_ge(pixel, OCR_PREPROCESS_BRIGHTNESS_EDGE_THRESHOLD)) {
				uint32_t x0 = x >= R ? x - R : 0, x1 = (x + R < w - R) ? x + R : w - R; /* ..=min(x+3, w-3) */
				uint32_t y0 = y >= R ? y - R : 0, y1 = (y + R < h - R) ? y + R : h - R;
				for (uint32_t xx = x0; xx <= x1 && !keep; ++xx)
					for (uint32_t yy = y0; yy <= y1; ++yy) {
						const uint8_t *q = brq_rgb + ((size_t)yy * w + xx) * 3;
						if (!ocr_brightness_all_ge(q, OCR_PREPROCESS_BRIGHTNESS_THRESHOLD)) continue;
						if (ocr_monochromaticy(q) <= OCR_PREPROCESS_MONOCHROMATICY_THRESHOLD) { keep = 1; break; }
					}
			}
			out[(size_t)y * w + x] = keep ? (uint8_t)(255 - luma8(pixel[0], pixel[1], pixel[2])) : 255;
		}
	}
	return 0;
}

/* ---- find_scales_preprocess: vision-cpu/src/lib.rs:233-251 ------------------------------- */
/* Rows < scales_start_y are left untouched (stale), as in the reference. */
ORC_API int orc_find_scales_preprocess(const uint8_t *brq_rgb, uint32_t w, uint32_t h, uint32_t scales_start_y, uint8_t *out) {
	if (scales_start_y > h) return -1;                      /* h - scales_start_y underflows in the reference */
	for (uint32_t y = scales_start_y; y < h; ++y)
		for (uint32_t x = 0; x < w; ++x) {
			const uint8_t *p = brq_rgb + ((size_t)y * w + x) * 3;
			out[(size_t)y * w + x] = luma8(p[0], p[1], p[2]) != 0 ? 255 : 0;
		}
	return 0;
}

/* ---- isolate_map_markers: vision-cpu/src/lib.rs:253-280 ---------------------------------- */
ORC_API void orc_isolate_map_markers(uint8_t *map_rgb, uint32_t w, uint32_t h) {
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)w * h; ++i) {
		uint8_t *p = map_rgb + (size_t)i * 3;
		if (!orc_is_any_map_marker_color(p[0], p[1], p[2])) { p[0] = 0; p[1] = 0; p[2] = 0; }
	}
}

/* ---- dilation: imageproc 0.22.0 (Cargo.lock:1503) ---------------------------------------- */
/* morphology::dilate_mut(img, Norm::L1, k): distance_transform_mut(img, L1) then
 * `*p = if *p <= k {255} else {0}`.  distance_transform.rs: foreground = non-zero pixels get 0,
 * background gets min(w+h,255); forward pass checks (x-1,y),(x,y-1); backward pass checks
 * (x+1,y),(x,y+1); check(): `if candidate+1 < current {current = candidate+1}` in u16.
 * Call site: vision-cpu/src/lib.rs:372. */
ORC_API void orc_dilate_l1_imageproc(uint8_t *img, uint32_t w, uint32_t h, uint8_t k) {
	uint32_t md = w + h; if (md > 255) md = 255;
	for (size_t i = 0; i < (size_t)w * h; ++i) img[i] = img[i] > 0 ? 0 : (uint8_t)md;
#define DT_CHECK(cx, cy, nx, ny) do { uint16_t cur = img[(size_t)(cy) * w + (cx)]; \
		uint16_t cand = (uint16_t)(img[(size_t)(ny) * w + (nx)] + 1u); \
		if (cand < cur) img[(size_t)(cy) * w + (cx)] = (uint8_t)cand; } while (0)
	for (uint32_t y = 0; y < h; ++y)
		for (uint32_t x = 0; x < w; ++x) {
			if (x > 0) DT_CHECK(x, y, x - 1, y);
			if (y > 0) DT_CHECK(x, y, x, y - 1);
		}
	for (uint32_t y = h; y-- > 0;)
		for (uint32_t x = w; x-- > 0;) {
			if (x < w - 1) DT_CHECK(x, y, x + 1, y);
			if (y < h - 1) DT_CHECK(x, y, x, y + 1);
		}
#undef DT_CHECK
	for (size_t i = 0; i < (size_t)w * h; ++i) img[i] = img[i] <= k ? 255 : 0;
}

/* Equivalent closed form for k = 1: 4-neighbour "cross" OR, clipped at the image edge.
 * Kept separately so a test can assert it equals the literal imageproc restatement. */
ORC_API void orc_dilate_cross(const uint8_t *in, uint32_t w, uint32_t h, uint8_t *out) {
	for (uint32_t y = 0; y < h; ++y)
		for (uint32_t x = 0; x < w; ++x) {
			int on = in[(size_t)y * w + x] != 0;
			if (!on && x > 0) on = in[(size_t)y * w + x - 1] != 0;
			if (!on && x + 1 < w) on = in[(size_t)y * w + x + 1] != 0;
			if (!on && y > 0) on = in[(size_t)(y - 1) * w + x] != 0;
			if (!on && y + 1 < h) on = in[(size_t)(y + 1) * w + x] != 0;
			out[(size_t)y * w + x] = on ? 255 : 0;
		}
}

/* ---- mask_marker_lines: vision-cpu/src/lib.rs:357-375 ------------------------------------ */
/* Works on the (possibly already isolated) RGB crop; the predicate is idempotent under
 * isolate_map_markers because (0,0,0) has s = NaN -> 0 < MIN_SAT. */
ORC_API void orc_mask_marker_lines(const uint8_t *map_rgb, uint32_t w, uint32_t h, uint8_t *lsd) {
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)w * h; ++i) {
		const uint8_t *p = map_rgb + (size_t)i * 3;
		lsd[i] = orc_is_any_map_marker_color(p[0], p[1], p[2]) ? 255 : 0;
	}
	orc_dilate_l1_imageproc(lsd, w, h, 1);
}

/* ---- lsd: vision-common/src/lsd.rs ------------------------------------------------------- */
#define IMG(img, w, x, y) ((img)[(size_t)(y) * (w) + (x)])

/* get_centre: lsd.rs:5-44.  unsafe_get_pixel is unchecked in the reference; coordinates are
 * clamped here (cannot trigger through find_lines) so the oracle never reads
 * out of bounds. */
static inline uint8_t px_clamped(const uint8_t *img, uint32_t w, uint32_t h, uint32_t x, uint32_t y) {
	if (x >= w) x = w - 1;
	if (y >= h) y = h - 1;
	return IMG(img, w, x, y);
}
ORC_API void orc_get_centre(const uint8_t *img, uint32_t w, uint32_t h, float ptx, float pty, float *ox, float *oy) {
	const float MAX_DIST = 5.0f;
	float left = ptx;
	while (left > 0.0f && fabsf(left - ptx) < MAX_DIST && px_clamped(img, w, h, f32_as_u32(left), f32_as_u32(pty)) == 255) left -= 1.0f;
	float right = ptx;
	while (right < (float)(w - 1) && fabsf(right - ptx) < MAX_DIST && px_clamped(img, w, h, f32_as_u32(right), f32_as_u32(pty)) == 255) right += 1.0f;
	float up = pty;
	while (up > 0.0f && fabsf(up - pty) < MAX_DIST && px_clamped(img, w, h, f32_as_u32(ptx), f32_as_u32(up)) == 255) up -= 1.0f;
	float down = pty;
	while (down < (float)(h - 1) && fabsf(down - pty) < MAX_DIST && px_clamped(img, w, h, f32_as_u32(ptx), f32_as_u32(down)) == 255) down += 1.0f;
	*ox = (left + right) / 2.0f;
	*oy = (up + down) / 2.0f;
}

/* Ray directions: vision-cpu/src/lib.rs:398-399,437: theta = ((i as f32)/10.0).to_radians();
 * dx = theta.cos(); dy = theta.sin().  f32::to_radians(x) = x * (PI_f32 / 180.0_f32);
 * cos/sin = platform libm cosf/sinf (glibc here). */
static float g_dx[3600], g_dy[3600];
static int g_trig_ready = 0;
static void trig_init(void) {
	if (g_trig_ready) return;
	const float PI_F = 3.14159265358979323846264338327950288f;
	const float k = PI_F / 180.0f;
	for (uint32_t i = 0; i < 3600; ++i) {
		float theta = ((float)i / 10.0f) * k;
		g_dx[i] = cosf(theta);
		g_dy[i] = sinf(theta);
	}
	g_trig_ready = 1;
}
ORC_API void orc_ray_table(float *dx, float *dy) {
	trig_init();
	memcpy(dx, g_dx, sizeof g_dx);
	memcpy(dy, g_dy, sizeof g_dy);
}

/* find_line_in_image closure: vision-cpu/src/lib.rs:388-432.  Returns the number of mask
 * samples taken (workload statistic, not part of the reference output). */
static inline uint32_t cast_ray(const uint8_t *img, uint32_t w, uint32_t h, float ptx, float pty, float max_gap, float dx, float dy,
                                float *x_end_o, float *y_end_o) {
	float x = ptx, y = pty;
	const float x_start = x, y_start = y;
	float x_end = x, y_end = y;
	float gap0 = 0.0f, gap1 = 0.0f, gap2 = 0.0f;
	float x_offset = 0.0f, y_offset = 0.0f;
	const float wf = (float)w, hf = (float)h;
	uint32_t steps = 0;
	while (x >= 0.0f && y >= 0.0f && x < wf && y < hf) {
		++steps;
		if (IMG(img, w, f32_as_u32(x), f32_as_u32(y)) == 255) {
			gap0 = 0.0f; gap1 = 0.0f; gap2 = 0.0f;
		} else if (gap0 >= max_gap) {
			x = gap1; y = gap2;
			break;
		} else if (gap0 == 0.0f) {
			gap0 = 1.0f; gap1 = x; gap2 = y;
		} else {
			gap0 += 1.0f;
		}
		x_offset += dx;
		y_offset += dy;
		x = x_offset + x_start;
		y = y_offset + y_start;
	}
	/* image.get_pixel_checked(x as u32, y as u32) == Some(Luma([0])) (util/src/image.rs:113-124) */
	uint32_t xi = f32_as_u32(x), yi = f32_as_u32(y);
	if (xi < w && yi < h && IMG(img, w, xi, yi) == 0) {
		x_end = x - dx;
		y_end = y - dy;
	}
	*x_end_o = x_end;
	*y_end_o = y_end;
	return steps;
}

static __thread uint64_t g_stat_steps = 0;  /* accumulated mask samples of the calling thread */

/* find_longest_line: vision-cpu/src/lib.rs:387-449.  rayon `reduce(identity, op)` with
 * op(a,b) = a if a.len > b.len else b folds in index order (rayon 1.5.3, Cargo.lock:2543), so
 * the result is max len^2 with ties going to the HIGHEST theta index. */
ORC_API void orc_find_longest_line(const uint8_t *img, uint32_t w, uint32_t h, float ptx, float pty, float max_gap, float line[4], float *len_sq) {
	trig_init();
	float best[4] = {0, 0, 0, 0}, best_len = 0.0f;         /* Default::default() identity */
	uint64_t steps = 0;
	for (uint32_t i = 0; i < 3600; ++i) {
		float xe, ye;
		steps += cast_ray(img, w, h, ptx, pty, max_gap, g_dx[i], g_dy[i], &xe, &ye);
		float ddx = ptx - xe, ddy = pty - ye;               /* p0.distance_sqr(&p1): geometry.rs:62-68 */
		float len = ddx * ddx + ddy * ddy;
		if (!(best_len > len)) { best[0] = ptx; best[1] = pty; best[2] = xe; best[3] = ye; best_len = len; }
	}
	g_stat_steps += steps;
	memcpy(line, best, sizeof best);
	*len_sq = best_len;
}

/* nearest_point_on_line: lsd.rs:47-58 */
static inline void nearest_point_on_line(float px, float py, float r0x, float r0y, float r1x, float r1y, float *ox, float *oy) {
	float dx = r1x - r0x, dy = r1y - r0y;
	if (dx == 0.0f && dy == 0.0f) { *ox = r0x; *oy = r0y; return; }
	float u = ((px - r0x) * dx + (py - r0y) * dy) / (dx * dx + dy * dy);
	*ox = r0x + u * dx;
	*oy = r0y + u * dy;
}

/* find_lines::<32>: lsd.rs:60-107.  stats (optional): [0]=rounds (find_longest_line calls),
 * [1]=mask samples, [2]=white pixels skipped by proximity, [3]=white pixels visited. */
ORC_API uint32_t orc_find_lines(const uint8_t *img, uint32_t w, uint32_t h, uint32_t max_gap_u, float lines[32][4], uint64_t stats[4]) {
	const float max_gap = (float)max_gap_u;
	uint32_t n = 0;
	uint64_t rounds = 0, skipped = 0, visited = 0;
	g_stat_steps = 0;
	for (uint32_t yi = 0; yi < h; ++yi) {
		for (uint32_t xi = 0; xi < w; ++xi) {
			if (IMG(img, w, xi, yi) != 255) continue;
			++visited;
			float x = (float)xi, y = (float)yi;
			int skip = 0;
			for (uint32_t l = 0; l < n; ++l) {
				float nx, ny;
				nearest_point_on_line(x, y, lines[l][0], lines[l][1], lines[l][2], lines[l][3], &nx, &ny);
				float ex = x - nx, ey = y - ny;
				if (ex * ex + ey * ey < 50.0f) { skip = 1; break; }
			}
			if (skip) { ++skipped; continue; }
			float cx, cy;
			orc_get_centre(img, w, h, x, y, &cx, &cy);
			float longest[4], max_length;
			orc_find_longest_line(img, w, h, cx, cy, max_gap, longest, &max_length);
			++rounds;
			if (max_length > 2500.0f) {
				orc_get_centre(img, w, h, longest[2], longest[3], &longest[2], &longest[3]);
				memcpy(lines[n], longest, sizeof longest);
				++n;
				if (n == 32) goto done;
			}
		}
	}
done:
	if (stats) { stats[0] = rounds; stats[1] = g_stat_steps; stats[2] = skipped; stats[3] = visited; }
	return n;
}

/* ---- meters-per-pixel: src/vision/mpx_ratio.rs:3-134 ------------------------------------- */
/* find_scale_width.  The reference reads (x, y..y+4) with unchecked get_pixel (UB below the
 * image, util/src/image.rs:136-140); here an out-of-image pixel counts as non-zero (the tick
 * test fails).  `right - left` wraps in u32 as a release build does (mpx_ratio.rs:58).
 * dbg (optional) = [left, y, right, y] of the accepted bar. Returns 1 = Some(ratio). */
static inline int scale_px_is_zero(const uint8_t *img, uint32_t w, uint32_t h, uint32_t x, uint32_t y) {
	if (x >= w || y >= h) return 0;
	return IMG(img, w, x, y) == 0;
}
ORC_API int orc_find_scale_width(uint32_t meters, uint32_t x, uint32_t y, const uint8_t *img, uint32_t w, uint32_t h, double *ratio, uint32_t dbg[4]) {
	const uint32_t MIN_SCALE_WIDTH = 10, MIN_SCALE_VERTICAL_BAR_HEIGHT = 4;
	if (y < MIN_SCALE_VERTICAL_BAR_HEIGHT) return 0;
	if (x >= w) return 0;                                   /* reference: unchecked read; defined here as None */
	uint32_t max_scale_y_offset = f64_round_as_u32((20.0 / 640.0) * (double)w);
	uint32_t y_end = (h < y + max_scale_y_offset) ? h : y + max_scale_y_offset;
	for (uint32_t yy = y; yy < y_end; ++yy) {
		if (IMG(img, w, x, yy) != 0) continue;
		/* Go right... (the `(y..y-4).rev()` half of the chain is an empty range) */
		uint32_t right = 0;
		for (uint32_t xx = x; xx < w; ++xx) {
			int all_zero = 1;
			for (uint32_t ty = yy; ty < yy + MIN_SCALE_VERTICAL_BAR_HEIGHT; ++ty)
				if (!scale_px_is_zero(img, w, h, xx, ty)) { all_zero = 0; break; }
			if (!all_zero) continue;
			right = xx;
			break;
		}
		if (right == 0) continue;
		right -= 1;
		uint32_t left = 0;
		for (uint32_t xx = x; xx-- > 0;) {
			int all_zero = 1;
			for (uint32_t ty = yy; ty < yy + MIN_SCALE_VERTICAL_BAR_HEIGHT; ++ty)
				if (!scale_px_is_zero(img, w, h, xx, ty)) { all_zero = 0; break; }
			if (!all_zero) continue;
			left = xx;
			break;
		}
		if (left == 0) continue;
		left += 1;
		uint32_t width = right - left;                       /* wrapping, as release Rust */
		if (width < MIN_SCALE_WIDTH) continue;
		if (dbg) { dbg[0] = left; dbg[1] = yy; dbg[2] = right; dbg[3] = yy; }
		*ratio = (double)meters / (double)width;
		return 1;
	}
	return 0;
}

/* calc_meters_to_px_ratio: mpx_ratio.rs:3,79-134.  scales = n x (meters, x, y), n <= 3.
 * Mean of the successful ones, summed in index order ((a+b)/2, (a+b+c)/3). */
ORC_API int orc_calc_meters_to_px_ratio(const uint32_t *scales, uint32_t n, const uint8_t *img, uint32_t w, uint32_t h, double *ratio) {
	if (n == 0 || n > 3) return 0;
	double sum = 0.0; uint32_t ok = 0;
	for (uint32_t i = 0; i < n; ++i) {
		double r;
		if (orc_find_scale_width(scales[i * 3], scales[i * 3 + 1], scales[i * 3 + 2], img, w, h, &r, NULL)) {
			sum = ok ? sum + r : r;
			++ok;
		}
	}
	if (!ok) return 0;
	*ratio = ok == 1 ? sum : sum / (double)ok;
	return 1;
}

/* ---- derived marker outputs: src/ui/mod.rs:131-140, src/ui/markers.rs:98 ------------------ */
ORC_API void orc_marker_new(const float line[4], double ratio, double *length_px, double *meters) {
	double ax = (double)line[0] - (double)line[2], ay = (double)line[1] - (double)line[3];
	double length = sqrt(ax * ax + ay * ay);               /* powi(2) + powi(2), sqrt */
	*length_px = length;
	*meters = length * ratio;
}
ORC_API float orc_marker_angle(const float line[4]) { return atan2f(line[1] - line[3], line[0] - line[2]); }

/* ---- find_minimap: src/vision/find_minimap.rs:8-146 (the step next to crop_to_map in process()) ---- */
/* get_edginess (find_minimap.rs:8-45): max over the 8 neighbours of sum |dB|+|dG|+|dR|, as f32 / 765.0 */
static float get_edginess(const uint8_t *bgra, uint32_t W, uint32_t ox, uint32_t oy, uint32_t x, uint32_t y) {
	static const int per[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 1}, {0, 1}, {1, 1}, {-1, 0}, {1, 0}};
	const uint8_t *p = bgra + ((size_t)(oy + y) * W + ox + x) * 4;
	uint16_t max = 0;
	for (int k = 0; k < 8; ++k) {
		const uint8_t *q = bgra + ((size_t)(oy + y + per[k][1]) * W + ox + x + per[k][0]) * 4;
		uint16_t s = (uint16_t)(abs_diff_i(p[0], q[0]) + abs_diff_i(p[1], q[1]) + abs_diff_i(p[2], q[2]));
		if (s > max) max = s;
	}
	return (float)max / 765.0f;
}
/* find_edge (find_minimap.rs:63-129); dir: 0 = Up, 1 = Down, 2 = Left, 3 = Right.  u32 arithmetic wraps as in a
 * release build. */
static uint32_t find_edge(const uint8_t *bgra, uint32_t W, uint32_t ox, uint32_t oy, uint32_t w, uint32_t h, uint32_t x, uint32_t y, int dir) {
	const float EDGINESS_THRESHOLD = 0.01f;
	uint32_t xy[2] = {x, y};
	const int c = dir < 2 ? 1 : 0, oc = dir < 2 ? 0 : 1;
	uint32_t c_max = dir < 2 ? h : w, oc_max = dir < 2 ? w : h;
	const int cod = (dir == 0 || dir == 2) ? -1 : 1;
	const uint32_t d = oc_max > xy[oc] ? oc_max - xy[oc] : xy[oc] - oc_max;
	const uint32_t min_line_length0 = d / 2u - 1u;
	c_max -= 3u; oc_max -= 3u;
	for (;;) {
		xy[c] = (uint32_t)((int32_t)xy[c] + cod);
		if (xy[c] > c_max) return c_max + 2u;
		else if (xy[c] < 3u) return 0u;
		if (get_edginess(bgra, W, ox, oy, xy[0], xy[1]) <= EDGINESS_THRESHOLD) {
			const uint32_t ret = xy[c];
			uint32_t p[2] = {xy[0], xy[1]};
			uint32_t min_line_length = min_line_length0;
			int ok = 1;
			while (min_line_length > 0u) {
				p[oc] = (uint32_t)((int32_t)p[oc] - cod);
				if (p[oc] < 3u || p[oc] > oc_max) { ok = 0; break; }
				if (get_edginess(bgra, W, ox, oy, p[0], p[1]) <= EDGINESS_THRESHOLD) min_line_length -= 1u;
				else { ok = 0; break; }
			}
			if (ok) return (uint32_t)((int32_t)ret - cod);
		}
	}
}
/* find_minimap on the map ROI of the frame (src/vision/mod.rs:85).  rect = {left, right, top, bottom} in ROI
 * coordinates; returns 1 = Some, 0 = None, -1 = geometry invalid. */
ORC_API int orc_find_minimap(const uint8_t *bgra, uint32_t W, uint32_t H, uint32_t rect[4]) {
	uint32_t mb[4];
	if (!orc_map_bounds(W, H, mb)) return -1;
	const uint32_t w = mb[2], h = mb[3];
	if (w < 3u || h < 3u) return 0;
	const uint32_t x = w / 2u, y = h / 2u;
	rect[0] = find_edge(bgra, W, mb[0], mb[1], w, h, x, y, 2);
	rect[1] = find_edge(bgra, W, mb[0], mb[1], w, h, x, y, 3);
	rect[2] = find_edge(bgra, W, mb[0], mb[1], w, h, x, y, 0);
	rect[3] = find_edge(bgra, W, mb[0], mb[1], w, h, x, y, 1);
	return 1;
}

/* ---- whole-frame driver (call order of src/vision/mod.rs:36-240) -------------------------- */
/* Used for the CPU baseline timing and for end-to-end goldens.  anchors = n x (meters,x,y)
 * OCR label anchors in BRQ coordinates (OCR itself is out of scope; anchors are inputs).
 * stages bit0: marker mask + LSD; bit1: ui_map; bit2: ocr_preprocess; bit3: scales + mpx. */
typedef struct {
	uint32_t map_open, n_lines;
	float lines[32][4];
	double mpx; uint32_t has_mpx; uint32_t n_mask_px;
	uint64_t rounds, steps;
} orc_frame_result;

ORC_API int orc_process_frame(const uint8_t *bgra, uint32_t W, uint32_t H, int grayscale, uint32_t max_gap, uint32_t stages,
                              const uint32_t *anchors, uint32_t n_anchors, uint32_t scales_start_y, orc_frame_result *res,
                              uint8_t *ui_rgba_o, uint8_t *lsd_o, uint8_t *ocr_o, uint8_t *scales_o) {
	memset(res, 0, sizeof *res);
	uint32_t roi[4];
	uint32_t mb[4];
	if (!orc_map_bounds(W, H, mb)) return -1;
	uint32_t w = mb[2], h = mb[3], bw = w / 2, bh = h / 2;
	uint8_t *map_rgb = (uint8_t *)malloc((size_t)w * h * 3);
	uint8_t *brq_rgb = (uint8_t *)malloc((size_t)bw * bh * 3);
	uint8_t *ui = ui_rgba_o ? ui_rgba_o : ((stages & 2u) ? (uint8_t *)malloc((size_t)w * h * 4) : NULL);
	int open = orc_crop_to_map(bgra, W, H, grayscale, ui, map_rgb, brq_rgb, roi);
	if (open == 1) {
		res->map_open = 1;
		if (stages & 1u) {
			uint8_t *lsd = lsd_o ? lsd_o : (uint8_t *)malloc((size_t)w * h);
			orc_isolate_map_markers(map_rgb, w, h);
			orc_mask_marker_lines(map_rgb, w, h, lsd);
			uint64_t st[4];
			res->n_lines = orc_find_lines(lsd, w, h, max_gap, res->lines, st);
			res->rounds = st[0]; res->steps = st[1];
			uint32_t c = 0;
			for (size_t i = 0; i < (size_t)w * h; ++i) c += lsd[i] == 255;
			res->n_mask_px = c;
			if (!lsd_o) free(lsd);
		}
		if (stages & 4u) {
			uint8_t *ocr = ocr_o ? ocr_o : (uint8_t *)malloc((size_t)bw * bh);
			orc_ocr_preprocess(brq_rgb, bw, bh, ocr);
			if (!ocr_o) free(ocr);
		}
		if (stages & 8u) {
			uint8_t *sc = scales_o ? scales_o : (uint8_t *)calloc((size_t)bw * bh, 1);
			if (orc_find_scales_preprocess(brq_rgb, bw, bh, scales_start_y, sc) == 0) {
				double r;
				if (orc_calc_meters_to_px_ratio(anchors, n_anchors, sc, bw, bh, &r)) { res->mpx = r; res->has_mpx = 1; }
			}
			if (!scales_o) free(sc);
		}
	}
	if (!ui_rgba_o && ui) free(ui);
	free(map_rgb); free(brq_rgb);
	return open;
}

/* Frame-parallel batch driver for the CPU baseline ("independent frames parallel across the
 * batch").  Per-frame work is single-threaded inside (nested OpenMP
 * regions are serialised), so `threads` is the number of cores actually used. */
ORC_API int orc_process_batch(const uint8_t *frames, uint32_t n, uint32_t W, uint32_t H, int grayscale, uint32_t max_gap, uint32_t stages,
                              const uint32_t *anchors, uint32_t n_anchors, uint32_t scales_start_y, orc_frame_result *res, int threads) {
	trig_init();
#ifdef _OPENMP
	omp_set_max_active_levels(1);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
	for (int64_t f = 0; f < (int64_t)n; ++f) {
		orc_process_frame(frames + (size_t)f * W * H * 4, W, H, grayscale, max_gap, stages,
		                  anchors ? anchors + (size_t)f * 9 : NULL, n_anchors, scales_start_y, &res[f], NULL, NULL, NULL, NULL);
	}
	return 0;
}

/* ---- capture hand-off (src/capture.rs:33-63) ----------------------------------------------------------------
 * `crc32fast::hash(&frame)` (capture.rs:44): crc32fast 1.3.2 is a crates.io dependency (Cargo.lock:621-622), not vendored
 * under /root/reference.  It implements the standard CRC-32/IEEE 802.3 (reflected polynomial 0xEDB88320, init and
 * final xor 0xFFFFFFFF; check value crc("123456789") = 0xCBF43926) -- the same function as zlib's crc32, which the
 * tests use as the second opinion.  Restated here bit by bit. */
ORC_API uint32_t orc_crc32(const uint8_t *data, uint64_t n) {
	uint32_t c = 0xFFFFFFFFu;
	for (uint64_t i = 0; i < n; ++i) {
		c ^= data[i];
		for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
	}
	return c ^ 0xFFFFFFFFu;
}

/* The capture loop's duplicate rule (capture.rs:34,44-47): `last_frame_crc32` starts at 0; a frame is delivered iff its
 * CRC differs from last_frame_crc32, which then takes its value.  keep[i] = 1 for delivered frames; returns their count
 * and leaves the final last_frame_crc32 in *last (in: initial value). */
ORC_API uint32_t orc_capture_dedupe(const uint32_t *crcs, uint32_t n, uint32_t *last, uint8_t *keep) {
	uint32_t kept = 0;
	for (uint32_t i = 0; i < n; ++i) {
		keep[i] = crcs[i] != *last;
		if (keep[i]) { *last = crcs[i]; ++kept; }
	}
	return kept;
}
