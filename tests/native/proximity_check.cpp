// Host check of csrc/smh_proximity.h (built and run by tests/test_proximity.py): the cheap signed-distance classifier in
// front of lsd.rs's near-line test must never disagree with the exact f32 test where it claims to be sure, and
// prox_filter_word must equal the bit-by-bit exact filter.  Random and adversarial segments over the 4K coordinate range.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "smh_proximity.h"

using namespace smh;

static uint32_t exact_filter(uint32_t surv, float px0, float py, const ProxLine &L) {
	for (uint32_t b = 0; b < 32; ++b)
		if (((surv >> b) & 1u) && near_line(px0 + (float)b, py, L.x0, L.y0, L.x1, L.y1)) surv &= ~(1u << b);
	return surv;
}

int main(int argc, char **argv) {
	const long iters = argc > 1 ? atol(argv[1]) : 400000;
	std::mt19937_64 rng(12345);
	std::uniform_real_distribution<double> U(0.0, 1.0);
	long words = 0, sure = 0, ring = 0, bad = 0;
	float max_near = 0.0f, min_far = 1e9f;   // extreme classifier distances of exactly-near / exactly-far pixels
	for (long it = 0; it < iters; ++it) {
		const double range = (it & 1) ? 4096.0 : 1400.0;
		// end points on the half-pixel grid like get_centre's results (lsd.rs:5-44), or arbitrary floats
		auto coord = [&](double r) { const double v = U(rng) * r; return (it % 3) ? (float)(floor(v * 2.0) / 2.0) : (float)v; };
		float x0 = coord(range), y0 = coord(range), x1, y1;
		const int kind = (int)(it % 7);
		if (kind == 0) { x1 = x0; y1 = y0; }                                   // degenerate
		else if (kind == 1) { x1 = x0 + (float)(U(rng) * 2000 - 1000); y1 = y0 + (float)((U(rng) - 0.5) * 2.0); }  // nearly horizontal
		else if (kind == 2) { y1 = y0 + (float)(U(rng) * 2000 - 1000); x1 = x0 + (float)((U(rng) - 0.5) * 2.0); }  // nearly vertical
		else if (kind == 3) { x1 = x0 + (float)((U(rng) - 0.5) * 120); y1 = y0 + (float)((U(rng) - 0.5) * 120); } // short (around the acceptance length)
		else { x1 = coord(range); y1 = coord(range); }
		const ProxLine L = prox_line(x0, y0, x1, y1);
		const double dx = (double)x1 - x0, dy = (double)y1 - y0, len = sqrt(dx * dx + dy * dy);
		for (int rep = 0; rep < 24; ++rep) {
			// a word whose pixels sit around the sqrt(50) boundary (adversarial), on the line, or anywhere
			double t = (U(rng) * 1.4 - 0.2), off;
			const int where = rep % 4;
			if (where == 0) off = 7.0710678 + (U(rng) - 0.5) * 0.5;
			else if (where == 1) off = -(7.0710678 + (U(rng) - 0.5) * 0.5);
			else if (where == 2) off = (U(rng) - 0.5) * 30.0;
			else off = (U(rng) - 0.5) * 3000.0;
			double cx, cy;
			if (len > 0) { cx = x0 + t * dx - off * dy / len; cy = y0 + t * dy + off * dx / len; }
			else { cx = x0 + off; cy = y0 + (U(rng) - 0.5) * 16; }
			const float py = (float)floor(cy), px0 = (float)(floor(cx) - (double)(rng() % 32));
			if (!(py > -64.0f && py < 4160.0f && px0 > -64.0f && px0 < 4160.0f)) continue;
			const uint32_t surv = (uint32_t)rng() | (uint32_t)(rng() << 7);
			const uint32_t got = prox_filter_word(surv, px0, py, L), want = exact_filter(surv, px0, py, L);
			++words;
			if (got != want) {
				if (++bad < 10) fprintf(stderr, "MISMATCH line (%.9g,%.9g)-(%.9g,%.9g) word px0=%.1f py=%.1f surv=%08x got=%08x want=%08x\n", x0, y0, x1, y1, px0, py, surv, got, want);
			}
			if (!L.degenerate) {
				const float s0 = (float)(((double)px0 - (double)L.x0) * L.dyl - ((double)py - (double)L.y0) * L.dxl);
				for (uint32_t b = 0; b < 32; ++b) {
					const float d = fabsf(s0 + (float)b * L.a);
					const bool ex = near_line(px0 + (float)b, py, x0, y0, x1, y1);
					if (ex && d > max_near) max_near = d;
					if (!ex && d < min_far) min_far = d;
					if (d < SMH_PROX_SURE_NEAR) { ++sure; if (!ex) { if (++bad < 10) fprintf(stderr, "sure-near but exact far: d=%.6f\n", d); } }
					else if (d > SMH_PROX_SURE_FAR) { ++sure; if (ex) { if (++bad < 10) fprintf(stderr, "sure-far but exact near: d=%.6f\n", d); } }
					else ++ring;
				}
			}
		}
	}
	printf("words %ld sure %ld ring %ld bad %ld max_near %.5f min_far %.5f\n", words, sure, ring, bad, max_near, min_far);
	return bad ? 1 : 0;
}
