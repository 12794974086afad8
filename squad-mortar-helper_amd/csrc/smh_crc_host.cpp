// smh_crc_host.cpp -- CRC-32/IEEE of a captured frame on the HOST cores (x86-64, carry-less multiply), for the ingest queue's
// region-of-interest upload mode (smh_runtime.cpp, SMHV_INGEST_ROI_UPLOAD): the reference's capture thread hashes the WHOLE
// frame (crc32fast::hash, src/capture.rs:34,44-47) and drops it when the hash equals the previous capture's, while the vision
// pipeline reads only the map ROI and the button rectangle (39 % of a 1080p frame).  Hashing on the device means uploading all
// of it first; hashing where the bytes are lets the PCIe link carry the 39 % -- and nothing at all for a duplicate.
//
// The algorithm is the published one ("Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Gopal et al., Intel 2009;
// the same folding crc32fast 1.3.2 and zlib-ng use): four 128-bit lanes folded 512 bits at a time, folded into one, reduced to
// 64 and then 32 bits (Barrett).  The folding constants are derived HERE from the polynomial (x^n mod P, bit-reflected) rather
// than copied from a table, and tests/test_host_abi.py checks the result against zlib for ragged lengths and alignments.
// Without PCLMULQDQ (or on another architecture) a slicing-by-8 table loop does the same job at a tenth of the speed.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../include/smh_vision_hip.h"
#include "../../include/smh_vision_hip_debug.h"

namespace {

constexpr uint32_t POLY_REFLECTED = 0xEDB88320u;               // CRC-32/IEEE 802.3, reflected

struct Tables {
	uint32_t t[8][256];
	Tables() {
		for (uint32_t i = 0; i < 256; ++i) {
			uint32_t c = i;
			for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ POLY_REFLECTED : c >> 1;
			t[0][i] = c;
		}
		for (uint32_t i = 0; i < 256; ++i)
			for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 255u];
	}
};
const Tables &tables() { static const Tables T; return T; }

// state -> state over `n` bytes (state = ~crc convention: start with 0xFFFFFFFF, finish with ~state)
uint32_t crc_table_update(uint32_t st, const uint8_t *p, size_t n) {
	const Tables &T = tables();
	while (n && ((uintptr_t)p & 7u)) { st = (st >> 8) ^ T.t[0][(st ^ *p++) & 255u]; --n; }
	while (n >= 8) {
		uint64_t v;
		memcpy(&v, p, 8);
		v ^= st;
		st = T.t[7][v & 255u] ^ T.t[6][(v >> 8) & 255u] ^ T.t[5][(v >> 16) & 255u] ^ T.t[4][(v >> 24) & 255u] ^
		     T.t[3][(v >> 32) & 255u] ^ T.t[2][(v >> 40) & 255u] ^ T.t[1][(v >> 48) & 255u] ^ T.t[0][(v >> 56) & 255u];
		p += 8; n -= 8;
	}
	while (n--) st = (st >> 8) ^ T.t[0][(st ^ *p++) & 255u];
	return st;
}

#if defined(__x86_64__)
}  // namespace
#include <immintrin.h>
namespace {

// x^n mod P as a bit-reflected 33-bit constant in the form the folding step wants: reflect(x^n mod P) << 1 (bit i of the result
// = coefficient of x^(32 - i)): the reflected-domain carry-less products come out one bit short, the shift puts it back.
uint64_t fold_constant(uint32_t n) {
	// r = x^n mod P in the normal (non-reflected) representation, computed bit by bit; P = 0x104C11DB7
	uint32_t r = 1u;                                            // x^0
	for (uint32_t i = 0; i < n; ++i) r = (r & 0x80000000u) ? (r << 1) ^ 0x04C11DB7u : r << 1;
	uint64_t refl = 0;
	for (int b = 0; b < 32; ++b) if ((r >> b) & 1u) refl |= 1ull << (31 - b);
	return refl << 1;
}
// floor(x^64 / P), reflected, 33 bits (Barrett constant)
uint64_t barrett_mu() {
	// polynomial long division of x^64 by P (33 bits)
	uint64_t q = 0, rem_hi = 1;                                 // remainder register holds the current top 33 bits
	unsigned __int128 num = (unsigned __int128)1 << 64, P = 0x104C11DB7ull;
	for (int i = 64; i >= 32; --i) {
		if ((num >> i) & 1) { q |= 1ull << (i - 32); num ^= P << (i - 32); }
	}
	(void)rem_hi;
	uint64_t refl = 0;                                          // q has 33 bits (x^32 .. x^0): reflect over 33 bits
	for (int b = 0; b <= 32; ++b) if ((q >> b) & 1ull) refl |= 1ull << (32 - b);
	return refl;
}
struct Consts { uint64_t k1, k2, k3, k4, k5, px, mu, w1, w2; };
const Consts &consts() {
	// distances of the folds in bits: 512 + 64 / 512 (four lanes), 128 + 64 / 128 (one lane), 64 (the 96 -> 64 step); w1 / w2: 2048 bits
	// (four 512-bit registers of four lanes each: the VPCLMULQDQ loop)
	static const Consts c = {fold_constant(4 * 128 + 32), fold_constant(4 * 128 - 32), fold_constant(128 + 32), fold_constant(128 - 32), fold_constant(64),
	                         // P reflected over 33 bits
	                         [] { uint64_t p = 0x104C11DB7ull, r = 0; for (int b = 0; b <= 32; ++b) if ((p >> b) & 1ull) r |= 1ull << (32 - b); return r; }(),
	                         barrett_mu(), fold_constant(16 * 128 + 32), fold_constant(16 * 128 - 32)};
	return c;
}

__attribute__((target("pclmul,sse4.1"))) inline __m128i fold(__m128i a, __m128i b, __m128i k) {
	return _mm_xor_si128(_mm_xor_si128(b, _mm_clmulepi64_si128(a, k, 0x00)), _mm_clmulepi64_si128(a, k, 0x11));
}

__attribute__((target("pclmul,sse4.1"))) uint32_t crc_clmul_finish(__m128i x3, __m128i x2, __m128i x1, __m128i x0, const __m128i *d, size_t n);
// state -> state over n bytes, n >= 64
__attribute__((target("pclmul,sse4.1"))) uint32_t crc_clmul_update(uint32_t st, const uint8_t *p, size_t n) {
	const Consts &c = consts();
	const __m128i *d = (const __m128i *)p;
	__m128i x3 = _mm_loadu_si128(d), x2 = _mm_loadu_si128(d + 1), x1 = _mm_loadu_si128(d + 2), x0 = _mm_loadu_si128(d + 3);
	d += 4; n -= 64;
	x3 = _mm_xor_si128(x3, _mm_cvtsi32_si128((int)st));
	const __m128i k1k2 = _mm_set_epi64x((long long)c.k2, (long long)c.k1);
	while (n >= 64) {
		x3 = fold(x3, _mm_loadu_si128(d), k1k2);
		x2 = fold(x2, _mm_loadu_si128(d + 1), k1k2);
		x1 = fold(x1, _mm_loadu_si128(d + 2), k1k2);
		x0 = fold(x0, _mm_loadu_si128(d + 3), k1k2);
		d += 4; n -= 64;
	}
	return crc_clmul_finish(x3, x2, x1, x0, d, n);
}
// the four 128-bit lanes (oldest first) folded into one, the remaining whole 16-byte blocks, 128 -> 64 -> 32 bits, the ragged tail
__attribute__((target("pclmul,sse4.1"))) uint32_t crc_clmul_finish(__m128i x3, __m128i x2, __m128i x1, __m128i x0, const __m128i *d, size_t n) {
	const Consts &c = consts();
	const __m128i k3k4 = _mm_set_epi64x((long long)c.k4, (long long)c.k3);
	__m128i x = fold(x3, x2, k3k4);
	x = fold(x, x1, k3k4);
	x = fold(x, x0, k3k4);
	while (n >= 16) { x = fold(x, _mm_loadu_si128(d), k3k4); ++d; n -= 16; }
	// 128 -> 64 bits, then 64 -> 32 by Barrett reduction
	const __m128i lo32 = _mm_set_epi32(0, 0, 0, -1);
	x = _mm_xor_si128(_mm_clmulepi64_si128(x, k3k4, 0x10), _mm_srli_si128(x, 8));
	x = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x, lo32), _mm_set_epi64x(0, (long long)c.k5), 0x00), _mm_srli_si128(x, 4));
	const __m128i pu = _mm_set_epi64x((long long)c.mu, (long long)c.px);
	const __m128i t1 = _mm_clmulepi64_si128(_mm_and_si128(x, lo32), pu, 0x10);
	const __m128i t2 = _mm_clmulepi64_si128(_mm_and_si128(t1, lo32), pu, 0x00);
	uint32_t out = (uint32_t)_mm_extract_epi32(_mm_xor_si128(x, t2), 1);
	if (n) out = crc_table_update(out, (const uint8_t *)d, n);
	return out;
}
bool have_clmul() {
	static const bool v = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	return v;
}

// The same folding 2048 bits at a time: four 512-bit registers, each four independent 128-bit lanes, one VPCLMULQDQ pair per
// register and step (Zen 4 / Zen 5, Ice Lake and later: where CPUID offers VPCLMULQDQ + AVX-512F/VL).  The sixteen lanes all fold
// by the same distance, 2048 bits; at the end the four registers are folded into one (512 bits apart), its four lanes go through the
// 128-bit finish above.  The ingest workers hash a frame in pieces of a few rows (61 KB): the set-up is nothing against that.
#define SMH_V512 __attribute__((target("avx512f,avx512vl,avx512bw,vpclmulqdq,pclmul,sse4.1")))
SMH_V512 inline __m512i fold512(__m512i a, __m512i b, __m512i k) {
	return _mm512_ternarylogic_epi64(b, _mm512_clmulepi64_epi128(a, k, 0x00), _mm512_clmulepi64_epi128(a, k, 0x11), 0x96);
}
// state -> state over n bytes, n >= 256
SMH_V512 uint32_t crc_vclmul_update(uint32_t st, const uint8_t *p, size_t n) {
	const Consts &c = consts();
	const __m512i *d = (const __m512i *)p;
	__m512i y0 = _mm512_loadu_si512(d), y1 = _mm512_loadu_si512(d + 1), y2 = _mm512_loadu_si512(d + 2), y3 = _mm512_loadu_si512(d + 3);
	d += 4; n -= 256;
	y0 = _mm512_xor_si512(y0, _mm512_zextsi128_si512(_mm_cvtsi32_si128((int)st)));
	const __m512i kw = _mm512_broadcast_i32x4(_mm_set_epi64x((long long)c.w2, (long long)c.w1));
	while (n >= 256) {
		y0 = fold512(y0, _mm512_loadu_si512(d), kw);
		y1 = fold512(y1, _mm512_loadu_si512(d + 1), kw);
		y2 = fold512(y2, _mm512_loadu_si512(d + 2), kw);
		y3 = fold512(y3, _mm512_loadu_si512(d + 3), kw);
		d += 4; n -= 256;
	}
	const __m512i k512 = _mm512_broadcast_i32x4(_mm_set_epi64x((long long)c.k2, (long long)c.k1));
	__m512i y = fold512(y0, y1, k512);
	y = fold512(y, y2, k512);
	y = fold512(y, y3, k512);
	while (n >= 64) { y = fold512(y, _mm512_loadu_si512(d), k512); ++d; n -= 64; }
	return crc_clmul_finish(_mm512_extracti32x4_epi32(y, 0), _mm512_extracti32x4_epi32(y, 1), _mm512_extracti32x4_epi32(y, 2), _mm512_extracti32x4_epi32(y, 3), (const __m128i *)d, n);
}
bool have_vclmul() {
	static const bool v = __builtin_cpu_supports("vpclmulqdq") && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
	                      __builtin_cpu_supports("avx512bw") && have_clmul();
	return v;
}
#else
bool have_clmul() { return false; }
bool have_vclmul() { return false; }
uint32_t crc_clmul_update(uint32_t st, const uint8_t *, size_t) { return st; }
uint32_t crc_vclmul_update(uint32_t st, const uint8_t *, size_t) { return st; }
#endif

}  // namespace

// Incremental form for the ingest workers (smh_runtime.cpp), which hash a frame row by row and copy the rows the pipeline
// reads while they are still in the cache: state in, state out; start with 0xFFFFFFFF, the CRC is ~state.
namespace smh {
uint32_t crc32_host_update(uint32_t st, const uint8_t *p, size_t n) {
	if (n == 0) return st;
	if (n >= 512 && have_vclmul()) return crc_vclmul_update(st, p, n);
	return have_clmul() && n >= 64 ? crc_clmul_update(st, p, n) : crc_table_update(st, p, n);
}
// which of the three loops this machine gets: 2 = VPCLMULQDQ (512-bit), 1 = PCLMULQDQ (128-bit), 0 = tables
int crc32_host_level() { return have_vclmul() ? 2 : (have_clmul() ? 1 : 0); }
}  // namespace smh

// diagnostic: the CRC by ONE of the three loops (2 = VPCLMULQDQ, 1 = PCLMULQDQ, 0 = tables; a level the machine does not have falls
// back to the next lower one) -- the tests run every loop the machine has against zlib; level < 0 only reports the machine's level
extern "C" SMHV_API int smhv_debug_crc32_host_level(const void *data, uint64_t nbytes, int level, uint32_t *crc) {
	const int have = smh::crc32_host_level();
	if (level < 0 || !data || !crc) return have;
	const uint8_t *p = (const uint8_t *)data;
	const size_t n = (size_t)nbytes;
	uint32_t st = 0xFFFFFFFFu;
	if (n) {
		if (level >= 2 && have >= 2 && n >= 512) st = crc_vclmul_update(st, p, n);
		else if (level >= 1 && have >= 1 && n >= 64) st = crc_clmul_update(st, p, n);
		else st = crc_table_update(st, p, n);
	}
	*crc = n ? ~st : 0u;
	return have;
}

// CRC-32/IEEE of nbytes at data (== crc32fast::hash == zlib crc32(0, ..)); any length, any alignment.  Host only: needs no device.
extern "C" SMHV_API uint32_t smhv_crc32_host(const void *data, uint64_t nbytes) {
	if (!data || nbytes == 0) return 0u;
	return ~smh::crc32_host_update(0xFFFFFFFFu, (const uint8_t *)data, (size_t)nbytes);
}
