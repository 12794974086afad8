// Probe: which CUs does a stream created with hipExtStreamCreateWithCUMask use on this device?
// build (on the GPU box): hipcc --offload-arch=gfx950 -O2 tools/native/cu_mask_probe.cpp -o tools/native/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ void k_where(unsigned *out, int spin) {
	unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
	unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
	volatile int x = 0;
	for (int i = 0; i < spin; ++i) x += i;
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main(int argc, char **argv) {
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	printf("device %s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
	const int nb = 8192;
	unsigned *d; hipMalloc(&d, nb * 8);
	std::vector<unsigned> h(2 * nb);
	auto run = [&](hipStream_t s, const char *name) -> std::set<unsigned> {
		hipMemsetAsync(d, 0, nb * 8, s);
		hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, s, d, 20000);
		hipStreamSynchronize(s);
		hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
		std::set<unsigned> cus; std::set<unsigned> xccs;
		for (int i = 0; i < nb; ++i) {
			unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
			unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
			cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
		}
		printf("%-28s distinct CUs used: %zu over %zu XCCs\n", name, cus.size(), xccs.size());
		return cus;
	};
	run(nullptr, "default stream");
	for (int variant = 0; variant < 4; ++variant) {
		std::vector<uint32_t> mask(8, 0u);
		const char *name = "";
		if (variant == 0) { for (int i = 0; i < 8; ++i) mask[i] = 0xFFFFFFFFu; name = "mask all 256"; }
		if (variant == 1) { for (int i = 0; i < 4; ++i) mask[i] = 0xFFFFFFFFu; name = "mask low 128 bits"; }
		if (variant == 2) { for (int i = 0; i < 8; ++i) mask[i] = 0x00FFFFFFu; name = "mask 24 of every 32 bits"; }
		if (variant == 3) { for (int i = 0; i < 8; ++i) mask[i] = 0x55555555u; name = "mask even bits"; }
		hipStream_t s;
		hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
		if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); continue; }
		run(s, name);
		hipStreamDestroy(s);
	}
	// complementary masks: every 32-bit word split at bit `cut`
	for (int cut : {8, 16, 24}) {
		std::vector<uint32_t> a(8), b(8);
		for (int i = 0; i < 8; ++i) { a[i] = (1u << cut) - 1u; b[i] = ~a[i]; }
		hipStream_t sa, sb;
		if (hipExtStreamCreateWithCUMask(&sa, 8, a.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&sb, 8, b.data()) != hipSuccess) { printf("create failed\n"); continue; }
		char na[64], nb_[64]; snprintf(na, 64, "low %d bits of each word", cut); snprintf(nb_, 64, "high %d bits of each word", 32 - cut);
		auto ca = run(sa, na), cb = run(sb, nb_);
		size_t common = 0; for (auto c : ca) common += cb.count(c);
		printf("   -> overlap %zu, union %zu\n", common, ca.size() + cb.size() - common);
		hipStreamDestroy(sa); hipStreamDestroy(sb);
	}
	return 0;
}
