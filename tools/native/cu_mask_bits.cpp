// Probe: which CU (XCC, shader engine, CU) does bit b of a hipExtStreamCreateWithCUMask mask select on this device?
// build (on the GPU box): hipcc --offload-arch=gfx950 -O2 tools/native/cu_mask_bits.cpp -o /tmp/cu_mask_bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_where(unsigned *out, int spin) {
	unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
	unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
	volatile int x = 0;
	for (int i = 0; i < spin; ++i) x += i;
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
	const int nb = 64;
	unsigned *d; hipMalloc(&d, nb * 8);
	std::vector<unsigned> h(2 * nb);
	for (int b = 0; b < 256; ++b) {
		if (b >= 72 && b % 32 != 0 && b % 32 != 31 && b != 255) continue;
		std::vector<uint32_t> mask(8, 0u);
		mask[b / 32] = 1u << (b % 32);
		hipStream_t s;
		if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("bit %d: create failed\n", b); continue; }
		hipMemsetAsync(d, 0, nb * 8, s);
		hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, s, d, 2000);
		hipStreamSynchronize(s);
		hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
		std::set<unsigned> cus;
		for (int i = 0; i < nb; ++i) { unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF; cus.insert((xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)); }
		printf("bit %3d ->", b);
		for (auto c : cus) printf(" xcc %u se %u sh %u cu %u;", c >> 12, (c >> 8) & 7, (c >> 4) & 1, c & 0xF);
		printf("\n");
		hipStreamDestroy(s);
	}
	return 0;
}
