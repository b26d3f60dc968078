// Does gfx950 execute scalar stores (s_store_dwordx4 + s_dcache_wb), and how fast?  Each wave writes 80 bytes
// (5 x dwordx4) per "slot" from SGPRs that a VALU compare has just produced.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t nslots, uint32_t seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
	const uint32_t nwaves = gridDim.x * 4;
	for (uint32_t slot = wave; slot < nslots; slot += nwaves) {
		uint32_t w[20];
#pragma unroll
		for (int r = 0; r < 10; r++) {
			const uint32_t x = (slot * 2654435761u + seed + r * 40503u) >> (lane & 15);
			const unsigned long long b = __ballot((x & 1) != 0);
			w[2 * r] = (uint32_t)b;
			w[2 * r + 1] = (uint32_t)(b >> 32);
		}
		uint32_t *dst = out + (size_t)slot * 20;
#pragma unroll
		for (int q = 0; q < 5; q++) {
			typedef uint32_t u4 __attribute__((ext_vector_type(4)));
			u4 v = { w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3] };
			asm volatile("s_nop 4\n\ts_store_dwordx4 %0, %1, %2" : : "s"(v), "s"(dst), "n"(16 * q) : "memory");
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
int main()
{
	const uint32_t n = 1000000;
	uint32_t *d; (void)hipMalloc(&d, (size_t)n * 80); (void)hipMemset(d, 0xee, (size_t)n * 80);
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, n, 7u);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	(void)hipEventRecord(a);
	for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, n, 7u);
	(void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	std::vector<uint32_t> h((size_t)n * 20);
	(void)hipMemcpy(h.data(), d, (size_t)n * 80, hipMemcpyDeviceToHost);
	size_t bad = 0;
	for (uint32_t slot = 0; slot < n; slot++)
		for (int r = 0; r < 10; r++) {
			unsigned long long want = 0;
			for (uint32_t lane = 0; lane < 64; lane++) {
				const uint32_t x = (slot * 2654435761u + 7u + r * 40503u) >> (lane & 15);
				want |= (unsigned long long)(x & 1) << lane;
			}
			if (h[(size_t)slot * 20 + 2 * r] != (uint32_t)want || h[(size_t)slot * 20 + 2 * r + 1] != (uint32_t)(want >> 32)) bad++;
		}
	printf("scalar stores: %zu bad of %u rounds; %.3f ms per launch (1M slots x 80 B)\n", bad, n * 10, ms / 10);
	return 0;
}
