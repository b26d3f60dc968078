// Microbenchmark 3: cycles per trellis step of tg_vit_block() alone (registers only: no loads, no history
// stores), as a function of waves per SIMD.  Build twice to compare two versions of vit_core.h:
//   hipcc --offload-arch=gfx950 -O3 -I<dir with vit_core.h> vit_core_rate.hip -o vit_core_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "vit_core.h"

// -DUSE_BM: branch metrics from the LDS table (tg_vit_block_bm) instead of arithmetic
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, int iters)
{
	tg_vit_state v;
	tg_vit_init(v);
	uint32_t x = seed * (threadIdx.x + 1) + blockIdx.x, acc = 0;
#ifdef USE_BM
	__shared__ uint32_t s_bm[TG_BM_WORDS];
	for (int i = threadIdx.x; i < 32; i += 64)
		tg_bm_entry(i >> 3, i & 7, s_bm + 8 * i);
	__syncthreads();
	auto bm = [&](int p, uint32_t e, uint32_t w[6]) {
		const uint4 a = *(const uint4 *)(s_bm + (8 * p + e) * 8);
		const uint2 b = *(const uint2 *)(s_bm + (8 * p + e) * 8 + 4);
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y;
	};
#define tg_vit_block tg_vit_block_bm
#define BMARG , bm
#else
#define BMARG
#endif
	tg_vit_leadin(v, x & 63);
	for (int it = 0; it < iters; it++) {
		uint32_t h[4];
		x = x * 1664525u + 1013904223u;
		tg_vit_block<false>(v, x >> 8, h BMARG);
		acc ^= h[0] ^ h[1] ^ h[2] ^ h[3];
		tg_vit_block<false>(v, x >> 20, h BMARG);
		acc += h[0] ^ h[1] ^ h[2] ^ h[3];
		if ((it & 3) == 3)
			tg_vit_normalize(v);
	}
	out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
	uint32_t *d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	for (int w : {1, 2, 3, 4, 8}) {
		const int iters = 2000, blocks = 256 * 4 * w;
		hipEvent_t a, b;
		(void)hipEventCreate(&a); (void)hipEventCreate(&b);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 12345u, 10);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(a);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 12345u, iters);
		(void)hipEventRecord(b);
		(void)hipEventSynchronize(b);
		float ms; (void)hipEventElapsedTime(&ms, a, b);
		const double steps = (double)iters * 16 * w;	/* wave-steps per SIMD */
		printf("waves/SIMD=%d  %7.3f ms  %.2f ns per wave-step per SIMD = %.1f cyc @2.4GHz\n", w, ms, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
	}
	return 0;
}
