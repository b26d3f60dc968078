// Microbenchmark 5: HBM read rate on MI355X for (a) a plain 16-B-per-lane streaming read and (b) the front end's
// pattern: one wave per 510-byte slot, two unaligned dwords per lane, several slots in flight per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t __attribute__((aligned(1))) u32u;
__global__ __launch_bounds__(256) void k_stream(const uint4 *in, size_t n16, uint32_t *out)
{
	uint32_t acc = 0;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
		const uint4 v = in[i];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
template <int DEPTH, int PITCH>
__global__ __launch_bounds__(256) void k_slots(const uint8_t *in, uint32_t nslots, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
	uint32_t acc = 0;
	for (uint32_t s = wave; s < nslots; s += nwaves * DEPTH) {
		uint32_t d0[DEPTH], d1[DEPTH];
#pragma unroll
		for (int q = 0; q < DEPTH; q++) {
			const uint32_t slot = s + q * nwaves;
			if (slot < nslots) {
				const uint8_t *b = in + (size_t)slot * PITCH;
				d0[q] = *(const u32u *)(b + 4 * lane);
				d1[q] = (lane < 63) ? *(const u32u *)(b + 256 + 4 * lane) : 0;
			} else d0[q] = d1[q] = 0;
		}
#pragma unroll
		for (int q = 0; q < DEPTH; q++) acc ^= d0[q] ^ d1[q];
	}
	if (acc == 0x12345678u) out[0] = acc;
}
template <typename F> static float timeit(F f)
{
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	f(); (void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	for (int i = 0; i < 10; i++) f();
	(void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main()
{
	const uint32_t n = 1000000; const size_t bytes = (size_t)n * 512 + 1024;
	uint8_t *d; uint32_t *o; (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 4); (void)hipMemset(d, 1, bytes);
	for (int blocks : {2048, 4096, 8192}) {
		float ms = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const uint4 *)d, (size_t)n * 510 / 16, o); });
		printf("streaming uint4 read, %5d blocks: %.1f us  %.2f TB/s\n", blocks, ms * 1e3, n * 510.0 / ms / 1e9);
	}
	for (int blocks : {2048, 4096}) {
		float ms = timeit([&] { hipLaunchKernelGGL((k_slots<1, 510>), dim3(blocks), dim3(256), 0, 0, d, n, o); });
		printf("slot pattern pitch 510 depth 1, %5d blocks: %.1f us  %.2f TB/s\n", blocks, ms * 1e3, n * 510.0 / ms / 1e9);
		ms = timeit([&] { hipLaunchKernelGGL((k_slots<3, 510>), dim3(blocks), dim3(256), 0, 0, d, n, o); });
		printf("slot pattern pitch 510 depth 3, %5d blocks: %.1f us  %.2f TB/s\n", blocks, ms * 1e3, n * 510.0 / ms / 1e9);
		ms = timeit([&] { hipLaunchKernelGGL((k_slots<6, 510>), dim3(blocks), dim3(256), 0, 0, d, n, o); });
		printf("slot pattern pitch 510 depth 6, %5d blocks: %.1f us  %.2f TB/s\n", blocks, ms * 1e3, n * 510.0 / ms / 1e9);
		ms = timeit([&] { hipLaunchKernelGGL((k_slots<3, 512>), dim3(blocks), dim3(256), 0, 0, d, n, o); });
		printf("slot pattern pitch 512 depth 3, %5d blocks: %.1f us  %.2f TB/s (aligned slots)\n", blocks, ms * 1e3, n * 510.0 / ms / 1e9);
	}
	return 0;
}
