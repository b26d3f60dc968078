// Microbenchmark (round 4): what does rocprofv3's FETCH_SIZE count for the trellis kernels' read pattern?
// MI355X_MICROARCH.md calibrates "FETCH_SIZE = half the bytes" for wide coalesced streaming reads (16 B per lane) only.
// k_vit<432> reads 18 dwords per lane from packed slots 80 bytes apart, every other slot (the others are k_vit<216>'s):
//   k_stream16      : 16 B per lane, coalesced, over the whole 80 MB array            (the calibrated case)
//   k_slots<1, 18>  : lane = slot, 18 dwords of every slot                            (every byte of 72 of 80 used)
//   k_slots<2, 18>  : lane = every other slot, 18 dwords                              (k_vit<432>: half the slots)
//   k_slots<2, 9>   : 9 dwords of every other slot                                    (one block of k_vit<216>'s)
// rocprofv3 --pmc FETCH_SIZE over this binary: KiB per launch next to the bytes each variant needs / touches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_stream16(const uint4 *in, size_t n16, uint32_t *out)
{
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
		const uint4 v = in[i];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

template <int SEL, int NW>
__global__ __launch_bounds__(64) void k_slots(const uint32_t *packed, uint32_t nslots, uint32_t *out)
{
	const uint32_t i = (blockIdx.x * 64 + threadIdx.x) * SEL;
	if (i >= nslots)
		return;
	const uint32_t *pw = packed + (size_t)i * 20;
	uint32_t acc = 0;
#pragma unroll
	for (int g = 0; g < NW; g++)
		acc ^= pw[g];
	if (acc == 0x12345678u)
		out[0] = acc;
}

int main()
{
	const uint32_t n = 1000000;
	uint32_t *d, *o;
	(void)hipMalloc(&d, (size_t)n * 80 + 4096);
	(void)hipMalloc(&o, 4);
	(void)hipMemset(d, 1, (size_t)n * 80);
	// a 600 MB sweep between the launches keeps the array out of the Infinity Cache (256 MB)
	uint4 *big;
	(void)hipMalloc(&big, (size_t)600 << 20);
	(void)hipMemset(big, 2, (size_t)600 << 20);
	for (int rep = 0; rep < 3; rep++) {
		hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, big, ((size_t)600 << 20) / 16, o);
		hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, (const uint4 *)d, (size_t)n * 80 / 16, o);
		hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, big, ((size_t)600 << 20) / 16, o);
		hipLaunchKernelGGL((k_slots<1, 18>), dim3((n + 63) / 64), dim3(64), 0, 0, d, n, o);
		hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, big, ((size_t)600 << 20) / 16, o);
		hipLaunchKernelGGL((k_slots<2, 18>), dim3((n / 2 + 63) / 64), dim3(64), 0, 0, d, n, o);
		hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, big, ((size_t)600 << 20) / 16, o);
		hipLaunchKernelGGL((k_slots<2, 9>), dim3((n / 2 + 63) / 64), dim3(64), 0, 0, d, n, o);
	}
	(void)hipDeviceSynchronize();
	printf("array: %u slots x 80 B = %.1f MB; k_slots<1,18> needs 72 MB, <2,18> 36 MB, <2,9> 18 MB\n", n, n * 80 / 1e6);
	return 0;
}
