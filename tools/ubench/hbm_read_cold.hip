// Microbenchmark (round 5): what the HBM read path gives when the input is COLD -- NB distinct 510 MB buffers taken in
// turn, as bench.py rotates its captures -- against the same buffer read over and over (half of which then sits in the
// 256 MB Infinity Cache).  (a) plain 16-B-per-lane streaming read, (b) the stream front end's group pattern: a wave takes
// groups of 2040 bytes (wave, wave + nwaves, ...), 16 bytes per lane from the aligned address below + 128 more, the next
// group requested before this one is consumed, (c) the same with the 320 bytes of packed slots written per group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
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
template <int STORE>
__global__ __launch_bounds__(256) void k_groups(const uint8_t *in, uint32_t ngroups, uint32_t *packed, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
	uint32_t acc = 0;
	if (wave >= ngroups) return;
	auto fetch = [&](uint32_t g, uint4 &a, uint4 &b, uint4 &c) {
		const uint8_t *p = in + 100 + (size_t)g * 2040;
		const uint8_t *b16 = p - ((uintptr_t)p & 15);
		a = *(const uint4 *)(b16 + 16 * lane);
		b = *(const uint4 *)(b16 + 1024 + 16 * lane);
		c = *(const uint4 *)(b16 + 2048 + 16 * (lane < 7 ? lane : 7));
	};
	uint4 a0, b0, c0, a1, b1, c1;
	uint32_t g = wave;
	fetch(g, a0, b0, c0);
	for (;;) {
		const uint32_t g1 = g + nwaves;
		fetch(g1 < ngroups ? g1 : g, a1, b1, c1);
		uint32_t v = a0.x ^ a0.y ^ a0.z ^ a0.w ^ b0.x ^ b0.y ^ b0.z ^ b0.w ^ c0.x ^ c0.y ^ c0.z ^ c0.w;
		acc ^= v;
		if (STORE == 1) { packed[(size_t)g * 80 + lane] = v; if (lane < 16) packed[(size_t)g * 80 + 64 + lane] = v; } else if (STORE == 2) packed[(size_t)g * 64 + lane] = v;
		if (g1 >= ngroups) break;
		const uint32_t g2 = g1 + nwaves;
		fetch(g2 < ngroups ? g2 : g1, a0, b0, c0);
		v = a1.x ^ a1.y ^ a1.z ^ a1.w ^ b1.x ^ b1.y ^ b1.z ^ b1.w ^ c1.x ^ c1.y ^ c1.z ^ c1.w;
		acc ^= v;
		if (STORE == 1) { packed[(size_t)g1 * 80 + lane] = v; if (lane < 16) packed[(size_t)g1 * 80 + 64 + lane] = v; } else if (STORE == 2) packed[(size_t)g1 * 64 + lane] = v;
		if (g2 >= ngroups) break;
		g = g2;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
	const int NB = 8;
	const uint32_t n = 1000000; const size_t bytes = (size_t)n * 510 + 4096;
	std::vector<uint8_t *> d(NB);
	uint32_t *o, *pk;
	std::vector<uint32_t *> pks(NB);
	for (int i = 0; i < NB; i++) (void)hipMalloc(&pks[i], (size_t)n * 80 + 4096);
	for (int i = 0; i < NB; i++) { (void)hipMalloc(&d[i], bytes); (void)hipMemset(d[i], 1 + i, bytes); }
	(void)hipMalloc(&o, 4); (void)hipMalloc(&pk, (size_t)n * 80 + 4096);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto run = [&](const char *name, int nb, auto launch) {
		for (int i = 0; i < 2 * NB; i++) launch(d[i % nb]);
		(void)hipDeviceSynchronize();
		float tot = 0, mn = 1e9f;
		for (int i = 0; i < 24; i++) {
			(void)hipEventRecord(e0);
			launch(d[i % nb]);
			(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1); tot += ms; if (ms < mn) mn = ms;
		}
		printf("%-44s %d buffer(s): mean %.1f us (min %.1f)  %.2f TB/s of input\n", name, nb, tot / 24 * 1e3, mn * 1e3, n * 510.0 / (tot / 24) / 1e9);
	};
	const uint32_t ngroups = n / 4;
	for (int nb : {1, NB}) {
		for (int blocks : {4096, 8192})
			run(blocks == 4096 ? "streaming uint4 read, 4096 blocks" : "streaming uint4 read, 8192 blocks", nb,
			    [&](uint8_t *p) { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const uint4 *)p, (size_t)n * 510 / 16, o); });
		for (int blocks : {2560, 3072, 5120})
			run(blocks == 2560 ? "group pattern, 2560 blocks" : blocks == 3072 ? "group pattern, 3072 blocks" : "group pattern, 5120 blocks", nb,
			    [&](uint8_t *p) { hipLaunchKernelGGL(k_groups<0>, dim3(blocks), dim3(256), 0, 0, p, ngroups, pk, o); });
		run("group pattern + packed-slot stores, 2560 blocks", nb,
		    [&](uint8_t *p) { hipLaunchKernelGGL(k_groups<1>, dim3(2560), dim3(256), 0, 0, p, ngroups, pk, o); });
		run("group pattern + stores, OUTPUT rotating too", nb,
		    [&](uint8_t *p) { int w = 0; for (int i = 0; i < NB; i++) if (d[i] == p) w = i; hipLaunchKernelGGL(k_groups<1>, dim3(2560), dim3(256), 0, 0, p, ngroups, pks[w], o); });
		run("group pattern + 64-byte packed slots, 2560 blocks", nb,
		    [&](uint8_t *p) { hipLaunchKernelGGL(k_groups<2>, dim3(2560), dim3(256), 0, 0, p, ngroups, pk, o); });
	}
	return 0;
}
