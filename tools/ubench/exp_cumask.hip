// Experiment: can a memory-bound front-end stand-in (front_buildup's four-slot-group kernel) and a VALU-bound trellis
// stand-in (vit_core.h blocks) share the GPU on DISJOINT CU sets (hipExtStreamCreateWithCUMask)?  Times each alone
// on its CU set, alone on the whole GPU, and both together.
#define main fb_main
#include "front_buildup.hip"
#undef main
#include "vit_core.h"

__global__ __launch_bounds__(64) void kv(uint32_t *out, uint32_t seed, int iters)
{
	tg_vit_state v;
	tg_vit_init(v);
	uint32_t x = seed * (threadIdx.x + 1) + blockIdx.x, acc = 0;
	tg_vit_leadin(v, x & 63);
	for (int it = 0; it < iters; it++) {
		uint32_t h[4];
		x = x * 1664525u + 1013904223u;
		tg_vit_block<false>(v, x >> 8, h);
		acc ^= h[0] ^ h[1] ^ h[2] ^ h[3];
		tg_vit_block<false>(v, x >> 20, h);
		acc += h[0] ^ h[1] ^ h[2] ^ h[3];
		if ((it & 3) == 3)
			tg_vit_normalize(v);
	}
	out[blockIdx.x * 64 + threadIdx.x] = acc;
}

static hipStream_t masked(int ncu_front, bool front)
{
	// diagonal pick: ((i % 8) + (i / 8)) % 8 < k gives 4 k CUs, 4 k / 8 per XCD under either bit -> CU mapping
	uint32_t m[8] = { 0 };
	const int k = ncu_front / 32;
	for (int i = 0; i < 256; i++) {
		const bool f = ((i % 8) + (i / 8)) % 8 < k;
		if (f == front)
			m[i / 32] |= 1u << (i % 32);
	}
	hipStream_t s;
	if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("cu mask stream failed\n"); exit(1); }
	return s;
}

int main()
{
	const uint32_t n = 1000000; const size_t bytes = (size_t)n * 510 + 4096;
	uint8_t *d; uint32_t *o, *p, *ov;
	(void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 4); (void)hipMalloc(&p, (size_t)n * 128 + 4096);
	(void)hipMalloc(&ov, 256 * 4 * 8 * 64 * 4);
	(void)hipMemset(d, 1, bytes);
	const int vblocks = 256 * 4 * 4, viters = 70;	// about the trellis work of one config-2 step
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	auto T = [&](hipStream_t sf, hipStream_t sv, bool runf, bool runv) {
		float best = 1e9f;
		for (int rep = 0; rep < 6; rep++) {
			(void)hipDeviceSynchronize();
			(void)hipEventRecord(a, 0);
			(void)hipStreamWaitEvent(sf, a, 0); (void)hipStreamWaitEvent(sv, a, 0);
			if (runf) hipLaunchKernelGGL(k8, dim3(8192), dim3(256), 0, sf, d, n, p, o);
			if (runv) hipLaunchKernelGGL(kv, dim3(vblocks), dim3(64), 0, sv, ov, 12345u, viters);
			hipEvent_t ef, ev2; (void)hipEventCreate(&ef); (void)hipEventCreate(&ev2);
			(void)hipEventRecord(ef, sf); (void)hipEventRecord(ev2, sv);
			(void)hipStreamWaitEvent(0, ef, 0); (void)hipStreamWaitEvent(0, ev2, 0);
			(void)hipEventRecord(b, 0);
			(void)hipEventSynchronize(b);
			float ms; (void)hipEventElapsedTime(&ms, a, b);
			if (rep && ms < best) best = ms;
			(void)hipEventDestroy(ef); (void)hipEventDestroy(ev2);
		}
		return best * 1e3f;
	};
	hipStream_t s0, s1; (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
	printf("whole GPU: front %.1f us, trellis %.1f us, both on two plain streams %.1f us\n", T(s0, s1, true, false), T(s0, s1, false, true), T(s0, s1, true, true));
	for (int ncu : {32, 64, 96}) {
		hipStream_t sf = masked(ncu, true), sv = masked(ncu, false);
		printf("front on %3d CUs %.1f us | trellis on %3d CUs %.1f us | together %.1f us\n", ncu, T(sf, sv, true, false), 256 - ncu,
		       T(sf, sv, false, true), T(sf, sv, true, true));
	}
	{	// how the mask bits map to CUs: full mask, lower 224 / 128 bits, every second bit
		auto mk = [&](auto pred) { uint32_t m[8] = { 0 }; for (int i = 0; i < 256; i++) if (pred(i)) m[i / 32] |= 1u << (i % 32);
			hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("mask failed\n"); exit(1); } return s; };
		hipStream_t full = mk([](int) { return true; }), lo224 = mk([](int i) { return i < 224; }), lo128 = mk([](int i) { return i < 128; });
		hipStream_t even = mk([](int i) { return (i & 1) == 0; }), hi128 = mk([](int i) { return i >= 128; });
		printf("trellis: full mask %.1f | bits 0..223 %.1f | bits 0..127 %.1f | bits 128..255 %.1f | even bits %.1f us\n",
		       T(s0, full, false, true), T(s0, lo224, false, true), T(s0, lo128, false, true), T(s0, hi128, false, true), T(s0, even, false, true));
		printf("front:   full mask %.1f | bits 0..223 %.1f | bits 0..127 %.1f | bits 128..255 %.1f | even bits %.1f us\n",
		       T(full, s1, true, false), T(lo224, s1, true, false), T(lo128, s1, true, false), T(hi128, s1, true, false), T(even, s1, true, false));
	}
	return 0;
}
