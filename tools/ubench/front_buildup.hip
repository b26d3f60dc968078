// Microbenchmark 6: where the front end's time goes.  The slot read pattern of hbm_read.hip (three slots in
// flight per wave, registers rotated), with the kernel's other stages added one at a time:
//   STAGE 0  loads only (xor-accumulate)
//   STAGE 1  + park in the wave's LDS window, read one dword back
//   STAGE 2  + ten LDS byte gathers (pseudo-random addresses inside the window)
//   STAGE 3  + ten ballots and twenty v_writelane
//   STAGE 4  + 80-byte store per slot
//   STAGE 5  + per-slot descriptor through a scalar load (offset | type << 56)
//   STAGE 6  = 5 with the descriptor through a broadcast vector load + v_readfirstlane (in-order vmcnt)
//   STAGE 7  = 5 + gather address set chosen by the descriptor's type (two sets, scalar branch)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t __attribute__((aligned(1))) u32u;

__device__ __forceinline__ uint64_t vdesc(const uint64_t *desc, uint32_t s)
{
	const uint64_t v = __builtin_nontemporal_load(desc + s);	/* same address in every lane */
	return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
}

template <int STAGE, int SW = 20, int SP = 20>
__global__ __launch_bounds__(256) void k(const uint8_t *in, const uint64_t *desc, uint32_t nslots, uint32_t *packed, uint32_t *out)
{
	__shared__ uint32_t s_win[4][128];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * 4 + wib, nwaves = gridDim.x * 4;
	uint32_t *mine = s_win[wib];
	const uint8_t *mine8 = (const uint8_t *)mine;
	uint32_t addr[10];
	for (int r = 0; r < 10; r++)
		addr[r] = (lane * 37 + r * 101 + (lane >> 3) * 7) % 510;
	uint32_t acc = 0;
	uint32_t a0, a1, b0, b1, c0, c1;
	uint32_t slot = wave;
	uint32_t addr2[10];
	for (int r = 0; r < 10; r++)
		addr2[r] = (lane * 41 + r * 97 + (lane >> 2) * 3) % 510;
	uint32_t ty = 0;
#define DESC(s) (STAGE == 6 ? vdesc(desc, s) : desc[s])
#define OFF(s) (STAGE >= 5 ? (size_t)((dtmp = DESC(s)) & 0x00ffffffffffffffull) : (size_t)(s) * 510)
	uint64_t dtmp = 0;
#define FETCH(R0, R1, s) { const uint8_t *b_ = in + OFF(s); R0 = *(const u32u *)(b_ + 4 * lane); R1 = *(const u32u *)(b_ + 256 + 4 * lane - (lane == 63 ? 2 : 0)); }
	if (slot >= nslots)
		return;
	const uint32_t last = slot + ((nslots - 1 - slot) / nwaves) * nwaves;	/* this wave's last slot */
#define CL(s) ((s) < nslots ? (s) : last)
	FETCH(a0, a1, slot) FETCH(b0, b1, CL(slot + nwaves)) FETCH(c0, c1, CL(slot + 2 * nwaves))
#define STEP(R0, R1)											\
	{												\
		uint32_t w_ = R0 ^ R1;									\
		if (STAGE >= 1) {									\
			mine[lane] = R0;								\
			mine[64 + lane] = R1;								\
		}											\
		FETCH(R0, R1, CL(slot + 3 * nwaves))							\
		ty = (uint32_t)(dtmp >> 56);								\
		if (STAGE >= 1)										\
			w_ ^= mine[(lane * 5) & 127];							\
		if (STAGE >= 2) {									\
			uint32_t by_[10];								\
			if (STAGE == 7 && (ty & 1)) {							\
				for (int r = 0; r < 10; r++)						\
					by_[r] = mine8[addr2[r]];					\
			} else {									\
				for (int r = 0; r < 10; r++)						\
					by_[r] = mine8[addr[r]];					\
			}										\
			if (STAGE >= 3) {								\
				unsigned long long bal_[10];						\
				for (int r = 0; r < 10; r++)						\
					bal_[r] = __ballot(by_[r] != 0);				\
				for (int r = 0; r < 10; r++)						\
					asm("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"	\
					    : "+v"(w_) : "s"((uint32_t)bal_[r]), "i"(2 * r), "s"((uint32_t)(bal_[r] >> 32)), "i"(2 * r + 1)); \
			} else {									\
				for (int r = 0; r < 10; r++)						\
					w_ ^= by_[r];							\
			}										\
		}											\
		if (STAGE >= 4) {									\
			if (lane < SW)									\
				packed[(size_t)slot * SP + lane] = w_;					\
		} else											\
			acc ^= w_;									\
		slot += nwaves;										\
	}
	while (slot < nslots) {		/* every slot is processed; requests past the end re-read the last slot */
		STEP(a0, a1)
		if (slot >= nslots) break;
		STEP(b0, b1)
		if (slot >= nslots) break;
		STEP(c0, c1)
	}
	if (acc == 0x12345678u) out[0] = acc;
}


// STAGE 8: as stage 4, but a wave takes four CONSECUTIVE slots (a group), stages their 4 x 20 output dwords in LDS
// and writes the 320 contiguous bytes with two store instructions
__global__ __launch_bounds__(256) void k8(const uint8_t *in, uint32_t nslots, uint32_t *packed, uint32_t *out)
{
	__shared__ uint32_t s_win[4][128];
	__shared__ uint32_t s_out[4][80];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * 4 + wib, nwaves = gridDim.x * 4;
	uint32_t *mine = s_win[wib];
	uint32_t *mo = s_out[wib];
	const uint8_t *mine8 = (const uint8_t *)mine;
	uint32_t addr[10];
	for (int r = 0; r < 10; r++)
		addr[r] = (lane * 37 + r * 101 + (lane >> 3) * 7) % 510;
	uint32_t a0, a1, b0, b1, c0, c1;
	const uint32_t ngroups = nslots / 4;
	uint32_t g = wave;	/* group index; slot = 4 g + j */
	if (g >= ngroups)
		return;
	const uint32_t lastg = g + ((ngroups - 1 - g) / nwaves) * nwaves;
	uint32_t j = 0;
#define SLOT(gg, jj) ((size_t)(4 * (gg) + (jj)) * 510)
#define FETCH8(R0, R1, off) { const uint8_t *b_ = in + (off); R0 = *(const u32u *)(b_ + 4 * lane); R1 = *(const u32u *)(b_ + 256 + 4 * lane - (lane == 63 ? 2 : 0)); }
	FETCH8(a0, a1, SLOT(g, 0)) FETCH8(b0, b1, SLOT(g, 1)) FETCH8(c0, c1, SLOT(g, 2))
	/* slot sequence per wave: (g,0) (g,1) (g,2) (g,3) (g+N,0) ...; the request three ahead */
#define STEP8(R0, R1)											\
	{												\
		mine[lane] = R0;									\
		mine[64 + lane] = R1;									\
		{											\
			uint32_t j3 = j + 3, g3 = g;							\
			if (j3 >= 4) { j3 -= 4; g3 += nwaves; }						\
			if (g3 >= ngroups) { g3 = lastg; j3 = 3; }					\
			FETCH8(R0, R1, SLOT(g3, j3))							\
		}											\
		uint32_t w_ = 0, by_[10];								\
		for (int r = 0; r < 10; r++)								\
			by_[r] = mine8[addr[r]];							\
		unsigned long long bal_[10];								\
		for (int r = 0; r < 10; r++)								\
			bal_[r] = __ballot(by_[r] != 0);						\
		for (int r = 0; r < 10; r++)								\
			asm("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"	\
			    : "+v"(w_) : "s"((uint32_t)bal_[r]), "i"(2 * r), "s"((uint32_t)(bal_[r] >> 32)), "i"(2 * r + 1)); \
		if (lane < 20)										\
			mo[j * 20 + lane] = w_;								\
		if (++j == 4) {										\
			uint32_t *dst = packed + (size_t)g * 80;					\
			dst[lane] = mo[lane];								\
			if (lane < 16)									\
				dst[64 + lane] = mo[64 + lane];						\
			j = 0;										\
			g += nwaves;									\
		}											\
	}
	while (g < ngroups) {		/* every group is processed; requests past the end re-read the last slot */
		STEP8(a0, a1)
		if (g >= ngroups) break;
		STEP8(b0, b1)
		if (g >= ngroups) break;
		STEP8(c0, c1)
	}
	if (a0 == 0x12345678u) out[0] = a0;
}

// group read patterns alone (2040 contiguous bytes per wave and step, two groups in flight, xor-accumulate):
// W = 16: two 16-byte loads per lane (8-byte aligned addresses), W = 8: four 8-byte loads, W = 4: eight dwords
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v4u __attribute__((aligned(1))) v4uu;
typedef v2u __attribute__((aligned(1))) v2uu;
template <int W>
__global__ __launch_bounds__(256) void kg(const uint8_t *in, uint32_t nslots, uint32_t *out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
	const uint32_t ngroups = nslots / 4;
	uint32_t acc = 0;
	for (uint32_t g = wave; g < ngroups; g += 2 * nwaves) {
		for (int q = 0; q < 2; q++) {
			const uint32_t gg = g + q * nwaves < ngroups ? g + q * nwaves : g;
			const uint8_t *b = in + (size_t)gg * 2040;
			if (W == 16) {
				const v4u x0 = *(const v4uu *)(b + 16 * lane);
				const v4u x1 = *(const v4uu *)(b + (lane == 63 ? 2024 : 1024 + 16 * lane));
				acc ^= x0.x ^ x0.y ^ x0.z ^ x0.w ^ x1.x ^ x1.y ^ x1.z ^ x1.w;
			} else if (W == 8) {
				for (int k = 0; k < 4; k++) {
					const uint32_t o = 512 * k + 8 * lane;
					const v2u x = *(const v2uu *)(b + (o + 8 > 2040 ? 2032 : o));
					acc ^= x.x ^ x.y;
				}
			} else {
				for (int k = 0; k < 8; k++) {
					const uint32_t o = 256 * k + 4 * lane;
					acc ^= *(const u32u *)(b + (o + 4 > 2040 ? 2036 : o));
				}
			}
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}

template <typename F> static float timeit(F f)
{
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for (int i = 0; i < 5; i++) f();
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	for (int i = 0; i < 20; i++) f();
	(void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 20;
}

int main()
{
	const uint32_t n = 1000000; const size_t bytes = (size_t)n * 510 + 4096;
	uint8_t *d; uint32_t *o, *p; uint64_t *ds;
	(void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 4); (void)hipMalloc(&p, (size_t)n * 128 + 4096); (void)hipMalloc(&ds, (size_t)n * 8);
	(void)hipMemset(d, 1, bytes);
	uint64_t *h = (uint64_t *)malloc((size_t)n * 8);
	for (uint32_t i = 0; i < n; i++) h[i] = (uint64_t)i * 510 | ((uint64_t)(i & 1) << 56);
	(void)hipMemcpy(ds, h, (size_t)n * 8, hipMemcpyHostToDevice);
	const int blocks = 8192;
#define RUN(S) { float ms = timeit([&] { hipLaunchKernelGGL((k<S>), dim3(blocks), dim3(256), 0, 0, d, ds, n, p, o); }); \
		 printf("stage %d: %.1f us  %.2f TB/s of input\n", S, ms * 1e3, n * 510.0 / ms / 1e9); }
#define RUNW(SW, SP) { float ms = timeit([&] { hipLaunchKernelGGL((k<4, SW, SP>), dim3(blocks), dim3(256), 0, 0, d, ds, n, p, o); }); \
		 printf("stage 4, %d dwords stored at a pitch of %d dwords: %.1f us\n", SW, SP, ms * 1e3); }
#define RUNG(W) { float ms = timeit([&] { hipLaunchKernelGGL((kg<W>), dim3(blocks), dim3(256), 0, 0, d, n, o); }); printf("group reads, %d bytes per load: %.1f us  %.2f TB/s\n", W, ms * 1e3, n * 510.0 / ms / 1e9); }
	RUNG(16) RUNG(8) RUNG(4) RUNG(16)
	RUNW(20, 20) RUNW(16, 16)
	{ float ms = timeit([&] { hipLaunchKernelGGL(k8, dim3(blocks / 4), dim3(256), 0, 0, d, n, p, o); }); printf("stage 8 (groups of 4 slots, 2048 blocks): %.1f us\n", ms * 1e3); }
	{ float ms = timeit([&] { hipLaunchKernelGGL(k8, dim3(blocks), dim3(256), 0, 0, d, n, p, o); }); printf("stage 8 (groups of 4 slots, 8192 blocks): %.1f us\n", ms * 1e3); }
	RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
	return 0;
}
