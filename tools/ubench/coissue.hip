// Do instructions of different classes from different waves of a SIMD issue side by side on gfx950?
//   hipcc --offload-arch=gfx950 -O2 -o coissue coissue.hip && ./coissue
// Per pattern: time of a block of 64 instructions per wave, at 1 / 2 / 4 / 8 waves per SIMD, as cycles per block per SIMD
// (2.4 GHz assumed).  If vector and scalar (or LDS) instructions of different waves overlap, the mixed block costs
// max(parts); if the SIMD issues one instruction per turn whatever its class, it costs their sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(x) x x x x x x x x
#define V "v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n"
#define V2 "v_xor_b32 %0, %0, %2\n"
#define S "s_add_u32 %1, %1, 3\n"
#define L "ds_read_b32 %4, %5\n"

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters)
{
	__shared__ uint32_t lds[64];
	lds[threadIdx.x] = threadIdx.x;
	uint32_t a = threadIdx.x, c = a * 7 + 1, d = a ^ 0x55, s = blockIdx.x, t = 0;
	const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)&lds[threadIdx.x];
	for (int i = 0; i < iters; i++) {
		if (MODE == 0)		// 64 VOP3
			asm volatile(R8(R8(V)) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 1)	// 64 SALU
			asm volatile(R8(R8(S)) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 2)	// 64 VOP3 + 64 SALU interleaved
			asm volatile(R8(R8(V S)) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 3)	// 64 VOP3 + 32 SALU
			asm volatile(R8(R8(V) R8(S V)) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 4)	// 64 VOP2
			asm volatile(R8(R8(V2)) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 5)	// 64 VOP2 + 64 SALU
			asm volatile(R8(R8(V2 S)) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 6)	// 16 LDS reads
			asm volatile(R8(L L) "s_waitcnt lgkmcnt(0)\n" : "+v"(a), "+s"(s), "=v"(t) : "v"(c), "v"(d), "v"(addr) : "scc");
		else if (MODE == 7)	// 64 VOP3 + 16 LDS reads
			asm volatile(R8(V V V V L V V V V L) "s_waitcnt lgkmcnt(0)\n" : "+v"(a), "+s"(s) : "v"(c), "v"(d), "v"(t), "v"(addr) : "scc");
		else if (MODE == 8)	// 64 VOP3 + 64 s_nop 0
			asm volatile(R8(R8(V "s_nop 0\n")) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
		else if (MODE == 9)	// 64 VOP3 + 64 s_waitcnt that never waits
			asm volatile(R8(R8(V "s_waitcnt lgkmcnt(15)\n")) : "+v"(a), "+s"(s) : "v"(c), "v"(d) : "scc");
	}
	out[blockIdx.x * 64 + threadIdx.x] = a ^ s ^ t;
}

template <int MODE>
static void run(const char *name, uint32_t *d_out)
{
	const int iters = 4000;
	printf("%-40s", name);
	for (int wps = 1; wps <= 8; wps *= 2) {
		const int blocks = 1024 * wps;
		hipEvent_t e0, e1;
		hipEventCreate(&e0);
		hipEventCreate(&e1);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 100);
		hipDeviceSynchronize();
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, iters);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("  %d w/SIMD: %7.1f", wps, ms * 1e6 * 2.4 / ((double)iters * wps));
	}
	printf("   cycles per block per SIMD\n");
}

int main()
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	uint32_t *d_out;
	hipMalloc(&d_out, 1024 * 8 * 64 * 4);
	run<0>("64 VOP3 (v_bitop3)", d_out);
	run<1>("64 SALU (s_add_u32)", d_out);
	run<2>("64 VOP3 + 64 SALU", d_out);
	run<3>("64 VOP3 + 32 SALU", d_out);
	run<4>("64 VOP2 (v_xor)", d_out);
	run<5>("64 VOP2 + 64 SALU", d_out);
	run<6>("16 ds_read_b32", d_out);
	run<7>("64 VOP3 + 16 ds_read_b32", d_out);
	run<8>("64 VOP3 + 64 s_nop 0", d_out);
	run<9>("64 VOP3 + 64 s_waitcnt (no wait)", d_out);
	return 0;
}
