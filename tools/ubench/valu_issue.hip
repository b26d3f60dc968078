// Issue-rate microbenchmark for the instructions the trellis kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
// For each instruction pattern: cycles per instruction per SIMD at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	uint32_t c = 0x00010001u + threadIdx.x;
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) {	// dependent v_pk_add_u16
			asm volatile(REP16("v_pk_add_u16 %0, %0, %1\n") : "+v"(a0) : "v"(c));
			asm volatile(REP16("v_pk_add_u16 %0, %0, %1\n") : "+v"(a0) : "v"(c));
			asm volatile(REP16("v_pk_add_u16 %0, %0, %1\n") : "+v"(a0) : "v"(c));
			asm volatile(REP16("v_pk_add_u16 %0, %0, %1\n") : "+v"(a0) : "v"(c));
		} else if (MODE == 1) {	// 8 independent chains of v_pk_add_u16
#define I8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
			asm volatile(I8("v_pk_add_u16") I8("v_pk_add_u16") I8("v_pk_add_u16") I8("v_pk_add_u16") I8("v_pk_add_u16") I8("v_pk_add_u16") I8("v_pk_add_u16") I8("v_pk_add_u16")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		} else if (MODE == 2) {	// dependent v_add_u32
			asm volatile(REP16("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(c));
			asm volatile(REP16("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(c));
			asm volatile(REP16("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(c));
			asm volatile(REP16("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(c));
		} else if (MODE == 3) {	// 8 independent v_add_u32
			asm volatile(I8("v_add_u32") I8("v_add_u32") I8("v_add_u32") I8("v_add_u32") I8("v_add_u32") I8("v_add_u32") I8("v_add_u32") I8("v_add_u32")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		} else if (MODE == 4) {	// the add-compare-select pattern: 8 butterflies, add add min, op_sel broadcasts
#define BF(d, a, b) "v_pk_add_u16 %8, " a ", %9 op_sel_hi:[0,1]\n" "v_pk_add_u16 %10, " b ", %9 op_sel:[1,0]\n" "v_pk_min_u16 " d ", %8, %10\n"
			uint32_t t0, t1;
			asm volatile(BF("%0", "%0", "%4") BF("%1", "%1", "%5") BF("%2", "%2", "%6") BF("%3", "%3", "%7")
				     BF("%4", "%0", "%4") BF("%5", "%1", "%5") BF("%6", "%2", "%6") BF("%7", "%3", "%7")
				     BF("%0", "%0", "%4") BF("%1", "%1", "%5") BF("%2", "%2", "%6") BF("%3", "%3", "%7")
				     BF("%4", "%0", "%4") BF("%5", "%1", "%5") BF("%6", "%2", "%6") BF("%7", "%3", "%7")
				     BF("%0", "%0", "%4") BF("%1", "%1", "%5") BF("%2", "%2", "%6") BF("%3", "%3", "%7")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "+v"(c), "=&v"(t1));
			asm volatile("v_pk_add_u16 %0, %0, %1\nv_pk_add_u16 %0, %0, %1\nv_pk_add_u16 %0, %0, %1\nv_pk_add_u16 %0, %0, %1\n" : "+v"(a0) : "v"(c));
		} else if (MODE == 5) {	// same pattern with distinct temporaries per butterfly (no WAR/WAW reuse)
			uint32_t t[16];
#define BG(d, a, b, x, y) "v_pk_add_u16 " x ", " a ", %24 op_sel_hi:[0,1]\n" "v_pk_add_u16 " y ", " b ", %24 op_sel:[1,0]\n"
#define BM(d, x, y) "v_pk_min_u16 " d ", " x ", " y "\n"
			asm volatile(BG("%0", "%0", "%4", "%8", "%9") BG("%1", "%1", "%5", "%10", "%11") BG("%2", "%2", "%6", "%12", "%13") BG("%3", "%3", "%7", "%14", "%15")
				     BG("%4", "%0", "%4", "%16", "%17") BG("%5", "%1", "%5", "%18", "%19") BG("%6", "%2", "%6", "%20", "%21") BG("%7", "%3", "%7", "%22", "%23")
				     BM("%0", "%8", "%9") BM("%1", "%10", "%11") BM("%2", "%12", "%13") BM("%3", "%14", "%15")
				     BM("%4", "%16", "%17") BM("%5", "%18", "%19") BM("%6", "%20", "%21") BM("%7", "%22", "%23")
				     BG("%0", "%0", "%4", "%8", "%9") BG("%1", "%1", "%5", "%10", "%11") BG("%2", "%2", "%6", "%12", "%13") BG("%3", "%3", "%7", "%14", "%15")
				     BG("%4", "%0", "%4", "%16", "%17") BG("%5", "%1", "%5", "%18", "%19") BG("%6", "%2", "%6", "%20", "%21") BG("%7", "%3", "%7", "%22", "%23")
				     BM("%0", "%8", "%9") BM("%1", "%10", "%11") BM("%2", "%12", "%13") BM("%3", "%14", "%15")
				     BM("%4", "%16", "%17") BM("%5", "%18", "%19") BM("%6", "%20", "%21") BM("%7", "%22", "%23")
				     BG("%0", "%0", "%4", "%8", "%9") BG("%1", "%1", "%5", "%10", "%11") BG("%2", "%2", "%6", "%12", "%13") BG("%3", "%3", "%7", "%14", "%15")
				     BM("%0", "%8", "%9") BM("%1", "%10", "%11") BM("%2", "%12", "%13") BM("%3", "%14", "%15")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
				       "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
				       "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15])
				     : "v"(c));
			asm volatile("v_pk_add_u16 %0, %0, %1\nv_pk_add_u16 %0, %0, %1\nv_pk_add_u16 %0, %0, %1\nv_pk_add_u16 %0, %0, %1\n" : "+v"(a0) : "v"(c));
		} else if (MODE == 6) {	// 8 independent v_pk_min_u16
			asm volatile(I8("v_pk_min_u16") I8("v_pk_min_u16") I8("v_pk_min_u16") I8("v_pk_min_u16") I8("v_pk_min_u16") I8("v_pk_min_u16") I8("v_pk_min_u16") I8("v_pk_min_u16")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		} else if (MODE == 7) {	// 8 independent v_and_b32 (plain VOP2)
			asm volatile(I8("v_and_b32") I8("v_and_b32") I8("v_and_b32") I8("v_and_b32") I8("v_and_b32") I8("v_and_b32") I8("v_and_b32") I8("v_and_b32")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		} else if (MODE == 8) {	// 8 independent v_pk_add_u16 with op_sel (broadcast of the low half)
#define I8S(op) op " %0, %0, %8 op_sel_hi:[0,1]\n" op " %1, %1, %8 op_sel_hi:[0,1]\n" op " %2, %2, %8 op_sel_hi:[0,1]\n" op " %3, %3, %8 op_sel_hi:[0,1]\n" op " %4, %4, %8 op_sel_hi:[0,1]\n" op " %5, %5, %8 op_sel_hi:[0,1]\n" op " %6, %6, %8 op_sel_hi:[0,1]\n" op " %7, %7, %8 op_sel_hi:[0,1]\n"
			asm volatile(I8S("v_pk_add_u16") I8S("v_pk_add_u16") I8S("v_pk_add_u16") I8S("v_pk_add_u16") I8S("v_pk_add_u16") I8S("v_pk_add_u16") I8S("v_pk_add_u16") I8S("v_pk_add_u16")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
		}
	}
	out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int MODE>
static void run(const char *name, int ninstr, uint32_t *d_out)
{
	const int iters = 20000;
	for (int wps = 1; wps <= 8; wps *= 2) {
		const int blocks = 1024 * wps;	// 256 CUs x 4 SIMDs x wps waves
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
		const double instr_per_simd = (double)ninstr * iters * wps;
		printf("%-44s waves/SIMD %d: %8.3f ms  %6.2f ns per instruction per SIMD (%5.2f cycles at 2.4 GHz)\n", name, wps, ms,
		       ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
	}
}

int main()
{
	uint32_t *d_out;
	hipMalloc(&d_out, 1024 * 8 * 64 * 4);
	run<0>("dependent v_pk_add_u16", 64, d_out);
	run<1>("8 independent v_pk_add_u16", 64, d_out);
	run<2>("dependent v_add_u32", 64, d_out);
	run<3>("8 independent v_add_u32", 64, d_out);
	run<7>("8 independent v_and_b32", 64, d_out);
	run<6>("8 independent v_pk_min_u16", 64, d_out);
	run<8>("8 independent v_pk_add_u16 op_sel", 64, d_out);
	run<4>("ACS pattern, 2 shared temporaries", 64, d_out);
	run<5>("ACS pattern, own temporaries", 64, d_out);
	return 0;
}
