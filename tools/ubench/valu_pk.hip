// Microbenchmark: issue rate of the trellis kernels' instruction mix on gfx950 as a function of
// waves per SIMD.  Every mode keeps 8 independent dependency chains per lane.
//   hipcc --offload-arch=gfx950 -O3 valu_pk.hip -o valu_pk && ./valu_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ us2 as2(uint32_t x) { return __builtin_bit_cast(us2, x); }
static __device__ __forceinline__ uint32_t asu(us2 x) { return __builtin_bit_cast(uint32_t, x); }

enum { M_PKADD, M_PKADD_SEL, M_PKMIN, M_PKMAD, M_ADD32, M_MULLO, M_PERM, M_BUTTERFLY, M_MIN32, M_AND, NMODES };
static const char *mode_name[NMODES] = { "v_pk_add_u16", "v_pk_add_u16 op_sel", "v_pk_min_u16", "v_pk_mad_u16", "v_add_u32",
					 "v_mul_lo_u32", "v_perm_b32", "2x pk_add(op_sel)+pk_min", "v_min_u32", "v_and_b32" };
static const int mode_instr[NMODES] = { 1, 1, 1, 1, 1, 1, 1, 3, 1, 1 };

template <int MODE>
__global__ void k(uint32_t *out, uint32_t seed, int iters)
{
	constexpr int ILP = 8;
	us2 z[ILP];
	for (int i = 0; i < ILP; i++) z[i] = as2(seed * (i + 1) + threadIdx.x);
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 8; r++) {
#pragma unroll
			for (int i = 0; i < ILP; i++) {
				const us2 o = z[(i + 3) & 7];
				if (MODE == M_PKADD) z[i] = z[i] + o;
				if (MODE == M_PKADD_SEL) z[i] = z[i].yy + o.yx;
				if (MODE == M_PKMIN) z[i] = __builtin_elementwise_min(z[i], o);
				if (MODE == M_PKMAD) z[i] = z[i].xx * o + as2(0x02000000u);
				if (MODE == M_ADD32) z[i] = as2(asu(z[i]) + asu(o));
				if (MODE == M_MULLO) z[i] = as2(asu(z[i]) * asu(o));
				if (MODE == M_PERM) z[i] = as2(__builtin_amdgcn_perm(asu(z[i]), asu(o), 0x06040200u));
				if (MODE == M_BUTTERFLY) z[i] = __builtin_elementwise_min(z[i].xx + o, z[(i + 5) & 7].yy + o.yx);
				if (MODE == M_MIN32) z[i] = as2(min(asu(z[i]), asu(o)));
				if (MODE == M_AND) z[i] = as2((asu(z[i]) & 0xff00ff00u) ^ asu(o));
			}
		}
	}
	uint32_t acc = 0;
	for (int i = 0; i < ILP; i++) acc ^= asu(z[i]);
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(int waves_per_simd, uint32_t *d)
{
	const int iters = 1000;
	const int blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks: one wave each
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, 12345u, 10);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, 12345u, iters);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	const double per_wave_instr = (double)iters * 8 * 8 * mode_instr[MODE] * (MODE == M_AND ? 2 : 1);
	const double ns = ms * 1e6 / (per_wave_instr * waves_per_simd);
	printf("%-28s waves/SIMD=%d  %7.3f ms  %.3f ns/instr/SIMD  = %.2f cyc @2.4GHz\n", mode_name[MODE], waves_per_simd, ms, ns, ns * 2.4);
}

int main()
{
	uint32_t *d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	for (int w : {1, 2, 4, 8}) {
		run<M_PKADD>(w, d); run<M_PKADD_SEL>(w, d); run<M_PKMIN>(w, d); run<M_PKMAD>(w, d); run<M_ADD32>(w, d);
		run<M_MULLO>(w, d); run<M_PERM>(w, d); run<M_BUTTERFLY>(w, d); run<M_MIN32>(w, d); run<M_AND>(w, d);
	}
	return 0;
}
