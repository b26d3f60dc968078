// Microbenchmark 2: candidates for a cheaper add-compare-select on gfx950.
//   - is a packed f16 min (bit patterns 0x0400..0x7bff are positive normal halfs, ordered like the integers)
//     issued at full rate, and is it bit-exact as an unsigned 16-bit min on those patterns?
//   - rates of v_add3_u32 / v_alignbit_b32 / v_min_f32 / v_pk_min_i16 next to v_add_u32 and v_pk_min_u16
//   hipcc --offload-arch=gfx950 -O3 valu_min.hip -o valu_min && ./valu_min
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ us2 as2(uint32_t x) { return __builtin_bit_cast(us2, x); }
static __device__ __forceinline__ uint32_t asu(us2 x) { return __builtin_bit_cast(uint32_t, x); }
static __device__ __forceinline__ uint32_t minh(uint32_t a, uint32_t b)
{
	uint32_t r;
	asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
static __device__ __forceinline__ uint32_t minh_x(uint32_t a, uint32_t b)	/* lo = min(a.lo, b.hi), hi = min(a.hi, b.lo) */
{
	uint32_t r;
	asm volatile("v_pk_min_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

enum { M_ADD32, M_PKMINU, M_PKMINH, M_PKMINH_X, M_PKMINI, M_MINF32, M_ADD3, M_ALIGNBIT, M_NEWBFLY, M_OLDBFLY, M_BFI, M_LSHLOR, NMODES };
static const char *mode_name[NMODES] = { "v_add_u32", "v_pk_min_u16", "v_pk_min_f16", "v_pk_min_f16 op_sel", "v_pk_min_i16", "v_min_f32",
					 "v_add3_u32", "v_alignbit_b32", "2x add32 + pk_min_f16", "2x pk_add_u16 + pk_min_u16", "v_bfi_b32", "v_lshl_or_b32" };
static const int mode_instr[NMODES] = { 1, 1, 1, 1, 1, 1, 1, 1, 3, 3, 1, 1 };

template <int MODE>
__global__ void k(uint32_t *out, uint32_t seed, int iters)
{
	constexpr int ILP = 8;
	uint32_t z[ILP];
	for (int i = 0; i < ILP; i++) z[i] = ((seed * (i + 1) + threadIdx.x) & 0x3fff3fffu) | 0x04000400u;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 8; r++) {
#pragma unroll
			for (int i = 0; i < ILP; i++) {
				const uint32_t o = z[(i + 3) & 7], p = z[(i + 5) & 7];
				if (MODE == M_ADD32) z[i] = z[i] + o;
				if (MODE == M_PKMINU) z[i] = asu(__builtin_elementwise_min(as2(z[i]), as2(o)));
				if (MODE == M_PKMINH) z[i] = minh(z[i], o);
				if (MODE == M_PKMINH_X) z[i] = minh_x(z[i], o);
				if (MODE == M_PKMINI) z[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ss2, z[i]), __builtin_bit_cast(ss2, o)));
				if (MODE == M_MINF32) z[i] = __builtin_bit_cast(uint32_t, __builtin_fminf(__builtin_bit_cast(float, z[i]), __builtin_bit_cast(float, o)));
				if (MODE == M_ADD3) z[i] = z[i] + o + p;
				if (MODE == M_ALIGNBIT) z[i] = __builtin_amdgcn_alignbit(z[i], o, 16);
				if (MODE == M_NEWBFLY) z[i] = minh(z[i] + 0x01000200u, o + 0x02000101u);
				if (MODE == M_OLDBFLY) z[i] = asu(__builtin_elementwise_min(as2(z[i]).xx + as2(0x01000200u), as2(o).yy + as2(0x02000101u)));
				if (MODE == M_BFI) z[i] = (z[i] & p) | (o & ~p);
				if (MODE == M_LSHLOR) z[i] = (z[i] << 3) | o;
			}
		}
	}
	uint32_t acc = 0;
	for (int i = 0; i < ILP; i++) acc ^= z[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(int waves_per_simd, uint32_t *d)
{
	const int iters = 1000;
	const int blocks = 256 * 4 * waves_per_simd;
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, 12345u, 10);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, 12345u, iters);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	const double per_wave_instr = (double)iters * 8 * 8 * mode_instr[MODE];
	const double ns = ms * 1e6 / (per_wave_instr * waves_per_simd);
	printf("%-28s waves/SIMD=%d  %7.3f ms  %.3f ns/instr/SIMD  = %.2f cyc @2.4GHz\n", mode_name[MODE], waves_per_simd, ms, ns, ns * 2.4);
}

/* exhaustive check: all pairs (a, b) of 16-bit patterns in [lo, hi] */
__global__ void k_check(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first)
{
	const uint32_t a = lo + blockIdx.x;
	if (a > hi) return;
	for (uint32_t b = lo + threadIdx.x; b <= hi; b += blockDim.x) {
		const uint32_t x = a | (b << 16), y = b | (a << 16);
		const uint32_t r = minh(x, y), want = (a < b ? a : b) * 0x10001u;
		const uint32_t rx = minh_x(x, x);	/* lo = min(a, b), hi = min(b, a) */
		if (r != want || rx != want) {
			if (atomicAdd(bad, 1ull) == 0) { first[0] = x; first[1] = r; first[2] = rx; }
		}
	}
}

int main()
{
	uint32_t *d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	unsigned long long *bad; uint32_t *first;
	(void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 12);
	const uint32_t ranges[3][2] = { { 0x0400, 0x7bff }, { 0x0000, 0x03ff }, { 0x0000, 0x7bff } };
	for (int r = 0; r < 3; r++) {
		(void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 12);
		hipLaunchKernelGGL(k_check, dim3(ranges[r][1] - ranges[r][0] + 1), dim3(256), 0, 0, ranges[r][0], ranges[r][1], bad, first);
		unsigned long long hb; uint32_t hf[3];
		(void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 12, hipMemcpyDeviceToHost);
		printf("v_pk_min_f16 as u16 min on patterns [0x%04x, 0x%04x]: %llu mismatching pairs", ranges[r][0], ranges[r][1], hb);
		if (hb) printf(" (e.g. in 0x%08x -> 0x%08x / op_sel 0x%08x)", hf[0], hf[1], hf[2]);
		printf("\n");
	}
	for (int w : {1, 2, 4, 8}) {
		run<M_ADD32>(w, d); run<M_PKMINU>(w, d); run<M_PKMINH>(w, d); run<M_PKMINH_X>(w, d); run<M_PKMINI>(w, d); run<M_MINF32>(w, d);
		run<M_ADD3>(w, d); run<M_ALIGNBIT>(w, d); run<M_NEWBFLY>(w, d); run<M_OLDBFLY>(w, d); run<M_BFI>(w, d); run<M_LSHLOR>(w, d);
	}
	return 0;
}
