// Microbenchmark 4: when does v_add_u32 issue at ~2.4 cycles/wave and when at ~4?  Variants of operand patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
enum { V_ACC3, V_ACC4, V_ACC1, V_RING, V_BFLY_LIT, V_BFLY_VGPR, V_BFLY_SGPR, V_ADD_LIT, V_ADD_SGPR, V_INDEP_MIX, V_BFLY_VGPR_SPLIT, V_BFLY_LIT_SPLIT, NV };
static const char *name[NV] = { "z[i] += z[i+3]", "z[i] += z[i+4] (same bank?)", "z[i] += z[i+1]", "y[i] = z[i] + z[i+3] (ring of 16)",
	"min(z+lit, o+lit)", "min(z+vK1, o+vK2)", "min(z+sK1, o+sK2)", "z[i] += literal", "z[i] += sgpr",
	"indep: 2 add chains + 1 pk_min chain", "16 adds (vK) then 8 mins", "16 adds (lit) then 8 mins" };
static const int instr[NV] = { 1, 1, 1, 1, 3, 3, 3, 1, 1, 3, 3, 3 };

template <int V>
__global__ void k(uint32_t *out, uint32_t seed, int iters, uint32_t s1, uint32_t s2)
{
	uint32_t z[8], y[8];
	for (int i = 0; i < 8; i++) { z[i] = ((seed * (i + 1) + threadIdx.x) & 0x3fff3fffu) | 0x04000400u; y[i] = z[i] ^ 0x55; }
	uint32_t k1 = s1 + (threadIdx.x & 1), k2 = s2 + (threadIdx.x & 2);
	asm volatile("" : "+v"(k1), "+v"(k2));
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 8; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (V == V_ACC3) z[i] += z[(i + 3) & 7];
				if (V == V_ACC4) z[i] += z[(i + 4) & 7];
				if (V == V_ACC1) z[i] += z[(i + 1) & 7];
				if (V == V_RING) { if (r & 1) z[i] = y[i] + y[(i + 3) & 7]; else y[i] = z[i] + z[(i + 3) & 7]; }
				if (V == V_BFLY_LIT) z[i] = pkmin(z[i] + 0x01000200u, z[(i + 3) & 7] + 0x02000101u);
				if (V == V_BFLY_VGPR) z[i] = pkmin(z[i] + k1, z[(i + 3) & 7] + k2);
				if (V == V_BFLY_SGPR) z[i] = pkmin(z[i] + s1, z[(i + 3) & 7] + s2);
				if (V == V_ADD_LIT) z[i] += 0x01000200u + i;
				if (V == V_ADD_SGPR) z[i] += s1;
			}
		}
	}
	if (V == V_INDEP_MIX) {
		uint32_t a[8], b[8], m[8];
		for (int i = 0; i < 8; i++) { a[i] = z[i]; b[i] = y[i]; m[i] = z[i] ^ y[i]; }
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int r = 0; r < 8; r++) {
#pragma unroll
				for (int i = 0; i < 8; i++) {
					a[i] += a[(i + 3) & 7];
					b[i] += b[(i + 5) & 7];
					m[i] = pkmin(m[i], m[(i + 3) & 7]);
				}
			}
		}
		for (int i = 0; i < 8; i++) z[i] = a[i] ^ b[i] ^ m[i];
	}
	if (V == V_BFLY_VGPR_SPLIT || V == V_BFLY_LIT_SPLIT) {
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int r = 0; r < 8; r++) {
				uint32_t t[8], u[8];
#pragma unroll
				for (int i = 0; i < 8; i++) {
					t[i] = z[i] + (V == V_BFLY_LIT_SPLIT ? 0x01000200u : k1);
					u[i] = z[(i + 3) & 7] + (V == V_BFLY_LIT_SPLIT ? 0x02000101u : k2);
				}
#pragma unroll
				for (int i = 0; i < 8; i++) asm volatile("" : "+v"(t[i]), "+v"(u[i]));
#pragma unroll
				for (int i = 0; i < 8; i++)
					z[i] = pkmin(t[i], u[i]);
			}
		}
	}
	uint32_t acc = 0;
	for (int i = 0; i < 8; i++) acc ^= z[i] ^ y[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int V>
static void run(int w, uint32_t *d)
{
	const int iters = 1000, blocks = 256 * 4 * w;
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(64), 0, 0, d, 12345u, 10, 0x01000200u, 0x02000101u);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(64), 0, 0, d, 12345u, iters, 0x01000200u, 0x02000101u);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	const double ns = ms * 1e6 / ((double)iters * 64 * instr[V] * w);
	printf("%-36s waves/SIMD=%d  %.2f cyc/instr @2.4GHz\n", name[V], w, ns * 2.4);
}

int main()
{
	uint32_t *d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	for (int w : {2, 8}) {
		run<V_ACC3>(w, d); run<V_ACC4>(w, d); run<V_ACC1>(w, d); run<V_RING>(w, d); run<V_BFLY_LIT>(w, d);
		run<V_BFLY_VGPR>(w, d); run<V_BFLY_SGPR>(w, d); run<V_ADD_LIT>(w, d); run<V_ADD_SGPR>(w, d); run<V_INDEP_MIX>(w, d); run<V_BFLY_VGPR_SPLIT>(w, d); run<V_BFLY_LIT_SPLIT>(w, d);
	}
	return 0;
}
