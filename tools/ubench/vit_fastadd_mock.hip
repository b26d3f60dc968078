// Microbenchmark (round 5): what would whole-word v_add_u32 in place of v_pk_add_u16 buy the difference-form trellis?
// On gfx950 v_add_u32 / v_and_b32 issue at ~2.5 cycles per wave in runs, v_pk_add_u16 / v_pk_min_u16 at ~4.4
// (profiles/r01_ubench_valu_issue_rates.txt).  A whole-word add cannot broadcast a half, so the states would have to be
// re-paired every step (4 phases, three of them broadcast-free) and every register would need its own table dword.
// This file does NOT compute a trellis in the mock forms: it runs the instruction mix such a kernel would have --
//   form 0: vit_core.h's difference form as it is (tg_vit_block_bmd)
//   form 1: the same number of adds / mins / table dwords, adds as v_add_u32 on whole registers in steps 0, 1, 2 of
//           every four (runs of 8 / 16 / 8 adds in front of 8 mins each), packed adds with op_sel in step 3; table
//           dwords per 4 steps: 4 + 16 + 4 + 8
//   form 2: as form 1 but every add a v_pk_add_u16 (the cost of the extra table reads alone)
// and prints cycles per trellis step per SIMD at 1..8 waves.  hipcc --offload-arch=gfx950 -O3 -I<csrc> vit_fastadd_mock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "vit_core.h"

template <bool FAST>
__device__ __forceinline__ tg_us2 mock_add(tg_us2 a, tg_us2 c)
{
	if (FAST)
		return tg_as_us2(tg_as_u32(a) + tg_as_u32(c));
	return tg_pk_add_sel<0, 1, 0, 1>(a, c);
}

// a 2-op step without broadcasts: 8 adds, 8 mins, 4 table dwords
template <bool FAST>
__device__ __forceinline__ void mock_d2(tg_vit_state &v, const uint32_t *w)
{
	tg_us2 Y[8], N[8];
#pragma unroll
	for (int m = 0; m < 4; m++) {
		Y[2 * m] = mock_add<FAST>(v.Z[4 + m], tg_as_us2(w[m & 1]));
		Y[2 * m + 1] = mock_add<FAST>(v.Z[4 + m], tg_as_us2(w[2 + (m & 1)]));
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int m = 0; m < 4; m++) {
		N[2 * m] = tg_pk_min_sel<0, 1, 0, 1>(v.Z[m], Y[2 * m]);
		N[2 * m + 1] = tg_pk_min_sel<0, 1, 0, 1>(v.Z[m], Y[2 * m + 1]);
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = N[(5 * k + 3) & 7];	/* (some re-pairing) */
}

// a 3-op step without broadcasts: 16 adds, 8 mins, 16 table dwords
template <bool FAST>
__device__ __forceinline__ void mock_d3(tg_vit_state &v, const uint32_t *w)
{
	tg_us2 X[8], Y[8], N[8];
#pragma unroll
	for (int m = 0; m < 4; m++) {
		X[2 * m] = mock_add<FAST>(v.Z[m], tg_as_us2(w[4 * m]));
		X[2 * m + 1] = mock_add<FAST>(v.Z[m], tg_as_us2(w[4 * m + 1]));
		Y[2 * m] = mock_add<FAST>(v.Z[4 + m], tg_as_us2(w[4 * m + 2]));
		Y[2 * m + 1] = mock_add<FAST>(v.Z[4 + m], tg_as_us2(w[4 * m + 3]));
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int k = 0; k < 8; k++)
		N[k] = tg_pk_min_sel<0, 1, 0, 1>(X[k], Y[k]);
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = N[(3 * k + 1) & 7];
}

template <int FORM>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, int iters)
{
	tg_vit_state v;
	tg_vit_init(v);
	uint32_t x = seed * (threadIdx.x + 1) + blockIdx.x, acc = 0;
	__shared__ __attribute__((aligned(16))) uint32_t s_bm[2048];
	if (threadIdx.x < 32) {
		uint32_t w[10];
		tg_bmd_entry(threadIdx.x >> 3, threadIdx.x & 7, w);
		tg_bmd_store(s_bm, (int)threadIdx.x, w);
	}
	for (int i = TG_BMD_WORDS + threadIdx.x; i < 2048; i += 64)
		s_bm[i] = 0x00010001u * (i & 3);
	__syncthreads();
	auto bmd = [&](int p, uint32_t o, uint32_t w[10]) {
		const uint8_t *q = (const uint8_t *)s_bm + o;
		const uint4 a = *(const uint4 *)(q + 4 * TG_BMD_A0 + 128 * p);
		const uint4 b = *(const uint4 *)(q + 4 * TG_BMD_A1 + 128 * p);
		const uint2 c = *(const uint2 *)(q + 4 * TG_BMD_A2 + 128 * p);
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; w[8] = c.x; w[9] = c.y;
	};
	// mock table: 8 entries per pair position, 32 dwords (128 bytes) per entry and cycle of two pairs
	auto cyc = [&](int c, uint32_t o, uint32_t w[32]) {
		const uint4 *q = (const uint4 *)((const uint8_t *)s_bm + 2048 + 1024 * c + 8 * o);	/* o = 16 e: 128 e */
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const uint4 a = q[i];
			w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w;
		}
	};
	// form 3: form 0 without its table reads (ten dwords that stay in registers): what the LDS costs the loop
	uint32_t fixed[10];
	{
		uint32_t w0[10];
		bmd(0, (x << 4) & 0x70u, w0);
#pragma unroll
		for (int i = 0; i < 10; i++) {
			fixed[i] = w0[i];
			asm volatile("" : "+v"(fixed[i]));
		}
	}
	auto bmreg = [&](int, uint32_t o, uint32_t w[10]) {
#pragma unroll
		for (int i = 0; i < 10; i++)
			w[i] = fixed[i];
		acc ^= o;	/* (keeps the index arithmetic alive) */
	};
	tg_vit_leadin_bmd(v, x & 63, bmd);
	for (int it = 0; it < iters; it++) {
		uint32_t h[4];
		x = x * 1664525u + 1013904223u;
		if (FORM == 3) {
			tg_vit_block_bmd<false>(v, x >> 8, h, bmreg);
			acc ^= h[0] ^ h[1] ^ h[2] ^ h[3];
			tg_vit_block_bmd<false>(v, x >> 20, h, bmreg);
			acc += h[0] ^ h[1] ^ h[2] ^ h[3];
		} else if (FORM == 0) {
			tg_vit_block_bmd<false>(v, x >> 8, h, bmd);
			acc ^= h[0] ^ h[1] ^ h[2] ^ h[3];
			tg_vit_block_bmd<false>(v, x >> 20, h, bmd);
			acc += h[0] ^ h[1] ^ h[2] ^ h[3];
		} else {
#pragma unroll
			for (int blk = 0; blk < 2; blk++) {
				const uint32_t tw = blk ? x >> 20 : x >> 8;
#pragma unroll
				for (int c = 0; c < 2; c++) {
					uint32_t w[32];
					cyc(c, (c ? tw >> 2 : tw << 4) & 0x70u, w);
					mock_d2<FORM == 1>(v, w);		/* 4 dwords */
					mock_d3<FORM == 1>(v, w + 4);		/* 16 */
					mock_d2<FORM == 1>(v, w + 20);		/* 4 */
					tg_acs_d3(v, w + 22);			/* 8, packed adds with op_sel */
				}
#pragma unroll
				for (int d = 0; d < 4; d++)
					h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
				tg_vit_clean(v);
				acc += h[0] ^ h[1] ^ h[2] ^ h[3];
			}
		}
		if ((it & 3) == 3)
			tg_vit_normalize_floor(v);
	}
	out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int FORM>
static void run(uint32_t *d)
{
	for (int w : {1, 2, 3, 4, 8}) {
		const int iters = 2000, blocks = 256 * 4 * w;
		hipEvent_t a, b;
		(void)hipEventCreate(&a); (void)hipEventCreate(&b);
		hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(64), 0, 0, d, 12345u, 10);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(a);
		hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(64), 0, 0, d, 12345u, iters);
		(void)hipEventRecord(b);
		(void)hipEventSynchronize(b);
		float ms; (void)hipEventElapsedTime(&ms, a, b);
		const double steps = (double)iters * 16 * w;	/* wave-steps per SIMD */
		printf("form %d  waves/SIMD=%d  %7.3f ms  %.2f ns per wave-step per SIMD = %.1f cyc @2.4GHz\n", FORM, w, ms, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
	}
}

int main()
{
	uint32_t *d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	run<0>(d);
	run<1>(d);
	run<2>(d);
	run<3>(d);
	return 0;
}
