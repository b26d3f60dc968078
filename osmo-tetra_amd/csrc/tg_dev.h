/*
 * tg_dev.h -- what the HIP translation units of the library share (round 5: tg_kernels.hip split by stage):
 *   tg_k_front.hip    slot / block / stream / soft front ends (rows D, I, U, F, S, B of SURVEY 8(a))
 *   tg_k_trellis.hip  k_vit, k_clean, k_bbk_blocks, k_burst(_ring), k_conv (rows X, V, C, R, L)
 *   tg_k_walk.hip     the synchroniser's walk on the device (row S; tg_walk_core.h)
 *   tg_k_aux.hip      tables, code fill, masks, item lists, device-walk mid stages, re-ordering, GSMTAP, stages
 * Every unit has its OWN copy of the constant tables (internal linkage: no relocatable device code); tgk_init() in
 * tg_k_aux.hip builds them once and hands them to each unit's upload function.
 *
 * No MFMA anywhere: there is no dense contraction on this path.  The trellis kernels are VALU-issue bound
 * packed-u16 integer work (one lane per trellis); the front kernels are byte gathers, part HBM, part issue bound
 * (DESIGN.md section 4).
 */
#ifndef TG_DEV_H
#define TG_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "tetra_gpu.h"
#include "tg_layout.h"
#include "vit_core.h"
#include "tg_internal.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

/* ------------------------------------------------------------------------- */
/* constant tables (uploaded once per process by tgk_init)                   */
/* ------------------------------------------------------------------------- */
#define TG_NBLKTYPES 5	/* block-mode front tables: TG_KIND_SB1 / _216 / _432 / _168, then BBK */
#define TG_BLK_BBK   4

struct tg_const_tables {
	uint16_t front_src[3][TG_PACKED_WORDS][32];	/* [NORM_1, NORM_2, SYNC][word][bit] -> slot byte offset */
	uint16_t mask_pos[TG_MASK_WORDS][32];		/* [mask word][bit] -> position in the LFSR sequence */
	uint16_t blk_src[TG_NBLKTYPES][TG_PACKED_WORDS][32];	/* block mode: [SB1, 216, 432, 168, BBK][word][bit] -> type-5 bit of the block */
	uint32_t lfsr_lin[432];				/* seq[n] = parity(init & lfsr_lin[n]) */
	uint32_t sb1_mask[5];				/* SB1 is always scrambled with init 3 */
	uint16_t crc_lsb[256];
	uint16_t crc_msb[256];
	/* CRC-16 as a linear map (k_burst): crc(bits) = crc_aff[kind] ^ XOR over the set bits i of crc_lin[kind][i], bits
	 * in the order they are fed (8 per decoded byte, LSB first; 8 (NBLK - 1) + 4 of them), kind = SB1 / 216 / 432 */
	uint16_t crc_lin[3][288];
	uint16_t crc_aff[4];
};


static __constant__ __attribute__((aligned(16))) tg_const_tables c_tab;
/* per unit: copy the host's tables into this unit's c_tab (and whatever else the unit keeps) */
extern "C" int tgk_upload_front(const tg_const_tables *host);
extern "C" int tgk_upload_trellis(const tg_const_tables *host);
extern "C" int tgk_upload_aux(const tg_const_tables *host);
extern "C" int tgk_upload_slot(const tg_const_tables *host);

#ifdef TG_TRACE
/* measurement build (tools/experiments/trace_untraced.py): the heavy kernels' workgroups leave (kind, first and last tick of the 100 MHz
 * clock) in a device array -- what runs beside what in the pipelined bench WITHOUT a profiler slowing the launching thread.
 * One array per unit; tgk_trace_read() (tg_k_aux.hip) reads them one after the other */
struct tg_trace_rec { uint32_t kind, block; unsigned long long t0, t1; };
#define TG_TRACE_CAP (1u << 19)
static __device__ tg_trace_rec g_trace[TG_TRACE_CAP];
static __device__ unsigned int g_trace_n;
static inline int tg_trace_read_unit(void *out, unsigned int *n, int reset)
{
	unsigned int cnt = 0;
	int rc = (int)hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_trace_n), sizeof(cnt));
	if (!rc && out)
		rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), (size_t)(cnt < TG_TRACE_CAP ? cnt : TG_TRACE_CAP) * sizeof(tg_trace_rec));
	if (n)
		*n = cnt < TG_TRACE_CAP ? cnt : TG_TRACE_CAP;
	const unsigned int z = 0;
	if (!rc && reset)
		rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &z, sizeof(z));
	return rc;
}
extern "C" int tgk_trace_read_front(void *out, unsigned int *n, int reset);
extern "C" int tgk_trace_read_trellis(void *out, unsigned int *n, int reset);
extern "C" int tgk_trace_read_slot(void *out, unsigned int *n, int reset);
#define TG_TRACE_BEGIN const unsigned long long tr_t0_ = wall_clock64()
#define TG_TRACE_END(KIND_, EVERY_) do { if (threadIdx.x == 0 && (blockIdx.x % (EVERY_)) == 0) {			\
		const unsigned int i_ = atomicAdd(&g_trace_n, 1u);							\
		if (i_ < TG_TRACE_CAP) { g_trace[i_].kind = (KIND_); g_trace[i_].block = blockIdx.x; g_trace[i_].t0 = tr_t0_;	\
					 g_trace[i_].t1 = wall_clock64(); } } } while (0)
#else
#define TG_TRACE_BEGIN do { } while (0)
#define TG_TRACE_END(KIND_, EVERY_) do { } while (0)
#endif

#define TG_LB_TBL 4096u		/* hash table slots for the scrambling codes of a device-walk batch (k_lists2) */
#define TG_DESC_TYPE(d) ((uint32_t)((d) >> 56))
#define TG_DESC_OFF(d)  ((d) & 0x00ffffffffffffffull)

__device__ __forceinline__ uint32_t spread4(uint32_t nib)
{
	/* 4 bits -> 4 bytes of 0/1 (bit 0 -> byte 0) */
	return ((nib & 15u) * 0x00204081u) & 0x01010101u;
}

#endif
