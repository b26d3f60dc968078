/*
 * tg_kernels.hip -- the HIP kernels of the TETRA lower-MAC receive path (gfx950).
 *
 *   k_front          : slot bytes -> packed, de-interleaved code words   (rows D, I, U of SURVEY 8(a))
 *   k_front_stream   : the same on a slot grid + training-sequence search  (rows F, S; config 3)
 *   k_front_blocks   : the same for blocks handed over on their own      (block mode, the tp_sap_udata_ind unit)
 *   k_front_soft, k_float_to_bits(_afc) : float phases / soft values in   (row B, config 5)
 *   k_vit<KIND,HMODE>: descramble + Viterbi + CRC-16 + type-1 output      (rows X, V, C, R, L)
 *   k_bbk_blocks     : BBK blocks of block mode                           (row R)
 *   k_fill_*         : forward-fill of the cell scrambling code           (row L, feedback loop 1)
 *   k_masks          : scrambling sequence -> masks in code-word layout   (row X)
 *   k_grid_*         : stream mode: per-slot arrays and item lists from classification words + bitmap
 *   k_conv<CODE,NCH> : generic trellis: any RCPC puncturer on either mother code
 *   k_burst<SB1_PASS>: small batches: one workgroup per burst, butterflies across a 16-lane DPP row
 *   k_walk           : the burst synchroniser's walk of every channel on the device (tg_walk_core.h; row S)
 *   k_cls_plain2, k_masks2, k_lb_scan, k_lists2 : device-walk batches: plain bitmap, SYNC list, code look-back, item lists
 *   k_reorder, k_gsmtap : ACELP re-ordering with the caller's tables, GSMTAP messages of a batch
 *   k_stages(_crc)   : the chain's intermediate bit strings step by step (the reference's DEBUGP lines; tg_stages.c)
 *
 * No MFMA anywhere: there is no dense contraction on this path.  The trellis kernels are VALU-issue bound
 * packed-u16 integer work (one lane per trellis); the front kernels are byte gathers, part HBM, part issue bound
 * (DESIGN.md section 4).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "tetra_gpu.h"
#include "tg_layout.h"
#include "vit_core.h"
#include "tg_internal.h"

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

__constant__ __attribute__((aligned(16))) tg_const_tables c_tab;

#ifdef TG_TRACE
/* measurement build (tools/trace_untraced.py): the heavy kernels' workgroups leave (kind, first and last tick of the 100 MHz
 * clock) in a device array -- what runs beside what in the pipelined bench WITHOUT a profiler slowing the launching thread */
struct tg_trace_rec { uint32_t kind, block; unsigned long long t0, t1; };
#define TG_TRACE_CAP (1u << 20)
__device__ tg_trace_rec g_trace[TG_TRACE_CAP];
__device__ unsigned int g_trace_n;
extern "C" int tgk_trace_read(void *out, unsigned int *n, int reset)
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
#define TG_TRACE_BEGIN const unsigned long long tr_t0_ = wall_clock64()
#define TG_TRACE_END(KIND_, EVERY_) do { if (threadIdx.x == 0 && (blockIdx.x % (EVERY_)) == 0) {			\
		const unsigned int i_ = atomicAdd(&g_trace_n, 1u);							\
		if (i_ < TG_TRACE_CAP) { g_trace[i_].kind = (KIND_); g_trace[i_].block = blockIdx.x; g_trace[i_].t0 = tr_t0_;	\
					 g_trace[i_].t1 = wall_clock64(); } } } while (0)
#else
#define TG_TRACE_BEGIN do { } while (0)
#define TG_TRACE_END(KIND_, EVERY_) do { } while (0)
#endif

/* clean-block fast path (k_clean): [0..4095] 12 received bits of an 8-step block -> g1 bits | g2 bits << 8;
 * [4096..8191] (state << 8 | g1 bits) -> input bits | expected g2 bits << 8 | next state << 12 */
__device__ uint16_t g_clean_lut[8192];

/* optional RM(30,14) decoder of the BBK (tg_rm.c): coset leaders by syndrome, generator parity rows */
__device__ const uint32_t *g_rm_leader;
__constant__ uint16_t c_rm_parity[14];

/* bb: bit p = p-th received BBK bit (descrambled).  Returns the corrected word in the same order. */
__device__ __forceinline__ uint32_t rm3014_correct(uint32_t bb, uint32_t &nerr)
{
	const uint32_t rx = __builtin_bitreverse32(bb & 0x3fffffffu) >> 2;	/* codeword bit 29 = first received bit */
	uint32_t syn = rx & 0xffff;
#pragma unroll
	for (int i = 0; i < 14; i++)
		syn ^= ((rx >> (29 - i)) & 1) ? c_rm_parity[i] : 0u;
	const uint32_t e = g_rm_leader[syn];
	nerr = __builtin_popcount(e);
	return __builtin_bitreverse32(rx ^ e) >> 2;
}

/* ------------------------------------------------------------------------- */
/* k_front                                                                   */
/* ------------------------------------------------------------------------- */
/*
 * One wavefront per slot, four independent wavefronts per workgroup.
 *   1. the 510 slot bytes are read from HBM exactly once, as two coalesced (possibly
 *      unaligned) dwords per lane, and parked in this wave's 512-byte LDS window
 *      (bytes 510/511 of the window are always zero: "no source" gathers point there);
 *   2. ten gather rounds: every lane picks one byte out of LDS (its LDS addresses for the
 *      three burst types live in VGPRs for the whole kernel), a 64-bit ballot collapses
 *      them: lanes 0..31 form one packed word, lanes 32..63 the next, and v_writelane drops
 *      the two dwords into lanes 2r and 2r+1 of the output register;
 *   3. the 80-byte packed slot goes out as one coalesced store.
 * Three slots are kept in flight per wave (the dwords of slot i+3 are requested as soon as slot i
 * is parked in LDS).  LDS operations of one wave execute in order: no barrier between the stages.
 */
typedef uint32_t __attribute__((aligned(1))) tg_u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) tg_u16_unaligned;
typedef uint16_t __attribute__((may_alias)) tg_u16_alias;

__device__ __forceinline__ void front_fetch(const uint8_t *base, uint32_t lane, uint32_t &d0, uint32_t &d1)
{
	d0 = *(const tg_u32_unaligned *)(base + 4 * lane);
	/* bytes 256..509: lane 63's dword would read 2 bytes past the slot, so it reads bytes 506..509
	 * instead (fixed up in front_park).  One unconditional load per half: hipcc's s_waitcnt insertion
	 * counts only loads it knows were issued, and a load under an exec branch is not one of them. */
	d1 = *(const tg_u32_unaligned *)(base + 256 + 4 * lane - (lane == 63 ? 2 : 0));
}

__device__ __forceinline__ uint32_t front_gather(const uint8_t *lds0, const uint32_t (&addr)[10])
{
	uint32_t myword = 0;
	uint32_t bytes[10];
#pragma unroll
	for (int r = 0; r < 10; r++)
		bytes[r] = lds0[addr[r]];	/* ten independent LDS reads in flight */
	unsigned long long bal[10];
#pragma unroll
	for (int r = 0; r < 10; r++)
		bal[r] = __ballot(bytes[r] != 0);
	/* the ballots live in SGPR pairs: drop their halves into lanes 2r, 2r+1.  hipcc pads no hazards for
	 * inline asm, and v_writelane reading an SGPR a VALU compare has just written needs wait states (seen
	 * on gfx950: without them the OLD value is read) -- one s_nop covers the youngest compare, the older
	 * ones are further back. */
	asm("s_nop 4\n\t"
	    "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
	    "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
	    "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9\n\tv_writelane_b32 %0, %11, 10\n\tv_writelane_b32 %0, %12, 11\n\t"
	    "v_writelane_b32 %0, %13, 12\n\tv_writelane_b32 %0, %14, 13\n\tv_writelane_b32 %0, %15, 14\n\tv_writelane_b32 %0, %16, 15\n\t"
	    "v_writelane_b32 %0, %17, 16\n\tv_writelane_b32 %0, %18, 17\n\tv_writelane_b32 %0, %19, 18\n\tv_writelane_b32 %0, %20, 19"
	    : "+v"(myword)
	    : "s"((uint32_t)bal[0]), "s"((uint32_t)(bal[0] >> 32)), "s"((uint32_t)bal[1]), "s"((uint32_t)(bal[1] >> 32)),
	      "s"((uint32_t)bal[2]), "s"((uint32_t)(bal[2] >> 32)), "s"((uint32_t)bal[3]), "s"((uint32_t)(bal[3] >> 32)),
	      "s"((uint32_t)bal[4]), "s"((uint32_t)(bal[4] >> 32)), "s"((uint32_t)bal[5]), "s"((uint32_t)(bal[5] >> 32)),
	      "s"((uint32_t)bal[6]), "s"((uint32_t)(bal[6] >> 32)), "s"((uint32_t)bal[7]), "s"((uint32_t)(bal[7] >> 32)),
	      "s"((uint32_t)bal[8]), "s"((uint32_t)(bal[8] >> 32)), "s"((uint32_t)bal[9]), "s"((uint32_t)(bal[9] >> 32)));
	return myword;
}

/* slot descriptor: byte offset in bits 0..55, burst type in bits 56..63 (one SMEM load per slot) */
#define TG_DESC_TYPE(d) ((uint32_t)((d) >> 56))
#define TG_DESC_OFF(d)  ((d) & 0x00ffffffffffffffull)

/* LDS swizzle of the 512-byte slot window: XOR the bank index with the 128-byte row number (a bijection),
 * which spreads the byte gathers of a round over the banks (offline count: 55 -> 38 LDS cycles per NORM_1 slot) */
__device__ __forceinline__ uint32_t front_swz(uint32_t a)
{
	return a ^ (((a >> 7) & 31u) << 2);
}

/* stage 1 of a slot: its two dwords go to the wave's LDS window (after this the data registers are free
 * for the next request); stage 2 (front_process): gather, pack, store */
__device__ __forceinline__ bool front_park(uint32_t *mine, uint32_t lane, uint32_t d0, uint32_t d1)
{
	mine[front_swz(4 * lane) >> 2] = d0;
	mine[front_swz(256 + 4 * lane) >> 2] = (lane == 63) ? (d1 >> 16) : d1;	/* window bytes 510/511 stay zero */
	/* any of the slot's 510 bytes other than 0 / 1 (the two dwords of the 64 lanes cover exactly the slot) */
	return __ballot(((d0 | d1) & 0xfefefefeu) != 0) != 0;
}

__device__ __forceinline__ void front_process(uint32_t slot, uint32_t type, uint32_t lane, bool nonbinary,
					       const uint8_t *lds0, const uint32_t (&a_n1)[10],
					       const uint32_t (&a_n2)[10], const uint32_t (&a_sb)[10],
					       uint32_t *stage, uint8_t *__restrict__ rec)
{
	uint32_t myword = 0;
	if (type == TG_BURST_NORM_1)
		myword = front_gather(lds0, a_n1);
	else if (type == TG_BURST_NORM_2)
		myword = front_gather(lds0, a_n2);
	else if (type == TG_BURST_SYNC)
		myword = front_gather(lds0, a_sb);
	else if (lane == 0) {
		/* not a burst we decode (NORM_3 / EXT are ignored like phy/tetra_burst.c:374-377):
		 * no trellis lane will touch this record, mark it */
		rec[(size_t)slot * TG_REC_BYTES + TG_REC_TYPE] = TG_BURST_NONE;
	}
	const uint32_t flags = nonbinary ? TG_FLAG_NONBINARY : 0;
	if (lane == TG_PW_META) {
		const uint32_t toff = (type == TG_BURST_SYNC) ? TG_SYNC_TRAIN_OFF : TG_NORM_TRAIN_OFF;
		myword = type | (flags << 8) | (toff << 16);
	}
	/* the packed slot waits in the wave's LDS staging row until its group of four is complete (front_flush) */
	if (lane < TG_PACKED_WORDS)
		stage[lane] = myword;
}

/* write cnt (1..4) consecutive packed slots, first = slot index 'first', from the wave's staging area: two
 * range-checked buffer stores (lanes past cnt * 80 bytes are dropped), 320 contiguous bytes for a full group */
__device__ __forceinline__ void front_flush(const uint32_t *mo, uint32_t lane, uint32_t first, uint32_t cnt,
					     uint32_t *__restrict__ packed)
{
	const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(packed + (size_t)first * TG_PACKED_WORDS, 0,
									       cnt * TG_PACKED_WORDS * 4, 0x00027000);
	__builtin_amdgcn_raw_buffer_store_b32(mo[lane], out, lane * 4, 0, 0);
	__builtin_amdgcn_raw_buffer_store_b32(mo[64 + lane], out, 256 + lane * 4, 0, 0);
}

__global__ __launch_bounds__(256)
void k_front(const uint8_t *__restrict__ stream, const uint64_t *__restrict__ slot_desc,
	     uint32_t nslots, uint32_t *__restrict__ packed, uint8_t *__restrict__ rec)
{
	__shared__ uint32_t s_slot[4][128];
	__shared__ uint32_t s_out[4][128];	/* per wave: four packed slots (80 dwords) on their way out */

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * 4 + wib;		/* wave-uniform: descriptors come through SMEM */
	const uint32_t nwaves = gridDim.x * 4;
	const uint32_t half = lane >> 5, bit = lane & 31;
	uint32_t *mine = s_slot[wib];
	uint32_t *mo = s_out[wib];
	const uint8_t *lds0 = (const uint8_t *)&s_slot[0][0];

	/* LDS byte address of this lane's source bit per round, for the three burst types */
	uint32_t a_n1[10], a_n2[10], a_sb[10];
#pragma unroll
	for (int r = 0; r < 10; r++) {
		const uint32_t o0 = c_tab.front_src[0][2 * r + half][bit];
		const uint32_t o1 = c_tab.front_src[1][2 * r + half][bit];
		const uint32_t o2 = c_tab.front_src[2][2 * r + half][bit];
		a_n1[r] = wib * 512 + front_swz(o0 == 0xffff ? 510 : o0);
		a_n2[r] = wib * 512 + front_swz(o1 == 0xffff ? 510 : o1);
		a_sb[r] = wib * 512 + front_swz(o2 == 0xffff ? 510 : o2);
	}

	/*
	 * Work assignment: a wave takes GROUPS of four consecutive slots (group g = wave, wave + nwaves, ...), one
	 * slot after the other; position t of its sequence is slot 4 (wave + (t >> 2) nwaves) + (t & 3).  The four
	 * packed slots of a group leave as 320 contiguous bytes (two store instructions per group instead of one
	 * 80-byte store per slot), and the slots a wave reads back to back are neighbours in memory.  Measured with
	 * the stages of this kernel in isolation (tools/ubench/front_buildup.hip): 126 us per 1 M slots with one
	 * 80-byte store per slot and slots dealt round-robin, 112 us this way, 97 us without any store -- the
	 * per-slot stores, not the gathers (3 us), were what kept the kernel off the read rate.
	 */
	const uint32_t ngroups = (nslots + 3) >> 2;
	if (wave >= ngroups)
		return;
	const uint32_t mygroups = (ngroups - wave + nwaves - 1) / nwaves;
	uint32_t T = 4 * mygroups;				/* length of this wave's slot sequence */
	if (wave + (mygroups - 1) * nwaves == ngroups - 1)
		T -= 4 * ngroups - nslots;			/* the last group of the batch may be short */
#define SLOT_OF(t) (4u * (wave + ((t) >> 2) * nwaves) + ((t) & 3u))

	/* TG_FRONT_DEPTH slots in flight per wave, registers rotated statically (no copies, so a wait only
	 * ever covers the oldest request): set k holds sequence position t = k (mod DEPTH) */
#ifndef TG_FRONT_DEPTH
#define TG_FRONT_DEPTH 3
#endif
	constexpr int DEPTH = TG_FRONT_DEPTH;
	const uint64_t none = (uint64_t)TG_BURST_NONE << 56;
	uint32_t t = 0;
	uint64_t dsc[DEPTH];
	uint32_t r0[DEPTH], r1[DEPTH];
#pragma unroll
	for (int k = 0; k < DEPTH; k++) {
		dsc[k] = none;
		r0[k] = r1[k] = 0;
		if ((uint32_t)k < T) {
			dsc[k] = slot_desc[SLOT_OF((uint32_t)k)];
			front_fetch(stream + TG_DESC_OFF(dsc[k]), lane, r0[k], r1[k]);
		}
	}

	/* the descriptor of the slot DEPTH positions ahead is itself requested one step early (dn): its scalar
	 * load then completes under this step's LDS round trip instead of stalling the wave right before the
	 * data loads that depend on it */
	uint64_t dn = none;
	if ((uint32_t)DEPTH < T)
		dn = slot_desc[SLOT_OF((uint32_t)DEPTH)];
#define FRONT_FLUSH_IF(last)										\
		if ((t & 3u) == 3u || (last))								\
			front_flush(mo, lane, slot_ - (t & 3u), (t & 3u) + 1u, packed);
#define FRONT_STEP(D, R0, R1)										\
	{												\
		if (t >= T)										\
			break;										\
		const uint32_t slot_ = SLOT_OF(t);							\
		const uint32_t type_ = TG_DESC_TYPE(D);							\
		const bool nb_ = front_park(mine, lane, R0, R1);					\
		if (t + DEPTH < T) {									\
			D = dn;										\
			front_fetch(stream + TG_DESC_OFF(D), lane, R0, R1);				\
		}											\
		if (t + DEPTH + 1 < T)									\
			dn = slot_desc[SLOT_OF(t + DEPTH + 1)];						\
		front_process(slot_, type_, lane, nb_, lds0, a_n1, a_n2, a_sb, mo + (t & 3u) * TG_PACKED_WORDS, rec); \
		FRONT_FLUSH_IF(t + 1 == T)								\
		t++;											\
	}
	/* main loop: every step has a slot to gather and one to request, nothing is conditional -- the
	 * register sets keep their roles across the back edge (no copies), so the s_waitcnt in front of a
	 * gather covers only that slot's two loads and the younger requests stay in flight.  (With the
	 * bounds checks inside, hipcc rotated one set through v_mov at the loop latch behind an
	 * s_waitcnt vmcnt(0): every third slot exposed a full HBM round trip.) */
#define FRONT_STEP_FULL(D, R0, R1)									\
	{												\
		const uint32_t slot_ = SLOT_OF(t);							\
		const uint32_t type_ = TG_DESC_TYPE(D);							\
		const bool nb_ = front_park(mine, lane, R0, R1);	/* waits for this set's two loads only */ \
		D = dn;											\
		front_fetch(stream + TG_DESC_OFF(D), lane, R0, R1);					\
		dn = slot_desc[SLOT_OF(t + DEPTH + 1)];							\
		front_process(slot_, type_, lane, nb_, lds0, a_n1, a_n2, a_sb, mo + (t & 3u) * TG_PACKED_WORDS, rec); \
		FRONT_FLUSH_IF(false)									\
		t++;											\
	}
	while (t + 2 * DEPTH < T) {	/* the last step of the body requests the descriptor at t + 2 DEPTH */
#pragma unroll
		for (int k = 0; k < DEPTH; k++)
			FRONT_STEP_FULL(dsc[k], r0[k], r1[k])
	}
#undef FRONT_STEP_FULL
	/* tail (at most 2 DEPTH slots per wave): the same steps with their bounds checks */
	static_assert(DEPTH == 3, "the tail is written out for three register sets");
	for (;;) {	/* (a loop over the sets with a flag instead of these breaks cost 27 VGPRs and three waves per SIMD) */
		FRONT_STEP(dsc[0], r0[0], r1[0])
		FRONT_STEP(dsc[1], r0[1], r1[1])
		FRONT_STEP(dsc[2], r0[2], r1[2])
	}
#undef FRONT_STEP
#undef FRONT_FLUSH_IF
#undef SLOT_OF
}

__device__ __forceinline__ uint32_t spread4(uint32_t nib)
{
	/* 4 bits -> 4 bytes of 0/1 (bit 0 -> byte 0) */
	return ((nib & 15u) * 0x00204081u) & 0x01010101u;
}

/* ------------------------------------------------------------------------- */
/* block mode: one type-5 block per item (the tp_sap_udata_ind() unit)        */
/* ------------------------------------------------------------------------- */
/*
 * k_front_blocks: the front end for blocks that arrive on their own (phy/tetra_burst.c:350-372 hands
 * tp_sap_udata_ind() one block at a time): wave per block, the block's 30..432 type-5 bytes go to the LDS
 * window, the same ballot gather as k_front with per-kind tables (de-interleave + 2/3 de-puncture order) fills
 * code words 0..17 (or the BBK word), word 19 = block type | flags << 8.  Descriptor = byte offset |
 * (uint64_t)table index << 56 | (uint64_t)tp_sap type << 48.  Reads never go past the block.
 */
__global__ __launch_bounds__(256)
void k_front_blocks(const uint8_t *__restrict__ bits, const uint64_t *__restrict__ desc, uint32_t nblocks,
		    uint32_t *__restrict__ packed)
{
	__shared__ uint32_t s_win[4][128];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * 4 + wib, nwaves = gridDim.x * 4;
	const uint32_t half = lane >> 5, bit = lane & 31;
	uint32_t *mine = s_win[wib];
	const uint8_t *mine8 = (const uint8_t *)mine;
	static const uint16_t lens[TG_NBLKTYPES] = { 120, 216, 432, 168, 30 };

	for (uint32_t b = wave; b < nblocks; b += nwaves) {
		const uint64_t d = desc[b];
		const uint32_t x = (uint32_t)(d >> 56), tptype = (uint32_t)(d >> 48) & 0xff;
		const uint8_t *base = bits + (d & 0x0000ffffffffffffull);
		const uint32_t len = lens[x];
		/* bytes 4 lane .. 4 lane + 3 and 256 + 4 lane ..: whole dwords inside the block, the 2-byte tail of a BBK */
		uint32_t d0 = 0, d1 = 0;
		if (4 * lane + 4 <= len)
			d0 = *(const tg_u32_unaligned *)(base + 4 * lane);
		else if (4 * lane + 2 <= len)
			d0 = *(const tg_u16_unaligned *)(base + 4 * lane);
		if (256 + 4 * lane + 4 <= len)
			d1 = *(const tg_u32_unaligned *)(base + 256 + 4 * lane);
		mine[lane] = d0;
		mine[64 + lane] = d1;
		uint32_t myword = 0, acc = 0;
#pragma unroll
		for (int r = 0; r < 10; r++) {
			const uint32_t o = c_tab.blk_src[x][2 * r + half][bit];
			const uint32_t byte = (o == 0xffff) ? 0u : (uint32_t)mine8[o];
			acc |= byte;
			const unsigned long long bal = __ballot(byte != 0);
			asm("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"
			    : "+v"(myword) : "s"((uint32_t)bal), "i"(2 * r), "s"((uint32_t)(bal >> 32)), "i"(2 * r + 1));
		}
		const uint32_t flags = __ballot(acc > 1) ? TG_FLAG_NONBINARY : 0;
		if (lane == TG_PW_META)
			myword = tptype | (flags << 8);
		if (lane < TG_PACKED_WORDS)
			packed[(size_t)b * TG_PACKED_WORDS + lane] = myword;
	}
}

/* BBK blocks: descramble, keep the first 14 bits (lower_mac/tetra_lower_mac.c:268-274), crc_ok = 1 */
__global__ __launch_bounds__(256)
void k_bbk_blocks(const uint32_t *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ packed,
		  const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec, int kflags)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nitems)
		return;
	const uint32_t b = items[i];
	const uint32_t midx = maskidx[b];
	const uint32_t meta = packed[(size_t)b * TG_PACKED_WORDS + TG_PW_META];
	uint32_t bb = packed[(size_t)b * TG_PACKED_WORDS + TG_PW_BBK] ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
	uint8_t *r = rec + (size_t)b * TG_REC_BYTES;
	uint32_t nerr = 0;
	if (kflags & TGK_F_RM)
		bb = rm3014_correct(bb, nerr);
	r[TG_REC_BBK_NERR] = (uint8_t)nerr;
	uint4 o;
	o.x = spread4(bb);
	o.y = spread4(bb >> 4);
	o.z = spread4(bb >> 8);
	o.w = spread4(bb >> 12) & 0x0000ffffu;
	*(uint4 *)(r + TG_REC_BBK) = o;
	r[TG_REC_TYPE] = (uint8_t)meta;
	r[TG_REC_FLAGS] = (uint8_t)(meta >> 8);
	r[TG_REC_CRC_OK] = 1;
	*(uint32_t *)(r + TG_REC_CODE) = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
	*(uint32_t *)(r + TG_REC_SLOT) = b;
}

/* ------------------------------------------------------------------------- */
/* k_front_stream: burst-sync correlation + demux on a slot grid               */
/* ------------------------------------------------------------------------- */
/*
 * Stream mode of the front end (BASELINE config 3).  Slots lie on a grid (anchor + 510 n); the
 * search window of slot n is what the reference's synchroniser would hold when it gets to that
 * slot while being fed 'chunk' bytes per call (phy/tetra_burst_sync.c:106-120):
 *     w = min(chunk * ceil((bs + 510) / chunk), len) - bs        (510 .. 573 for chunk = 64)
 * Per wave and slot: the view's 640 ... 1088 bytes -> LDS; ten ... seventeen 64-bit ballots turn them into a bit string held in
 * SGPRs; every lane then tests one window position per round against y (38 bits), n and p (22 bits)
 * with two v_alignbit_b32 -- the first hit in ascending position is tetra_find_train_seq()'s answer
 * (phy/tetra_burst.c:269-339).  Positions 0..255 are always scanned (the expected hits sit at 214
 * and 244), the rest only if nothing was found.  If the burst is decodable (SYNC at 214, NORM at 244)
 * the same LDS window feeds the gather of k_front.
 */
__device__ __forceinline__ uint32_t pattern_bits(const uint8_t *seq, int from, int n)
{
	uint32_t v = 0;
	for (int i = 0; i < n; i++)
		v |= (uint32_t)seq[from + i] << i;
	return v;
}

struct tg_stream_params {
	uint64_t anchor;	/* stream offset of grid slot 0 */
	uint64_t len;		/* stream length in bytes */
	uint32_t nslots;
	uint32_t chunk;		/* bytes per tetra_burst_sync_in() call being emulated */
	int32_t cshift;		/* log2(chunk) when it is a power of two, else -1 */
	uint32_t y32, y6, n22, p22;
	uint32_t q22, x22;	/* first 22 bits of the other two sequences the reference's look-ahead filter passes */
	/* several recorded channels in one grid (BASELINE config 4: a GPU's share of the channels in one batch): channel
	 * c owns grid slots gbase .. gbase + ncls - 1 (gbase a multiple of 32, the slots up to the next channel's gbase
	 * are padding and never decoded); its stream lies at byte d_off of the buffer, anchor / len are relative to it */
	const struct tg_chan_ent *chan;
	uint32_t nchan;		/* 0: one stream, the fields above */
	uint64_t pbit;		/* packed ingest (per-position form): bit position of the channel's stream position 0 in the packed buffer */
};

/* channel of grid slot 'slot' (nchan <= 64: one table word per lane, a ballot counts the channels that start at or
 * before the slot); wave-uniform */
__device__ __forceinline__ uint32_t chan_of_slot(const tg_chan_ent *chan, uint32_t nchan, uint32_t slot, uint32_t lane)
{
	const uint32_t gb = lane < nchan ? chan[lane].gbase : 0xffffffffu;
	return (uint32_t)__builtin_popcountll(__ballot(gb <= slot)) - 1u;
}

/*
 * One grid slot through the per-position search: the wave's view (TG_VIEW_OF: 510 + what two feeds of the replay add to a
 * window + a sequence's 38, rounded up to 64: 640 / 832 / 1088 bytes) goes to LDS, ten to seventeen ballots turn it into a
 * bit string in SGPRs, every lane tests one window position per round.  This is the exact form for ANY slot
 * (stream end, windows longer than the slot, bytes other than 0 / 1, nothing found where a burst should be): the
 * round-1 kernel ran it on every slot (k_front_stream_v1, kept for A/B runs), the packed-bit kernel below hands
 * it the slots it cannot settle (k_front_stream_fix).
 */
/* VIEWT: the view in bytes when the caller knows it at compile time (k_front_stream_fix is built for each of the three:
 * the 64-byte feeds of the metric then run the ten-round code), 0: TG_VIEW_OF(prm.chunk) at run time, arrays for the largest */
template <bool PACKED = false, int VIEWT = 0>
__device__ __forceinline__ void front_stream_slot(const uint8_t *__restrict__ stream, const tg_stream_params &prm, uint32_t slot,
						  uint32_t lane, uint32_t half, uint32_t bit, uint32_t wbase, uint32_t *mine,
						  const uint8_t *lds0, const uint32_t (&a_n1)[10], const uint32_t (&a_n2)[10],
						  const uint32_t (&a_sb)[10], uint32_t &myword, uint32_t &clsword, uint32_t &ysword)
{
	const uint64_t bs = prm.anchor + (uint64_t)slot * TG_SLOT_BITS;
	uint32_t d0, d1, d2, d3, d4;
	/* how far the view reaches depends on the feeds (TG_VIEW_OF: 640 / 832 / 1088 bytes for feeds of up to 64 / 128 / 256);
	 * its far end lies up to 578 bytes past the slot: read only where the buffer's slack covers it -- bytes past the stream's
	 * end count as zeros anyway */
	const uint32_t view = VIEWT ? (uint32_t)VIEWT : TG_VIEW_OF(prm.chunk);
	const bool ok3 = view > 768 && bs + 768 + 4 * lane + 16 <= prm.len + TG_STREAM_SLACK;
	const bool ok4 = view > 1024 && lane < 16 && bs + 1024 + 4 * lane + 16 <= prm.len + TG_STREAM_SLACK;
	if (PACKED) {
		/* packed ingest: 'stream' is the packed buffer and prm.anchor counts from the channel's bit 0, whose position in the
		 * buffer the caller has added to... the bit position of the slot: every lane fetches the two bytes that hold its
		 * four bits of each 256-byte third of the view and spreads them to the bytes the unpacked stream would have */
		const uint64_t b0 = prm.pbit + bs + 4 * lane;
		auto nib = [&](uint64_t b) {
			const uint32_t w = *(const tg_u16_unaligned *)(stream + (b >> 3));
			return spread4((w >> (b & 7)) & 15u);
		};
		d0 = nib(b0);
		d1 = nib(b0 + 256);
		d2 = (bs + 512 + 4 * lane + 16 <= prm.len + TG_STREAM_SLACK) ? nib(b0 + 512) : 0u;
		d3 = ok3 ? nib(b0 + 768) : 0u;
		d4 = ok4 ? nib(b0 + 1024) : 0u;
	} else {
		const uint8_t *base = stream + bs;
		/* (the buffer carries TG_STREAM_SLACK readable bytes of slack) */
		d0 = *(const tg_u32_unaligned *)(base + 4 * lane);
		d1 = *(const tg_u32_unaligned *)(base + 256 + 4 * lane);
		d2 = (bs + 512 + 4 * lane + 4 <= prm.len + TG_STREAM_SLACK) ? *(const tg_u32_unaligned *)(base + 512 + 4 * lane) : 0u;
		d3 = ok3 ? *(const tg_u32_unaligned *)(base + 768 + 4 * lane) : 0u;
		d4 = ok4 ? *(const tg_u32_unaligned *)(base + 1024 + 4 * lane) : 0u;
	}

	uint64_t fed = bs + TG_SLOT_BITS + prm.chunk - 1;
	fed = prm.cshift >= 0 ? (fed >> prm.cshift) << prm.cshift : (fed / prm.chunk) * prm.chunk;
	if (fed > prm.len)
		fed = prm.len;
	const uint32_t w = (uint32_t)(fed - bs);			/* search window, >= 510 */
	const uint32_t wv = w < view ? w : view;	/* what we can see of it */
	const uint64_t rest = prm.len - bs;
	const uint32_t vis = rest < view ? (uint32_t)rest : view;	/* stream bytes in view */

	mine[lane] = d0;
	mine[64 + lane] = d1;
	constexpr uint32_t ROWDW = (VIEWT ? VIEWT : TG_STREAM_VIEW) / 4;	/* the row's data dwords: a lane's dword goes there if it lies inside */
	if (128 + lane < ROWDW)
		mine[128 + lane] = d2;
	if (192 + lane < ROWDW)
		mine[192 + lane] = d3;
	if (256 + lane < ROWDW)
		mine[256 + lane] = d4;

	/* bytes -> bit string in SGPRs (bit i of B[r] = byte 64 r + i); bytes past the stream end read as 0
	 * (every test below bounds itself by the window, so bytes past the window need no masking) */
	constexpr int NR = (VIEWT ? VIEWT : TG_STREAM_VIEW) / 64;	/* rounds of 64 window positions: at most 17 (10 with feeds of up to 64 bytes) */
	unsigned long long B[NR + 1];
	if (vis == view) {	/* everywhere but at the very end of the stream: no per-lane bound */
#pragma unroll
		for (int r = 0; r < NR; r++)
			B[r] = 64u * r < view ? __ballot(lds0[wbase + 64 * r + lane] != 0) : 0ull;
	} else {
#pragma unroll
		for (int r = 0; r < NR; r++)
			B[r] = 64u * r < vis ? __ballot(lds0[wbase + 64 * r + lane] != 0 && 64u * r + lane < vis) : 0ull;
	}
	B[NR] = 0;
	/* a byte other than 0 / 1 inside the search window: from the three dwords of the lane (byte k of dword q
	 * is window byte 256 q + 4 lane + k), bytes at or past the window end masked off */
	uint32_t anyb;
	{
		const uint32_t p0 = 4 * lane, p1 = 256 + 4 * lane, p2 = 512 + 4 * lane, p3 = 768 + 4 * lane, p4 = 1024 + 4 * lane;
		const uint32_t k0 = wv > p0 ? wv - p0 : 0, k1 = wv > p1 ? wv - p1 : 0, k2 = wv > p2 ? wv - p2 : 0, k3 = wv > p3 ? wv - p3 : 0;
		const uint32_t k4 = wv > p4 ? wv - p4 : 0, m4 = k4 >= 4 ? 0xffffffffu : ((1u << (8 * k4)) - 1u);
		const uint32_t m0 = k0 >= 4 ? 0xffffffffu : ((1u << (8 * k0)) - 1u);
		const uint32_t m1 = k1 >= 4 ? 0xffffffffu : ((1u << (8 * k1)) - 1u);
		const uint32_t m2 = k2 >= 4 ? 0xffffffffu : ((1u << (8 * k2)) - 1u);
		const uint32_t m3 = k3 >= 4 ? 0xffffffffu : ((1u << (8 * k3)) - 1u);
		anyb = (((d0 & m0) | (d1 & m1) | (d2 & m2) | (d3 & m3) | (d4 & m4)) & 0xfefefefeu) ? 2u : 0u;
	}

	uint32_t rc = TG_BURST_NONE, offs = 0, flags = 0;
	uint32_t ys = TG_YS_NONE;	/* where SYNC sequences start inside this slot, window or not */
	bool found = false, inview = false;	/* inview: a sequence that ends inside the view, inside the window or not */
	uint32_t voffs = 0, vtype = 0;		/* the first of those */
#pragma unroll
	for (int r = 0; r < NR; r++) {
		const bool full = (r < 4 || !found) && 64u * r < wv;
		const bool look = r >= 7 && !found && 64u * r < vis;	/* nothing so far: anything in the rest of the view? */
		if (full || r < 8 || look) {
			const uint32_t c = 64 * r + lane;
			const uint32_t b0 = (uint32_t)B[r], b1 = (uint32_t)(B[r] >> 32);
			const uint32_t b2 = (uint32_t)B[r + 1], b3 = (uint32_t)(B[r + 1] >> 32);
			const uint32_t w0 = half ? b1 : b0, w1 = half ? b2 : b1;
			const uint32_t win = __builtin_amdgcn_alignbit(w1, w0, bit);
			/* the last 6 bits of the 38-bit SYNC sequence are only looked at where its first 32 match
			 * (wave-uniform branch: almost never taken outside a SYNC burst's round) */
			bool y38 = (win == prm.y32);
			if (__ballot(y38)) {
				const uint32_t w2 = half ? b3 : b2;
				const uint32_t win2 = __builtin_amdgcn_alignbit(w2, w1, bit);
				y38 = y38 && ((win2 & 0x3f) == prm.y6);
			}
			if (r < 8) {
				const unsigned long long my = __ballot(y38 && c < TG_SLOT_BITS && c + 38 <= vis);
				if (my) {
					if (ys == TG_YS_NONE)
						ys = 64 * r + __builtin_ctzll(my);
					else
						ys |= TG_YS_MULTI;
					if (my & (my - 1))
						ys |= TG_YS_MULTI;
				}
			}
			if (look && !inview) {	/* (rounds 0..6: whatever starts there ends inside every window) */
				const bool vy = y38 && c + 38 <= vis, vn = (win & 0x3fffff) == prm.n22 && c + 22 <= vis;
				const bool vp = (win & 0x3fffff) == prm.p22 && c + 22 <= vis;
				const unsigned long long mv = __ballot(vy || vn || vp);
				if (mv) {
					const uint32_t l0 = __builtin_ctzll(mv);
					inview = true;
					voffs = 64 * r + l0;
					vtype = __builtin_amdgcn_readlane(vy ? (uint32_t)TG_BURST_SYNC : vn ? (uint32_t)TG_BURST_NORM_1 : (uint32_t)TG_BURST_NORM_2, l0);
				}
			}
			if (full) {
				/* the window holds at least 510 bytes: rounds 0..6 (c + 38 <= 485) need no bound */
				const bool in38 = (r < 7) || (c + 38 <= w), in22 = (r < 7) || (c + 22 <= w);
				const bool isy = y38 && in38;
				const bool isn = ((win & 0x3fffff) == prm.n22) && in22;
				const bool isp = ((win & 0x3fffff) == prm.p22) && in22;
				const bool any = isy || isn || isp;
				unsigned long long m = __ballot(any && c >= 21);
				if (r == 0) {
					/* positions below 21: the reference gates every position with a 22-bit look-ahead window that
					 * is primed with in[0..19] and then fed in[cur + 21], i.e. until cur = 21 it holds the stream
					 * with in[20] missing (phy/tetra_burst.c:289-297).  A sequence that starts there counts iff that
					 * skewed window equals the first 22 bits of ANY of the five training sequences: e_0..e_21 =
					 * in[c-1..19], in[21..c+21] (c = 0: a zero, in[0..19], in[21]) */
					const unsigned long long S = B[0];
					const uint32_t cc = lane < 21 ? lane : 20;
					uint32_t X;
					if (cc == 0)
						X = (((uint32_t)S & 0xfffffu) << 1) | ((uint32_t)(S >> 21) & 1u) << 21;
					else
						X = ((uint32_t)(S >> (cc - 1)) & ((1u << (21 - cc)) - 1u)) |
						    (((uint32_t)(S >> 21) & ((1u << (cc + 1)) - 1u)) << (21 - cc));
					const bool gate = X == (prm.y32 & 0x3fffffu) || X == prm.n22 || X == prm.p22 || X == prm.q22 || X == prm.x22;
					const unsigned long long me = __ballot(any && c < 21 && gate);
					if (me)
						m = me;		/* the first accepted one wins over anything from 21 on */
				}
				if (!found && m) {
					const uint32_t l0 = __builtin_ctzll(m);
					offs = 64 * r + l0;
					const uint32_t ty = isy ? TG_BURST_SYNC : isn ? TG_BURST_NORM_1 : TG_BURST_NORM_2;
					rc = __builtin_amdgcn_readlane(ty, l0);
					found = true;
				}
			}
		}
	}
	if (__ballot(anyb > 1))
		flags |= TG_CLS_NONBINARY;
	if (!found && w > view)
		flags |= TG_CLS_CLIPPED;
	if (!found && !inview)
		flags |= TG_CLS_NOVIEW;
	const uint32_t metaoffs = offs;
	if (!found && inview) {		/* what a longer window finds first (tg_layout.h) */
		offs = voffs;
		flags |= (vtype + 1u) << TG_CLS_VIEWHIT_SHIFT;
	}

	/* what tetra_burst_sync_in() would hand to tetra_burst_rx_cb() (phy/tetra_burst_sync.c:121-141) */
	uint32_t dtype = TG_BURST_NONE;
	if (rc == TG_BURST_SYNC && offs == TG_SYNC_TRAIN_OFF)
		dtype = TG_BURST_SYNC;
	else if ((rc == TG_BURST_NORM_1 || rc == TG_BURST_NORM_2) && offs == TG_NORM_TRAIN_OFF)
		dtype = rc;

	myword = 0;
	if (dtype == TG_BURST_NORM_1)
		myword = front_gather(lds0, a_n1);
	else if (dtype == TG_BURST_NORM_2)
		myword = front_gather(lds0, a_n2);
	else if (dtype == TG_BURST_SYNC)
		myword = front_gather(lds0, a_sb);
	if (lane == TG_PW_META)
		myword = dtype | (((flags & TG_CLS_NONBINARY) ? TG_FLAG_NONBINARY : 0u) << 8) | (metaoffs << 16);
	clsword = rc | (offs << 8) | (flags << 24);
	ysword = ys;
}

#define STREAM_SLOT_TABLES(VIEWB)									\
	constexpr int WINDW = (VIEWB) / 4 + 4;	/* the view's dwords + one zero pad row */			\
	__shared__ uint32_t s_slot[4][WINDW];								\
	const uint32_t lane = threadIdx.x & 63;								\
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);				\
	const uint32_t wave = blockIdx.x * 4 + wib;							\
	const uint32_t nwaves = gridDim.x * 4;								\
	const uint32_t half = lane >> 5, bit = lane & 31;						\
	uint32_t *mine = s_slot[wib];									\
	const uint8_t *lds0 = (const uint8_t *)&s_slot[0][0];						\
	const uint32_t wbase = wib * WINDW * 4;								\
	if (lane < 4)											\
		mine[(VIEWB) / 4 + lane] = 0;	/* "no source" gathers read this */			\
	uint32_t a_n1[10], a_n2[10], a_sb[10];								\
	_Pragma("unroll")										\
	for (int r = 0; r < 10; r++) {									\
		const uint32_t o0 = c_tab.front_src[0][2 * r + half][bit];				\
		const uint32_t o1 = c_tab.front_src[1][2 * r + half][bit];				\
		const uint32_t o2 = c_tab.front_src[2][2 * r + half][bit];				\
		a_n1[r] = wbase + (o0 == 0xffff ? (VIEWB) : o0);						\
		a_n2[r] = wbase + (o1 == 0xffff ? (VIEWB) : o1);						\
		a_sb[r] = wbase + (o2 == 0xffff ? (VIEWB) : o2);						\
	}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
void k_front_stream_v1(const uint8_t *__restrict__ stream, tg_stream_params prm,
		       uint32_t *__restrict__ packed, uint32_t *__restrict__ cls, uint16_t *__restrict__ ysum)
{
	STREAM_SLOT_TABLES(TG_STREAM_VIEW)
	__shared__ uint32_t s_out[4][128];	/* per wave: four packed slots on their way out, then their cls / ysum words */
	uint32_t *mo = s_out[wib];

	/* groups of four neighbouring grid slots per wave; packed slots, classification words and SYNC summaries are
	 * staged in LDS and written once per group (as k_front: per-slot stores cost more than the search saves) */
	const uint32_t ngroups = (prm.nslots + 3) >> 2;
	if (wave >= ngroups)
		return;
	const uint32_t mygroups = (ngroups - wave + nwaves - 1) / nwaves;
	uint32_t T = 4 * mygroups;
	if (wave + (mygroups - 1) * nwaves == ngroups - 1)
		T -= 4 * ngroups - prm.nslots;
	for (uint32_t t = 0; t < T; t++) {
		const uint32_t slot = 4u * (wave + (t >> 2) * nwaves) + (t & 3u);
		uint32_t myword, clsword, ys;
		front_stream_slot(stream, prm, slot, lane, half, bit, wbase, mine, lds0, a_n1, a_n2, a_sb, myword, clsword, ys);
		if (lane < TG_PACKED_WORDS)
			mo[(t & 3u) * TG_PACKED_WORDS + lane] = myword;
		if (lane == 0) {
			mo[80 + (t & 3u)] = clsword;
			mo[84 + (t & 3u)] = ys;
		}
		if ((t & 3u) == 3u || t + 1 == T) {
			const uint32_t cnt = (t & 3u) + 1u, first = slot - (t & 3u);
			front_flush(mo, lane, first, cnt, packed);
			if (lane < cnt) {
				cls[first + lane] = mo[80 + lane];
				if (ysum)
					ysum[first + lane] = (uint16_t)mo[84 + lane];
			}
		}
	}
}

/* classification word of a slot the packed-bit kernel leaves to k_front_stream_fix (never a valid word: offsets stay below 1088) */
#define TG_CLS_DEFER 0xffffffffu

/* second pass of the packed-bit front end: every slot the first pass deferred (it appended them to a list: defer[0] =
 * count, slots from defer[TG_DEFER_LIST]) goes through the exact per-position search, one wave per list entry at a
 * time.  Deferred slots are rare (damaged training sequences, the end of a stream); the grid is sized for about one
 * entry per wave, a wave takes entries wave, wave + nwaves, ... */
#define TG_DEFER_LIST 16
template <bool PACKED, int VIEWT>
__global__ __launch_bounds__(256)
void k_front_stream_fix(const uint8_t *__restrict__ stream, tg_stream_params prm,
			uint32_t *__restrict__ packed, uint32_t *__restrict__ cls, uint16_t *__restrict__ ysum,
			const uint32_t *__restrict__ defer)
{
	STREAM_SLOT_TABLES(VIEWT)
	const uint32_t count = defer[0];
	for (uint32_t e = wave; e < count; e += nwaves) {
		{
			const uint32_t slot = defer[TG_DEFER_LIST + e];
			uint32_t myword, clsword, ys;
			if (prm.nchan) {
				const uint32_t c = chan_of_slot(prm.chan, prm.nchan, slot, lane);
				const uint32_t i = slot - prm.chan[c].gbase;
				if (i >= prm.chan[c].ncls) {	/* padding behind a channel's last slot: nothing there */
					myword = 0;
					clsword = TG_BURST_NONE;
					ys = TG_YS_NONE;
				} else {
					tg_stream_params q = prm;
					q.anchor = prm.chan[c].anchor;
					q.len = prm.chan[c].len;
					q.pbit = prm.chan[c].d_off & ~TG_CHAN_PACKED;
					front_stream_slot<PACKED, VIEWT>(PACKED ? stream : stream + prm.chan[c].d_off, q, i, lane, half, bit, wbase, mine, lds0,
								  a_n1, a_n2, a_sb, myword, clsword, ys);
				}
			} else
				front_stream_slot<PACKED, VIEWT>(stream, prm, slot, lane, half, bit, wbase, mine, lds0, a_n1, a_n2, a_sb, myword, clsword, ys);
			if (lane < TG_PACKED_WORDS)
				packed[(size_t)slot * TG_PACKED_WORDS + lane] = myword;
			if (lane == 0) {
				cls[slot] = clsword;
				if (ysum)
					ysum[slot] = (uint16_t)ys;
			}
		}
	}
}

/*
 * k_front_stream: the stream front end on packed bits.
 *
 * The grid slots of a stream are contiguous, so a wave takes GROUPS of four neighbouring slots = 2040 contiguous
 * stream bytes (+ look-ahead), fetched as 16 bytes per lane from a 16-byte aligned base -- the access pattern that
 * reaches the HBM read rate -- and turned into bits at once: two chained v_dot4_u32_u8 (weights 1,2,4,8 / 16..128)
 * make 8 bits of 8 bytes.  The group's 2176-bit string is parked in LDS (272 bytes); everything after works on bits:
 *   - lane (k, i) = (slot of the group, 32-position column) re-aligns its slot: W0..W2 = bits 32 i .. 32 i + 95 of
 *     slot k (two LDS reads, three v_alignbit_b32); W0 also goes back to LDS as the slot-aligned 512-bit window the
 *     gather reads;
 *   - training-sequence search, bit-parallel: t_j = the slot's bit string shifted down by j (one v_alignbit_b32),
 *     match mask of a sequence = AND of t_j over its 1-bits AND NOT (OR of t_j over its 0-bits); y (38 bits), n
 *     and p (22 bits) share the t_j: ~100 vector instructions give the exact match masks of all three sequences at
 *     all 4 x 512 positions (the per-position form needs ~8 per 64 positions and pattern);
 *   - ballots of the (masked) match words + s_ff1 / v_readlane give, per slot, tetra_find_train_seq()'s answer
 *     restricted to positions 21..472 (every window holds the slot's own 510 bytes, so a match that ends inside the
 *     slot is valid whatever the window), the "hit below 21" flag and the SYNC summary of the slot;
 *   - the de-interleaving gather reads single bytes of the 64-byte window (16 dwords in 16 banks: conflict-free,
 *     the byte form had 2-3 way conflicts), isolates its bit with a per-lane mask and ballots as before.
 * Anything this cannot settle exactly -- nothing found up to position 472, a byte other than 0 / 1 in the group, the
 * last groups of the stream -- is marked TG_CLS_DEFER and redone by k_front_stream_fix with the per-position form.
 */
static constexpr uint8_t TSQ_N[22] = { 1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0 };
static constexpr uint8_t TSQ_P[22] = { 0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0 };
static constexpr uint8_t TSQ_Y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };

template <int N> static constexpr uint64_t tsq_bits(const uint8_t (&seq)[N])
{
	uint64_t v = 0;
	for (int i = 0; i < N; i++)
		v |= (uint64_t)seq[i] << i;
	return v;
}

#define TG_GROUP_SLOTS   4
#define TG_GROUP_BYTES   (TG_GROUP_SLOTS * TG_SLOT_BITS)	/* 2040 */
#define TG_GROUP_LOAD    2176					/* bytes fetched per group: 2 x 1024 + 128 */
#define TG_FAST_LAST_POS (TG_SLOT_BITS - 38)			/* 472: a 38-bit match starting here still ends inside the slot */

__device__ __forceinline__ uint32_t bytes16_to_bits(const uint4 &x)
{
	const uint32_t lo = __builtin_amdgcn_udot4(x.y, 0x80402010u, __builtin_amdgcn_udot4(x.x, 0x08040201u, 0u, false), false);
	const uint32_t hi = __builtin_amdgcn_udot4(x.w, 0x80402010u, __builtin_amdgcn_udot4(x.z, 0x08040201u, 0u, false), false);
	return lo | (hi << 8);
}

/*
 * The gather of round 3: a lane owns one BYTE of the packed slot (60 of its 80 bytes carry bits: three per code word,
 * the lead-in bits of the two blocks, four BBK bytes) and collects its eight bits in eight rounds of one LDS byte read
 * and ONE vector instruction.  What makes one instruction enough: the slot's bit window lies in LDS eight times,
 * version s shifted down by s bits, so that window bit p is bit 0 of byte p >> 3 of version p & 7 -- the wanted bit
 * arrives at a fixed position, and v_alignbit_b32 (acc:byte >> 1) shifts it into the accumulator's top while the
 * accumulator moves down: after eight rounds the top byte holds the lane's output byte, round r at bit r.  No masks,
 * no compares, no ballots, no v_writelane: 8 + 1 instructions per slot instead of 41, plus 7 alignbits and 7 LDS
 * stores per GROUP for the shifted copies.  Layout: slot k at k * TG_VER_SLOT dwords (= 16 mod 32: the copies' stores
 * are conflict-free), version s at s * TG_VER_STRIDE dwords inside it (the byte reads' conflicts were counted over the
 * three gather tables for every stride: 57 LDS cycles for the 48 half-wave reads at 24, 69 at 64); dword 16 of version 0
 * stays zero: where "no source" points.
 */
#define TG_VER_STRIDE 24	/* dwords between the versions of a slot's window */
#define TG_VER_SLOT   208	/* dwords per slot: 8 versions + pad */
/* (one asm block per burst type and slot of the group: the slot's offset is the reads' immediate, the eight reads are
 * in flight together and each shift waits for its own byte only; written as asm because hipcc otherwise merges the
 * three burst types' gathers into one tail behind eight register moves / adds per slot.  X only makes the blocks differ.) */
template <int KOFF, int X>
__device__ __forceinline__ uint32_t front_gather_bytes(const uint32_t (&a)[8])
{
	uint32_t acc, t0, t1, t2, t3, t4, t5, t6, t7;
	asm volatile("; gather %18\n\t"
		     "ds_read_u8 %1, %9 offset:%17\n\tds_read_u8 %2, %10 offset:%17\n\tds_read_u8 %3, %11 offset:%17\n\t"
		     "ds_read_u8 %4, %12 offset:%17\n\tds_read_u8 %5, %13 offset:%17\n\tds_read_u8 %6, %14 offset:%17\n\t"
		     "ds_read_u8 %7, %15 offset:%17\n\tds_read_u8 %8, %16 offset:%17\n\t"
		     "s_waitcnt lgkmcnt(7)\n\tv_lshlrev_b32 %0, 31, %1\n\t"
		     "s_waitcnt lgkmcnt(6)\n\tv_alignbit_b32 %0, %2, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(5)\n\tv_alignbit_b32 %0, %3, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(4)\n\tv_alignbit_b32 %0, %4, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(3)\n\tv_alignbit_b32 %0, %5, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(2)\n\tv_alignbit_b32 %0, %6, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(1)\n\tv_alignbit_b32 %0, %7, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(0)\n\tv_alignbit_b32 %0, %8, %0, 1\n\t"
		     "v_lshrrev_b32 %0, 24, %0"
		     : "=&v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
		     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "n"(KOFF), "n"(X)
		     : "memory");
	return acc;
}

struct tg_group_data {
	uint4 a, b, c;	/* bytes 16 l .., 1024 + 16 l .., 2048 + 16 min(l, 7) .. of the group's aligned range */
	uint32_t a0;	/* the group starts a0 bytes into that range */
	bool fast;	/* all four windows of the group lie inside the stream */
};

#ifndef TG_STREAM_WPE
#define TG_STREAM_WPE 6	/* waves per SIMD (80 VGPRs: 6 fit).  With the grid at two rounds of resident workgroups (launch_stream_front):
			 * 4 -> 161-168 us per 1 M slots, 5 -> 156-161, 6 -> 155-159, 8 (64 VGPRs) -> 195-200 (tools/front_grid.sh) */
#endif
/* acc & (t0 == p0) & (t1 == p1), sel = 2 p0 + p1 (a constant once the caller's loop is unrolled): one v_bitop3_b32 */
__device__ __forceinline__ uint32_t tsq_and2(uint32_t acc, uint32_t t0, uint32_t t1, int sel)
{
	switch (sel) {
	case 0: return __builtin_amdgcn_bitop3_b32(acc, t0, t1, 0x10);
	case 1: return __builtin_amdgcn_bitop3_b32(acc, t0, t1, 0x20);
	case 2: return __builtin_amdgcn_bitop3_b32(acc, t0, t1, 0x40);
	default: return __builtin_amdgcn_bitop3_b32(acc, t0, t1, 0x80);
	}
}

#ifndef TGS_ABLATE
#define TGS_ABLATE 0	/* measurement builds only (tools/front_ablate.sh): 1 no stores, 2 every group from one address, 4 no
			 * gathers, 8 no search and no classification, 16 no shifted copies, 32 no classification, 64 no atomic for the deferred slots,
			 * 128 classification kept but the gather always NORM_1's, 256 no classification but the gather's type varies -- the kernel's
			 * results are wrong with any of them */
#endif
#ifdef TGS_TIMING
/* measurement build: reference-clock ticks (s_memtime, 100 MHz) a wave spends between the marks of a group, summed over
 * all waves; tools/front_phases.py */
__device__ unsigned long long g_tgs_acc[8];
extern "C" int tgk_front_stream_stamps(unsigned long long *out, int reset)
{
	static const unsigned long long z[8] = { 0 };
	int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tgs_acc), sizeof(g_tgs_acc));
	if (!rc && reset)
		rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tgs_acc), z, sizeof(z));
	return rc;
}
#define TGS_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
			 tgs_acc[i] += t_ - tgs_last; tgs_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TGS_MARK(i) do { } while (0)
#endif
#ifndef TG_STREAM_WPB
#define TG_STREAM_WPB 4	/* waves per workgroup (they share nothing: each has its own staging areas) */
#endif
template <bool PACKED>
__global__ __launch_bounds__(64 * TG_STREAM_WPB) __attribute__((amdgpu_waves_per_eu(TG_STREAM_WPE, TG_STREAM_WPE)))
void k_front_stream(const uint8_t *__restrict__ stream, tg_stream_params prm,
		    uint32_t *__restrict__ packed, uint32_t *__restrict__ cls, uint16_t *__restrict__ ysum,
		    uint32_t *__restrict__ defer)
{
	constexpr uint64_t PY = tsq_bits(TSQ_Y), PN = tsq_bits(TSQ_N), PP = tsq_bits(TSQ_P);
	__shared__ __attribute__((aligned(16))) uint32_t s_bits[TG_STREAM_WPB][72];	/* per wave: the group's bit string (68 dwords used; packed ingest: 72, 16 bytes per lane) */
	__shared__ uint32_t s_win[TG_STREAM_WPB][4 * TG_VER_SLOT];	/* per wave: four slots x eight shifted copies of the 512-bit window */
	__shared__ uint32_t s_out[TG_STREAM_WPB][160];	/* per wave: four packed slots on their way out, then their cls / ysum words (+ the idle lanes' dump) */

#ifdef TGS_TIMING
	unsigned long long tgs_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tgs_last = __builtin_amdgcn_s_memtime();
#endif
	TG_TRACE_BEGIN;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * TG_STREAM_WPB + wib;
	const uint32_t nwaves = gridDim.x * TG_STREAM_WPB;
	const uint32_t col = lane & 15;			/* 32-position column of the lane's slot */
	uint32_t *bits = s_bits[wib];
	uint32_t *win = s_win[wib];
	uint32_t *mo = s_out[wib];

	/* the lane's byte of the packed slot: lanes 0..53 byte l % 3 of code word l / 3, 54 / 55 the lead-in bits of the two
	 * blocks (byte 3 of words 0 and 9), 56..59 the BBK word, 60..63 none; per burst type and round the LDS byte that
	 * carries the wanted bit at its bit 0 */
	const uint32_t ow = lane < 54 ? lane / 3 : lane == 54 ? 0u : lane == 55 ? (uint32_t)TG_PW_BLK2 : (uint32_t)TG_PW_BBK;
	const uint32_t ob = lane < 54 ? lane % 3 : lane < 56 ? 3u : lane - 56;
	const uint32_t obyte = lane < 60 ? 4 * ow + ob : 4 * 88 + (lane - 60);	/* (the idle lanes write behind the staged slots: < 640 with the last slot's offset) */
	uint32_t g_adr[3][8];
	/* (the asm block takes LDS addresses as the hardware sees them: the array's offset inside the workgroup's LDS) */
	const uint32_t ver0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)&s_win[0][0];
	{
		/* the lane's 24 table entries = 8 consecutive ushorts of three rows: three 16-byte loads in flight together (one
		 * load and one wait per entry cost every wave ~24 memory latencies before its first group: 186 -> 174 us) */
		uint4 row[3];
#pragma unroll
		for (int x = 0; x < 3; x++)
			row[x] = *(const uint4 *)&c_tab.front_src[x][lane < 60 ? ow : 0][lane < 60 ? 8 * ob : 0];
#pragma unroll
		for (int x = 0; x < 3; x++) {
			const uint32_t w4[4] = { row[x].x, row[x].y, row[x].z, row[x].w };
#pragma unroll
			for (int r = 0; r < 8; r++) {
				const uint32_t o = lane < 60 ? (w4[r >> 1] >> (16 * (r & 1))) & 0xffffu : 0xffffu;
				g_adr[x][r] = ver0 + wib * (4 * TG_VER_SLOT * 4) + (o == 0xffff ? 64u : (o & 7) * (TG_VER_STRIDE * 4) + (o >> 3));
			}
		}
#pragma unroll
		for (int x = 0; x < 3; x++)
#pragma unroll
			for (int r = 0; r < 8; r++)
				asm volatile("" : "+v"(g_adr[x][r]));	/* the whole address in the register: the slot's offset is the immediate */
	}
	if (lane < 4)
		win[lane * TG_VER_SLOT + 16] = 0;	/* "no source" reads this */
	for (int i = lane; i < 128; i += 64)
		mo[i] = 0;				/* bytes of the staged slots that nobody owns stay zero */
	/* which positions of the lane's column count: main search 21..472, "early" 0..20, SYNC summary 0..509 */
	const uint32_t vmain = (col == 0) ? 0xffe00000u : (col == 14) ? 0x01ffffffu : (col == 15) ? 0u : 0xffffffffu;
	const uint32_t vearly = (col == 0) ? 0x001fffffu : 0u;
	const uint32_t vys = (col == 15) ? 0x3fffffffu : 0xffffffffu;
	const uint32_t pos0 = (lane >> 4) * TG_SLOT_BITS + 32 * col;	/* first bit of the column inside the group */

	const uint32_t ngroups = (prm.nslots + 3) >> 2;
	if (wave >= ngroups)
		return;

	/* request a group: 16 bytes per lane from the 16-byte aligned address below the group's first byte.  Groups the
	 * fast path may not touch (their windows or the exact form's 640-byte views reach past the stream) fetch group 0
	 * instead, so that every step issues the same loads */
	/* multi-channel batches: the channel a wave is in changes a handful of times over its groups, so its table entry
	 * is kept in scalar registers and looked up again only when a group falls outside [cg0, cg1) */
	uint32_t cg0 = 1, cg1 = 0, cncls = 0;
	uint64_t cfirst = 0, cspan = 0;		/* stream offset of the channel's grid slot 0; bytes from there to its end */
	auto fetch = [&](uint32_t g, tg_group_data &d) {
		uint64_t gb, first;
		if (prm.nchan) {
			const uint32_t s0 = 4u * g;
			if (s0 < cg0 || s0 >= cg1) {
				/* (readfirstlane: the values are wave-uniform and must live in scalar registers, so that the wait
				 * for these loads stays inside this rarely taken branch and does not drain the prefetch) */
				const uint32_t c = chan_of_slot(prm.chan, prm.nchan, s0, lane);
				const tg_chan_ent e = prm.chan[c];
				const uint32_t nxt = c + 1 < prm.nchan ? prm.chan[c + 1].gbase : prm.nslots;
				cg0 = __builtin_amdgcn_readfirstlane(e.gbase);
				cg1 = __builtin_amdgcn_readfirstlane(nxt);
				cncls = __builtin_amdgcn_readfirstlane(e.ncls);
				const uint64_t f = (e.d_off & ~TG_CHAN_PACKED) + e.anchor, sp = e.len - e.anchor;
				cfirst = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)f) |
					 ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(f >> 32)) << 32);
				cspan = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)sp) |
					((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32);
			}
			const uint32_t i0 = s0 - cg0;
			first = cfirst;
			gb = first + (uint64_t)i0 * TG_SLOT_BITS;
			d.fast = i0 + 4u <= cncls && (uint64_t)i0 * TG_SLOT_BITS + TG_GROUP_BYTES + TG_VIEW_OF(prm.chunk) <= cspan;
		} else {
			first = prm.anchor;
			gb = prm.anchor + (uint64_t)g * TG_GROUP_BYTES;
			d.fast = gb + TG_GROUP_BYTES + TG_VIEW_OF(prm.chunk) <= prm.len;
		}
		if (PACKED) {
			/* packed ingest: the stream lies in memory one bit per position, so a group is 255 bytes: eighteen lanes
			 * fetch 16 bytes each from the aligned address below its first bit, a0 = how many bits in the group starts */
			const uint64_t gbit = d.fast ? gb : first;
			const uint8_t *p = stream + (gbit >> 3);
			const uint32_t ab = (uint32_t)((uintptr_t)p & 15);
			d.a0 = 8 * ab + (uint32_t)(gbit & 7);
			d.a = *(const uint4 *)(p - ab + 16 * (lane < 18 ? lane : 17));
			d.b = d.c = make_uint4(0, 0, 0, 0);	/* (unused here; left unset they keep the whole struct in scratch memory) */
			return;
		}
#if TGS_ABLATE & 2
		const uint8_t *p = stream + first + 2040u * (wave & 1023u);
#else
		const uint8_t *p = stream + (d.fast ? gb : first);
#endif
		d.a0 = (uint32_t)((uintptr_t)p & 15);
		const uint8_t *base16 = p - d.a0;
		d.a = *(const uint4 *)(base16 + 16 * lane);
		d.b = *(const uint4 *)(base16 + 1024 + 16 * lane);
		d.c = *(const uint4 *)(base16 + 2048 + 16 * (lane < 7 ? lane : 7));
	};

	auto work = [&](uint32_t g, const tg_group_data &cur) {
		TGS_MARK(0);	/* since the last mark: the next group's fetch issued */
		/* bytes other than 0 / 1 anywhere in the group: not for this kernel */
		bool defer_all;
		if (PACKED) {
			defer_all = !cur.fast;
			TGS_MARK(1);
			if (lane < 18)		/* the bits are the bit string: 288 bytes, as they came */
				((uint4 *)bits)[lane] = cur.a;
		} else {
			const uint32_t orall = cur.a.x | cur.a.y | cur.a.z | cur.a.w | cur.b.x | cur.b.y | cur.b.z | cur.b.w |
					       cur.c.x | cur.c.y | cur.c.z | cur.c.w;
			defer_all = !cur.fast || __ballot((orall & 0xfefefefeu) != 0) != 0;

			TGS_MARK(1);	/* the group's bytes are here */
			/* bytes -> bits -> LDS */
			tg_u16_alias *b16 = (tg_u16_alias *)bits;
			b16[lane] = (uint16_t)bytes16_to_bits(cur.a);
			b16[64 + lane] = (uint16_t)bytes16_to_bits(cur.b);
			if (lane < 8)
				b16[128 + lane] = (uint16_t)bytes16_to_bits(cur.c);
		}
		/* the lane's column of its slot: 96 bits from position pos0 + a0 of the string */
		uint32_t W0, W1, W2;
		{
			const uint32_t p = pos0 + cur.a0;
			const uint32_t *q = bits + (p >> 5);
			const uint32_t D0 = q[0], D1 = q[1], D2 = q[2], D3 = q[3];
			W0 = __builtin_amdgcn_alignbit(D1, D0, p);
			W1 = __builtin_amdgcn_alignbit(D2, D1, p);
			W2 = __builtin_amdgcn_alignbit(D3, D2, p);
		}
		TGS_MARK(2);	/* bits through LDS, the lane's column */
		{
			uint32_t *v = win + (lane >> 4) * TG_VER_SLOT + col;
			v[0] = W0;
#pragma unroll
			for (int sft = 1; sft < ((TGS_ABLATE & 16) ? 1 : 8); sft++)
				v[sft * TG_VER_STRIDE] = __builtin_amdgcn_alignbit(W1, W0, sft);
		}

		/* match masks of the three sequences at the column's 32 positions */
		/* one accumulator per sequence, two positions per step: acc & (t_j == p_j) & (t_j+1 == p_j+1) is one
		 * three-input logic instruction (v_bitop3_b32) whatever the two pattern bits are */
		uint32_t my = vys, mn = 0xffffffffu, mp = 0xffffffffu;
#pragma unroll
		for (int j = 0; j < ((TGS_ABLATE & 8) ? 2 : 38); j += 2) {
			const uint32_t t0 = (j == 0) ? W0 : (j < 32) ? __builtin_amdgcn_alignbit(W1, W0, j)
					  : (j == 32) ? W1 : __builtin_amdgcn_alignbit(W2, W1, j - 32);
			const int k = j + 1;
			const uint32_t t1 = (k < 32) ? __builtin_amdgcn_alignbit(W1, W0, k)
					  : (k == 32) ? W1 : __builtin_amdgcn_alignbit(W2, W1, k - 32);
			/* truth table index = acc << 2 | t0 << 1 | t1: the one entry with acc = 1, t0 = p_j, t1 = p_k */
#define TSQ_STEP(acc, P) acc = tsq_and2(acc, t0, t1, 2 * (int)(((P) >> j) & 1) + (int)(((P) >> k) & 1))
			TSQ_STEP(my, PY);
			if (j < 22) {
				TSQ_STEP(mn, PN);
				TSQ_STEP(mp, PP);
			}
#undef TSQ_STEP
		}
		const uint32_t any = my | mn | mp;

		TGS_MARK(3);	/* shifted copies stored, match masks, ballots */
#if !(TGS_ABLATE & (8 | 32 | 256))
		/* per slot (= 16-lane row): the first hit and the SYNC summary by reductions inside the row -- every lane makes a
		 * key of its own first hit ((position << 2 | type) in the high half, first y position in the low half: one
		 * v_pk_min_u16 reduces both) and a count word (hit below 21 in the high half, number of y hits in the low), four
		 * rotate-and-combine steps (DPP row_ror 8 4 2 1) leave the row's result in all of its lanes.  Vector
		 * instructions only: the form with ballots, per-lane shifts of them and the LDS crossbar cost 24 us per 1 M
		 * slots in round trips between the vector unit, scalar registers and LDS (TGS_ABLATE), this one */
		typedef unsigned short cls_us2 __attribute__((ext_vector_type(2)));
		const uint32_t hm = any & vmain;
		const uint32_t hb = (uint32_t)__builtin_ctz(hm | 0x80000000u);
		const uint32_t ht = ((my >> hb) & 1) ? (uint32_t)TG_BURST_SYNC : ((mn >> hb) & 1) ? (uint32_t)TG_BURST_NORM_1 : (uint32_t)TG_BURST_NORM_2;
		const uint32_t hkey = hm ? (((32u * col + hb) << 2) | ht) : 0xffffu;
		const uint32_t ykey = my ? (32u * col + (uint32_t)__builtin_ctz(my | 0x80000000u)) : 0xffffu;
		uint32_t rmin = (hkey << 16) | ykey;
		uint32_t rsum = (((any & vearly) != 0) ? 0x10000u : 0u) + (uint32_t)__builtin_popcount(my);	/* (<= 510 y hits: the halves do not meet) */
#define ROW_STEP(CTRL)													\
		{													\
			const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rmin, (CTRL), 0xf, 0xf, true);	\
			const uint32_t u_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rsum, (CTRL), 0xf, 0xf, true);	\
			const cls_us2 m_ = __builtin_elementwise_min(__builtin_bit_cast(cls_us2, rmin), __builtin_bit_cast(cls_us2, t_));	\
			rmin = __builtin_bit_cast(uint32_t, m_);							\
			rsum += u_;											\
		}
		ROW_STEP(0x128)	/* row_ror:8 */
		ROW_STEP(0x124)
		ROW_STEP(0x122)
		ROW_STEP(0x121)
#undef ROW_STEP
		const uint32_t k16 = rmin >> 16, yfirst = rmin & 0xffffu, ycnt = rsum & 0xffffu;
		const uint32_t offs = k16 >> 2, rc = k16 & 3u;
		uint32_t ys = ycnt ? (yfirst | (ycnt > 1 ? (uint32_t)TG_YS_MULTI : 0u)) : (uint32_t)TG_YS_NONE;
		/* a sequence below offset 21 is accepted or not by the reference's skewed look-ahead window: the exact pass
		 * evaluates that rule (rare: a payload coincidence, about ten slots in a million) */
		const bool dfr = defer_all || k16 == 0xffffu || (rsum >> 16) != 0;
		uint32_t dtype = TG_BURST_NONE;
		if (rc == TG_BURST_SYNC ? offs == TG_SYNC_TRAIN_OFF : offs == TG_NORM_TRAIN_OFF)
			dtype = rc;
		if (dfr)
			dtype = TG_BURST_NONE;
#define CLS_OWNER      ((lane & 15u) == 0u)	/* the lane that writes the slot's words */
#define CLS_SLOT       (lane >> 4)
#define CLS_LANE_OF(K) (16 * (K))
#endif
#if TGS_ABLATE & (8 | 32 | 256)
#define CLS_OWNER      (lane < 4u)
#define CLS_SLOT       lane
#define CLS_LANE_OF(K) (K)
		/* (measurement builds: every slot "a NORM_1 burst at its place", whatever the search said) */
		const bool dfr = false;
		const uint32_t dtype = TG_BURST_NORM_1;
		const uint32_t clsword = TG_BURST_NORM_1 | (TG_NORM_TRAIN_OFF << 8);
		const uint32_t meta = (dtype | (TG_NORM_TRAIN_OFF << 16)) ^ ((TGS_ABLATE & (32 | 256)) ? (any & 1u) : 0u);
		uint32_t ys = TG_YS_NONE;
#else
		const uint32_t clsword = dfr ? TG_CLS_DEFER : (rc | (offs << 8));
		const uint32_t meta = dfr ? 0u : (dtype | (offs << 16));
#endif

		const uint32_t first = 4u * g;
		const uint32_t cnt = (prm.nslots - first < 4u) ? prm.nslots - first : 4u;
#define STREAM_SLOT_K(K)												\
		{													\
			const uint32_t dt = (TGS_ABLATE & 128) ? (uint32_t)TG_BURST_NORM_1 :					\
					    (TGS_ABLATE & 256) ? (((g + (K)) & 1) ? (uint32_t)TG_BURST_NORM_1 : (uint32_t)TG_BURST_NORM_2) : \
					    (uint32_t)__builtin_amdgcn_readlane(dtype, CLS_LANE_OF(K));		\
			uint32_t mybyte = 0;										\
			if (TGS_ABLATE & 4)											\
				mybyte = dt;											\
			else if (dt == TG_BURST_NORM_1)									\
				mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 0>(g_adr[0]);			\
			else if (dt == TG_BURST_NORM_2)									\
				mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 1>(g_adr[1]);			\
			else if (dt == TG_BURST_SYNC)									\
				mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 2>(g_adr[2]);			\
			((uint8_t *)mo)[(K) * TG_PACKED_WORDS * 4 + obyte] = (uint8_t)mybyte;				\
		}
		TGS_MARK(4);	/* classification of the four slots */
		STREAM_SLOT_K(0)
		STREAM_SLOT_K(1)
		STREAM_SLOT_K(2)
		STREAM_SLOT_K(3)
#undef STREAM_SLOT_K
		TGS_MARK(5);	/* the four gathers */
		if (CLS_OWNER) {
			mo[CLS_SLOT * TG_PACKED_WORDS + TG_PW_META] = meta;
			mo[80 + CLS_SLOT] = clsword;
			mo[84 + CLS_SLOT] = ys;
		}
		{	/* slots this pass could not settle: onto the list of k_front_stream_fix (one atomic per group that has any) */
			const bool mine = CLS_OWNER && CLS_SLOT < cnt && dfr;
			const unsigned long long dm = __ballot(mine);
			if (dm) {
				uint32_t pos = 0;
				if (lane == 0 && !(TGS_ABLATE & 64))
					pos = atomicAdd(defer, (uint32_t)__builtin_popcountll(dm));
				pos = __builtin_amdgcn_readfirstlane(pos);
				if (mine)
					defer[TG_DEFER_LIST + pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u))] = first + CLS_SLOT;
			}
		}
#undef CLS_OWNER
#undef CLS_SLOT
#undef CLS_LANE_OF
		if (!(TGS_ABLATE & 1) || prm.nslots == 0xffffffffu)
			front_flush(mo, lane, first, cnt, packed);
		if (lane < cnt && (!(TGS_ABLATE & 1) || prm.nslots == 0xffffffffu)) {
			cls[first + lane] = mo[80 + lane];
			if (ysum)
				ysum[first + lane] = (uint16_t)mo[84 + lane];
		}
		TGS_MARK(6);	/* staged stores */
	};

	/* two register sets with fixed roles: the next group is requested before this one is worked on, no copies */
	tg_group_data dA, dB;
	uint32_t g = wave;
	fetch(g, dA);
	for (;;) {
		const uint32_t gB = g + nwaves;
		fetch(gB < ngroups ? gB : g, dB);
		work(g, dA);
		if (gB >= ngroups)
			break;
		const uint32_t gA = gB + nwaves;
		fetch(gA < ngroups ? gA : gB, dA);
		work(gB, dB);
		if (gA >= ngroups)
			break;
		g = gA;
	}
	TG_TRACE_END(0u, (TG_STREAM_WPB <= 4 ? 4u / TG_STREAM_WPB : 1u));
#ifdef TGS_TIMING
	if (lane == 0)
		for (int i = 0; i < 8; i++)
			atomicAdd(&g_tgs_acc[i], tgs_acc[i]);
#endif
}

/* ------------------------------------------------------------------------- */
/* soft input (BASELINE config 5): float phases -> bits / soft values, soft gather  */
/* ------------------------------------------------------------------------- */
/*
 * k_float_to_bits: the slicer of float_to_bits.c:33-72 (no AFC), 4 symbols per lane:
 *   phi > 2 -> +3 (0,1)   phi > 0 -> +1 (0,0)   phi < -2 -> -3 (1,1)   else -1 (1,0)   (NaN -> (1,0))
 * and, optionally, our soft values: soft0 = sat(rint(64 phi)), soft1 = sat(rint(64 (2 - |phi|))).
 */
__device__ __forceinline__ uint32_t slice_sym(float f)
{
	const uint32_t b0 = !(f > 0.0f);				/* first bit: 1 for the two negative symbols (and NaN) */
	const uint32_t b1 = (f > 2.0f) || (f < -2.0f);			/* second bit: 1 for the outer symbols */
	return b0 | (b1 << 8);
}

__device__ __forceinline__ int32_t sat127(float x)
{
	if (x != x)
		return 0;
	x = fminf(fmaxf(x, -127.0f), 127.0f);
	return (int32_t)__builtin_rintf(x);
}

__device__ __forceinline__ uint32_t soft_sym(float f)
{
	const int32_t s0 = sat127(64.0f * f), s1 = sat127(64.0f * (2.0f - fabsf(f)));
	return ((uint32_t)s0 & 0xff) | (((uint32_t)s1 & 0xff) << 8);
}

__global__ __launch_bounds__(256)
void k_float_to_bits(const float *__restrict__ in, unsigned long long n, uint8_t *__restrict__ bits, int8_t *__restrict__ soft)
{
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x * 4;
	for (unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
		if (i + 4 <= n) {
			const float4 f = *(const float4 *)(in + i);
			uint2 o;
			o.x = slice_sym(f.x) | (slice_sym(f.y) << 16);
			o.y = slice_sym(f.z) | (slice_sym(f.w) << 16);
			*(uint2 *)(bits + 2 * i) = o;
			if (soft) {
				uint2 q;
				q.x = soft_sym(f.x) | (soft_sym(f.y) << 16);
				q.y = soft_sym(f.z) | (soft_sym(f.w) << 16);
				*(uint2 *)(soft + 2 * i) = q;
			}
		} else {
			for (unsigned long long k = i; k < n; k++) {
				const uint32_t b = slice_sym(in[k]);
				bits[2 * k] = (uint8_t)b;
				bits[2 * k + 1] = (uint8_t)(b >> 8);
				if (soft) {
					const uint32_t q = soft_sym(in[k]);
					soft[2 * k] = (int8_t)q;
					soft[2 * k + 1] = (int8_t)(q >> 8);
				}
			}
		}
	}
}

/*
 * k_float_to_bits_afc: the pseudo-AFC of float_to_bits.c:142-146 is a sequential IIR with a float state
 * and a double intermediate, so bit-exactness needs the same operation order: one lane per channel
 * walks its symbols.  Contraction is switched off explicitly (no fma may be formed).
 */
__global__ void k_float_to_bits_afc(const float *__restrict__ in, unsigned long long n, uint8_t *__restrict__ bits,
				    float filter_val, float filter_goal, float *__restrict__ state)
{
#pragma clang fp contract(off)
	if (blockIdx.x || threadIdx.x)
		return;
	float filter = *state;
	const double keep = 1.0 - (double)filter_val;
	for (unsigned long long i = 0; i < n; i++) {
		const float fl = in[i];
		if ((fl > -5.0f) && (fl < 5.0f)) {
			const double a = __dmul_rn((double)filter, keep);
			const float b = __fmul_rn(__fsub_rn(fl, filter_goal), filter_val);
			filter = (float)__dadd_rn(a, (double)b);
		}
		const uint32_t s = slice_sym(__fsub_rn(fl, filter));
		bits[2 * i] = (uint8_t)s;
		bits[2 * i + 1] = (uint8_t)(s >> 8);
	}
	*state = filter;
}

/*
 * k_front_soft: the demux/de-interleave gather of k_front for int8 soft values.  Output per slot:
 * a 512-byte area, every block as [6 lead-in values, 2 pad][12 values] x NBLK in type-3 order
 * (first block at 0, second at TG_SOFT_AREA2, BBK at TG_SOFT_BBK), plus the meta word of the packed slot.
 */
struct tg_soft_tables {
	uint16_t src[3][TG_SOFT_SLOT_BYTES];	/* [NORM_1, NORM_2, SYNC][area byte] -> slot byte offset, 0xffff = zero */
};
__device__ tg_soft_tables g_soft_tab;

/* F32: the input is the float phase stream itself (one float = the two stream positions 2 k, 2 k + 1; a slot offset
 * is a position in that 2-values-per-symbol stream, odd offsets included): the wave loads the slot's 256 symbols,
 * applies soft_sym() and parks the 512 soft values where the int8 variant parks the bytes it fetched -- float_to_bits
 * and the gather in one pass, neither the bit stream nor the soft stream goes through memory. */
template <bool F32>
__global__ __launch_bounds__(256)
void k_front_soft(const void *__restrict__ in, unsigned long long nin, const uint64_t *__restrict__ slot_desc, uint32_t nslots,
		  uint32_t *__restrict__ area, uint32_t *__restrict__ packed, uint8_t *__restrict__ rec)
{
	constexpr uint32_t ROW = F32 ? 136 : 128;	/* dwords per wave; F32: bytes 512..543 stay zero ("no source") */
	constexpr uint32_t NOSRC = F32 ? 520u : 510u;
	__shared__ uint32_t s_slot[4][ROW];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * 4 + wib;
	const uint32_t nwaves = gridDim.x * 4;
	uint32_t *mine = s_slot[wib];
	const uint8_t *lds0 = (const uint8_t *)&s_slot[0][0];
	if (F32 && lane < 8)
		mine[128 + lane] = 0;

	/* this lane assembles area bytes 4 lane .. 4 lane + 3 and 256 + 4 lane ..: their LDS source addresses per burst
	 * type stay in registers (int8 variant: window byte 510 is zero = "no source") */
	uint32_t adr[3][8];
#pragma unroll
	for (int x = 0; x < 3; x++)
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint32_t o = g_soft_tab.src[x][256 * (q >> 2) + 4 * lane + (q & 3)];
			adr[x][q] = wib * (4 * ROW) + (o == 0xffff ? NOSRC : o);
		}

	/* groups of four neighbouring slots per wave (as k_front), the next slot's data requested before this
	 * one is gathered; past the end of the sequence the last slot is requested again, so that every step issues
	 * the same memory operations and the waits stay exact */
	const uint32_t ngroups = (nslots + 3) >> 2;
	if (wave >= ngroups)
		return;
	const uint32_t mygroups = (ngroups - wave + nwaves - 1) / nwaves;
	uint32_t T = 4 * mygroups;
	if (wave + (mygroups - 1) * nwaves == ngroups - 1)
		T -= 4 * ngroups - nslots;
#define SLOT_OF(t) (4u * (wave + ((t) >> 2) * nwaves) + ((t) & 3u))
	/* two slots' data in flight per wave: the loop body is written out for the two register sets */
	uint64_t d[2];
	uint32_t n0[2] = { 0, 0 }, n1[2] = { 0, 0 };
	float fv[2][4] = { { 0.0f, 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f, 0.0f } };
	auto fetch = [&](uint64_t dd, int h) {
		if (F32) {
			/* wave-uniform base + 32-bit lane offsets; symbol 255 of the window belongs to the next slot (it is
			 * read for odd offsets only) and may lie past the end of the input: clamp, any value will do */
			const unsigned long long f0 = TG_DESC_OFF(dd) >> 1;
			const float *base = (const float *)in + f0;
			const unsigned long long room = nin - 1 - f0;
			const uint32_t lim = room < 255 ? (uint32_t)room : 255u;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint32_t k = 64 * j + lane;
				fv[h][j] = base[k < lim ? k : lim];
			}
		} else {
			front_fetch((const uint8_t *)in + TG_DESC_OFF(dd), lane, n0[h], n1[h]);
		}
	};
	d[0] = slot_desc[SLOT_OF(0u)];
	d[1] = slot_desc[SLOT_OF(T > 1 ? 1u : 0u)];
	fetch(d[0], 0);
	fetch(d[1], 1);
	uint64_t dn = slot_desc[SLOT_OF(T > 2 ? 2u : T - 1)];
	for (uint32_t t0 = 0; t0 < T; t0 += 2) {
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const uint32_t t = t0 + h;
			if (t >= T)
				break;
			const uint32_t slot = SLOT_OF(t);
			const uint32_t type = TG_DESC_TYPE(d[h]);
			uint32_t odd = 0;
			if (F32) {
				odd = (uint32_t)TG_DESC_OFF(d[h]) & 1u;
#pragma unroll
				for (int j = 0; j < 4; j++)
					((tg_u16_alias *)mine)[64 * j + lane] = (uint16_t)soft_sym(fv[h][j]);
			} else {
				mine[lane] = n0[h];
				mine[64 + lane] = (lane == 63) ? (n1[h] >> 16) : n1[h];	/* lane 63 fetched bytes 506..509 */
			}
			d[h] = dn;
			fetch(d[h], h);
			dn = slot_desc[SLOT_OF(t + 3 < T ? t + 3 : T - 1)];
			uint32_t w0 = 0, w1 = 0;
			if (type == TG_BURST_NORM_1 || type == TG_BURST_NORM_2 || type == TG_BURST_SYNC) {
				uint32_t by[8];
				const int x = (type == TG_BURST_NORM_1) ? 0 : (type == TG_BURST_NORM_2) ? 1 : 2;
				/* (an odd offset is the LDS instruction's immediate, not an address add) */
#define SOFT_GATHER(X, ODD)									\
				_Pragma("unroll") for (int q = 0; q < 8; q++)			\
					by[q] = lds0[adr[X][q] + ODD];
				if (odd) {
					if (x == 0) { SOFT_GATHER(0, 1) } else if (x == 1) { SOFT_GATHER(1, 1) } else { SOFT_GATHER(2, 1) }
				} else {
					if (x == 0) { SOFT_GATHER(0, 0) } else if (x == 1) { SOFT_GATHER(1, 0) } else { SOFT_GATHER(2, 0) }
				}
#undef SOFT_GATHER
				w0 = by[0] | (by[1] << 8) | (by[2] << 16) | (by[3] << 24);
				w1 = by[4] | (by[5] << 8) | (by[6] << 16) | (by[7] << 24);
			} else if (lane == 0) {
				rec[(size_t)slot * TG_REC_BYTES + TG_REC_TYPE] = TG_BURST_NONE;
			}
			/* no store sits under a branch (exact s_waitcnt, see k_front): an ignored burst type writes zeros to its
			 * area, which nothing reads, and the meta word goes through a one-dword buffer range (lane 0 only) */
			uint32_t *dst = area + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4);
			dst[lane] = w0;
			dst[64 + lane] = w1;
			const uint32_t toff = (type == TG_BURST_SYNC) ? TG_SYNC_TRAIN_OFF : TG_NORM_TRAIN_OFF;
			const __amdgpu_buffer_rsrc_t mw = __builtin_amdgcn_make_buffer_rsrc(packed + (size_t)slot * TG_PACKED_WORDS + TG_PW_META,
											      0, 4, 0x00027000);
			__builtin_amdgcn_raw_buffer_store_b32(type | (toff << 16), mw, lane * 4, 0, 0);
		}
	}
#undef SLOT_OF
}

/* ------------------------------------------------------------------------- */
/* k_vit<KIND>                                                               */
/* ------------------------------------------------------------------------- */
#define TG_LB_TBL 4096u		/* hash table slots for the scrambling codes of a device-walk batch (k_lists2) */
template <int KIND> struct vit_cfg;
template <> struct vit_cfg<TG_KIND_SB1> { enum { NBLK = 10, TYPE1 = 60, MW = 0 }; };
template <> struct vit_cfg<TG_KIND_216> { enum { NBLK = 18, TYPE1 = 124, MW = TG_MW_216 }; };
template <> struct vit_cfg<TG_KIND_432> { enum { NBLK = 36, TYPE1 = 268, MW = TG_MW_432 }; };
template <> struct vit_cfg<TG_KIND_168> { enum { NBLK = 14, TYPE1 = 92, MW = TG_MW_168 }; };

/* MSB-first value of 'len' consecutive decoded bits starting at bit n0 (bits are held
 * LSB-first in od[]): the reference's bits_to_uint(type2 + n0, len), tetra_common.c:31-39 */
__device__ __forceinline__ uint32_t field_msb(uint32_t lo, uint32_t hi, int sh, int len)
{
	const unsigned long long two = (unsigned long long)lo | ((unsigned long long)hi << 32);
	const uint32_t f = (uint32_t)(two >> sh) & ((1u << len) - 1);
	return __builtin_bitreverse32(f) >> (32 - len);
}
#define FIELD_MSB(od, n0, len) field_msb((od)[(n0) >> 5], (od)[((n0) >> 5) + 1], (n0) & 31, (len))

typedef uint32_t tg_v32 __attribute__((ext_vector_type(32)));

#define TG_STAGE_PITCH 20	/* dwords per lane in the record staging area: 16 + 4 (dwordx4 rows of neighbouring lanes in different banks) */

/* byte 's' (0..15) of the 16 history bytes held in four dwords: two v_perm_b32 + one select */
__device__ __forceinline__ uint32_t hist_byte(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t s)
{
	const uint32_t sel = s & 7;
	const uint32_t lo = __builtin_amdgcn_perm(w1, w0, sel);
	const uint32_t hi = __builtin_amdgcn_perm(w3, w2, sel);
	return ((s & 8) ? hi : lo) & 0xff;
}

/*
 * What follows the decoded bits of a block, shared by the trellis kernels and the clean-block fast path:
 * CRC-16, type-1 bits at one byte per bit, record header / BBK / SYNC-PDU fields, optional wire record.
 * od[]: decoded type-2 bits, LSB first (bit i = input bit i of the encoder).
 */
template <int KIND, int HMODE>
__device__ __forceinline__ void vit_finish(const uint32_t (&od)[(vit_cfg<KIND>::NBLK + 3) / 4 + 1], const uint16_t *s_crc, bool valid,
					    uint32_t slot, uint32_t which, uint32_t idx, uint32_t midx,
					    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks,
					    uint8_t *__restrict__ rec, uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code,
					    uint8_t *__restrict__ wire, const uint32_t *__restrict__ softarea, int kflags,
					    uint32_t *stage = nullptr)
{
	constexpr int NBLK = vit_cfg<KIND>::NBLK;
	constexpr int TYPE1 = vit_cfg<KIND>::TYPE1;
	const bool block_mode = kflags & TGK_F_BLOCK;
	/* CRC-16 over type1 + 16 bits = (NBLK-1) bytes + 4 bits (lower_mac/tetra_lower_mac.c:258) */
	uint32_t crc = 0xffff;
#pragma unroll
	for (int i = 0; i < NBLK - 1; i++) {
		const uint32_t byte = (od[i >> 2] >> ((i & 3) * 8)) & 0xff;
		crc = ((crc << 8) & 0xffff) ^ s_crc[256 + (crc >> 8)] ^ s_crc[byte];
	}
	{
		const uint32_t nib = (od[(NBLK - 1) >> 2] >> (((NBLK - 1) & 3) * 8)) & 15;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			crc ^= ((nib >> i) & 1) << 15;
			crc = (crc & 0x8000) ? (((crc << 1) ^ 0x1021) & 0xffff) : ((crc << 1) & 0xffff);
		}
	}
	const uint32_t crc_ok = (crc == 0x1d0f);

	const bool wire_only = kflags & TGK_F_WIREONLY;
	if (KIND == TG_KIND_432 && stage && !block_mode && !wire_only) {
		/* SCH/F: the lane owns the whole 320-byte record.  Written 16 bytes at a time per lane, every store
		 * instruction touches 64 cache lines and every line is filled from memory before it is complete
		 * (FETCH_SIZE 3x the input).  Instead the record goes out in five 64-byte pieces through LDS: each lane parks
		 * its four dwordx4 of the piece, then lane l stores quarter (l & 3) of the pieces of records (l >> 2) + 16 i --
		 * four lanes = one complete 64-byte segment, sixteen records per store instruction.  Lanes past the end
		 * of the list hold a copy of the last item and store the same bytes again. */
		const uint32_t lane = threadIdx.x & 63;
		uint32_t *st_slot = stage + 64 * TG_STAGE_PITCH;	/* the 64 slot numbers */
		st_slot[lane] = slot;
		const uint32_t meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
		uint32_t bbraw;
		if (HMODE == 2) {
			const uint32_t *sb = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + TG_SOFT_BBK / 4;
			bbraw = 0;
#pragma unroll
			for (int q = 0; q < 4; q++)
				bbraw |= ((((sb[q] >> 7) & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
		} else
			bbraw = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_BBK];
		uint32_t bb = bbraw ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
		uint32_t nerr = 0;
		if (kflags & TGK_F_RM)
			bb = rm3014_correct(bb, nerr);
		const uint32_t code = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
		auto bits16 = [&](int q) {	/* type-1 bits 16 q .. 16 q + 15, one per byte */
			const uint32_t hw = (od[q >> 1] >> ((q & 1) * 16)) & 0xffff;
			uint4 o;
			o.x = spread4(hw);
			o.y = spread4(hw >> 4);
			o.z = spread4(hw >> 8);
			o.w = (q * 16 + 12 < TYPE1) ? spread4(hw >> 12) : 0u;
			return o;
		};
		uint4 *mine = (uint4 *)(stage + lane * TG_STAGE_PITCH);
#pragma unroll
		for (int c = 0; c < 5; c++) {
			if (c == 0) {
				/* bytes 0..15: type, flags, crc_ok[2], crc[2], code, slot; 16..31: SYNC fields (none), BBK errors */
				mine[0] = make_uint4((meta & 0xffffu) | (crc_ok << 16), crc, code, slot);
				mine[1] = make_uint4(0u, 0u, 0u, nerr);
				mine[2] = make_uint4(spread4(bb), spread4(bb >> 4), spread4(bb >> 8), spread4(bb >> 12) & 0x0000ffffu);
				mine[3] = bits16(0);
			} else {
#pragma unroll
				for (int i = 0; i < 4; i++)
					mine[i] = bits16(4 * c - 3 + i);
			}
			__builtin_amdgcn_s_waitcnt(0xc07f);	/* lgkmcnt(0): single wave, LDS visible */
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t rr = (lane >> 2) + 16 * i;
				const uint4 v = *(const uint4 *)(stage + rr * TG_STAGE_PITCH + 4 * (lane & 3));
				uint4 *dst = (uint4 *)(rec + (size_t)st_slot[rr] * TG_REC_BYTES + 64 * c + 16 * (lane & 3));
				*dst = v;
			}
			__builtin_amdgcn_wave_barrier();
		}
		if (!valid)
			return;
		if (wire) {
			uint32_t *wr = (uint32_t *)(wire + (size_t)slot * TG_WIRE_BYTES);
			uint32_t *wb = wr + TG_WIRE_W_BITS1;
			constexpr int NWD = (TYPE1 + 31) / 32;
#pragma unroll
			for (int q = 0; q < NWD - 1; q++)
				wb[q] = od[q];
			wb[NWD - 1] = (od[NWD - 1] & ((1u << (TYPE1 & 31)) - 1)) | (crc << TG_WIRE_SCHF_CRC_SHIFT);
			wr[0] = (meta & 0xff) | (((meta >> 8) & 0xff) << 8) | ((bb & 0x3fff) << 16);
		}
		return;
	}

	if (!valid)
		return;

	/* ---- outputs ---- */
	uint8_t *r = rec + (size_t)slot * TG_REC_BYTES;
	if (!wire_only) {
		uint4 *dst = (uint4 *)(r + (which ? TG_REC_BITS2 : TG_REC_BITS1));
		constexpr int NST = (TYPE1 + 15) / 16;
#pragma unroll
		for (int q = 0; q < NST; q++) {
			const uint32_t hw = (od[q >> 1] >> ((q & 1) * 16)) & 0xffff;
			uint4 o;
			o.x = spread4(hw);
			o.y = spread4(hw >> 4);
			o.z = spread4(hw >> 8);
			o.w = (q * 16 + 12 < TYPE1) ? spread4(hw >> 12) : 0u;	/* TYPE1 = 12 mod 16 */
#ifdef TG_EXP_NOSTORE
			if (q == 0 || hw == 0x12345u)
#endif
			dst[q] = o;
		}
	}
	if (!wire_only) {
		r[TG_REC_CRC_OK + which] = (uint8_t)crc_ok;
		*(uint16_t *)(r + TG_REC_CRC + 2 * which) = (uint16_t)crc;
	}

	/* optional bit-packed copy for transport (wave-uniform branch): tg_layout.h "Wire record".  The lanes of a slot
	 * write disjoint bytes: each block its payload words and its half (SCH/F: its field) of w[9], the primary lane
	 * the header word */
	uint32_t *wr = wire ? (uint32_t *)(wire + (size_t)slot * TG_WIRE_BYTES) : nullptr;
	if (wr) {
		uint32_t *wb = wr + (which ? TG_WIRE_W_BITS2 : TG_WIRE_W_BITS1);
		constexpr int NWD = (TYPE1 + 31) / 32;
		if (KIND == TG_KIND_432) {
#pragma unroll
			for (int q = 0; q < NWD - 1; q++)
				wb[q] = od[q];
			wb[NWD - 1] = (od[NWD - 1] & ((1u << (TYPE1 & 31)) - 1)) | (crc << TG_WIRE_SCHF_CRC_SHIFT);
		} else {
#pragma unroll
			for (int q = 0; q < NWD; q++)
				wb[q] = (q == NWD - 1) ? (od[q] & ((1u << (TYPE1 & 31)) - 1)) : od[q];
			((uint16_t *)(wr + TG_WIRE_W_CRC))[which] = (uint16_t)crc;
			if (KIND == TG_KIND_SB1) {	/* SB1 fills w[1..2]; w[3..4] are nobody else's */
				wr[3] = 0;
				wr[4] = 0;
			}
		}
	}

	if (KIND == TG_KIND_SB1) {
		/* SYNC PDU fields, lower_mac/tetra_lower_mac.c:284-297 */
		const uint32_t cc = FIELD_MSB(od, 4, 6), tn = FIELD_MSB(od, 10, 2) + 1;
		const uint32_t fn = FIELD_MSB(od, 12, 5), mn = FIELD_MSB(od, 17, 6);
		const uint32_t mcc = FIELD_MSB(od, 31, 10), mnc = FIELD_MSB(od, 41, 14);
		const uint32_t code = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
		if (!wire_only) {
			*(uint32_t *)(r + TG_REC_SBF0) = cc | (tn << 8) | (fn << 16) | (mn << 24);
			*(uint32_t *)(r + TG_REC_SBF1) = mcc | (mnc << 16);
			*(uint32_t *)(r + TG_REC_SBCODE) = code;
		}
		if (kflags & TGK_F_LOOKBACK) {
			/* device-walk batches (see k_lists2): sb_ok = one bit per grid slot "SB1 passed its CRC", sb_code = the slot's
			 * mask-table entry, masks = the batch's code table (open addressing, 0 = free: a code ends in binary 11) */
			/* one table access per DISTINCT code of the wave (a recording has one cell: every lane brings the same code, and
			 * a hundred thousand compare-and-swaps on one word would serialise): the first lane of each group looks its
			 * code up -- a plain read first, the atomic only while the slot reads free -- and hands the slot to the others */
			uint32_t *tbl = const_cast<uint32_t *>(masks);
			const bool live = valid && crc_ok;
			uint32_t myh = 0;
			unsigned long long todo = __ballot(live);
			while (todo) {
				const uint32_t l0 = (uint32_t)__builtin_ctzll(todo);
				const uint32_t c0 = __builtin_amdgcn_readlane(code, l0);
				uint32_t h = (c0 * 2654435761u) >> 20, probe = 0;
				if ((threadIdx.x & 63) == l0) {
					for (; probe < TG_LB_TBL; probe++) {
						uint32_t old = __hip_atomic_load(&tbl[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						if (old == 0u)
							old = atomicCAS(&tbl[h], 0u, c0);
						if (old == 0u || old == c0)
							break;
						h = (h + 1) & (TG_LB_TBL - 1);
					}
					if (probe == TG_LB_TBL) {	/* more codes than the table holds: the batch is handed to the host path */
						atomicOr(&tbl[TG_LB_TBL], 1u);
						h = 0;
					}
				}
				h = __builtin_amdgcn_readlane(h, l0);
				const bool mine = live && code == c0;
				if (mine)
					myh = h;
				todo &= ~__ballot(mine);
			}
			if (live) {
				sb_code[slot] = 1u + ((uint32_t)kflags >> 8) + myh;
				atomicOr(&sb_ok[slot >> 5], 1u << (slot & 31));
			}
		} else {
			sb_ok[idx] = crc_ok;
			sb_code[idx] = code;
		}
		if (block_mode && !wire_only) {	/* a block on its own: this lane also writes the header */
			r[TG_REC_TYPE] = (uint8_t)packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
			*(uint32_t *)(r + TG_REC_CODE) = 3u;
			*(uint32_t *)(r + TG_REC_SLOT) = slot;
		}
	} else if (block_mode) {
		if (wire_only)
			return;
		/* block mode (tgpu_plan_load_blocks): one block per record, no burst around it */
		const uint32_t meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
		r[TG_REC_TYPE] = (uint8_t)meta;
		r[TG_REC_FLAGS] = (uint8_t)(meta >> 8);
		*(uint32_t *)(r + TG_REC_CODE) = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
		*(uint32_t *)(r + TG_REC_SLOT) = slot;
	} else {
		/* BBK + header are written by the lane that owns the slot's "primary" block:
		 * SCH/F for NORM_1, BLK1 for NORM_2, SB2 for SYNC */
		const uint32_t meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
		const uint32_t btype = meta & 0xff;
		const bool primary = (KIND == TG_KIND_432) || (btype == TG_BURST_SYNC ? which == 1 : which == 0);
		if (primary) {
			uint32_t bbraw;
			if (HMODE == 2) {
				/* hard decision of the first 16 BBK soft values: bit = (value < 0) */
				const uint32_t *sb = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + TG_SOFT_BBK / 4;
				bbraw = 0;
#pragma unroll
				for (int q = 0; q < 4; q++)
					bbraw |= ((((sb[q] >> 7) & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
			} else
				bbraw = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_BBK];
			uint32_t bb = bbraw ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
			uint32_t nerr = 0;
			if (kflags & TGK_F_RM)		/* non-default: minimum-distance decoding of the (30,14) word first */
				bb = rm3014_correct(bb, nerr);
			if (!wire_only) {
				r[TG_REC_BBK_NERR] = (uint8_t)nerr;
				uint4 o;
				o.x = spread4(bb);
				o.y = spread4(bb >> 4);
				o.z = spread4(bb >> 8);
				o.w = spread4(bb >> 12) & 0x0000ffffu;	/* 14 type-1 bits (tetra_lower_mac.c:268-274) */
				*(uint4 *)(r + TG_REC_BBK) = o;
				r[TG_REC_TYPE] = (uint8_t)btype;
				r[TG_REC_FLAGS] = (uint8_t)(meta >> 8);
				*(uint32_t *)(r + TG_REC_CODE) = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
				*(uint32_t *)(r + TG_REC_SLOT) = slot;
			}
			if (wr)
				wr[0] = btype | (((meta >> 8) & 0xff) << 8) | ((bb & 0x3fff) << 16);
		}
	}
}

/*
 * k_clean<KIND>: optional pre-pass (tgpu_plan_set_fastpath).  A block whose received bits are exactly a code word
 * needs no trellis search: all four generators contain the term 1 and g1 is received at every step, so any other
 * path differs from the received word at the first step where its input differs -- the zero-distance path is the
 * unique minimum whatever the tie rule, and the decoder's answer is that path.  Its input bits follow from the g1
 * stream alone (G1 = 1 + D + D^4: u_k = r1_k ^ u_(k-1) ^ u_(k-4), start state 0), and it is the received word iff
 * the g2 bits it implies (G2 = 1 + D^2 + D^3 + D^4) equal the received ones.  Per 8-step block two table
 * look-ups in LDS do both (g_clean_lut).  Clean blocks are finished here (same vit_finish as the trellis kernels);
 * the others are appended to a list for k_vit, which then reads its item count from the device.
 * Results are identical with or without this pass.
 */
template <int KIND>
__global__ __launch_bounds__(256)
void k_clean(const uint32_t *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ packed,
	     const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec,
	     uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code, uint8_t *__restrict__ wire,
	     uint32_t *__restrict__ dirty_items, uint32_t *__restrict__ dirty_count, int kflags)
{
	constexpr int NBLK = vit_cfg<KIND>::NBLK;
	constexpr int NW = NBLK / 2;
	constexpr int NOD = (NBLK + 3) / 4;
	__shared__ uint16_t s_lut[8192];
	__shared__ uint16_t s_crc[512];
	for (int i = threadIdx.x; i < 8192 / 8; i += 256)
		((uint4 *)s_lut)[i] = ((const uint4 *)g_clean_lut)[i];
	for (int i = threadIdx.x; i < 256; i += 256) {
		s_crc[i] = c_tab.crc_lsb[i];
		s_crc[256 + i] = c_tab.crc_msb[i];
	}
	__syncthreads();
	const uint16_t *lutA = s_lut, *lutB = s_lut + 4096;
	const uint32_t lane = threadIdx.x & 63;

	for (uint32_t base = blockIdx.x * 256; base < nitems; base += gridDim.x * 256) {
		uint32_t idx = base + threadIdx.x;
		const bool valid = idx < nitems;
		if (!valid)
			idx = nitems - 1;
		uint32_t slot, which, item = items[idx];
		if (KIND == TG_KIND_216) {
			slot = item >> 1;
			which = item & 1;
		} else {
			slot = item;
			which = 0;
		}
		const uint32_t *pw = packed + (size_t)slot * TG_PACKED_WORDS + (which ? TG_PW_BLK2 : TG_PW_BLK1);
		const uint32_t midx = maskidx[slot];
		const uint32_t *mw = masks + (size_t)midx * TG_MASK_WORDS + vit_cfg<KIND>::MW;

		uint32_t od[NOD + 1];
#pragma unroll
		for (int i = 0; i <= NOD; i++)
			od[i] = 0;
		uint32_t state = 0, dirty = 0;
		uint32_t cur = pw[0] ^ mw[0];
		{	/* the four lead-in steps: six received bits, two g2 checks, four input bits */
			const uint32_t a = lutA[(cur >> 24) & 63];
			const uint32_t b = lutB[a & 0xff];
			dirty |= ((b >> 8) ^ (a >> 8)) & 3;
			od[0] = b & 15;
			state = tg_brev4(b & 15);
		}
#pragma unroll
		for (int d = 0; d < NW; d++) {
			if (d)
				cur = pw[d] ^ mw[d];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const bool last = (d == NW - 1) && h;		/* four steps + the flush steps */
				const uint32_t x = (h ? cur >> 12 : cur) & (last ? 0x3fu : 0xfffu);
				const uint32_t a = lutA[x];
				const uint32_t b = lutB[(state << 8) | (a & 0xff)];
				dirty |= ((b >> 8) ^ (a >> 8)) & (last ? 3u : 15u);
				const uint32_t u = b & (last ? 15u : 255u);
				constexpr int dummy = 0;
				(void)dummy;
				const int upos = 4 + 8 * (2 * d + h);
				od[upos >> 5] |= u << (upos & 31);
				if ((upos & 31) > 24 && !last)
					od[(upos >> 5) + 1] |= u >> (32 - (upos & 31));
				state = b >> 12;
			}
		}
		/* not a code word: hand the item to the trellis kernel (one atomic per wave) */
		const bool isdirty = valid && dirty != 0;
		const unsigned long long dm = __ballot(isdirty);
		if (dm) {
			uint32_t pos = 0;
			if (lane == (uint32_t)__builtin_ctzll(dm))
				pos = atomicAdd(dirty_count, (uint32_t)__builtin_popcountll(dm));
			pos = __shfl(pos, __builtin_ctzll(dm));
			if (isdirty)
				dirty_items[pos + __builtin_popcountll(dm & ((1ull << lane) - 1))] = item;
		}
		vit_finish<KIND, 1>(od, s_crc, valid && dirty == 0, slot, which, idx, midx, packed, masks, rec, sb_ok, sb_code, wire,
				    nullptr, kflags);
	}
}

/*
 * HMODE 1: survivor history in VGPRs -- chunks of 32 registers (8 blocks) written through
 *          the VGPR index mode (s_set_gpr_idx_on) with a wave-uniform block index, read back
 *          with static indices by the fully unrolled traceback.  No LDS for the trellis at
 *          all, so occupancy is set by registers: 2 waves/SIMD for SCH/F, 4 for the 216 blocks.
 * HMODE 2: soft input (BASELINE config 5): int8 soft values from k_front_soft's per-slot area instead of
 *          packed bits, 32-bit correlation metrics (tg_svit_*), history in VGPRs as in mode 1.
 */
template <int KIND, int HMODE>
__global__ __launch_bounds__(64, (HMODE == 2) ? (KIND == TG_KIND_SB1 ? 4 : 2) : (KIND == TG_KIND_432 ? 3 : 4))
void k_vit(const uint32_t *__restrict__ items, uint32_t nitems,
	   const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks,
	   const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec,
	   uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code, uint8_t *__restrict__ wire,
	   const uint32_t *__restrict__ softarea, int kflags, const uint32_t *__restrict__ nitems_dev)
{
	TG_TRACE_BEGIN;
	constexpr int NBLK = vit_cfg<KIND>::NBLK;
	constexpr int NW = NBLK / 2;			/* code words */
	if (nitems_dev) {	/* after k_clean: the list of blocks that still need the trellis was counted on the device */
		nitems = *nitems_dev;
		if (blockIdx.x * 64 >= nitems)
			return;
	}
	constexpr int NOD = (NBLK + 3) / 4;		/* dwords of decoded bits */
	constexpr int NCH = (NBLK + 7) / 8;		/* history chunks of 8 blocks */

	__shared__ uint16_t s_crc[512];
	/* record staging of the SCH/F kernel (vit_finish): 64 lanes x four dwordx4 at a pitch of 20 dwords + 64 slot numbers */
	__shared__ __attribute__((aligned(16))) uint32_t s_stage[(KIND == TG_KIND_432) ? 64 * TG_STAGE_PITCH + 64 : 4];

	const uint32_t lane = threadIdx.x;
	for (int i = lane; i < 256; i += 64) {
		s_crc[i] = c_tab.crc_lsb[i];
		s_crc[256 + i] = c_tab.crc_msb[i];
	}

	uint32_t idx = blockIdx.x * 64 + lane;
	const bool valid = idx < nitems;
	if (!valid)
		idx = nitems - 1;

	/* item: SB1 kernel -> position in the SYNC-slot list (items[] = slot ids);
	 *       216 kernel -> slot<<1 | which;  432 kernel -> slot id */
	uint32_t slot, which;
	if (KIND == TG_KIND_216) {
		const uint32_t it = items[idx];
		slot = it >> 1;
		which = it & 1;
	} else {
		slot = items[idx];
		which = 0;
	}

	const uint32_t *pw = packed + (size_t)slot * TG_PACKED_WORDS + (which ? TG_PW_BLK2 : TG_PW_BLK1);
	const uint32_t *mw;
	uint32_t midx = 0;
	if (KIND == TG_KIND_SB1) {
		mw = c_tab.sb1_mask;
	} else {
		midx = maskidx[slot];
		mw = masks + (size_t)midx * TG_MASK_WORDS + vit_cfg<KIND>::MW;
	}

	uint32_t od[NOD + 1];
#pragma unroll
	for (int i = 0; i <= NOD; i++)
		od[i] = 0;

	/* hard input: the block's descrambled code words are fetched in one burst (the 80-byte pitch means every lane
	 * touches its own cache lines; back to back they are fetched from HBM once) and parked in LDS, one column per
	 * lane; the trellis loop then has no global loads.  Reading them one per 16 steps instead refetched the same
	 * lines ~4x (FETCH_SIZE 183 MB for 42 MB of input on 500 k SCH/F blocks). */
	__shared__ uint32_t s_cw[(HMODE != 2) ? NW * 64 : 1];
	/* branch-metric table (vit_core.h, tg_bm_entry): six dwords per step pair and received triple */
	__shared__ __attribute__((aligned(16))) uint32_t s_bm[(HMODE != 2) ? TG_BM_WORDS : TG_PSOFT_TAB];
	auto bm = [&](int p, uint32_t e, uint32_t w[6]) {
		const uint32_t *q = s_bm + (8 * p + e) * 8;
		const uint4 a = *(const uint4 *)q;
		const uint2 b = *(const uint2 *)(q + 4);
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y;
	};
	tg_vit_state v;
	uint32_t cur = 0;
	if (HMODE != 2) {
		if (lane < 32)
			tg_bm_entry(lane >> 3, lane & 7, s_bm + 8 * lane);
#pragma unroll
		for (int g = 0; g < NW; g++)
			s_cw[g * 64 + lane] = pw[g] ^ mw[g];
		__syncthreads();
		tg_vit_init(v);
		cur = s_cw[lane];
		tg_vit_leadin_bm(v, cur >> 24, bm);
	}

	if (HMODE == 2) {
		/* soft input: 6 dwords (2 x 12 int8 = 16 trellis steps) per iteration from this block's soft area; the packed
		 * 16-bit soft trellis of vit_core.h (tg_pvit_*), branch metrics from the 512-entry table in LDS */
		for (int i = lane; i < TG_PSOFT_TAB; i += 64)
			s_bm[i] = tg_psoft_entry(i);
		__syncthreads();
		auto tab = [&](uint32_t idx) { return s_bm[idx]; };
		const uint32_t *sw = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + (which ? TG_SOFT_AREA2 / 4 : 0);
		/* the half-slot blocks: the wave's 64 soft areas (224 B each) come in ONCE, as 14 direct-to-LDS loads of 1 KB in
		 * which 14 neighbouring lanes cover one area (whole lines, each fetched one time), and the trellis reads its
		 * values from LDS.  Walking the areas 24 bytes per 16 steps per lane touched every line five times over ~20 us
		 * with a working set of the L2's size: 738 MB of traffic for 400 MB.  14 KB per wave: two waves per SIMD (what
		 * the SCH/F kernel runs with as well; the trellis has the instruction-level parallelism for it). */
		constexpr bool STAGED = (KIND == TG_KIND_216);
		constexpr int AREA_DW = (TG_SOFT_LEADIN_BYTES + TG_SOFT_BLOCK_BYTES * NBLK) / 4;	/* 56 */
		__shared__ __attribute__((aligned(16))) uint32_t s_soft[STAGED ? 64 * AREA_DW : 4];
		if (STAGED) {
			static_assert(!STAGED || AREA_DW % 4 == 0, "whole 16-byte pieces");
			constexpr int PIECES = AREA_DW / 4;	/* 14 */
			const uint32_t myoff = (uint32_t)(sw - softarea);
#pragma unroll
			for (int k = 0; k < PIECES; k++) {
				const uint32_t i = lane + 64u * k, r = i / PIECES, pc = i - PIECES * r;
				const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * r), (int)myoff);
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(softarea + (size_t)off + 4 * pc),
								 (__attribute__((address_space(3))) void *)(s_soft + 256 * k), 16, 0, 0);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		auto rd = [&](int i) { return STAGED ? s_soft[lane * AREA_DW + i] : sw[i]; };
		tg_pvit_state sv;
		tg_pvit_init(sv);
		{
			const uint32_t lw[2] = { rd(0), rd(1) };
			tg_pvit_leadin(sv, lw, (mw[0] >> 24) & 0x3f, tab);
		}
		/* software pipeline: the values (and mask word) of iteration g + 2 are loaded from global memory during
		 * iteration g; the table entries of a block are fetched from LDS while the block before it runs (the
		 * scheduling barriers keep the compiler from sinking the LDS reads to their first use) */
		uint32_t cw[6], nx[6], m = mw[0], nm = mw[NW > 1 ? 1 : 0];
#pragma unroll
		for (int q = 0; q < 6; q++) {
			cw[q] = rd(2 + q);
			nx[q] = rd(2 + (NW > 1 ? 6 : 0) + q);
		}
		uint32_t ta[12], tb[12];
		tg_psoft_fetch<0, 12>(cw, m & 0xfff, tab, ta);
		tg_v32 H[NCH];
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			const int nblk_c = (NBLK - 8 * c >= 8) ? 8 : (NBLK - 8 * c);
			const int nit = nblk_c / 2;
			const bool lastchunk = (c == NCH - 1);
			const int nloop = lastchunk ? nit - 1 : nit;
#pragma unroll 1
			for (int it = 0; it < nloop; it++) {
				const int g = 4 * c + it;
				const int g2 = (g + 2 < NW) ? g + 2 : NW - 1;
				uint32_t nn[6];
#pragma unroll
				for (int q = 0; q < 6; q++)
					nn[q] = rd(2 + 6 * g2 + q);
				const uint32_t nnm = mw[g2];
				tg_psoft_fetch<0, 12>(cw + 3, (m >> 12) & 0xfff, tab, tb);
				__builtin_amdgcn_sched_barrier(0);
				uint32_t h[4];
				tg_pvit_block<false>(sv, ta, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + d] = h[d];
				__builtin_amdgcn_sched_barrier(0);
				tg_psoft_fetch<0, 12>(nx, nm & 0xfff, tab, ta);
				__builtin_amdgcn_sched_barrier(0);
				tg_pvit_block<false>(sv, tb, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + 4 + d] = h[d];
				tg_vit_normalize(sv);	/* once per 16 steps: twelve metric bits stay exact (vit_core.h) */
#pragma unroll
				for (int q = 0; q < 6; q++) {
					cw[q] = nx[q];
					nx[q] = nn[q];
				}
				m = nm;
				nm = nnm;
			}
			if (lastchunk) {
				uint32_t h[4];
				tg_psoft_fetch<0, 6>(cw + 3, (m >> 12) & 0xfff, tab, tb);
				tg_pvit_block<false>(sv, ta, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + d] = h[d];
				tg_pvit_block<true>(sv, tb, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + 4 + d] = h[d];
			}
		}
		/* traceback from state 0, one nibble per four-step block */
		uint32_t pos = 0;
#pragma unroll
		for (int b = NBLK - 1; b >= 0; b--) {
			const int c = b >> 3, o = 4 * (b & 7);
			const uint32_t hi = tg_ptrace_hop(H[c][o + 2], H[c][o + 3], pos);
			const uint32_t lo = tg_ptrace_hop(H[c][o], H[c][o + 1], pos);
			od[b >> 2] |= (lo | (hi << 4)) << ((b & 3) * 8);
		}
	} else {
		tg_v32 H[NCH];
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			const int nblk_c = (NBLK - 8 * c >= 8) ? 8 : (NBLK - 8 * c);
			const int nit = nblk_c / 2;
			const bool lastchunk = (c == NCH - 1);
			const int nloop = lastchunk ? nit - 1 : nit;
#pragma unroll 1
			for (int it = 0; it < nloop; it++) {
				const int g = 4 * c + it;
				const uint32_t nxt = s_cw[(g + 1) * 64 + lane];
				uint32_t h[4];
				tg_vit_block_bm<false>(v, cur, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + d] = h[d];
				tg_vit_block_bm<false>(v, cur >> 12, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + 4 + d] = h[d];
				if (KIND == TG_KIND_432 && g == 8)
					tg_vit_normalize(v);
				cur = nxt;
			}
			if (lastchunk) {
				uint32_t h[4];
				tg_vit_block_bm<false>(v, cur, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + d] = h[d];
				tg_vit_block_bm<true>(v, cur >> 12, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + 4 + d] = h[d];
			}
		}
		/* block-wise traceback from state 0, all register indices static */
		uint32_t s = 0;
#pragma unroll
		for (int b = NBLK - 1; b >= 0; b--) {
			const int c = b >> 3, o = 4 * (b & 7);
			const uint32_t byte = hist_byte(H[c][o], H[c][o + 1], H[c][o + 2], H[c][o + 3], s);
			od[b >> 2] |= byte << ((b & 3) * 8);
			s = tg_brev4(byte);
		}
	}

	__syncthreads();	/* s_crc visible (single wave, but keep the compiler honest) */
	vit_finish<KIND, HMODE>(od, s_crc, valid, slot, which, idx, midx, packed, masks, rec, sb_ok, sb_code, wire, softarea, kflags,
				(KIND == TG_KIND_432) ? s_stage : nullptr);
	TG_TRACE_END(1u + (uint32_t)KIND, 8u);
}

/* ------------------------------------------------------------------------- */
/* k_burst: one workgroup per burst, the 16 states of a trellis across 16 lanes (small batches) */
/* ------------------------------------------------------------------------- */
/*
 * The kernel shape BASELINE.json's north star names: one workgroup per burst, its type-5 bits staged in LDS, the
 * 16-state add-compare-select as a butterfly ACROSS LANES.  It exists for small batches (the drop-in channel API
 * with a handful of bursts per flush), where the lane-per-trellis kernels leave 63 of 64 lanes idle and a flush is a
 * chain of eight launches: here a flush is two (k_burst<true> for the SB1 blocks of the SYNC slots, then k_burst<false>
 * for everything).
 *
 *   - descriptors and channels of the slot and the 255 before it + the channels' carry-in codes -> LDS in one parallel
 *     load (small batches keep them in mapped host memory: every dependent read would be a PCIe round trip), then the
 *     slot's 510 bytes -> LDS; every thread de-interleaves, de-punctures (2/3: the order of the bits is the type-3
 *     order) and descrambles its share of a block: r[i] = byte[(a (i + 1)) mod K] != 0, XOR bit (same position) of
 *     the scrambling sequence in its linear form (parity(code & lfsr_lin[pos]), lower_mac/tetra_scramb.c:34-50);
 *   - the scrambling code of the slot = the SYNC PDU of the latest SYNC slot at or before it (same channel) whose SB1
 *     passed its CRC, else the channel's carry-in (lower_mac/tetra_lower_mac.c:179-186, 291-300): pass 1 leaves
 *     (crc_ok, code) per SYNC slot, pass 2's workgroups look backwards through them -- no forward-fill launches;
 *   - trellis: a lane of a 16-lane row holds one state as metric << 8 | survivor byte (the lane-per-trellis word, one
 *     state per lane): same tie rule and register-exchange history as vit_core.h, so the 8-step blocks, the 16
 *     history bytes per block (one ds_write_b8 per lane) and the block-wise traceback are the same too.  The
 *     butterflies run in place with DPP partner exchanges and increments prepared by all threads (comment at s_inc
 *     below).  Row 0 of wave 0 decodes the slot's first block, row 1 the second, side by side;
 *   - CRC-16 as the linear map it is (c_tab.crc_lin), type-1 bits at one byte per bit, BBK, header: the record of the
 *     lane-per-trellis path, byte for byte (tests/test_gpu_parity.py::test_burst_kernel_equals_batch_kernels); the
 *     burst type is written last -- behind a system-wide fence when the owner polls it (marks).
 */
/* block parameters per kind on the device (tg_layout.h's host inlines: lower_mac/tetra_lower_mac.c:55-102) */
__device__ __forceinline__ uint32_t tgb_K(int kind)    { return kind == TG_KIND_SB1 ? 120u : kind == TG_KIND_216 ? 216u : 432u; }
__device__ __forceinline__ uint32_t tgb_a(int kind)    { return kind == TG_KIND_SB1 ? 11u : kind == TG_KIND_216 ? 101u : 103u; }
__device__ __forceinline__ uint32_t tgb_nblk(int kind) { return kind == TG_KIND_SB1 ? 10u : kind == TG_KIND_216 ? 18u : 36u; }
__device__ __forceinline__ uint32_t tgb_t1(int kind)   { return kind == TG_KIND_SB1 ? 60u : kind == TG_KIND_216 ? 124u : 268u; }

__device__ __forceinline__ uint32_t tgb_out_g12(uint32_t p, uint32_t u)
{
	/* (g1, g2) of the transition from state p with input u: out(j, 0) = {0,11,6,13,5,14,3,8} (lower_mac/viterbi_cch.c:35-40),
	 * g1 = bit 3, g2 = bit 2; complemented for u = 1 and for p >= 8 (every generator holds 1 and D^4) */
	const uint32_t tab = 0x83e5d6b0u;			/* nibble j = out(j, 0) */
	uint32_t o = (tab >> (4 * (p & 7))) & 15u;
	if (u)
		o ^= 15u;
	if (p & 8)
		o ^= 15u;
	return o >> 2;						/* bit 1 = g1, bit 0 = g2 */
}

#define TGB_MAX_STEPS (4 + 8 * 36)	/* the SCH/F trellis: 292 steps */

#ifdef TGB_TIMING	/* experiment build: phase time stamps of workgroup 0 (tools/flush_lat.c prints them) */
__device__ unsigned long long g_tgb_stamp[16];
#define TGB_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tgb_stamp[k] = wall_clock64(); } while (0)
extern "C" int tgk_burst_stamps(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tgb_stamp), sizeof(g_tgb_stamp));
}
#else
#define TGB_STAMP(k) do { } while (0)
#endif

/* the body: slot i of the batch by the calling workgroup (256 threads).  SB1_PASS is a constant at k_burst's two call sites;
 * the ring kernel below calls it with both values in turn (one copy of the LDS areas: the function's own) */
__device__ __forceinline__
void burst_body(const bool SB1_PASS, const uint32_t i, const uint8_t *__restrict__ stream, const uint64_t *__restrict__ slot_desc,
		const uint32_t *__restrict__ slot_chan, const uint32_t *__restrict__ chan_code, uint32_t nslots, uint32_t nchan,
		uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code, uint8_t *__restrict__ rec, uint32_t *__restrict__ maskidx,
		uint32_t *__restrict__ masks, int marks)
{
	__shared__ uint32_t s_slot32[128];
	/* the descriptors and channels of this slot and the 255 before it, the channels' carry-in codes: small batches keep
	 * them in mapped host memory, where every dependent read is a PCIe round trip -- fetch them in one */
	__shared__ uint64_t s_desc[256];
	__shared__ uint32_t s_chan[256], s_ccode[64];
	__shared__ uint8_t s_r[2][432 + 16];		/* received type-3 bits per block, descrambled */
	__shared__ uint8_t s_hist[2][37][16];		/* [.][36]: spare row for a block that only runs along */
	__shared__ uint8_t s_od[2][40];			/* decoded bytes (8 bits per trellis block) */
	__shared__ uint32_t s_code, s_nonbin;
	uint8_t *s_slot = (uint8_t *)s_slot32;
	const uint32_t tid = threadIdx.x;
	const uint32_t back = i < 255u ? i : 255u;
	TGB_STAMP(0);
	if (tid <= back) {
		s_desc[tid] = slot_desc[i - tid];
		s_chan[tid] = slot_chan[i - tid];
	}
	if (tid < 64 && tid < nchan)
		s_ccode[tid] = chan_code[tid];
	__syncthreads();
	const uint64_t d = s_desc[0];
	TGB_STAMP(1);
	const uint32_t type = TG_DESC_TYPE(d);
	const uint8_t *base = stream + TG_DESC_OFF(d);
	uint8_t *r = rec + (size_t)i * TG_REC_BYTES;
	if (SB1_PASS && type != TG_BURST_SYNC)
		return;
	/* a burst type this path does not decode: no record, but the slot still has a code in force (the batch kernels'
	 * forward fill gives every slot one, and tgpu_plan_final_codes() reads the channel's last slot whatever its type) */
	const bool ignored = type != TG_BURST_SYNC && type != TG_BURST_NORM_1 && type != TG_BURST_NORM_2;
	if (tid == 0)
		s_nonbin = 0;
	__syncthreads();
	if (!ignored) {	/* the slot -> LDS (byte loads: any alignment), non-binary test */
		const uint32_t b0 = base[2 * tid < 510 ? 2 * tid : 509], b1 = base[2 * tid + 1 < 510 ? 2 * tid + 1 : 509];
		if (2 * tid < 510)
			s_slot[2 * tid] = (uint8_t)b0;
		if (2 * tid + 1 < 510)
			s_slot[2 * tid + 1] = (uint8_t)b1;
		if ((b0 | b1) > 1)
			s_nonbin = 1;
	}
	/* the code in force for this slot: look backwards through the SYNC slots of the batch (pass 1 left their results) */
	if (tid == 0) {
		const uint32_t ch = s_chan[0];
		uint32_t code = ch < 64 ? s_ccode[ch] : chan_code[ch];
		if (!SB1_PASS) {
			bool open = true;		/* still inside the channel's run and no good SYNC slot seen */
			for (uint32_t t = 0; t <= back && open; t++) {
				if (s_chan[t] != ch)
					open = false;
				else if (TG_DESC_TYPE(s_desc[t]) == TG_BURST_SYNC && sb_ok[i - t]) {
					code = sb_code[i - t];
					open = false;
				}
			}
			if (open)			/* (a run longer than the window: the rest from memory) */
				for (int j = (int)i - 256; j >= 0 && slot_chan[j] == ch; j--)
					if (TG_DESC_TYPE(slot_desc[j]) == TG_BURST_SYNC && sb_ok[j]) {
						code = sb_code[j];
						break;
					}
		}
		s_code = code;
		if (ignored) {
			r[TG_REC_TYPE] = TG_BURST_NONE;
			maskidx[i] = i;
			masks[(size_t)i * TG_MASK_WORDS + TG_MW_CODE] = code;
		}
	}
	if (ignored)		/* (workgroup-uniform; SB1_PASS never gets here with one) */
		return;
	__syncthreads();
	const uint32_t code = s_code;
	TGB_STAMP(2);

	/* blocks of this burst: kind and where its type-4 bits sit in the slot (phy/tetra_burst.c:31-47) */
	int kind[2] = { -1, -1 };
	uint32_t o1[2] = { 0, 0 }, o2[2] = { 0, 0 }, bcode[2] = { code, code };
	if (type == TG_BURST_SYNC) {
		kind[0] = TG_KIND_SB1; o1[0] = TG_SB_BLK1_OFF; bcode[0] = 3;		/* lower_mac/tetra_scramb.h:14 */
		if (!SB1_PASS) { kind[1] = TG_KIND_216; o1[1] = TG_SB_BLK2_OFF; }
	} else if (type == TG_BURST_NORM_2) {
		kind[0] = TG_KIND_216; o1[0] = TG_NDB_BLK1_OFF;
		kind[1] = TG_KIND_216; o1[1] = TG_NDB_BLK2_OFF;
	} else {
		kind[0] = TG_KIND_432; o1[0] = TG_NDB_BLK1_OFF; o2[0] = TG_NDB_BLK2_OFF;
	}
#pragma unroll
	for (int b = 0; b < 2; b++) {
		if (kind[b] < 0)
			continue;
		const uint32_t K = tgb_K(kind[b]), a = tgb_a(kind[b]);
		for (uint32_t t3 = tid; t3 < K; t3 += 256) {
			const uint32_t j = (a * (t3 + 1)) % K;		/* type3[i] = type4[(a (i + 1)) mod K] */
			const uint32_t byte = s_slot[j < 216 ? o1[b] + j : o2[b] + j - 216];
			s_r[b][t3] = (uint8_t)((byte != 0) ^ (__popc(bcode[b] & c_tab.lfsr_lin[j]) & 1));
		}
	}
	__syncthreads();

	/*
	 * The trellis: in-place butterflies across a 16-lane row, partner exchange with DPP row operations.
	 * Predecessors j and j + 8 (they differ in the oldest state bit) produce 2 j and 2 j + 1 (which differ in the
	 * newest): if the two lanes of such a pair swap words and each keeps the better candidate of its successor, no
	 * word ever has to travel further -- the lane <-> state map rotates by one bit per step instead (state of lane L
	 * before step k = rotl4(L, k mod 4); the pair's lanes differ in physical bit 3 - k mod 4) and is the identity
	 * again every four steps, in particular wherever history bytes are extracted.  The partner's word arrives with
	 * one DPP move (row_ror:8, quad permutes for bits 1 and 0) or two (row_shl:4 / row_shr:4 under bank masks for
	 * bit 2).  What a lane adds to its own and to its partner's word at step k -- branch metric of its successor from
	 * either predecessor, the tie / decision bit on the candidate from j + 8 -- does not depend on the metrics: all 256
	 * threads prepare these increments for the whole block up front (one 32-bit word per step and lane in LDS), and the
	 * serial loop is read, exchange, two adds, one min per step.
	 */
	__shared__ uint32_t s_inc[2][TGB_MAX_STEPS][16];
	TGB_STAMP(3);
	{
		const uint32_t L = tid & 15, kq = tid >> 4;		/* lane of the row; step index modulo 16 */
		const uint32_t c = kq & 3;				/* = k mod 4 for every step this thread prepares */
		const uint32_t sig = ((L << c) | (L >> (4 - c))) & 15;	/* the lane's state before such a step */
		const uint32_t pv = (L >> (3 - c)) & 1;		/* 0: it holds predecessor j and becomes 2 j; 1: j + 8 -> 2 j + 1 */
		const uint32_t e0 = tgb_out_g12(sig & 7, pv);		/* expected (g1, g2) from predecessor j; from j + 8: the complement */
		const bool odd = kq & 1;				/* one received bit (g1) instead of two */
#pragma unroll
		for (int b = 0; b < 2; b++) {
			if (kind[b] < 0)
				continue;
			const uint32_t nst = 4 + 8 * tgb_nblk(kind[b]);
			const uint8_t *rr = s_r[b];
			constexpr int NIT = (TGB_MAX_STEPS + 15) / 16;
			uint32_t ra[NIT], rb[NIT];		/* all the received bits first: one LDS latency, not one per step */
#pragma unroll
			for (int it = 0; it < NIT; it++) {
				const uint32_t k = kq + 16 * it;
				const uint32_t p3 = (k + 4 < nst) ? 3 * (k >> 1) : 0;	/* (the last four steps are the flush: nothing received) */
				ra[it] = rr[odd ? p3 + 2 : p3];
				rb[it] = rr[p3 + 1];
			}
#pragma unroll
			for (int it = 0; it < NIT; it++) {
				const uint32_t k = kq + 16 * it;
				if (k >= nst)
					break;
				const uint32_t tie = 1u << (k < 4 ? k : (k - 4) & 7);
				uint32_t d0 = 0, d1 = 0;
				if (k + 4 < nst) {
					if (odd) {
						d0 = ra[it] ^ (e0 >> 1);
						d1 = 1 - d0;
					} else {
						const uint32_t x0 = ((ra[it] << 1) | rb[it]) ^ e0;
						d0 = (x0 & 1) + (x0 >> 1);
						d1 = 2 - d0;
					}
				}
				/* low half: what the lane adds to its own word for its own successor; high half: what it adds
				 * to its own word for the PARTNER's successor (the other input bit: expected bits complemented, so
				 * the distances swap).  A word from predecessor j + 8 carries the tie / decision bit. */
				const uint32_t ca = d0 << 8, cb = d1 << 8;
				s_inc[b][k][L] = pv ? ((cb + tie) | ((ca + tie) << 16)) : (ca | (cb << 16));
			}
		}
	}
	__syncthreads();

	if (tid < 64) {		/* wave 0: row 0 = first block, row 1 = second */
		const uint32_t lane = tid;
		TGB_STAMP(4);
		const uint32_t row = lane >> 4, st = lane & 15;
		const int mykind = row < 2 ? kind[row] : -1;
		const uint32_t nblk = mykind >= 0 ? tgb_nblk(mykind) : 0;
		const uint32_t nblk_max = max(kind[0] >= 0 ? tgb_nblk(kind[0]) : 0u, kind[1] >= 0 ? tgb_nblk(kind[1]) : 0u);
		const uint32_t *inc = &s_inc[row & 1][0][st];
		uint32_t W = (st == 0) ? 0u : (1000u << 8);
#define TGB_ACS(C, w)												\
		{												\
			/* own candidate; the candidate for the partner, which the partner picks up with a DPP move */	\
			uint32_t x, g, P;									\
			asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"	\
			    : "=v"(x) : "v"(W), "v"(w));							\
			asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"	\
			    : "=v"(g) : "v"(W), "v"(w));							\
			if ((C) == 0)										\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x128, 0xf, 0xf, false);	/* row_ror:8 */	\
			else if ((C) == 1) {									\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x104, 0xf, 0x5, false);	/* row_shl:4, banks 0, 2 */	\
				P = (uint32_t)__builtin_amdgcn_update_dpp((int)P, (int)g, 0x114, 0xf, 0xa, false);	/* row_shr:4, banks 1, 3 */	\
			} else if ((C) == 2)									\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x4e, 0xf, 0xf, false);	/* quad_perm:[2,3,0,1] */	\
			else											\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0xb1, 0xf, 0xf, false);	/* quad_perm:[1,0,3,2] */	\
			W = x < P ? x : P;									\
		}
		{	/* four lead-in steps (type-3 bits 0..5) */
			const uint32_t w0 = inc[0], w1 = inc[16], w2 = inc[32], w3 = inc[48];
			TGB_ACS(0, w0) TGB_ACS(1, w1) TGB_ACS(2, w2) TGB_ACS(3, w3)
			W &= ~0xffu;
		}
		/* blocks of eight steps on twelve bits (vit_core.h), two per iteration with alternating register sets so that
		 * a block's increments are read from LDS while the block before it runs; a row with fewer blocks than the
		 * other runs along, its history bytes going to the spare row of s_hist */
		uint32_t wa[8], wb[8];
#define TGB_FETCH(dst, blk)											\
		{												\
			const uint32_t *q = inc + 16 * (4 + 8 * ((blk) < nblk ? (blk) : 0));			\
			_Pragma("unroll") for (int k = 0; k < 8; k++)						\
				dst[k] = q[16 * k];								\
		}
#define TGB_BLOCK(wv, blk)											\
		{												\
			TGB_ACS(0, wv[0]) TGB_ACS(1, wv[1]) TGB_ACS(2, wv[2]) TGB_ACS(3, wv[3])			\
			TGB_ACS(0, wv[4]) TGB_ACS(1, wv[5]) TGB_ACS(2, wv[6]) TGB_ACS(3, wv[7])			\
			s_hist[row & 1][(blk) < nblk ? (blk) : 36][st] = (uint8_t)W;				\
			W &= ~0xffu;										\
		}
		TGB_FETCH(wa, 0u)
		for (uint32_t b = 0; b < nblk_max; b += 2) {
			TGB_FETCH(wb, b + 1)
			TGB_BLOCK(wa, b)
			if (b + 1 >= nblk_max)
				break;
			TGB_FETCH(wa, b + 2)
			TGB_BLOCK(wb, b + 1)
		}
#undef TGB_FETCH
#undef TGB_BLOCK
#undef TGB_ACS
		TGB_STAMP(5);
		/* block-wise traceback from state 0 (row leaders) */
		if (st == 0 && mykind >= 0) {
			uint32_t sidx = 0;
			for (int b = (int)nblk - 1; b >= 0; b--) {
				const uint32_t byte = s_hist[row][b][sidx];
				s_od[row][b] = (uint8_t)byte;
				sidx = tg_brev4(byte);
			}
		}
	}
	__syncthreads();

	TGB_STAMP(6);
	/* CRC-16 per block (thread 0 / 1), then the record */
	__shared__ uint32_t s_crcv[2], s_okv[2];
	/* the CRC is linear in the decoded bits: every thread takes the bits i = tid and tid + 256 of both blocks, XORs
	 * their table vectors, the waves fold theirs (one memory latency + a reduction instead of 36 dependent look-ups) */
	__shared__ uint32_t s_cpart[4];
	{
		uint32_t v = 0;			/* block 0 in the low half, block 1 in the high half */
#pragma unroll
		for (int b = 0; b < 2; b++) {
			if (kind[b] < 0)
				continue;
			const uint32_t nbits = 8 * (tgb_nblk(kind[b]) - 1) + 4;
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const uint32_t ib = tid + 256 * h;
				if (ib < nbits && ((s_od[b][ib >> 3] >> (ib & 7)) & 1))
					v ^= (uint32_t)c_tab.crc_lin[kind[b]][ib] << (16 * b);
			}
		}
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1)
			v ^= (uint32_t)__shfl_xor((int)v, m, 64);
		if ((tid & 63) == 0)
			s_cpart[tid >> 6] = v;
	}
	__syncthreads();
	if (tid < 2 && kind[tid] >= 0) {
		const uint32_t all = s_cpart[0] ^ s_cpart[1] ^ s_cpart[2] ^ s_cpart[3];
		const uint32_t crc = ((all >> (16 * tid)) & 0xffff) ^ c_tab.crc_aff[kind[tid]];
		s_crcv[tid] = crc;
		s_okv[tid] = (crc == 0x1d0f);
	}
	__syncthreads();
	TGB_STAMP(7);
	if (SB1_PASS) {
		if (tid == 0) {
			const uint8_t *od = s_od[0];
			uint32_t w0 = od[0] | (od[1] << 8) | (od[2] << 16) | ((uint32_t)od[3] << 24);
			uint32_t w1 = od[4] | (od[5] << 8) | (od[6] << 16) | ((uint32_t)od[7] << 24);
			const uint32_t ow[3] = { w0, w1, 0 };
			const uint32_t cc = FIELD_MSB(ow, 4, 6), mcc = FIELD_MSB(ow, 31, 10), mnc = FIELD_MSB(ow, 41, 14);
			sb_ok[i] = s_okv[0];
			sb_code[i] = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
		}
		return;
	}
	/* type-1 bits at one byte per bit: block 0 at @48, block 1 at @176 */
#pragma unroll
	for (int b = 0; b < 2; b++) {
		if (kind[b] < 0)
			continue;
		const uint32_t n1 = tgb_t1(kind[b]);
		const uint32_t span = (n1 + 15) & ~15u;			/* the batch kernels store whole 16-byte groups, zero padded */
		uint8_t *dst = r + (b ? TG_REC_BITS2 : TG_REC_BITS1);
		for (uint32_t k = tid; k < span; k += 256)
			dst[k] = k < n1 ? (uint8_t)((s_od[b][k >> 3] >> (k & 7)) & 1) : 0;
	}
	if (tid < 32) {		/* BBK: 30 bits in stream order, descrambled, the first 14 kept (tetra_lower_mac.c:268-274) */
		uint8_t bit = 0;
		if (tid < 14) {
			const uint32_t pos = (type == TG_BURST_SYNC) ? TG_SB_BBK_OFF + tid : TG_NDB_BBK1_OFF + tid;
			bit = (uint8_t)((s_slot[pos] != 0) ^ (__popc(code & c_tab.lfsr_lin[tid]) & 1));
		}
		if (tid < 16)
			r[TG_REC_BBK + tid] = bit;
	}
	if (tid == 0) {
		r[TG_REC_FLAGS] = s_nonbin ? TG_FLAG_NONBINARY : 0;
		r[TG_REC_CRC_OK] = (uint8_t)s_okv[0];
		r[TG_REC_CRC_OK + 1] = kind[1] >= 0 ? (uint8_t)s_okv[1] : 0;
		*(uint16_t *)(r + TG_REC_CRC) = (uint16_t)s_crcv[0];
		*(uint16_t *)(r + TG_REC_CRC + 2) = kind[1] >= 0 ? (uint16_t)s_crcv[1] : 0;
		*(uint32_t *)(r + TG_REC_CODE) = code;
		*(uint32_t *)(r + TG_REC_SLOT) = i;
		r[TG_REC_BBK_NERR] = 0;
		maskidx[i] = i;		/* what tgpu_plan_final_codes() reads: the code in force at this slot */
		masks[(size_t)i * TG_MASK_WORDS + TG_MW_CODE] = code;
		if (type == TG_BURST_SYNC) {
			const uint8_t *od = s_od[0];
			uint32_t w0 = od[0] | (od[1] << 8) | (od[2] << 16) | ((uint32_t)od[3] << 24);
			uint32_t w1 = od[4] | (od[5] << 8) | (od[6] << 16) | ((uint32_t)od[7] << 24);
			const uint32_t ow[3] = { w0, w1, 0 };
			const uint32_t cc = FIELD_MSB(ow, 4, 6), tn = FIELD_MSB(ow, 10, 2) + 1;
			const uint32_t fn = FIELD_MSB(ow, 12, 5), mn = FIELD_MSB(ow, 17, 6);
			const uint32_t mcc = FIELD_MSB(ow, 31, 10), mnc = FIELD_MSB(ow, 41, 14);
			*(uint32_t *)(r + TG_REC_SBF0) = cc | (tn << 8) | (fn << 16) | (mn << 24);
			*(uint32_t *)(r + TG_REC_SBF1) = mcc | (mnc << 16);
			*(uint32_t *)(r + TG_REC_SBCODE) = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
		}
	}
	/* the burst type is the record's completion mark: written last, after every thread's stores are visible system
	 * wide.  A host that holds the records in mapped memory (the channel API's small batches) presets the byte to
	 * TG_REC_PENDING and polls it instead of paying for a stream synchronise */
	if (marks)		/* (wave-uniform; on records in device memory the system-wide fence would only cost cache write-backs) */
		__threadfence_system();
	__syncthreads();
	if (tid == 0)
		*(volatile uint8_t *)(r + TG_REC_TYPE) = (uint8_t)type;
	TGB_STAMP(8);
}

template <bool SB1_PASS>
__global__ __launch_bounds__(256)
void k_burst(const uint8_t *__restrict__ stream, const uint64_t *__restrict__ slot_desc, const uint32_t *__restrict__ slot_chan,
	     const uint32_t *__restrict__ chan_code, uint32_t nslots, uint32_t nchan, uint32_t *__restrict__ sb_ok,
	     uint32_t *__restrict__ sb_code, uint8_t *__restrict__ rec, uint32_t *__restrict__ maskidx, uint32_t *__restrict__ masks,
	     int marks)
{
	burst_body(SB1_PASS, blockIdx.x, stream, slot_desc, slot_chan, chan_code, nslots, nchan, sb_ok, sb_code, rec, maskidx, masks, marks);
}

/*
 * k_burst_ring: the same decode by workgroups that STAY (TGPU_OPT_RING; the channel API's flushes of up to TG_RING_MAX
 * bursts).  A flush through k_burst is a kernel launch (7.5 us from the host's call to the first instruction's result back
 * on the host, tools/ubench/persist_rtt.hip) and two dependent reads over PCIe (descriptors, then the slot); a kernel that is
 * already running sees a request in mapped host memory after one PCIe read and answers in 2.9 us.
 *   - the request = one cache line the host fills (struct tg_ring_msg: descriptors, carry-in code, count) and numbers last;
 *     workgroup 0 polls it, takes the line in one 16-lane load (the number stands in both 32-byte halves: a half that shows
 *     the new number shows its new fields), hands it to the other workgroups through a box in device memory and all decode
 *     slot blockIdx.x: pass 1 (SB1 of SYNC slots), a barrier across the workgroups when more than one slot may need the
 *     result, pass 2; records and completion marks as k_burst writes them (the host polls the marks);
 *   - the workgroups leave on the host's stop request or when no request has come for idle_ticks of the 100 MHz clock
 *     (workgroup 0 decides and tells the others through the box; it clears `alive` last): a channel that falls silent frees
 *     its compute units, and the next flush starts the kernel again -- from `served`, so a request that was posted while the
 *     workgroups were leaving is not lost.
 */
__device__ __forceinline__ void ring_barrier(uint32_t *bar, uint32_t &epoch, uint32_t G)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		epoch++;
		__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * G)
			__builtin_amdgcn_s_sleep(1);
	}
	__syncthreads();
}

__global__ __launch_bounds__(256)
void k_burst_ring(tg_ring_msg *ring, tg_ring_box *box, const uint8_t *__restrict__ stream, uint32_t *__restrict__ sb_ok,
		  uint32_t *__restrict__ sb_code, uint8_t *__restrict__ rec, uint32_t *__restrict__ maskidx,
		  uint32_t *__restrict__ masks, uint32_t start_seq, uint32_t launch_no, unsigned long long idle_ticks)
{
	__shared__ uint32_t s_cmd[4];
	const uint32_t tid = threadIdx.x, w = blockIdx.x, G = gridDim.x;
	uint32_t last = start_seq, epoch = box->bar / G;	/* (the counter only grows: a new launch goes on where the last one stopped) */
	for (;;) {
		if (tid < 64) {
			uint32_t r = last, n = 0, hs = 0;
			if (w == 0) {
				const unsigned long long t0 = wall_clock64();
				for (;;) {
					/* the request line: lane l takes dword l */
					const uint32_t v = tid < 16 ? __hip_atomic_load((uint32_t *)ring + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
					const uint32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 15);
					if (r0 == r1 && r0 != last) {
						r = r0;
						n = __builtin_amdgcn_readlane(v, 1);
						hs = __builtin_amdgcn_readlane(v, 2);
						if (r != TG_RING_STOP) {
							if (tid >= 3 && tid < 12)	/* code, four descriptors */
								((uint32_t *)box)[tid] = v;
							if (tid == 1 || tid == 2)
								((uint32_t *)box)[tid] = v;
						}
						break;
					}
					if (wall_clock64() - t0 > idle_ticks) {
						r = TG_RING_STOP;
						break;
					}
				}
				TGB_STAMP(9);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				if (tid == 0) {		/* (leaving is told in a word of its own: the next launch must not find a stale "stop" in seq) */
					if (r == TG_RING_STOP)
						__hip_atomic_store(&box->stop, launch_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
					else
						__hip_atomic_store(&box->seq, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
				}
			} else {
				if (tid == 0)
					for (;;) {
						r = __hip_atomic_load(&box->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
						if (r != last)
							break;
						if (__hip_atomic_load(&box->stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == launch_no) {
							r = TG_RING_STOP;
							break;
						}
						__builtin_amdgcn_s_sleep(2);
					}
				r = __builtin_amdgcn_readfirstlane(r);
				n = box->n;
				hs = box->have_sync;
			}
			if (tid == 0) {
				s_cmd[0] = r;
				s_cmd[1] = n;
				s_cmd[2] = hs;
			}
		}
		__syncthreads();
		const uint32_t r = s_cmd[0], n = s_cmd[1], hs = s_cmd[2];
		if (r == TG_RING_STOP)
			break;
		/* (what the host and workgroup 0 wrote is read past this compute unit's vector cache) */
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		if (hs) {
			if (w < n)
				burst_body(true, w, stream, box->desc, box->chan, &box->code, n, 1u, sb_ok, sb_code, rec, maskidx, masks, 0);
			if (n > 1) {
				__threadfence();
				ring_barrier(&box->bar, epoch, G);
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			} else
				__syncthreads();
		}
		if (w < n)
			burst_body(false, w, stream, box->desc, box->chan, &box->code, n, 1u, sb_ok, sb_code, rec, maskidx, masks, 1);
		__syncthreads();
		last = r;
		if (w == 0 && tid == 0)
			__hip_atomic_store(&ring->served, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (w == 0 && tid == 0)
		__hip_atomic_store(&ring->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int tgk_burst_ring(struct tg_ring_msg *d_ring, struct tg_ring_box *d_box, const uint8_t *d_stream, uint32_t nwg,
			      uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_rec, uint32_t *d_maskidx, uint32_t *d_masks,
			      uint32_t start_seq, uint32_t launch_no, unsigned long long idle_ticks, void *stream)
{
	if (!nwg || nwg > TG_RING_MAX || !launch_no)
		return -1;
	hipLaunchKernelGGL(k_burst_ring, dim3(nwg), dim3(256), 0, (hipStream_t)stream, d_ring, d_box, d_stream, d_sb_ok, d_sb_code, d_rec,
			   d_maskidx, d_masks, start_seq, launch_no, idle_ticks);
	return (int)hipGetLastError();
}

/* one bit per grid slot: the classification word alone says "delivered" (a training sequence of the right type at its
 * nominal offset, no EARLY21 / NONBINARY flag) -- what the host walk's steady state tests, 32 slots to a word */
__global__ __launch_bounds__(256)
void k_cls_plain(const uint32_t *__restrict__ cls, uint32_t n, uint32_t *__restrict__ plain)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	const uint32_t v = i < n ? cls[i] & 0x03ffffffu : 0xffu;
	const bool ok = v == (TG_BURST_SYNC | TG_SYNC_TRAIN_OFF << 8) || v == (TG_BURST_NORM_1 | TG_NORM_TRAIN_OFF << 8) ||
			v == (TG_BURST_NORM_2 | TG_NORM_TRAIN_OFF << 8);
	const unsigned long long b = __ballot(ok);
	const uint32_t lane = threadIdx.x & 63, w = i >> 5;
	if (lane == 0 && 32 * w < n)
		plain[w] = (uint32_t)b;
	if (lane == 32 && 32 * w < n)
		plain[w] = (uint32_t)(b >> 32);
}

extern "C" int tgk_cls_plain(const uint32_t *d_cls, uint32_t n, uint32_t *d_plain, void *stream)
{
	if (!n)
		return 0;
	hipLaunchKernelGGL(k_cls_plain, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_cls, n, d_plain);
	return (int)hipGetLastError();
}

extern "C" int tgk_burst(const uint8_t *d_stream, const uint64_t *d_slot_desc, const uint32_t *d_slot_chan,
			 const uint32_t *d_chan_code, uint32_t nslots, uint32_t nchan, int have_sync, uint32_t *d_sb_ok,
			 uint32_t *d_sb_code, uint8_t *d_rec, uint32_t *d_maskidx, uint32_t *d_masks, int marks, void *stream)
{
	if (!nslots)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	if (have_sync)
		hipLaunchKernelGGL((k_burst<true>), dim3(nslots), dim3(256), 0, s, d_stream, d_slot_desc, d_slot_chan, d_chan_code, nslots,
				   nchan, d_sb_ok, d_sb_code, d_rec, d_maskidx, d_masks, 0);
	hipLaunchKernelGGL((k_burst<false>), dim3(nslots), dim3(256), 0, s, d_stream, d_slot_desc, d_slot_chan, d_chan_code, nslots,
			   nchan, d_sb_ok, d_sb_code, d_rec, d_maskidx, d_masks, marks);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* generic trellis: any RCPC puncturer on either mother code  */
/* ------------------------------------------------------------------------- */
/*
 * k_conv<CODE, NCH, G3>: tetra_rcpc_depunct() + conv_cch_decode() / conv_tch_decode() for a batch of equally
 * shaped blocks (lower_mac/tetra_conv_enc.c:226-248, viterbi_cch.c:58-66, viterbi_tch.c:56-64).  One lane per
 * block, 64 blocks per wave.  The wave's 64 * type3_len received bytes (1 bit per byte, 0xff = erased, the
 * reference's depunct buffer convention) are one contiguous range: read with coalesced dwords, reduced to 2-bit
 * classes (0 bit / 1 bit / erased), four to an LDS byte.  The step program (tg_conv.h: which type-3 byte carries
 * g1 / g2 / g3 of each step) is uniform: it is fetched with scalar loads and all the index / shift / presence
 * arithmetic runs on the scalar unit.  Per step the vector unit does 3 x (address add, LDS byte read, 24-bit
 * multiply, mask), 10 packed adds for the four branch-metric pairs and their tie variants and 24 for the
 * add-compare-select (tg_step_gen).  History: 16 bytes per 8 steps in
 * VGPRs (NCH chunks of 32 registers, as k_vit), block-wise traceback, decoded bits transposed through the same
 * LDS range and written out as one contiguous range.  Metrics are renormalised every 64 steps.
 */
template <int CODE, int NCH, bool G3>
__global__ __launch_bounds__(64)
void k_conv(const uint8_t *__restrict__ type3, unsigned long long nblocks, uint32_t t3len, uint32_t L,
	    const uint32_t *__restrict__ steps, uint8_t *__restrict__ type2, uint32_t rawoff)
{
	extern __shared__ uint32_t s_dyn[];
	uint8_t *s_in = (uint8_t *)s_dyn;
	const uint32_t lane = threadIdx.x;
	const unsigned long long blk0 = (unsigned long long)blockIdx.x * 64ull;
	const uint32_t nvalid = (nblocks - blk0 < 64ull) ? (uint32_t)(nblocks - blk0) : 64u;
	const uint32_t nq = (t3len + 3) >> 2;		/* class bytes per block (LDS row pitch) */

	{	/* stage: four received bytes -> one class byte (tg_conv_pack4) */
		const uint8_t *src = type3 + blk0 * t3len;
		if ((t3len & 3) == 0 && ((uintptr_t)type3 & 3) == 0) {
			/* rows are whole dwords and blk0 * t3len is a multiple of 64: one flat, coalesced dword range */
			const uint32_t ndw = nvalid * nq;
			for (uint32_t w = lane; w < ndw; w += 64)
				s_in[w] = (uint8_t)tg_conv_pack4(((const uint32_t *)src)[w]);
		} else {
			/* odd row length or base: the raw bytes go to a second LDS range first (flat, dwords when the base
			 * allows), then (row, quad) pairs flat over the wave are reduced from there; the row index by a
			 * corrected float division (all values < 2^15) */
			uint8_t *raw = s_in + rawoff;
			const uint32_t nbytes = nvalid * t3len;
			uint32_t done = 0;
			if (((uintptr_t)type3 & 3) == 0) {
				const uint32_t ndw = nbytes >> 2;
				for (uint32_t w = lane; w < ndw; w += 64)
					((uint32_t *)raw)[w] = ((const uint32_t *)src)[w];
				done = ndw << 2;
			}
			for (uint32_t i = done + lane; i < nbytes; i += 64)
				raw[i] = src[i];
			__syncthreads();
			const uint32_t nitem = nvalid * nq;
			const float inv = 1.0f / (float)nq;
			for (uint32_t w = lane; w < nitem; w += 64) {
				uint32_t r = (uint32_t)(((float)w + 0.5f) * inv);
				r -= (r * nq > w);
				r += ((r + 1) * nq <= w);
				const uint32_t q = w - r * nq;
				const uint8_t *row = raw + r * t3len;
				/* one (unaligned) LDS dword read; bytes past the row (the raw range has 4 spare bytes) -> erased */
				uint32_t x;
				__builtin_memcpy(&x, row + 4 * q, 4);
				const uint32_t nv = t3len - 4 * q;
				x |= (nv < 4) ? (0xffffffffu << (8 * nv)) : 0u;
				s_in[w] = (uint8_t)tg_conv_pack4(x);
			}
		}
	}
	__syncthreads();

	const uint8_t *mine = s_in + (lane < nvalid ? lane : 0) * nq;
	auto fetch = [&](uint32_t q) -> uint32_t { return mine[q]; };

	const uint32_t nblk = (L + 7) >> 3;
	tg_vit_state v;
	uint32_t h[4];
	tg_vit_init(v);
	tg_conv_block<CODE, G3, 4>(v, steps, 4, fetch, h);
	tg_v32 H[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		if (8u * c < nblk) {
			if (c)
				tg_vit_normalize(v);
			const uint32_t nb = (nblk - 8u * c < 8u) ? nblk - 8u * c : 8u;
#pragma unroll 1
			for (uint32_t it = 0; it < nb; it++) {
				const uint32_t b = 8u * c + it;
				const uint32_t left = L - 8u * b;
				if (left >= 8)
					tg_conv_block<CODE, G3, 8>(v, steps + 3 * (4 + 8 * b), 8, fetch, h);
				else
					tg_conv_block<CODE, G3, 0>(v, steps + 3 * (4 + 8 * b), (int)left, fetch, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][4 * it + d] = h[d];
			}
		}
	}

	__syncthreads();	/* the received bytes are dead: the same LDS range now takes the decoded bits */
	uint8_t *outl = s_in + lane * L;
	const bool al = (L & 3) == 0;
	uint32_t s = 0;
#pragma unroll
	for (int b = 8 * NCH - 1; b >= 0; b--) {
		if ((uint32_t)b < nblk) {
			const int c = b >> 3, o = 4 * (b & 7);
			const uint32_t byte = hist_byte(H[c][o], H[c][o + 1], H[c][o + 2], H[c][o + 3], s);
			const uint32_t nbits = (L - 8u * b < 8u) ? L - 8u * b : 8u;
			if (al) {	/* L = 0 mod 4: a block holds 8 or 4 bits */
				*(uint32_t *)(outl + 8 * b) = spread4(byte);
				if (nbits > 4)
					*(uint32_t *)(outl + 8 * b + 4) = spread4(byte >> 4);
			} else {
				for (uint32_t i = 0; i < nbits; i++)
					outl[8 * b + i] = (uint8_t)((byte >> i) & 1);
			}
			s = tg_brev4(byte);
		}
	}
	__syncthreads();
	{
		uint8_t *dst = type2 + blk0 * L;
		const uint32_t nbytes = nvalid * L;
		uint32_t done = 0;
		if (((uintptr_t)type2 & 3) == 0) {
			const uint32_t ndw = nbytes >> 2;
			for (uint32_t w = lane; w < ndw; w += 64)
				((uint32_t *)dst)[w] = s_dyn[w];
			done = ndw << 2;
		}
		for (uint32_t q = done + lane; q < nbytes; q += 64)
			dst[q] = s_in[q];
	}
}

/* ------------------------------------------------------------------------- */
/* k_reorder: a fixed index map applied to every block of a batch (ACELP re-ordering, tg_reorder.c)  */
/* ------------------------------------------------------------------------- */
/* out[b][j] = in[b][src[j]] where src[j] >= 0; destinations without a source keep what d_out held (the reference's
 * behaviour for a table that names a position never, lower_mac/tch_reordering.c:94-117).  One lane per output byte:
 * stores are consecutive, loads stay inside the block's row. */
__global__ __launch_bounds__(256)
void k_reorder(const uint8_t *__restrict__ in, unsigned long long nblocks, uint32_t nbits, const int32_t *__restrict__ src,
	       uint8_t *__restrict__ out)
{
	const unsigned long long total = nblocks * nbits;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
		const unsigned long long b = i / nbits;
		const uint32_t j = (uint32_t)(i - b * nbits);
		const int32_t s = src[j];
		if (s >= 0)
			out[i] = in[b * nbits + (uint32_t)s];
	}
}

extern "C" int tgk_reorder(const uint8_t *d_in, unsigned long long nblocks, uint32_t nbits, const int32_t *d_src, uint8_t *d_out,
			   void *stream)
{
	if (!nblocks)
		return 0;
	unsigned long long blocks = (nblocks * nbits + 255) / 256;
	if (blocks > 256 * 32)
		blocks = 256 * 32;
	hipLaunchKernelGGL(k_reorder, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, d_in, nblocks, nbits, d_src, d_out);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* scrambling-code forward fill: inclusive running max over (chan<<32 | entry) */
/* ------------------------------------------------------------------------- */
#define FILL_BLOCK 1024

/*
 * A decoded SYNC slot brings its own mask-table entry only when its code is news: the k-th SYNC slot of the batch is
 * "redundant" when the one before it (same channel) decoded to the same code -- the running maximum then keeps the
 * earlier entry, k_masks never computes this one, and all the slots of a cell share one 160-byte entry (a 1 M-slot
 * recording used to build and read 125 k identical ones).
 */
__device__ __forceinline__ bool sb_redundant(uint32_t k, const uint32_t *sb_ok, const uint32_t *sb_code,
					     const uint32_t *list_sb, const uint32_t *slot_chan)
{
	return k > 0 && sb_ok[k - 1] && sb_code[k] == sb_code[k - 1] && slot_chan[list_sb[k]] == slot_chan[list_sb[k - 1]];
}

__device__ __forceinline__ unsigned long long fill_key(uint32_t i, const uint32_t *slot_chan, const int32_t *slot_sbord,
						       const uint32_t *sb_ok, const uint32_t *sb_code, const uint32_t *list_sb,
						       uint32_t nchan)
{
	const uint32_t ch = slot_chan[i];
	const int32_t k = slot_sbord[i];
	/* entry ids: 0 = zero mask, 1+ch = channel carry-in, 1+nchan+k = k-th SYNC slot of the batch */
	uint32_t e = 1 + ch;
	if (k >= 0 && sb_ok[k] && !sb_redundant((uint32_t)k, sb_ok, sb_code, list_sb, slot_chan))
		e = 1 + nchan + (uint32_t)k;
	return ((unsigned long long)ch << 32) | e;
}

__device__ __forceinline__ unsigned long long wave_incl_max(unsigned long long v, uint32_t lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long o = __shfl_up(v, d);
		if (lane >= (uint32_t)d && o > v)
			v = o;
	}
	return v;
}

/* phase 1: per-block maximum */
__global__ __launch_bounds__(FILL_BLOCK)
void k_fill_reduce(const uint32_t *slot_chan, const int32_t *slot_sbord, const uint32_t *sb_ok, const uint32_t *sb_code,
		   const uint32_t *list_sb, uint32_t nchan, uint32_t nslots, unsigned long long *block_max)
{
	__shared__ unsigned long long sm[FILL_BLOCK / 64];
	const uint32_t i = blockIdx.x * FILL_BLOCK + threadIdx.x;
	unsigned long long v = (i < nslots) ? fill_key(i, slot_chan, slot_sbord, sb_ok, sb_code, list_sb, nchan) : 0ull;
	const uint32_t lane = threadIdx.x & 63;
	v = wave_incl_max(v, lane);
	if (lane == 63)
		sm[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long m = 0;
		for (int w = 0; w < FILL_BLOCK / 64; w++)
			m = sm[w] > m ? sm[w] : m;
		block_max[blockIdx.x] = m;
	}
}

/* phase 2: exclusive running max over the block maxima (one workgroup, 1024 maxima per pass) */
__global__ __launch_bounds__(FILL_BLOCK)
void k_fill_scan(unsigned long long *block_max, uint32_t nblocks)
{
	__shared__ unsigned long long sm[FILL_BLOCK / 64];
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long carry = 0;
	for (uint32_t base = 0; base < nblocks; base += FILL_BLOCK) {
		const uint32_t i = base + threadIdx.x;
		const unsigned long long v = (i < nblocks) ? block_max[i] : 0ull;
		const unsigned long long inc = wave_incl_max(v, lane);
		if (lane == 63)
			sm[w] = inc;
		__syncthreads();
		unsigned long long pre = carry, tot = carry;
		for (uint32_t q = 0; q < FILL_BLOCK / 64; q++) {
			if (q < w)
				pre = sm[q] > pre ? sm[q] : pre;
			tot = sm[q] > tot ? sm[q] : tot;
		}
		unsigned long long exc = __shfl_up(inc, 1);
		if (lane == 0)
			exc = 0;
		if (pre > exc)
			exc = pre;
		if (i < nblocks)
			block_max[i] = exc;
		carry = tot;
		__syncthreads();
	}
}

/* phase 3: in-block scan with carry-in, write the mask entry of every slot */
__global__ __launch_bounds__(FILL_BLOCK)
void k_fill_apply(const uint32_t *slot_chan, const int32_t *slot_sbord, const uint32_t *sb_ok, const uint32_t *sb_code,
		  const uint32_t *list_sb, uint32_t nchan, uint32_t nslots, const unsigned long long *block_excl, uint32_t *maskidx)
{
	__shared__ unsigned long long sm[FILL_BLOCK / 64];
	const uint32_t i = blockIdx.x * FILL_BLOCK + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long v = (i < nslots) ? fill_key(i, slot_chan, slot_sbord, sb_ok, sb_code, list_sb, nchan) : 0ull;
	v = wave_incl_max(v, lane);
	if (lane == 63)
		sm[w] = v;
	__syncthreads();
	unsigned long long pre = block_excl[blockIdx.x];
	for (uint32_t q = 0; q < w; q++)
		pre = sm[q] > pre ? sm[q] : pre;
	if (pre > v)
		v = pre;
	if (i < nslots)
		maskidx[i] = (uint32_t)v;	/* chan is monotone along the array, so the max stays inside the channel */
}

/* ------------------------------------------------------------------------- */
/* k_masks: one wavefront per mask-table entry                               */
/* ------------------------------------------------------------------------- */
__global__ __launch_bounds__(256)
void k_masks(const uint32_t *chan_code, uint32_t nchan, const uint32_t *sb_ok, const uint32_t *sb_code,
	     uint32_t nsb, const uint32_t *nsb_dev, const uint32_t *list_sb, const uint32_t *slot_chan, uint32_t *masks)
{
	if (nsb_dev)		/* the number of SYNC slots was counted on the device (nsb = its upper bound) */
		nsb = *nsb_dev;
	/* a wavefront keeps the linear-form masks of its 18 x 64 output bits in registers, looks at 64 entries at a
	 * time and builds the ones a slot can point at: entry 0, the channel carry-ins, and the SYNC slots that decoded
	 * (CRC) to a code other than their predecessor's (sb_redundant) -- per built entry 18 x (and, popcount, ballot)
	 * and one 160-byte store */
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t half = lane >> 5, bit = lane & 31;
	const uint32_t nent = 1 + nchan + nsb;
	uint32_t lin[TG_MW_ROUNDS];
#pragma unroll
	for (int r = 0; r < TG_MW_ROUNDS; r++) {
		const uint16_t pos = c_tab.mask_pos[2 * r + half][bit];
		lin[r] = (pos != 0xffff) ? c_tab.lfsr_lin[pos] : 0u;
	}
	for (uint32_t e0 = wave * 64; e0 < nent; e0 += nwaves * 64) {
		const uint32_t e = e0 + lane;
		uint32_t mycode = 0;
		bool need = e < nent;
		if (e >= 1 && e <= nchan)
			mycode = chan_code[e - 1];
		else if (e > nchan && e < nent) {
			const uint32_t k = e - 1 - nchan;
			need = sb_ok[k] && !sb_redundant(k, sb_ok, sb_code, list_sb, slot_chan);
			mycode = sb_code[k];
		}
		unsigned long long todo = __ballot(need);
		while (todo) {
			const uint32_t l = (uint32_t)__builtin_ctzll(todo);
			todo &= todo - 1;
			const uint32_t code = __builtin_amdgcn_readlane(mycode, l);
			uint32_t myword = 0;
#pragma unroll
			for (int r = 0; r < TG_MW_ROUNDS; r++) {
				const unsigned long long bal = __ballot(__popc(code & lin[r]) & 1);
				myword = (lane == (uint32_t)(2 * r)) ? (uint32_t)bal : myword;
				myword = (lane == (uint32_t)(2 * r + 1)) ? (uint32_t)(bal >> 32) : myword;
			}
			if (lane == TG_MW_CODE)
				myword = code;
			if (lane < TG_MASK_WORDS)
				masks[(size_t)(e0 + l) * TG_MASK_WORDS + lane] = myword;
		}
	}
}

/* ------------------------------------------------------------------------- */
/* host-side launch layer                                                    */
/* ------------------------------------------------------------------------- */
static uint32_t lfsr_next(uint32_t *st)
{
	/* Fibonacci LFSR of lower_mac/tetra_scramb.c:34-50, taps 32 26 23 22 16 12 11 10 8 7 5 4 2 1 */
	static const int taps[14] = { 32, 26, 23, 22, 16, 12, 11, 10, 8, 7, 5, 4, 2, 1 };
	uint32_t s = *st, fb = 0;
	for (int i = 0; i < 14; i++)
		fb ^= s >> (32 - taps[i]);
	fb &= 1;
	*st = (s >> 1) | (fb << 31);
	return fb;
}

static void build_clean_lut(uint16_t *t)
{
	/* received bits of an 8-step block: a(g1,g2) b(g1) a b a b a b -> bits 0,1 | 2 | 3,4 | 5 | 6,7 | 8 | 9,10 | 11 */
	static const int g1pos[8] = { 0, 2, 3, 5, 6, 8, 9, 11 }, g2pos[4] = { 1, 4, 7, 10 };
	for (uint32_t x = 0; x < 4096; x++) {
		uint32_t g1 = 0, g2 = 0;
		for (int i = 0; i < 8; i++)
			g1 |= ((x >> g1pos[i]) & 1) << i;
		for (int i = 0; i < 4; i++)
			g2 |= ((x >> g2pos[i]) & 1) << i;
		t[x] = (uint16_t)(g1 | (g2 << 8));
	}
	for (uint32_t st = 0; st < 16; st++)
		for (uint32_t g1 = 0; g1 < 256; g1++) {
			uint32_t h = st;	/* last four input bits, newest in bit 0 */
			uint32_t u8 = 0, g2 = 0;
			for (int k = 0; k < 8; k++) {
				const uint32_t u = ((g1 >> k) ^ h ^ (h >> 3)) & 1;			/* G1 = 1 + D + D^4 */
				if (!(k & 1))
					g2 |= ((u ^ (h >> 1) ^ (h >> 2) ^ (h >> 3)) & 1) << (k >> 1);	/* G2 = 1 + D^2 + D^3 + D^4 */
				u8 |= u << k;
				h = ((h << 1) | u) & 15;
			}
			t[4096 + (st << 8 | g1)] = (uint16_t)(u8 | (g2 << 8) | (h << 12));
		}
}

static void build_tables(tg_const_tables *t)
{
	memset(t, 0, sizeof(*t));
	const int btypes[3] = { TG_BURST_NORM_1, TG_BURST_NORM_2, TG_BURST_SYNC };
	for (int x = 0; x < 3; x++)
		for (int w = 0; w < TG_PACKED_WORDS; w++)
			for (int p = 0; p < 32; p++) {
				int o = tg_packed_src(btypes[x], w, p);
				t->front_src[x][w][p] = (o < 0) ? 0xffff : (uint16_t)o;
			}
	/* mask layout: which LFSR output index scrambles each packed bit */
	for (int w = 0; w < TG_MASK_WORDS; w++)
		for (int p = 0; p < 32; p++) {
			int pos = -1;
			if (w < TG_MW_216)
				pos = tg_codeword_src(TG_KIND_432, w - TG_MW_432, p);
			else if (w < TG_MW_BBK)
				pos = tg_codeword_src(TG_KIND_216, w - TG_MW_216, p);
			else if (w == TG_MW_BBK)
				pos = (p < 30) ? p : -1;
			else if (w >= TG_MW_168 && w < TG_MW_168 + 7)
				pos = tg_codeword_src(TG_KIND_168, w - TG_MW_168, p);
			t->mask_pos[w][p] = (pos < 0) ? 0xffff : (uint16_t)pos;
		}
	/* block mode: code-word bit -> type-5 bit of a block handed over on its own */
	for (int x = 0; x < TG_NBLKTYPES; x++)
		for (int w = 0; w < TG_PACKED_WORDS; w++)
			for (int p = 0; p < 32; p++) {
				int o = -1;
				if (x == TG_BLK_BBK)
					o = (w == TG_PW_BBK && p < 30) ? p : -1;
				else if (w < tg_kind_nblk(x) / 2)
					o = tg_codeword_src(x, w, p);
				t->blk_src[x][w][p] = (o < 0) ? 0xffff : (uint16_t)o;
			}
	/* linear form of the LFSR: run it on the 32 unit vectors */
	for (int b = 0; b < 32; b++) {
		uint32_t st = 1u << b;
		for (int n = 0; n < 432; n++)
			if (lfsr_next(&st))
				t->lfsr_lin[n] |= 1u << b;
	}
	/* SB1 mask for init = 3 (lower_mac/tetra_scramb.h:14) */
	{
		uint8_t seq[120];
		uint32_t st = 3;
		for (int n = 0; n < 120; n++)
			seq[n] = (uint8_t)lfsr_next(&st);
		for (int d = 0; d < 5; d++)
			for (int p = 0; p < 32; p++) {
				int j = tg_codeword_src(TG_KIND_SB1, d, p);
				if (j >= 0 && seq[j])
					t->sb1_mask[d] |= 1u << p;
			}
	}
	{
		static const int nblk_of[3] = { 10, 18, 36 };	/* TG_KIND_SB1, _216, _432 */
		for (int kind = 0; kind < 3; kind++) {
			const int nbits = 8 * (nblk_of[kind] - 1) + 4;
			uint16_t c = 0xffff;
			for (int i = 0; i < nbits; i++)
				c = tg_crc16_step_bits(c, 0, 1);
			t->crc_aff[kind] = c;
			for (int i = 0; i < 288; i++) {
				uint16_t v = 0;
				if (i < nbits) {
					v = tg_crc16_step_bits(0, 1, 1);
					for (int k = i + 1; k < nbits; k++)
						v = tg_crc16_step_bits(v, 0, 1);
				}
				t->crc_lin[kind][i] = v;
			}
		}
		t->crc_aff[3] = 0;
	}
	tg_crc16_make_table(t->crc_lsb);
	for (int x = 0; x < 256; x++) {
		int rv = 0;
		for (int i = 0; i < 8; i++)
			if (x & (1 << i))
				rv |= 0x80 >> i;
		t->crc_msb[x] = t->crc_lsb[rv];
	}
}

/* survivor-history placement of the trellis kernels: 1 = VGPRs (default), 0 = LDS.
 * Tuning knob for A/B runs (environment TGPU_HIST_MODE), both are bit-identical. */

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

static void build_soft_tables(tg_soft_tables *t)
{
	const int btypes[3] = { TG_BURST_NORM_1, TG_BURST_NORM_2, TG_BURST_SYNC };
	memset(t, 0xff, sizeof(*t));
	for (int x = 0; x < 3; x++) {
		const int bt = btypes[x];
		struct { int kind, base, wbase; } blk[2];
		int nb = 0;
		if (bt == TG_BURST_NORM_1) {
			blk[nb++] = { TG_KIND_432, 0, TG_PW_BLK1 };
		} else if (bt == TG_BURST_NORM_2) {
			blk[nb++] = { TG_KIND_216, 0, TG_PW_BLK1 };
			blk[nb++] = { TG_KIND_216, TG_SOFT_AREA2, TG_PW_BLK2 };
		} else {
			blk[nb++] = { TG_KIND_SB1, 0, TG_PW_BLK1 };
			blk[nb++] = { TG_KIND_216, TG_SOFT_AREA2, TG_PW_BLK2 };
		}
		for (int b = 0; b < nb; b++) {
			const int K = tg_kind_K(blk[b].kind), a = tg_kind_a(blk[b].kind);
			for (int i = 0; i < K; i++) {
				const int j = (a * (i + 1)) % K;	/* type3[i] = type4[j] */
				const int q = blk[b].base + (i < 6 ? i : TG_SOFT_LEADIN_BYTES + (i - 6));
				t->src[x][q] = (uint16_t)tg_block_stream_off(bt, blk[b].wbase, j);
			}
		}
		for (int p = 0; p < 30; p++)
			t->src[x][TG_SOFT_BBK + p] = (uint16_t)tg_bbk_stream_off(bt, p);
	}
}

extern "C" int tgk_init(void)
{
	static tg_const_tables host;
	static tg_soft_tables shost;
	build_soft_tables(&shost);
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_soft_tab), &shost, sizeof(shost)));
	build_tables(&host);
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &host, sizeof(host)));
	{
		static uint16_t lut[8192];
		build_clean_lut(lut);
		HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_clean_lut), lut, sizeof(lut)));
	}
	return 0;
}

extern "C" int tgk_front(const uint8_t *d_stream, const uint64_t *d_slot_desc,
			 uint32_t nslots, uint32_t *d_packed, uint8_t *d_rec, void *stream)
{
	if (!nslots)
		return 0;
	uint32_t blocks = (nslots + 3) / 4;
	uint32_t cap = 256 * 32;	/* measured on MI355X (tools/exp_front_grid.py): 2048 156 us, 4096 154 us, 8192 144 us */
	if (tgi_option(TGPU_OPT_FRONT_BLOCKS) > 0)
		cap = (uint32_t)tgi_option(TGPU_OPT_FRONT_BLOCKS);
	if (blocks > cap)
		blocks = cap;
	hipLaunchKernelGGL(k_front, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
			   d_stream, d_slot_desc, nslots, d_packed, d_rec);
	return (int)hipGetLastError();
}

extern "C" int tgk_clean(int kind, const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed, const uint32_t *d_masks,
			 const uint32_t *d_maskidx, uint8_t *d_rec, uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_wire,
			 uint32_t *d_dirty_items, uint32_t *d_dirty_count, int flags, void *stream)
{
	if (!nitems)
		return 0;
	uint32_t blocks = (nitems + 255) / 256;
	if (blocks > 256 * 8)
		blocks = 256 * 8;
	hipStream_t s = (hipStream_t)stream;
	if (kind == TG_KIND_216)
		hipLaunchKernelGGL((k_clean<TG_KIND_216>), dim3(blocks), dim3(256), 0, s, d_items, nitems, d_packed, d_masks, d_maskidx, d_rec,
				   d_sb_ok, d_sb_code, d_wire, d_dirty_items, d_dirty_count, flags);
	else if (kind == TG_KIND_432)
		hipLaunchKernelGGL((k_clean<TG_KIND_432>), dim3(blocks), dim3(256), 0, s, d_items, nitems, d_packed, d_masks, d_maskidx, d_rec,
				   d_sb_ok, d_sb_code, d_wire, d_dirty_items, d_dirty_count, flags);
	else
		return -1;
	return (int)hipGetLastError();
}

extern "C" int tgk_rm_enable(const uint32_t *h_leader, const uint16_t *h_parity)
{
	static uint32_t *d_leader;	/* one table per process, never freed */
	if (!d_leader) {
		HIPCHK(hipMalloc((void **)&d_leader, 65536 * 4));
		HIPCHK(hipMemcpy(d_leader, h_leader, 65536 * 4, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_rm_leader), &d_leader, sizeof(d_leader)));
		HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_rm_parity), h_parity, 14 * 2));
	}
	return 0;
}

extern "C" int tgk_front_blocks(const uint8_t *d_bits, const uint64_t *d_desc, uint32_t nblocks, uint32_t *d_packed, void *stream)
{
	if (!nblocks)
		return 0;
	uint32_t blocks = (nblocks + 3) / 4;
	if (blocks > 256 * 16)
		blocks = 256 * 16;
	hipLaunchKernelGGL(k_front_blocks, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_bits, d_desc, nblocks, d_packed);
	return (int)hipGetLastError();
}

extern "C" int tgk_bbk_blocks(const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed, const uint32_t *d_masks,
			      const uint32_t *d_maskidx, uint8_t *d_rec, int flags, void *stream)
{
	if (!nitems)
		return 0;
	hipLaunchKernelGGL(k_bbk_blocks, dim3((nitems + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_items, nitems,
			   d_packed, d_masks, d_maskidx, d_rec, flags);
	return (int)hipGetLastError();
}

static const uint8_t tsq_n[22] = { 1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0 };
static const uint8_t tsq_p[22] = { 0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0 };
static const uint8_t tsq_y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };
static const uint8_t tsq_q[22] = { 1,0,1,1,0,1,1,1,0,0,0,0,0,1,1,0,1,0,1,1,0,1 };
static const uint8_t tsq_x[30] = { 1,0,0,1,1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0,0,0,1,1 };

static uint32_t host_pattern_bits(const uint8_t *seq, int from, int n)
{
	uint32_t v = 0;
	for (int i = 0; i < n; i++)
		v |= (uint32_t)seq[from + i] << i;
	return v;
}

static void stream_patterns(tg_stream_params &prm, uint32_t chunk);

/* both passes of the packed-bit front end; d_defer: scratch of TG_DEFER_WORDS(nslots) dwords (count + list) */
/* per-kernel timing: an event to be recorded right in front of the next k_front_stream launch of this thread (behind the
 * clearing of the deferred-slot counter, which is a launch of its own) */
static __thread void *tl_front_ev_start;
extern "C" void tgk_front_stream_ev_start(void *ev)
{
	tl_front_ev_start = ev;
}

static int launch_stream_front(const uint8_t *d_stream, const tg_stream_params &prm, uint32_t *d_packed, uint32_t *d_cls,
			       uint16_t *d_ysum, uint32_t *d_defer, hipStream_t s, void *ev_mid, bool packed_input = false)
{
	const uint32_t nslots = prm.nslots;
	uint32_t blocks = ((nslots + 3) / 4 + 3) / 4;	/* a wave per group of four slots */
	uint32_t cap = 256 * 2 * TG_STREAM_WPE;	/* two rounds of what the 256 CUs hold: one round (persistent waves) is 3-4 % slower, four rounds 5 % */
	if (tgi_option(TGPU_OPT_FRONT_BLOCKS) > 0)
		cap = (uint32_t)tgi_option(TGPU_OPT_FRONT_BLOCKS);
	if (blocks > cap)
		blocks = cap;
	HIPCHK(hipMemsetAsync(d_defer, 0, 4, s));
	if (tl_front_ev_start) {
		HIPCHK(hipEventRecord((hipEvent_t)tl_front_ev_start, s));
		tl_front_ev_start = nullptr;
	}
	const dim3 fgrid(blocks * (4 / TG_STREAM_WPB)), fblock(64 * TG_STREAM_WPB);
	if (packed_input)
		hipLaunchKernelGGL(k_front_stream<true>, fgrid, fblock, 0, s, d_stream, prm, d_packed, d_cls, d_ysum, d_defer);
	else
		hipLaunchKernelGGL(k_front_stream<false>, fgrid, fblock, 0, s, d_stream, prm, d_packed, d_cls, d_ysum, d_defer);
	if (ev_mid)
		HIPCHK(hipEventRecord((hipEvent_t)ev_mid, s));
	uint32_t fblocks = (nslots / 128 + 3) / 4 + 1;	/* about a wave per deferred slot at 1 % of them */
	if (fblocks > 256 * 16)
		fblocks = 256 * 16;
#define FIX_LAUNCH(P, V) hipLaunchKernelGGL((k_front_stream_fix<P, V>), dim3(fblocks), dim3(256), 0, s, d_stream, prm, d_packed, d_cls, d_ysum, d_defer)
	const uint32_t view = TG_VIEW_OF(prm.chunk);	/* (the kernel built for this view) */
	if (packed_input) {
		if (view == 640) FIX_LAUNCH(true, 640); else if (view == 832) FIX_LAUNCH(true, 832); else FIX_LAUNCH(true, 1088);
	} else {
		if (view == 640) FIX_LAUNCH(false, 640); else if (view == 832) FIX_LAUNCH(false, 832); else FIX_LAUNCH(false, 1088);
	}
#undef FIX_LAUNCH
	return (int)hipGetLastError();
}

/* several channels in one grid: d_chan = device copy of nchan (<= 64) tg_chan_ent, nslots = the grid's total size */
extern "C" int tgk_front_stream_multi(const uint8_t *d_base, const struct tg_chan_ent *d_chan, uint32_t nchan, uint32_t nslots,
				      uint32_t chunk, uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum, uint32_t *d_defer,
				      void *stream, void *ev_mid, int packed_input)
{
	if (!nslots)
		return 0;
	if (!chunk || !nchan || nchan > 64 || (nslots & 31))
		return -1;
	tg_stream_params prm;
	memset(&prm, 0, sizeof(prm));
	prm.nslots = nslots;
	prm.chan = d_chan;
	prm.nchan = nchan;
	stream_patterns(prm, chunk);
	return launch_stream_front(d_base, prm, d_packed, d_cls, d_ysum, d_defer, (hipStream_t)stream, ev_mid, packed_input != 0);
}

static void stream_patterns(tg_stream_params &prm, uint32_t chunk)
{
	prm.chunk = chunk;
	prm.cshift = (chunk & (chunk - 1)) ? -1 : __builtin_ctz(chunk);
	prm.y32 = host_pattern_bits(tsq_y, 0, 32);
	prm.y6 = host_pattern_bits(tsq_y, 32, 6);
	prm.n22 = host_pattern_bits(tsq_n, 0, 22);
	prm.p22 = host_pattern_bits(tsq_p, 0, 22);
	prm.q22 = host_pattern_bits(tsq_q, 0, 22);
	prm.x22 = host_pattern_bits(tsq_x, 0, 22);
}

/* ev_mid (optional hipEvent_t): recorded between the packed-bit kernel and its fix-up pass (per-kernel timing) */
extern "C" int tgk_front_stream(const uint8_t *d_stream, uint64_t anchor, uint64_t len, uint32_t nslots,
				uint32_t chunk, uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum, uint32_t *d_defer,
				void *stream, void *ev_mid)
{
	if (!nslots)
		return 0;
	if (!chunk)
		return -1;
	tg_stream_params prm;
	memset(&prm, 0, sizeof(prm));
	prm.anchor = anchor;
	prm.len = len;
	prm.nslots = nslots;
	stream_patterns(prm, chunk);
	const int v1 = tgi_option(TGPU_OPT_STREAM_EXACT) != 0;	/* the per-position kernel on every slot (tests hold the two forms against each other) */
	hipStream_t s = (hipStream_t)stream;
	if (v1 || nslots < 16) {	/* (a handful of slots: the packed-bit kernel's group fetch wants 2176 readable bytes) */
		uint32_t blocks = (nslots + 3) / 4;
		if (blocks > 256 * 8)
			blocks = 256 * 8;
		hipLaunchKernelGGL(k_front_stream_v1, dim3(blocks), dim3(256), 0, s, d_stream, prm, d_packed, d_cls, d_ysum);
		if (ev_mid)
			HIPCHK(hipEventRecord((hipEvent_t)ev_mid, s));
		return (int)hipGetLastError();
	}
	return launch_stream_front(d_stream, prm, d_packed, d_cls, d_ysum, d_defer, s, ev_mid);
}

static int launch_front_soft(bool f32, const void *d_in, unsigned long long nin, const uint64_t *d_slot_desc, uint32_t nslots,
			     uint32_t *d_area, uint32_t *d_packed, uint8_t *d_rec, void *stream)
{
	if (!nslots)
		return 0;
	uint32_t blocks = (nslots + 3) / 4;
	uint32_t cap = 256 * 8;
	if (tgi_option(TGPU_OPT_FRONT_BLOCKS) > 0)
		cap = (uint32_t)tgi_option(TGPU_OPT_FRONT_BLOCKS);
	if (blocks > cap)
		blocks = cap;
	if (f32)
		hipLaunchKernelGGL(k_front_soft<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_in, nin, d_slot_desc,
				   nslots, d_area, d_packed, d_rec);
	else
		hipLaunchKernelGGL(k_front_soft<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_in, nin, d_slot_desc,
				   nslots, d_area, d_packed, d_rec);
	return (int)hipGetLastError();
}

extern "C" int tgk_front_soft(const int8_t *d_soft, const uint64_t *d_slot_desc, uint32_t nslots,
			      uint32_t *d_area, uint32_t *d_packed, uint8_t *d_rec, void *stream)
{
	return launch_front_soft(false, d_soft, 0, d_slot_desc, nslots, d_area, d_packed, d_rec, stream);
}

/* float phases in (nfloats symbols; slot offsets count stream positions, two per symbol) */
extern "C" int tgk_front_soft_f32(const float *d_phi, unsigned long long nfloats, const uint64_t *d_slot_desc, uint32_t nslots,
				  uint32_t *d_area, uint32_t *d_packed, uint8_t *d_rec, void *stream)
{
	return launch_front_soft(true, d_phi, nfloats, d_slot_desc, nslots, d_area, d_packed, d_rec, stream);
}

extern "C" int tgk_float_to_bits(const float *d_in, unsigned long long n, uint8_t *d_bits, int8_t *d_soft, void *stream)
{
	if (!n)
		return 0;
	unsigned long long blocks = (n / 4 + 255) / 256;
	if (blocks > 256 * 16)
		blocks = 256 * 16;
	if (!blocks)
		blocks = 1;
	hipLaunchKernelGGL(k_float_to_bits, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, d_in, n, d_bits, d_soft);
	return (int)hipGetLastError();
}

extern "C" int tgk_float_to_bits_afc(const float *d_in, unsigned long long n, uint8_t *d_bits, float filter_val,
				     float filter_goal, float *d_state, void *stream)
{
	hipLaunchKernelGGL(k_float_to_bits_afc, dim3(1), dim3(64), 0, (hipStream_t)stream, d_in, n, d_bits, filter_val,
			   filter_goal, d_state);
	return (int)hipGetLastError();
}

extern "C" int tgk_vit(int kind, const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed,
		       const uint32_t *d_masks, const uint32_t *d_maskidx, uint8_t *d_rec,
		       uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_wire, const uint32_t *d_soft, int flags,
		       const uint32_t *d_nitems /* NULL, or the device-side item count (nitems = its upper bound) */, void *stream)
{
	if (!nitems)
		return 0;
	const dim3 grid((nitems + 63) / 64), block(64);
	hipStream_t s = (hipStream_t)stream;
#define VIT_LAUNCH(K, H) hipLaunchKernelGGL((k_vit<K, H>), grid, block, 0, s, d_items, nitems, d_packed, d_masks, d_maskidx, d_rec, d_sb_ok, d_sb_code, d_wire, d_soft, flags, d_nitems)
	const int hm = d_soft ? 2 : 1;	/* survivor history in VGPRs (vit_core.h); 2 = the soft-input trellis */
	switch (kind) {
	case TG_KIND_SB1:
		if (hm == 2) VIT_LAUNCH(TG_KIND_SB1, 2); else VIT_LAUNCH(TG_KIND_SB1, 1);
		break;
	case TG_KIND_216:
		if (hm == 2) VIT_LAUNCH(TG_KIND_216, 2); else VIT_LAUNCH(TG_KIND_216, 1);
		break;
	case TG_KIND_432:
		if (hm == 2) VIT_LAUNCH(TG_KIND_432, 2); else VIT_LAUNCH(TG_KIND_432, 1);
		break;
	case TG_KIND_168:	/* hard input only */
		if (hm == 2) return -1; else VIT_LAUNCH(TG_KIND_168, 1);
		break;
	default:
		return -1;
	}
#undef VIT_LAUNCH
	return (int)hipGetLastError();
}

extern "C" int tgk_conv(int code, int g3, const uint8_t *d_type3, unsigned long long nblocks, uint32_t t3len, uint32_t L,
			const uint32_t *d_steps, uint8_t *d_type2, void *stream)
{
	if (!nblocks)
		return 0;
	const uint32_t nch = ((L + 7) / 8 + 7) / 8;
	const uint32_t span = (t3len + 3) / 4 > L ? (t3len + 3) / 4 : L;	/* class bytes in, decoded bits out */
	const uint32_t rawoff = (64 * span + 3) & ~3u;
	/* rows that are not whole dwords (or an odd base) are staged through a raw copy behind the working range */
	const bool odd = (t3len & 3) || ((uintptr_t)d_type3 & 3);
	const size_t lds = (size_t)rawoff + (odd ? ((((size_t)64 * t3len + 3) & ~(size_t)3) + 4) : 0);
	const unsigned long long nwg = (nblocks + 63) / 64;
	if (nch < 1 || nch > 8 || nwg > 0x7fffffffull || lds > 160 * 1024)
		return -1;
	hipStream_t s = (hipStream_t)stream;
	dim3 grid((unsigned)nwg), block(64);
#define CONV_LAUNCH(C, N, G) do {											\
		if (lds > 48 * 1024)										\
			HIPCHK(hipFuncSetAttribute((const void *)k_conv<C, N, G>,				\
						   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));	\
		hipLaunchKernelGGL((k_conv<C, N, G>), grid, block, lds, s, d_type3, nblocks, t3len, L, d_steps, d_type2, rawoff);	\
	} while (0)
#define CONV_CODE(C, G) switch (nch) {										\
	case 1: CONV_LAUNCH(C, 1, G); break; case 2: CONV_LAUNCH(C, 2, G); break; case 3: CONV_LAUNCH(C, 3, G); break;	\
	case 4: CONV_LAUNCH(C, 4, G); break; case 5: CONV_LAUNCH(C, 5, G); break; case 6: CONV_LAUNCH(C, 6, G); break;	\
	case 7: CONV_LAUNCH(C, 7, G); break; default: CONV_LAUNCH(C, 8, G); break; }
	if (code && g3)
		CONV_CODE(1, true)
	else if (code)
		CONV_CODE(1, false)
	else if (g3)
		CONV_CODE(0, true)
	else
		CONV_CODE(0, false)
#undef CONV_CODE
#undef CONV_LAUNCH
	return (int)hipGetLastError();
}

extern "C" int tgk_fill(const uint32_t *d_slot_chan, const int32_t *d_slot_sbord, const uint32_t *d_sb_ok, const uint32_t *d_sb_code,
			const uint32_t *d_list_sb, uint32_t nchan, uint32_t nslots, unsigned long long *d_block_tmp,
			uint32_t *d_maskidx, void *stream)
{
	if (!nslots)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const uint32_t nblocks = (nslots + FILL_BLOCK - 1) / FILL_BLOCK;
	hipLaunchKernelGGL(k_fill_reduce, dim3(nblocks), dim3(FILL_BLOCK), 0, s, d_slot_chan, d_slot_sbord, d_sb_ok, d_sb_code, d_list_sb,
			   nchan, nslots, d_block_tmp);
	hipLaunchKernelGGL(k_fill_scan, dim3(1), dim3(FILL_BLOCK), 0, s, d_block_tmp, nblocks);
	hipLaunchKernelGGL(k_fill_apply, dim3(nblocks), dim3(FILL_BLOCK), 0, s, d_slot_chan, d_slot_sbord, d_sb_ok, d_sb_code, d_list_sb,
			   nchan, nslots, d_block_tmp, d_maskidx);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* stream mode: the plan's per-slot arrays and item lists, built on the device  */
/* ------------------------------------------------------------------------- */
/*
 * Grid slot g is decoded iff the host walk marked it delivered (bit g of 'bits'); its burst type is the
 * classification word's.  Lists keep slot order (the forward fill of the scrambling code relies on the
 * SYNC ordinals growing with g): counts per 1024-slot block, exclusive scan over the blocks, emit.
 * blk[] : 3 words per block (sb, 216-items, 432-items), turned into exclusive bases in place; the three
 * totals follow at blk[3 * nblocks].
 */
#define GRID_BLOCK 1024

__device__ __forceinline__ uint32_t grid_type(const uint32_t *cls, const uint32_t *bits, uint32_t g, uint32_t n)
{
	if (g >= n || !((bits[g >> 5] >> (g & 31)) & 1))
		return TG_BURST_NONE;
	return cls[g] & 0xff;
}

__global__ __launch_bounds__(GRID_BLOCK)
void k_grid_count(const uint32_t *__restrict__ cls, const uint32_t *__restrict__ bits, uint32_t n, uint32_t *__restrict__ blk)
{
	__shared__ uint32_t sm[GRID_BLOCK / 64][3];
	const uint32_t g = blockIdx.x * GRID_BLOCK + threadIdx.x;
	const uint32_t t = grid_type(cls, bits, g, n);
	const uint32_t nsb = __builtin_popcountll(__ballot(t == TG_BURST_SYNC));
	const uint32_t nn2 = __builtin_popcountll(__ballot(t == TG_BURST_NORM_2));
	const uint32_t nn1 = __builtin_popcountll(__ballot(t == TG_BURST_NORM_1));
	if ((threadIdx.x & 63) == 0) {
		sm[threadIdx.x >> 6][0] = nsb;
		sm[threadIdx.x >> 6][1] = nsb + 2 * nn2;
		sm[threadIdx.x >> 6][2] = nn1;
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		uint32_t a = 0;
		for (int w = 0; w < GRID_BLOCK / 64; w++)
			a += sm[w][threadIdx.x];
		blk[3 * blockIdx.x + threadIdx.x] = a;
	}
}

__global__ __launch_bounds__(GRID_BLOCK)
void k_grid_scan(uint32_t *blk, uint32_t nblocks)
{
	/* exclusive prefix sums of the three per-block counts, 1024 blocks per pass, totals behind the last block */
	__shared__ uint32_t sm[GRID_BLOCK / 64][3];
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t carry[3] = { 0, 0, 0 };
	for (uint32_t base = 0; base < nblocks; base += GRID_BLOCK) {
		const uint32_t i = base + threadIdx.x;
		uint32_t v[3], inc[3];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			v[c] = (i < nblocks) ? blk[3 * i + c] : 0u;
			inc[c] = v[c];
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t o = __shfl_up(inc[c], d);
				if (lane >= (uint32_t)d)
					inc[c] += o;
			}
			if (lane == 63)
				sm[w][c] = inc[c];
		}
		__syncthreads();
#pragma unroll
		for (int c = 0; c < 3; c++) {
			uint32_t pre = carry[c], tot = carry[c];
			for (uint32_t q = 0; q < GRID_BLOCK / 64; q++) {
				if (q < w)
					pre += sm[q][c];
				tot += sm[q][c];
			}
			if (i < nblocks)
				blk[3 * i + c] = pre + inc[c] - v[c];
			carry[c] = tot;
		}
		__syncthreads();
	}
	if (threadIdx.x < 3)
		blk[3 * nblocks + threadIdx.x] = carry[threadIdx.x];
}

__global__ __launch_bounds__(GRID_BLOCK)
void k_grid_emit(const uint32_t *__restrict__ cls, const uint32_t *__restrict__ bits, uint32_t n,
		 const uint32_t *__restrict__ blk, uint32_t *__restrict__ slot_chan, int32_t *__restrict__ slot_sbord,
		 uint32_t *__restrict__ list_sb, uint32_t *__restrict__ list_216, uint32_t *__restrict__ list_432,
		 const tg_chan_ent *__restrict__ chan, uint32_t nchan)
{
	__shared__ uint32_t sm[GRID_BLOCK / 64][3];
	__shared__ uint32_t s_gbase[64];
	if (threadIdx.x < 64)
		s_gbase[threadIdx.x] = (chan && threadIdx.x < nchan) ? chan[threadIdx.x].gbase : 0xffffffffu;
	const uint32_t g = blockIdx.x * GRID_BLOCK + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t t = grid_type(cls, bits, g, n);
	const unsigned long long msb = __ballot(t == TG_BURST_SYNC), mn2 = __ballot(t == TG_BURST_NORM_2);
	const unsigned long long mn1 = __ballot(t == TG_BURST_NORM_1);
	const unsigned long long below = (1ull << lane) - 1;
	if (lane == 0) {
		sm[w][0] = __builtin_popcountll(msb);
		sm[w][1] = __builtin_popcountll(msb) + 2 * __builtin_popcountll(mn2);
		sm[w][2] = __builtin_popcountll(mn1);
	}
	__syncthreads();
	uint32_t bsb = blk[3 * blockIdx.x], b216 = blk[3 * blockIdx.x + 1], b432 = blk[3 * blockIdx.x + 2];
	for (uint32_t q = 0; q < w; q++) {
		bsb += sm[q][0];
		b216 += sm[q][1];
		b432 += sm[q][2];
	}
	const uint32_t psb = bsb + __builtin_popcountll(msb & below);
	const uint32_t p216 = b216 + __builtin_popcountll(msb & below) + 2 * __builtin_popcountll(mn2 & below);
	const uint32_t p432 = b432 + __builtin_popcountll(mn1 & below);
	if (g < n) {
		uint32_t c = 0;		/* channels own consecutive slot ranges: the last one that starts at or before g */
		for (uint32_t q = 1; q < nchan; q++)
			c += s_gbase[q] <= g;
		slot_chan[g] = c;
		slot_sbord[g] = (t == TG_BURST_SYNC) ? (int32_t)psb : -1;
	}
	if (t == TG_BURST_SYNC) {
		list_sb[psb] = g;
		list_216[p216] = (g << 1) | 1;	/* SB2 */
	} else if (t == TG_BURST_NORM_2) {
		list_216[p216] = g << 1;
		list_216[p216 + 1] = (g << 1) | 1;
	} else if (t == TG_BURST_NORM_1)
		list_432[p432] = g;
}

extern "C" int tgk_grid_lists(const uint32_t *d_cls, const uint32_t *d_bits, uint32_t n, uint32_t *d_blk,
			      uint32_t *d_slot_chan, int32_t *d_slot_sbord, uint32_t *d_list_sb, uint32_t *d_list_216,
			      uint32_t *d_list_432, const struct tg_chan_ent *d_chan, uint32_t nchan, void *stream)
{
	if (!n)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const uint32_t nblocks = (n + GRID_BLOCK - 1) / GRID_BLOCK;
	hipLaunchKernelGGL(k_grid_count, dim3(nblocks), dim3(GRID_BLOCK), 0, s, d_cls, d_bits, n, d_blk);
	hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(GRID_BLOCK), 0, s, d_blk, nblocks);
	hipLaunchKernelGGL(k_grid_emit, dim3(nblocks), dim3(GRID_BLOCK), 0, s, d_cls, d_bits, n, d_blk, d_slot_chan, d_slot_sbord,
			   d_list_sb, d_list_216, d_list_432, d_chan, nchan);
	return (int)hipGetLastError();
}

extern "C" int tgk_masks_dev(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_sb_ok,
			     const uint32_t *d_sb_code, uint32_t nsb, const uint32_t *d_nsb, const uint32_t *d_list_sb,
			     const uint32_t *d_slot_chan, uint32_t *d_masks, void *stream)
{
	const uint32_t nent = 1 + nchan + nsb;
	uint32_t blocks = ((nent + 63) / 64 + 3) / 4;	/* a wave per 64 entries */
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(k_masks, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_chan_code, nchan, d_sb_ok, d_sb_code, nsb,
			   d_nsb, d_list_sb, d_slot_chan, d_masks);
	return (int)hipGetLastError();
}

extern "C" int tgk_masks(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_sb_ok,
			 const uint32_t *d_sb_code, uint32_t nsb, const uint32_t *d_list_sb, const uint32_t *d_slot_chan,
			 uint32_t *d_masks, void *stream)
{
	return tgk_masks_dev(d_chan_code, nchan, d_sb_ok, d_sb_code, nsb, NULL, d_list_sb, d_slot_chan, d_masks, stream);
}

/* ------------------------------------------------------------------------- */
/* k_walk: the burst synchroniser's walk over a classified grid, on the device */
/* ------------------------------------------------------------------------- */
/*
 * tetra_burst_sync_in() (phy/tetra_burst_sync.c:54-154) per channel, from the classification words, the SYNC
 * summaries and k_cls_plain's bitmap, without the host: what tg_stream.c:sync_walk() computes in grid mode
 * (delivered bitmap, events, counts, final state), in the node form of tg_walk_core.h.
 *
 * One workgroup of 1024 threads per channel; bitmap, node list and arrival pointers live in LDS:
 *   A  the channel's plain bitmap -> LDS; nodes = its zero bits; per-word prefix counts (block scan)
 *   B  node list (slot of the i-th node)
 *   C  every node through tgw_run(), one lane each (a dozen dependent reads of cls / ysum per node: latency bound,
 *      hidden by the thousand lanes); the stream's head (the first lock, found on the host) likewise; arrival slot
 *      -> index of the first node at or after it
 *   D  which nodes does the walk visit?  reachability from the head's arrival along the arrival pointers: pointer
 *      doubling, marks {succ^n(head) : n < 2^r} after r rounds
 *   E  delivered bitmap = plain bitmap - spans of the visited nodes [slot, arrival) + their own deliveries
 *   F  bitmap -> global, number of delivered bursts, last delivered slot
 *   G  events of the visited nodes in slot order (block scan of the counts), bursts handled but not delivered,
 *      those after the last delivery (tail_tn_adds), final state
 * A channel with more than TGW_NCAP nodes or TGW_WCAP bitmap words, or whose walk meets something only the bytes can
 * settle (tg_walk_core.h), reports TGW_FALLBACK: the host walk takes over.
 */
#include "tg_walk_core.h"
static_assert(sizeof(tgw_rec) <= TGW_REC_BYTES, "TGW_REC_BYTES");

#define TGW_THREADS 1024
#define TGW_LDS_BYTES (TGW_WCAP * 4 + TGW_NCAP * 4 + TGW_WCAP * 2 + 2 * (TGW_NCAP + 8) * 2 + 2 * (TGW_NCAP + 8))	/* MODE 0, full caps */
#define TGW_LDS2_BYTES(wcap, ncap) ((wcap) * 4u + 2u * ((ncap) + 8u) * 2u + 2u * ((ncap) + 8u))	/* MODE 2 */
#ifndef TGW_THREADS_LIGHT
#define TGW_THREADS_LIGHT 256
#endif
#ifndef TGW_NODES_THREADS
#define TGW_NODES_THREADS 256
#endif
/* TGW_THREADS_LIGHT: the three-launch form: a workgroup that takes one wave slot per SIMD and 20-70 KB of LDS finds a
				 * place beside the heavy kernels of the other batches; 1024 threads and 128 KB wait for a nearly empty
				 * compute unit */

__device__ __forceinline__ uint32_t tgw_block_excl_scan(uint32_t v, uint32_t *sm /* 17 words */, uint32_t &total)
{
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(inc, d);
		if (lane >= (uint32_t)d)
			inc += o;
	}
	__syncthreads();	/* sm may still be read from the previous scan */
	if (lane == 63)
		sm[w] = inc;
	__syncthreads();
	uint32_t pre = 0, tot = 0;
	for (uint32_t q = 0; q < blockDim.x / 64; q++) {
		const uint32_t x = sm[q];
		if (q < w)
			pre += x;
		tot += x;
	}
	total = tot;
	return pre + inc - v;
}

#ifdef TGW_TIMING
__device__ unsigned long long g_tgw_stamp[64][12];
#define TGW_STAMP(i) do { if (threadIdx.x == 0) g_tgw_stamp[blockIdx.x][i] = __builtin_readcyclecounter(); } while (0)
extern "C" int tgk_walk_stamps(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tgw_stamp), sizeof(g_tgw_stamp));
}
#else
#define TGW_STAMP(i) do { } while (0)
#endif

/*
 * The body, for both homes of its working arrays: BIG = false, LDS (a channel of up to TGW_WCAP bitmap words and TGW_NCAP
 * nodes: 16-bit node indices, loop counts known at compile time); BIG = true, global memory (k_walk_big: a channel
 * beyond that -- a recording of more than 262 144 slots -- with the caps the plan's scratch area was made for; the same
 * steps at L2 latency, a millisecond or two for a million slots, on one compute unit beside the other batches' kernels).
 */
/* MODE 0: all of it in one launch.  MODE 1 + k_walk_nodes + MODE 2: the node pass (C) -- a dozen dependent reads per node, the
 * longest stretch of the walk -- as a launch of its own over the whole chip: MODE 1 runs A and B and leaves node list, word
 * prefixes and the node count in global memory (tg_walk_tmp), k_walk_nodes takes every node of every channel through tgw_run()
 * (256 nodes per workgroup), MODE 2 runs A and B again (the LDS form; the other keeps its arrays), picks the arrival pointers
 * up and goes on with D .. G. */
struct tg_walk_tmp {
	uint32_t *nslot, *wpre, *J;	/* this channel's node list, per-word prefix counts, arrival pointers */
	uint32_t *meta;			/* this channel's {node count or ~0: nothing to do, head, fallback flag, reason} */
};
template <bool BIG, typename idx_t, int MODE>
__device__ __forceinline__ void walk_body(uint32_t c, uint32_t *bm, uint32_t *nslot, idx_t *wpre, idx_t *Ja, idx_t *Jb, uint8_t *mark,
					  const tg_walk_tmp tmp, const bool skip,
					  tgw_rec *recs, tgpu_sync_event_rec_dev *ev_big, const uint32_t wcap, const uint32_t ncap,
					  const uint32_t rootidx /* the head's record: recs[rootidx] */, const uint32_t evcap, const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan,
					  const tg_walk_root *__restrict__ roots, uint32_t chunk, uint32_t cshift,
					  const uint32_t *__restrict__ g_cls, const uint16_t *__restrict__ g_ysum,
					  const uint32_t *__restrict__ g_plain, uint32_t *__restrict__ g_bits, uint32_t *__restrict__ g_bits2,
					  tg_walk_sum *__restrict__ sums, tgpu_sync_event_rec_dev *__restrict__ g_eager)
{
	__shared__ uint32_t sm[20];
	__shared__ uint32_t s_head, s_fb, s_why, s_nd, s_tail, s_last, s_lastdel, s_ns;

	const uint32_t tid = threadIdx.x, NT = blockDim.x;
	/* the light form (the LDS form split in three launches, MODE 1 / 2): MODE 1 keeps nothing in LDS -- node list and word
	 * prefixes go straight to the hand-over area --, MODE 2 keeps the bitmap, the arrival pointers and the marks */
	constexpr bool LIGHT1 = !BIG && MODE == 1, LIGHT2 = !BIG && MODE == 2;
	const tg_chan_ent ce = chan[c];
	const tg_walk_root rt = roots[c];
	tg_walk_sum *sum = sums + c;
	/* event e of the channel: the first TGW_EVEAGER in the block that is copied to the host with the batch, the rest behind */
	tgpu_sync_event_rec_dev *ev_eager = g_eager + (size_t)c * TGW_EVEAGER;
	const uint32_t ncls = ce.ncls, W = (ncls + 31) >> 5, w0 = ce.gbase >> 5;

	if (tid == 0) {
		s_head = 0xffffffffu;
		s_fb = 0;
		s_why = 0;
		s_nd = s_tail = 0;
		s_last = 0;		/* 1 + index of the last visited node (0: only the head run) */
		s_lastdel = 0;
		s_ns = 0;
	}
	__syncthreads();
	if (MODE == 1 && tid == 0)
		tmp.meta[0] = 0xffffffffu;	/* until A and B are through: nothing for k_walk_nodes / MODE 2 to do */
	if (!ncls || W > wcap || skip) {	/* nothing classified (the host settles such a channel), too long for this form's arrays, or left to the long form */
		if (tid == 0) {
			sum->nslots = sum->nevents = sum->tail_tn_adds = sum->burst_seq = 0;
			sum->final_state = TGW_S_UNLOCKED;
			sum->status = ncls ? TGW_FALLBACK : TGW_OK;
			sum->why = ncls ? TGW_WHY_SIZE : 0;	/* (k_walk_big overwrites this where it runs) */
			sum->nnodes = 0;
		}
		return;
	}
	tgw_chan wc;
	wc.cls = g_cls + ce.gbase;
	wc.ysum = g_ysum + ce.gbase;
	wc.packed = (ce.d_off & TG_CHAN_PACKED) != 0;
	wc.sbit = ce.d_off & ~TG_CHAN_PACKED;
	wc.s = wc.packed ? d_base : d_base + ce.d_off;
	wc.len = ce.len;
	wc.anchor = ce.anchor;
	wc.ncalls = (ce.len + chunk - 1) >> cshift;
	wc.ncls = ncls;
	wc.chunk = chunk;
	wc.cshift = cshift;

	TGW_STAMP(0);
	/* A: bitmap -> LDS (bits at and past ncls read "plain" so that they are no nodes; they are cleared again in F) */
	const uint32_t WPT = (W + NT - 1) / NT;	/* consecutive words per thread */
	uint32_t cnt = 0;
	uint32_t N, base = 0;
	auto plain_word = [&](uint32_t w) -> uint32_t {
		uint32_t v = g_plain[w0 + w];
		if (w == W - 1 && (ncls & 31))
			v |= ~0u << (ncls & 31);
		return v;
	};
	if (BIG && MODE == 2) {		/* (bitmap, node list and prefixes are where MODE 1 left them) */
		N = tmp.meta[0];
		if (N == 0xffffffffu)
			return;
	} else if (LIGHT2) {		/* (node count from MODE 1; ~0: that launch has settled the channel's summary already) */
		N = tmp.meta[0];
		if (N == 0xffffffffu)
			return;
		for (uint32_t w = tid; w < W; w += NT)
			bm[w] = plain_word(w);
	} else {
	for (uint32_t q = 0; q < WPT; q++) {
		const uint32_t w = WPT * tid + q;
		if (w < W) {
			const uint32_t v = plain_word(w);
			if (!LIGHT1)
				bm[w] = v;
			cnt += __popc(~v);
		}
	}
	base = tgw_block_excl_scan(cnt, sm, N);
	}
	if (N > ncap) {
		if (tid == 0) {
			sum->nslots = sum->nevents = sum->tail_tn_adds = sum->burst_seq = 0;
			sum->final_state = TGW_S_UNLOCKED;
			sum->status = TGW_FALLBACK;
			sum->why = TGW_WHY_NODES;
			sum->nnodes = N;
		}
		return;
	}
	/* B: prefix counts per word, node list */
	if (MODE != 2) {
	for (uint32_t q = 0; q < WPT; q++) {
		const uint32_t w = WPT * tid + q;
		if (w < W) {
			uint32_t z;
			if (LIGHT1) {
				tmp.wpre[w] = base;
				z = ~plain_word(w);
			} else {
				wpre[w] = (idx_t)base;
				z = ~bm[w];
			}
			while (z) {
				const uint32_t b = __builtin_ctz(z);
				z &= z - 1;
				if (LIGHT1)
					tmp.nslot[base++] = 32 * w + b;
				else
					nslot[base++] = 32 * w + b;
			}
		}
	}
	}
	__syncthreads();
	if (MODE == 1) {	/* the lists are k_walk_nodes' now (BIG: the body's arrays are the hand-over area) */
		if (tid == 0)
			tmp.meta[0] = N;
		return;
	}
	auto rank = [&](uint32_t t) -> uint32_t {	/* index of the first node at or after grid slot t */
		if (t >= ncls)
			return N;
		const uint32_t w = t >> 5;
		return (uint32_t)wpre[w] + __popc(~bm[w] & ((1u << (t & 31)) - 1u));
	};
	TGW_STAMP(1);
	/* C: every node, and the stream's head */
	if (MODE == 2) {	/* (done by k_walk_nodes) */
		if (!BIG)
			for (uint32_t i = tid; i < N; i += NT)
				Ja[i] = (idx_t)tmp.J[i];
		if (tid == 0) {
			s_head = tmp.meta[1];
			s_fb = tmp.meta[2];
			s_why = tmp.meta[3];
		}
	} else {
	for (uint32_t i = tid; i < N; i += NT) {
		const uint64_t bs = wc.anchor + (uint64_t)nslot[i] * TG_SLOT_BITS;
		const uint64_t kc = (bs + TG_SLOT_BITS + chunk - 1) >> cshift;
		tgw_rec r;
		tgw_run(&wc, TGW_S_LOCKED, bs, bs + TG_SLOT_BITS, kc - 1, &r);
		recs[i] = r;
		Ja[i] = (idx_t)(r.status == TGW_OK ? rank(r.next) : N);
	}
	if (tid == NT - 1) {
		tgw_rec r;
		tgw_run(&wc, TGW_S_KNOW_FSTART, rt.found_bs, wc.anchor, rt.found_k, &r);
		recs[rootidx] = r;
		if (r.status != TGW_OK) {
			s_fb = 1;
			s_why = r.why;
		} else
			s_head = rank(r.next);
	}
	}
	if (tid == 0)
		Ja[N] = Jb[N] = (idx_t)N;
	for (uint32_t i = tid; i <= N; i += NT)
		mark[i] = 0;
	__syncthreads();
	TGW_STAMP(2);
	/* D: reachability from the head along the arrival pointers */
	{
		const uint32_t head = s_head;
		if (tid == 0 && head < N)
			mark[head] = 1;
		__syncthreads();
		idx_t *J = Ja, *Jn = Jb;
		for (uint32_t span = 1; span <= N; span <<= 1) {
			for (uint32_t v = tid; v < N; v += NT)
				if (mark[v] && J[v] < N)
					mark[J[v]] = 1;
			for (uint32_t v = tid; v < N; v += NT) {
				const uint32_t j = J[v];
				Jn[v] = j < N ? J[j] : (idx_t)N;
			}
			__syncthreads();
			idx_t *t = J;
			J = Jn;
			Jn = t;
		}
	}
	TGW_STAMP(3);
	/* E: spans of the visited nodes (and of the head run) leave the bitmap, their own deliveries enter it */
	auto clear_span = [&](uint32_t from, uint32_t to) {	/* grid slots [from, to) */
		if (to > ncls)
			to = ncls;
		while (from < to) {
			const uint32_t w = from >> 5, b = from & 31;
			const uint32_t n = (32 - b < to - from) ? 32 - b : to - from;
			const uint32_t m = (n == 32 ? 0xffffffffu : ((1u << n) - 1u)) << b;
			atomicAnd(&bm[w], ~m);
			from += n;
		}
	};
	for (uint32_t i = tid; i < N; i += NT)
		if (mark[i]) {
			const tgw_rec *r = recs + i;
			if (r->status != TGW_OK) {
				s_fb = 1;
				s_why = r->why;
			}
			clear_span(MODE == 2 ? tmp.nslot[i] : nslot[i], r->next);
			atomicMax(&s_last, i + 1);
		}
	if (tid == NT - 1 && !s_fb)
		clear_span(0, recs[rootidx].next);
	__syncthreads();
	if (s_fb) {
		if (tid == 0) {
			sum->nslots = sum->nevents = sum->tail_tn_adds = sum->burst_seq = 0;
			sum->final_state = TGW_S_UNLOCKED;
			sum->status = TGW_FALLBACK;
			sum->why = s_why;
			sum->nnodes = N;
		}
		return;
	}
	for (uint32_t i = tid; i < N + 1; i += NT) {
		const bool root = (i == N);
		if (root || mark[i]) {
			const tgw_rec *r = recs + (root ? rootidx : i);
			for (uint32_t d = 0; d < r->ndel; d++)
				atomicOr(&bm[r->del[d] >> 5], 1u << (r->del[d] & 31));
		}
	}
	__syncthreads();
	TGW_STAMP(4);
	/* F: bitmap out, delivered bursts, last delivered slot */
	{
		uint32_t ns = 0, lastd = 0xffffffffu;
		for (uint32_t q = 0; q < WPT; q++) {
			const uint32_t w = WPT * tid + q;
			if (w < W) {
				uint32_t v = bm[w];
				if (w == W - 1 && (ncls & 31))
					v &= (1u << (ncls & 31)) - 1u;
				g_bits[w0 + w] = v;
				g_bits2[w0 + w] = v;
				ns += __popc(v);
				if (v)
					lastd = 32 * w + 31 - __builtin_clz(v);
			}
		}
		if (ns)
			atomicAdd(&s_ns, ns);
		if (lastd != 0xffffffffu)
			atomicMax(&s_lastdel, lastd + 1);	/* 1 + last delivered slot, 0 = none */
	}
	__syncthreads();
	TGW_STAMP(5);
	/* G: events in slot order */
	const uint32_t NPT = (N + NT - 1) / NT;
	const tgw_rec *root = recs + rootidx;
	uint32_t ecnt = 0;
	for (uint32_t q = 0; q < NPT; q++) {
		const uint32_t i = NPT * tid + q;
		if (i < N && mark[i])
			ecnt += recs[i].nev;
	}
	uint32_t etot;
	uint32_t eoff = tgw_block_excl_scan(ecnt, sm, etot) + root->nev;
	etot += root->nev;
	const uint32_t lastdel = s_lastdel;
	uint32_t nd = 0, tail = 0;
	auto emit = [&](const tgw_rec *r, uint32_t at) {
		for (uint32_t e = 0; e < r->nev; e++) {
			if (at + e < evcap) {
				tgpu_sync_event_rec_dev *o = at + e < TGW_EVEAGER ? ev_eager + at + e : ev_big + at + e;
				o->ev = (int32_t)r->ev[e][0];
				o->bitnum = r->ev[e][1];
				o->arg = r->ev[e][2];
			}
			if (r->evslot[e] != TGW_NOSLOT) {
				nd++;
				if (r->evslot[e] + 1 > lastdel)
					tail++;
			}
		}
	};
	if (tid == NT - 1)
		emit(root, 0);
	for (uint32_t q = 0; q < NPT; q++) {
		const uint32_t i = NPT * tid + q;
		if (i < N && mark[i]) {
			emit(recs + i, eoff);
			eoff += recs[i].nev;
		}
	}
	if (nd)
		atomicAdd(&s_nd, nd);
	if (tail)
		atomicAdd(&s_tail, tail);
	__syncthreads();
	if (tid == 0) {
		const tgw_rec *lastrec = s_last ? recs + (s_last - 1) : root;
		sum->nslots = s_ns;
		sum->nevents = etot;
		sum->final_state = lastrec->next == TGW_END ? lastrec->end_state : TGW_S_LOCKED;
		sum->tail_tn_adds = s_tail;
		sum->burst_seq = s_ns + s_nd;
		sum->status = etot > evcap ? TGW_FALLBACK : TGW_OK;
		sum->why = etot > evcap ? TGW_WHY_EVENTS : 0;
		sum->nnodes = N;
	}
	TGW_STAMP(6);
}

/* layout of the split form's hand-over area (the LDS form): per channel four words of meta data, then node list, word prefixes
 * and arrival pointers as 32-bit words */
#define TGW_TMP_META_BYTES 1024u	/* 64 channels x {N, head, fallback, reason} */
#define TGW_TMP_CHAN_WORDS (TGW_NCAP + TGW_WCAP + TGW_NCAP + 8u)
__device__ __forceinline__ tg_walk_tmp walk_tmp_small(uint8_t *d_tmp, uint32_t c)
{
	tg_walk_tmp t;
	uint32_t *w = (uint32_t *)(d_tmp + TGW_TMP_META_BYTES) + (size_t)c * TGW_TMP_CHAN_WORDS;
	t.nslot = w;
	t.wpre = w + TGW_NCAP;
	t.J = w + TGW_NCAP + TGW_WCAP;
	t.meta = (uint32_t *)d_tmp + 4 * c;
	return t;
}
__device__ __forceinline__ tg_walk_tmp walk_tmp_big(uint8_t *slot, const tg_walk_big_layout &L, uint8_t *d_tmp, uint32_t c)
{
	tg_walk_tmp t;
	t.nslot = (uint32_t *)(slot + L.o_nslot);
	t.wpre = (uint32_t *)(slot + L.o_wpre);
	t.J = (uint32_t *)(slot + L.o_ja);
	t.meta = (uint32_t *)d_tmp + 4 * c;
	return t;
}

template <int MODE>
__global__ __launch_bounds__(TGW_THREADS)
void k_walk(const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan, const tg_walk_root *__restrict__ roots,
	    uint32_t chunk, uint32_t cshift, const uint32_t *__restrict__ g_cls, const uint16_t *__restrict__ g_ysum,
	    const uint32_t *__restrict__ g_plain, uint32_t *__restrict__ g_bits, uint32_t *__restrict__ g_bits2,
	    tg_walk_sum *__restrict__ sums, tgpu_sync_event_rec_dev *__restrict__ g_eager,
	    tgpu_sync_event_rec_dev *__restrict__ g_evbig, tgw_rec *__restrict__ g_recs, uint8_t *__restrict__ d_tmp,
	    unsigned long long skip_mask, uint32_t wcap, uint32_t ncap, uint32_t rec_stride)
{
	/* working arrays in LDS, laid out for the caps of this launch (tgk_walk): MODE 0 all of them, MODE 2 the bitmap, the two
	 * arrival-pointer arrays and the marks, MODE 1 none */
	extern __shared__ uint32_t s_dyn[];
	uint32_t *bm = s_dyn;
	uint32_t *nslot = bm + wcap;
	uint16_t *wpre = (uint16_t *)(nslot + ncap);
	uint16_t *Ja = (MODE == 2) ? (uint16_t *)(bm + wcap) : wpre + wcap, *Jb = Ja + ncap + 8;
	uint8_t *mark = (uint8_t *)(Jb + ncap + 8);
	const uint32_t c = blockIdx.x;
	tg_walk_tmp tmp = { nullptr, nullptr, nullptr, nullptr };
	if (MODE)
		tmp = walk_tmp_small(d_tmp, c);
	walk_body<false, uint16_t, MODE>(c, bm, nslot, wpre, Ja, Jb, mark, tmp, ((skip_mask >> c) & 1) != 0, g_recs + (size_t)c * rec_stride,
					 g_evbig + (size_t)c * TGW_EVCAP, wcap, ncap < rec_stride - 1 ? ncap : rec_stride - 1, rec_stride - 1, TGW_EVCAP, d_base, chan, roots, chunk, cshift, g_cls,
					 g_ysum, g_plain, g_bits, g_bits2, sums, g_eager);
}

/* the channels of the batch that are too long for the form above (their indices in `big`), one workgroup each, working
 * arrays in the plan's scratch area (tg_walk_big_layout): runs behind k_walk, which has reported them as TGW_WHY_SIZE */
template <int MODE>
__global__ __launch_bounds__(TGW_THREADS)
void k_walk_big(tg_walk_big big, uint8_t *__restrict__ scratch, const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan,
		const tg_walk_root *__restrict__ roots, uint32_t chunk, uint32_t cshift, const uint32_t *__restrict__ g_cls,
		const uint16_t *__restrict__ g_ysum, const uint32_t *__restrict__ g_plain, uint32_t *__restrict__ g_bits,
		uint32_t *__restrict__ g_bits2, tg_walk_sum *__restrict__ sums, tgpu_sync_event_rec_dev *__restrict__ g_eager,
		uint8_t *__restrict__ d_tmp)
{
	tg_walk_big_layout L;
	tg_walk_big_offsets(big.wcap, big.ncap, big.evcap, &L);
	uint8_t *base = scratch + (size_t)blockIdx.x * L.slot_bytes;
	const uint32_t c = big.chan[blockIdx.x];
	tg_walk_tmp tmp = { nullptr, nullptr, nullptr, nullptr };
	if (MODE)
		tmp = walk_tmp_big(base, L, d_tmp, c);
	walk_body<true, uint32_t, MODE>(c, (uint32_t *)(base + L.o_bm), (uint32_t *)(base + L.o_nslot), (uint32_t *)(base + L.o_wpre),
					(uint32_t *)(base + L.o_ja), (uint32_t *)(base + L.o_jb), base + L.o_mark, tmp, false, (tgw_rec *)(base + L.o_recs),
					(tgpu_sync_event_rec_dev *)(base + L.o_ev), big.wcap, big.ncap, big.ncap, big.evcap, d_base, chan, roots, chunk,
					cshift, g_cls, g_ysum, g_plain, g_bits, g_bits2, sums, g_eager);
}

/* phase C of the walk over the whole chip: workgroup (x, y) takes nodes 256 x .. of channel y (BIG: of the y-th long channel)
 * through tgw_run(); its first thread also runs the stream's head */
template <bool BIG>
__global__ __launch_bounds__(256)
void k_walk_nodes(tg_walk_big big, uint8_t *__restrict__ scratch, uint8_t *__restrict__ d_tmp, tgw_rec *__restrict__ g_recs,
		  const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan, const tg_walk_root *__restrict__ roots,
		  uint32_t chunk, uint32_t cshift, const uint32_t *__restrict__ g_cls, const uint16_t *__restrict__ g_ysum,
		  const uint32_t *__restrict__ g_plain, uint32_t rec_stride)
{
	const uint32_t c = BIG ? big.chan[blockIdx.y] : blockIdx.y;
	tg_walk_tmp tmp;
	tgw_rec *recs;
	uint32_t ncap;
	if (BIG) {
		tg_walk_big_layout L;
		tg_walk_big_offsets(big.wcap, big.ncap, big.evcap, &L);
		uint8_t *slot = scratch + (size_t)blockIdx.y * L.slot_bytes;
		tmp = walk_tmp_big(slot, L, d_tmp, c);
		recs = (tgw_rec *)(slot + L.o_recs);
		ncap = big.ncap;
	} else {
		tmp = walk_tmp_small(d_tmp, c);
		recs = g_recs + (size_t)c * rec_stride;
		ncap = rec_stride - 1;
	}
	const uint32_t N = tmp.meta[0];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (N == 0xffffffffu || (i >= N && i != 0))
		return;
	const tg_chan_ent ce = chan[c];
	const uint32_t ncls = ce.ncls, W = (ncls + 31) >> 5, w0 = ce.gbase >> 5;
	tgw_chan wc;
	wc.cls = g_cls + ce.gbase;
	wc.ysum = g_ysum + ce.gbase;
	wc.packed = (ce.d_off & TG_CHAN_PACKED) != 0;
	wc.sbit = ce.d_off & ~TG_CHAN_PACKED;
	wc.s = wc.packed ? d_base : d_base + ce.d_off;
	wc.len = ce.len;
	wc.anchor = ce.anchor;
	wc.ncalls = (ce.len + chunk - 1) >> cshift;
	wc.ncls = ncls;
	wc.chunk = chunk;
	wc.cshift = cshift;
	auto rank = [&](uint32_t t) -> uint32_t {	/* index of the first node at or after grid slot t */
		if (t >= ncls)
			return N;
		const uint32_t w = t >> 5;
		uint32_t v = g_plain[w0 + w];
		if (w == W - 1 && (ncls & 31))
			v |= ~0u << (ncls & 31);
		return tmp.wpre[w] + __popc(~v & ((1u << (t & 31)) - 1u));
	};
	if (i < N) {
		const uint64_t bs = wc.anchor + (uint64_t)tmp.nslot[i] * TG_SLOT_BITS;
		const uint64_t kc = (bs + TG_SLOT_BITS + chunk - 1) >> cshift;
		tgw_rec r;
		tgw_run(&wc, TGW_S_LOCKED, bs, bs + TG_SLOT_BITS, kc - 1, &r);
		recs[i] = r;
		tmp.J[i] = r.status == TGW_OK ? rank(r.next) : N;
	}
	if (i == 0) {
		const tg_walk_root rt = roots[c];
		tgw_rec r;
		tgw_run(&wc, TGW_S_KNOW_FSTART, rt.found_bs, wc.anchor, rt.found_k, &r);
		recs[ncap] = r;
		tmp.meta[1] = r.status == TGW_OK ? rank(r.next) : 0xffffffffu;
		tmp.meta[2] = r.status != TGW_OK;
		tmp.meta[3] = r.status != TGW_OK ? r.why : 0u;
	}
}

extern "C" int tgk_walk(const uint8_t *d_base, const struct tg_chan_ent *d_chan, const struct tg_walk_root *d_roots, uint32_t nchan,
			uint32_t chunk, const uint32_t *d_cls, const uint16_t *d_ysum, const uint32_t *d_plain, uint32_t *d_bits,
			uint32_t *d_bits2, struct tg_walk_sum *d_sums, void *d_eager, void *d_evbig, void *d_recs, void *d_tmp,
			unsigned long long skip_mask, uint32_t wcap, uint32_t ncap, uint32_t rec_stride, int wide, void *stream)
{
	if (!nchan)
		return 0;
	if (!chunk || (chunk & (chunk - 1)) || nchan > 64 || rec_stride < 2 || rec_stride > TGW_NCAP + 1)
		return -1;
	if (!d_tmp || wide || !wcap || !ncap || wcap > TGW_WCAP || ncap > TGW_NCAP) {
		wcap = TGW_WCAP;
		ncap = TGW_NCAP;
	}
	wcap = (wcap + 1u) & ~1u;
	ncap = (ncap + 255u) & ~255u;
	const uint32_t cshift = (uint32_t)__builtin_ctz(chunk);
	hipStream_t s = (hipStream_t)stream;
#define WALK_ARGS d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, d_bits, d_bits2, d_sums, \
		  (tgpu_sync_event_rec_dev *)d_eager, (tgpu_sync_event_rec_dev *)d_evbig, (tgw_rec *)d_recs, (uint8_t *)d_tmp, skip_mask, \
		  wcap, ncap, rec_stride
	if (!d_tmp) {
		HIPCHK(hipFuncSetAttribute((const void *)k_walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TGW_LDS_BYTES));
		hipLaunchKernelGGL(k_walk<0>, dim3(nchan), dim3(TGW_THREADS), TGW_LDS_BYTES, s, WALK_ARGS);
		return (int)hipGetLastError();
	}
	static __thread int attr_set_dev = -1;	/* (the attribute is per device and process: once per thread and device is enough) */
	int dev = 0;
	HIPCHK(hipGetDevice(&dev));
	if (attr_set_dev != dev) {
		HIPCHK(hipFuncSetAttribute((const void *)k_walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TGW_LDS_BYTES));
		HIPCHK(hipFuncSetAttribute((const void *)k_walk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TGW_LDS_BYTES));
		attr_set_dev = dev;
	}
	const uint32_t nt = wide ? TGW_THREADS : TGW_THREADS_LIGHT;
	hipLaunchKernelGGL(k_walk<1>, dim3(nchan), dim3(nt), wide ? TGW_LDS_BYTES : 0, s, WALK_ARGS);
	tg_walk_big none = {};
	hipLaunchKernelGGL(k_walk_nodes<false>, dim3(ncap / TGW_NODES_THREADS, nchan), dim3(TGW_NODES_THREADS), 0, s, none, (uint8_t *)nullptr, (uint8_t *)d_tmp,
			   (tgw_rec *)d_recs, d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, rec_stride);
	hipLaunchKernelGGL(k_walk<2>, dim3(nchan), dim3(nt), wide ? TGW_LDS_BYTES : TGW_LDS2_BYTES(wcap, ncap), s, WALK_ARGS);
#undef WALK_ARGS
	return (int)hipGetLastError();
}

extern "C" int tgk_walk_big(const struct tg_walk_big *big, void *d_scratch, const uint8_t *d_base, const struct tg_chan_ent *d_chan,
			    const struct tg_walk_root *d_roots, uint32_t chunk, const uint32_t *d_cls, const uint16_t *d_ysum,
			    const uint32_t *d_plain, uint32_t *d_bits, uint32_t *d_bits2, struct tg_walk_sum *d_sums, void *d_eager, void *d_tmp,
			    void *stream)
{
	if (!big || !big->n)
		return 0;
	if (big->n > TGW_BIG_MAX || !d_scratch || !chunk || (chunk & (chunk - 1)))
		return -1;
	const uint32_t cshift = (uint32_t)__builtin_ctz(chunk);
	hipStream_t s = (hipStream_t)stream;
#define WALK_ARGS *big, (uint8_t *)d_scratch, d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, d_bits, d_bits2, d_sums, \
		  (tgpu_sync_event_rec_dev *)d_eager, (uint8_t *)d_tmp
	if (!d_tmp) {
		hipLaunchKernelGGL(k_walk_big<0>, dim3(big->n), dim3(TGW_THREADS), 0, s, WALK_ARGS);
		return (int)hipGetLastError();
	}
	hipLaunchKernelGGL(k_walk_big<1>, dim3(big->n), dim3(TGW_THREADS), 0, s, WALK_ARGS);
	hipLaunchKernelGGL(k_walk_nodes<true>, dim3((big->ncap + 255) / 256, big->n), dim3(256), 0, s, *big, (uint8_t *)d_scratch,
			   (uint8_t *)d_tmp, (tgw_rec *)nullptr, d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, 0u);
	hipLaunchKernelGGL(k_walk_big<2>, dim3(big->n), dim3(TGW_THREADS), 0, s, WALK_ARGS);
#undef WALK_ARGS
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* k_gsmtap: GSMTAP messages of a decoded batch          */
/* ------------------------------------------------------------------------- */
/*
 * What the reference's upper MAC sends for every CRC-OK block it is indicated (tetra_upper_mac.c:483-488 ->
 * tetra_gsmtap.c:31-63): 16-byte GSMTAP v2 header (type TETRA_I1, timeslot tn - 1, frame number ((hn 60) + mn) 18 + fn in
 * network order, channel sub-type) + the block's type-1 bits packed MSB first.  One thread per (slot, block of the
 * burst in the reference's order: SB1 BBK SB2 / BBK BLK1 BLK2 / BBK SCH-F); message k of slot i at msgs + (3 i + k) *
 * TG_GSMTAP_STRIDE, its length in lens[3 i + k] (0: none -- the block failed its CRC, the burst has no such block, the
 * slot was not decoded, or the caller marks the burst as traffic and the block is one the reference dumps instead).
 * times[i] = the PHY clock when the burst comes in (after the time steps of tetra_burst_sync_in()); a SYNC burst whose
 * SB1 passes its CRC sets tn / fn / mn from its PDU for all three of its blocks (tetra_lower_mac.c:291-302, 332).
 * Logical channels as tetra_lower_mac.c:170-173, 303, 315-319: SB1 BSCH, BBK AACH, SCH-F SCH_F, SB2 BNCH in the BNCH
 * frame, else (SB2, NDB) unknown (sub-type 0).  The first indication of a block only: further PDUs of the same block
 * (tetra_lower_mac.c:330-352) depend on the upper MAC's return value -- the host's tgpu_gsmtap_makemsg() with an offset.
 */
#define TG_GSMTAP_STRIDE 52
__global__ __launch_bounds__(256)
void k_gsmtap(const uint8_t *__restrict__ rec, const tg_tdma_time_dev *__restrict__ times, const uint8_t *__restrict__ traffic,
	      uint32_t nslots, uint8_t *__restrict__ msgs, uint8_t *__restrict__ lens)
{
	const uint32_t id = blockIdx.x * 256 + threadIdx.x;
	if (id >= 3 * nslots)
		return;
	const uint32_t i = id / 3, k = id % 3;
	const uint8_t *r = rec + (size_t)i * TG_REC_BYTES;
	uint8_t *m = msgs + (size_t)id * TG_GSMTAP_STRIDE;
	const uint32_t type = r[TG_REC_TYPE];
	/* block k of the burst: 0 SB1 / 1 BBK / 2 first block / 3 second block, 4 none */
	uint32_t what = 4;
	if (type == TG_BURST_SYNC)
		what = k == 0 ? 0u : k == 1 ? 1u : 3u;
	else if (type == TG_BURST_NORM_2)
		what = k == 0 ? 1u : k == 1 ? 2u : 3u;
	else if (type == TG_BURST_NORM_1)
		what = k == 0 ? 1u : k == 1 ? 2u : 4u;
	const uint8_t *bits = r + TG_REC_BBK;
	uint32_t nbits = 14, ok = 1, sub = 2 /* GSMTAP_TETRA_AACH */;
	tg_tdma_time_dev tm = times[i];
	if (type == TG_BURST_SYNC && r[TG_REC_CRC_OK]) {
		const uint32_t f0 = *(const uint32_t *)(r + TG_REC_SBF0);
		tm.tn = (f0 >> 8) & 0xff;
		tm.fn = (f0 >> 16) & 0xff;
		tm.mn = f0 >> 24;
	}
	const bool is_traffic = traffic && traffic[i];
	if (what == 0) {
		bits = r + TG_REC_BITS1;
		nbits = 60;
		ok = r[TG_REC_CRC_OK];
		sub = 1;	/* BSCH */
	} else if (what == 2) {
		bits = r + TG_REC_BITS1;
		nbits = type == TG_BURST_NORM_1 ? 268 : 124;
		ok = r[TG_REC_CRC_OK];
		sub = type == TG_BURST_NORM_1 ? 5u : 0u;	/* SCH_F; an NDB half has no channel yet (tetra_lower_mac.c:312) */
		if (is_traffic && type == TG_BURST_NORM_1)
			what = 4;			/* dumped, not indicated (tetra_lower_mac.c:198) */
	} else if (what == 3) {
		bits = r + TG_REC_BITS2;
		nbits = 124;
		ok = r[TG_REC_CRC_OK + 1];
		sub = (type == TG_BURST_SYNC && tm.fn == 18 && tm.tn == 4 - ((tm.mn + 3) % 4)) ? 6u : 0u;	/* BNCH (:122-127, 170) */
		if (is_traffic && (traffic[i] & 2) == 0)
			what = 4;			/* second block of a traffic slot that was not stolen */
	}
	if (what == 4 || !ok) {
		lens[id] = 0;
		return;
	}
	const uint32_t fn = ((tm.hn * 60u) + tm.mn) * 18u + tm.fn;
	const uint32_t nbytes = (nbits + 7) >> 3;
	m[0] = 2;	/* GSMTAP_VERSION */
	m[1] = 4;	/* header length in words */
	m[2] = 5;	/* GSMTAP_TYPE_TETRA_I1 */
	m[3] = (uint8_t)(tm.tn - 1);
	m[4] = m[5] = 0;
	m[6] = m[7] = 0;
	m[8] = (uint8_t)(fn >> 24);
	m[9] = (uint8_t)(fn >> 16);
	m[10] = (uint8_t)(fn >> 8);
	m[11] = (uint8_t)fn;
	m[12] = (uint8_t)sub;
	m[13] = m[14] = m[15] = 0;
	for (uint32_t b = 0; b < nbytes; b++) {
		uint32_t v = 0;
		for (uint32_t q = 0; q < 8; q++)
			if (8 * b + q < nbits && bits[8 * b + q])
				v |= 0x80u >> q;
		m[16 + b] = (uint8_t)v;
	}
	lens[id] = (uint8_t)(16 + nbytes);
}

extern "C" int tgk_gsmtap(const uint8_t *d_rec, const void *d_times, const uint8_t *d_traffic, uint32_t nslots, uint8_t *d_msgs,
			  uint8_t *d_lens, void *stream)
{
	if (!nslots)
		return 0;
	hipLaunchKernelGGL(k_gsmtap, dim3((3 * nslots + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_rec,
			   (const tg_tdma_time_dev *)d_times, d_traffic, nslots, d_msgs, d_lens);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* k_stages: the lower MAC's steps one by one (the reference's DEBUGP lines)   */
/* ------------------------------------------------------------------------- */
/*
 * What the product kernels fold into one gather table and a trellis, as the separate steps of
 * lower_mac/tetra_lower_mac.c:175-254 -- for looking inside a block, and as a second formulation the tests hold the fused
 * kernels against.  One workgroup per block, a thread per bit, one byte per bit:
 *   type4[i]   = type5[i] ^ seq(code)[i]                 (tetra_scramb_bits, :178-186; seq in its linear form)
 *   type3[i]   = type4[(a (i + 1)) mod K]                (block_deinterleave, :245)
 *   type3dp    = 0xff everywhere, then type3[j] at position 8 (j / 3) + {0, 1, 4}[j mod 3]
 *                                                        (tetra_rcpc_depunct with the 2/3 puncturer, :249-250)
 * type2 is the generic trellis' (tgpu_conv_execute on type3), the CRC k_stages_crc's.  a == 0 (BBK): type4 only.
 */
__global__ __launch_bounds__(256)
void k_stages(const uint8_t *__restrict__ type5, const uint32_t *__restrict__ codes, uint32_t fixed_code, uint32_t K, uint32_t a,
	      uint32_t mother_len, uint8_t *__restrict__ type4, uint8_t *__restrict__ type3, uint8_t *__restrict__ type3dp)
{
	__shared__ uint8_t s4[432];
	const size_t blk = blockIdx.x;
	const uint32_t code = codes ? codes[blk] : fixed_code;
	for (uint32_t i = threadIdx.x; i < K; i += 256) {
		const uint8_t b = (uint8_t)((type5[blk * K + i] != 0) ^ (__popc(code & c_tab.lfsr_lin[i]) & 1));
		s4[i] = b;
		type4[blk * K + i] = b;
	}
	if (!a)
		return;
	for (uint32_t i = threadIdx.x; i < mother_len; i += 256)
		type3dp[blk * mother_len + i] = 0xff;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < K; i += 256) {
		const uint8_t b = s4[(a * (i + 1)) % K];
		type3[blk * K + i] = b;
		const uint32_t r = i % 3;
		type3dp[blk * mother_len + 8 * (i / 3) + (r == 2 ? 4 : r)] = b;
	}
}

/* CRC-16 of a block's first n bits, bit by bit (x^16 + x^12 + x^5 + 1, register preset to ones, no final complement:
 * crc16_ccitt_bits() of lower_mac/crc_simple.c; a good block leaves 0x1d0f, crc_simple.h), a thread per block */
__global__ __launch_bounds__(256)
void k_stages_crc(const uint8_t *__restrict__ type2, unsigned long long nblocks, uint32_t type2_len, uint32_t n, uint16_t *__restrict__ crc)
{
	const unsigned long long blk = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	if (blk >= nblocks)
		return;
	uint32_t reg = 0xffff;
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t fb = ((reg >> 15) ^ type2[blk * type2_len + i]) & 1u;
		reg = (reg << 1) & 0xffff;
		if (fb)
			reg ^= 0x1021;
	}
	crc[blk] = (uint16_t)reg;
}

extern "C" int tgk_stages(const uint8_t *d_type5, const uint32_t *d_codes, uint32_t fixed_code, unsigned long long nblocks, uint32_t K,
			  uint32_t a, uint32_t mother_len, uint8_t *d_type4, uint8_t *d_type3, uint8_t *d_type3dp, void *stream)
{
	if (!nblocks)
		return 0;
	if (K > 432 || nblocks > 0x7fffffffull)
		return -1;	/* TGPU_EINVAL */
	hipLaunchKernelGGL(k_stages, dim3((uint32_t)nblocks), dim3(256), 0, (hipStream_t)stream, d_type5, d_codes, fixed_code, K, a,
			   mother_len, d_type4, d_type3, d_type3dp);
	return (int)hipGetLastError();
}

extern "C" int tgk_stages_crc(const uint8_t *d_type2, unsigned long long nblocks, uint32_t type2_len, uint32_t n, uint16_t *d_crc,
			      void *stream)
{
	if (!nblocks)
		return 0;
	hipLaunchKernelGGL(k_stages_crc, dim3((uint32_t)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_type2, nblocks,
			   type2_len, n, d_crc);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* device-walk batches: everything between the front end and the trellis kernels in five small launches */
/* ------------------------------------------------------------------------- */
/*
 * The host-walk path builds ordered lists (count / scan / emit), decodes SB1 over the ordered SYNC list, forward-fills
 * the code over all slots (reduce / scan / apply) and builds mask entries per SYNC ordinal: ten launches, ~85 us of small
 * kernels per 1 M slots.  With the walk on the device none of that order is needed:
 *   k_cls_plain2  plain bitmap (as k_cls_plain) + the list of SYNC-classified slots (wave-aggregated append, any order)
 *                 + the channel of every 32-slot word                                    -- before the walk
 *   k_vit<SB1>    over that list, beside the walk (side stream): a block that passes its CRC sets its slot's bit in
 *                 'okbits' and takes a mask-table entry for its code from a small hash table (codes in play are few:
 *                 one per cell), remembered per slot                                     (tg_lb, vit_finish)
 *   k_masks2      the scrambling masks of the entries in use (carry-ins + hash table)
 *   k_lb_scan     per 32-slot word: the latest word at or before it (same channel) that holds a delivered SYNC slot
 *                 with a good SB1 -- one workgroup, running maximum; also every channel's code after the batch
 *   k_lists2      per delivered slot: the mask entry of the latest such SYNC slot at or before it (this slot included:
 *                 an SB1 sets the code for the BBK and SB2 of its own burst, tetra_lower_mac.c:179-186, 291-300), else
 *                 the channel's carry-in; and the slot's items appended to the 216 / 432 lists (wave-aggregated, any
 *                 order -- records are addressed by slot)
 * An undelivered SYNC slot may be decoded (its SB1 costs 84 trellis steps) but never counts: the look-back ANDs okbits
 * with the delivered bitmap.
 */
#define TG_MID_CHUNKS 4		/* 1024-slot chunks per workgroup of k_cls_plain2 / k_lists2: one atomic per 4096 slots and list */
__global__ __launch_bounds__(1024)
void k_cls_plain2(const uint32_t *__restrict__ cls, uint32_t n, uint32_t *__restrict__ plain, uint32_t *__restrict__ list_sb,
		  uint32_t *__restrict__ cnt_sb, uint8_t *__restrict__ word_chan, const tg_chan_ent *__restrict__ chan, uint32_t nchan)
{
	/* (a single word takes ~90 atomics per microsecond: one per wave would be 15 000 of them) */
	__shared__ uint32_t s_cnt[TG_MID_CHUNKS][16], s_base;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned long long sbm[TG_MID_CHUNKS];
	bool sb[TG_MID_CHUNKS];
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		const uint32_t i = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
		const uint32_t w = i >> 5;
		const uint32_t v = i < n ? cls[i] & 0x03ffffffu : 0xffu;
		sb[j] = v == (TG_BURST_SYNC | TG_SYNC_TRAIN_OFF << 8);
		const bool ok = sb[j] || v == (TG_BURST_NORM_1 | TG_NORM_TRAIN_OFF << 8) || v == (TG_BURST_NORM_2 | TG_NORM_TRAIN_OFF << 8);
		const unsigned long long b = __ballot(ok);
		sbm[j] = __ballot(sb[j]);
		if (lane == 0 && 32 * w < n)
			plain[w] = (uint32_t)b;
		if (lane == 32 && 32 * w < n)
			plain[w] = (uint32_t)(b >> 32);
		/* the wave's 64 slots lie in one or two channels (grids start at multiples of 32) */
		if ((lane == 0 || lane == 32) && 32 * w < n) {
			uint32_t c = 0;
			for (uint32_t q = 1; q < nchan; q++)
				c += chan[q].gbase <= i;
			word_chan[w] = (uint8_t)c;
		}
		if (lane == 0)
			s_cnt[j][wv] = (uint32_t)__builtin_popcountll(sbm[j]);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int q = 0; q < TG_MID_CHUNKS * 16; q++)
			tot += s_cnt[0][q];
		s_base = tot ? atomicAdd(cnt_sb, tot) : 0u;
	}
	__syncthreads();
	uint32_t pos = s_base;
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		uint32_t mine = pos;
		for (uint32_t q = 0; q < 16; q++) {
			if (q < wv)
				mine += s_cnt[j][q];
			pos += s_cnt[j][q];
		}
		if (sb[j])
			list_sb[mine + __builtin_popcountll(sbm[j] & ((1ull << lane) - 1))] = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
	}
}

__global__ __launch_bounds__(256)
void k_masks2(const uint32_t *__restrict__ chan_code, uint32_t nchan, const uint32_t *__restrict__ tbl, uint32_t *__restrict__ masks)
{
	/* as k_masks: a wavefront keeps the linear-form masks of its 18 x 64 output bits in registers and builds, of 64 entries
	 * at a time, the ones in use: entry 0, the channel carry-ins, the occupied slots of the code table */
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t half = lane >> 5, bit = lane & 31;
	const uint32_t nent = 1 + nchan + TG_LB_TBL;
	uint32_t lin[TG_MW_ROUNDS];
#pragma unroll
	for (int r = 0; r < TG_MW_ROUNDS; r++) {
		const uint16_t pos = c_tab.mask_pos[2 * r + half][bit];
		lin[r] = (pos != 0xffff) ? c_tab.lfsr_lin[pos] : 0u;
	}
	for (uint32_t e0 = wave * 64; e0 < nent; e0 += nwaves * 64) {
		const uint32_t e = e0 + lane;
		uint32_t mycode = 0;
		bool need = e < nent;
		if (e >= 1 && e <= nchan)
			mycode = chan_code[e - 1];
		else if (e > nchan && e < nent) {
			mycode = tbl[e - 1 - nchan];
			need = mycode != 0;
		}
		unsigned long long todo = __ballot(need);
		while (todo) {
			const uint32_t l = (uint32_t)__builtin_ctzll(todo);
			todo &= todo - 1;
			const uint32_t code = __builtin_amdgcn_readlane(mycode, l);
			uint32_t myword = 0;
#pragma unroll
			for (int r = 0; r < TG_MW_ROUNDS; r++) {
				const unsigned long long bal = __ballot(__popc(code & lin[r]) & 1);
				myword = (lane == (uint32_t)(2 * r)) ? (uint32_t)bal : myword;
				myword = (lane == (uint32_t)(2 * r + 1)) ? (uint32_t)(bal >> 32) : myword;
			}
			if (lane == TG_MW_CODE)
				myword = code;
			if (lane < TG_MASK_WORDS)
				masks[(size_t)(e0 + l) * TG_MASK_WORDS + lane] = myword;
		}
	}
}

/* latest delivered SYNC slot with a good SB1 at or before grid slot g in g's channel, or 0xffffffff */
__device__ __forceinline__ uint32_t tg_lb_find(uint32_t g, const uint32_t *__restrict__ okbits, const uint32_t *__restrict__ dbits,
					       const uint32_t *__restrict__ prevw, const uint8_t *__restrict__ word_chan)
{
	const uint32_t w = g >> 5;
	uint32_t m = okbits[w] & dbits[w] & (0xffffffffu >> (31 - (g & 31)));
	uint32_t ww = w;
	if (!m) {
		if (!w)
			return 0xffffffffu;
		const uint32_t p = prevw[w - 1];	/* 1 + the latest word <= w - 1 that has one, in that word's channel; 0: none */
		if (!p || word_chan[p - 1] != word_chan[w])
			return 0xffffffffu;
		ww = p - 1;
		m = okbits[ww] & dbits[ww];
	}
	return 32 * ww + 31 - __builtin_clz(m);
}

#define LBS_THREADS 1024
__global__ __launch_bounds__(LBS_THREADS)
void k_lb_scan(const uint32_t *__restrict__ okbits, const uint32_t *__restrict__ dbits, const uint8_t *__restrict__ word_chan,
	       uint32_t nwords, uint32_t *__restrict__ prevw, const tg_chan_ent *__restrict__ chan, uint32_t nchan,
	       const uint32_t *__restrict__ chan_code, const uint32_t *__restrict__ slot_entry, const uint32_t *__restrict__ masks,
	       const uint32_t *__restrict__ tbl, uint32_t *__restrict__ final_code)
{
	/* one workgroup per channel (a channel's words never look into another's): running maximum of (word + 1 if the word has
	 * a delivered good SYNC slot, else 0) over the channel's words, 1024 consecutive words per pass (coalesced loads, the
	 * next pass's in flight during the scan) */
	__shared__ uint32_t sm[LBS_THREADS / 64];
	const uint32_t c = blockIdx.x;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t w0 = chan[c].gbase >> 5, wn = (chan[c].ncls + 31) >> 5;
	(void)nwords;
	uint32_t carry = 0;
	uint32_t w = threadIdx.x;
	uint32_t has = w < wn ? (okbits[w0 + w] & dbits[w0 + w]) : 0u;
	for (uint32_t base = 0; base < wn; base += LBS_THREADS) {
		const uint32_t wnext = base + LBS_THREADS + threadIdx.x;
		const uint32_t hnext = wnext < wn ? (okbits[w0 + wnext] & dbits[w0 + wnext]) : 0u;
		uint32_t inc = has ? w0 + w + 1 : 0u;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t o = __shfl_up(inc, d);
			if (lane >= (uint32_t)d && o > inc)
				inc = o;
		}
		__syncthreads();
		if (lane == 63)
			sm[wv] = inc;
		__syncthreads();
		uint32_t pre = carry, tot = carry;
		for (uint32_t q = 0; q < LBS_THREADS / 64; q++) {
			if (q < wv)
				pre = sm[q] > pre ? sm[q] : pre;
			tot = sm[q] > tot ? sm[q] : tot;
		}
		if (w < wn)
			prevw[w0 + w] = inc > pre ? inc : pre;
		carry = tot;
		w = wnext;
		has = hnext;
	}
	(void)word_chan;
	/* the code in force after the batch: the channel's latest delivered good SYNC slot, else its carry-in */
	if (threadIdx.x == 0) {
		uint32_t code = chan_code[c];
		if (carry) {
			const uint32_t ww = carry - 1;
			const uint32_t m = okbits[ww] & dbits[ww];
			code = masks[(size_t)slot_entry[32 * ww + 31 - __builtin_clz(m)] * TG_MASK_WORDS + TG_MW_CODE];
		}
		final_code[c] = code;
		if (c == 0)
			final_code[64] = tbl[TG_LB_TBL];	/* != 0: the batch had more codes than the table holds */
	}
}

__global__ __launch_bounds__(1024)
void k_lists2(const uint32_t *__restrict__ cls, const uint32_t *__restrict__ dbits, uint32_t n, const uint32_t *__restrict__ okbits,
	      const uint32_t *__restrict__ prevw, const uint8_t *__restrict__ word_chan, const uint32_t *__restrict__ slot_entry,
	      uint32_t *__restrict__ maskidx, uint32_t *__restrict__ list_216, uint32_t *__restrict__ list_432,
	      uint32_t *__restrict__ cnt /* [1]: 216 items, [2]: 432 items */)
{
	__shared__ uint32_t s_c216[TG_MID_CHUNKS][16], s_c432[TG_MID_CHUNKS][16], s_b216, s_b432;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long below = (1ull << lane) - 1;
	uint32_t t[TG_MID_CHUNKS], in216[TG_MID_CHUNKS], in432[TG_MID_CHUNKS];
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		const uint32_t g = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
		t[j] = TG_BURST_NONE;
		if (g < n && ((dbits[g >> 5] >> (g & 31)) & 1))
			t[j] = cls[g] & 0xff;
		if (t[j] != TG_BURST_NONE) {
			const uint32_t gs = tg_lb_find(g, okbits, dbits, prevw, word_chan);
			maskidx[g] = gs != 0xffffffffu ? slot_entry[gs] : 1u + word_chan[g >> 5];
		}
		const unsigned long long msb = __ballot(t[j] == TG_BURST_SYNC), mn2 = __ballot(t[j] == TG_BURST_NORM_2);
		const unsigned long long mn1 = __ballot(t[j] == TG_BURST_NORM_1);
		in216[j] = __builtin_popcountll(msb & below) + 2 * __builtin_popcountll(mn2 & below);
		in432[j] = __builtin_popcountll(mn1 & below);
		if (lane == 0) {
			s_c216[j][wv] = __builtin_popcountll(msb) + 2 * __builtin_popcountll(mn2);
			s_c432[j][wv] = __builtin_popcountll(mn1);
		}
	}
	__syncthreads();
	if (threadIdx.x < 2) {		/* one atomic per workgroup and list */
		uint32_t tot = 0;
		for (int q = 0; q < TG_MID_CHUNKS * 16; q++)
			tot += threadIdx.x ? s_c432[0][q] : s_c216[0][q];
		const uint32_t base = tot ? atomicAdd(cnt + 1 + threadIdx.x, tot) : 0u;
		if (threadIdx.x)
			s_b432 = base;
		else
			s_b216 = base;
	}
	__syncthreads();
	uint32_t r216 = s_b216, r432 = s_b432;
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		const uint32_t g = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
		uint32_t p216 = r216, p432 = r432;
		for (uint32_t q = 0; q < 16; q++) {
			if (q < wv) {
				p216 += s_c216[j][q];
				p432 += s_c432[j][q];
			}
			r216 += s_c216[j][q];
			r432 += s_c432[j][q];
		}
		p216 += in216[j];
		p432 += in432[j];
		if (t[j] == TG_BURST_SYNC)
			list_216[p216] = (g << 1) | 1;		/* SB2 */
		else if (t[j] == TG_BURST_NORM_2) {
			list_216[p216] = g << 1;
			list_216[p216 + 1] = (g << 1) | 1;
		} else if (t[j] == TG_BURST_NORM_1)
			list_432[p432] = g;
	}
}

extern "C" int tgk_cls_plain2(const uint32_t *d_cls, uint32_t n, uint32_t *d_plain, uint32_t *d_list_sb, uint32_t *d_cnt_sb,
			      uint8_t *d_word_chan, const struct tg_chan_ent *d_chan, uint32_t nchan, void *stream)
{
	if (!n)
		return 0;
	hipLaunchKernelGGL(k_cls_plain2, dim3((n + 1024 * TG_MID_CHUNKS - 1) / (1024 * TG_MID_CHUNKS)), dim3(1024), 0, (hipStream_t)stream, d_cls, n, d_plain, d_list_sb, d_cnt_sb,
			   d_word_chan, d_chan, nchan);
	return (int)hipGetLastError();
}

extern "C" int tgk_masks2(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_tbl, uint32_t *d_masks, void *stream)
{
	const uint32_t nent = 1 + nchan + TG_LB_TBL;
	hipLaunchKernelGGL(k_masks2, dim3(((nent + 63) / 64 + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_chan_code, nchan, d_tbl, d_masks);
	return (int)hipGetLastError();
}

extern "C" int tgk_lb_scan(const uint32_t *d_okbits, const uint32_t *d_dbits, const uint8_t *d_word_chan, uint32_t nwords,
			   uint32_t *d_prevw, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_chan_code,
			   const uint32_t *d_slot_entry, const uint32_t *d_masks, const uint32_t *d_tbl, uint32_t *d_final_code, void *stream)
{
	if (!nwords)
		return 0;
	hipLaunchKernelGGL(k_lb_scan, dim3(nchan), dim3(LBS_THREADS), 0, (hipStream_t)stream, d_okbits, d_dbits, d_word_chan, nwords, d_prevw,
			   d_chan, nchan, d_chan_code, d_slot_entry, d_masks, d_tbl, d_final_code);
	return (int)hipGetLastError();
}

extern "C" int tgk_lists2(const uint32_t *d_cls, const uint32_t *d_dbits, uint32_t n, const uint32_t *d_okbits, const uint32_t *d_prevw,
			  const uint8_t *d_word_chan, const uint32_t *d_slot_entry, uint32_t *d_maskidx, uint32_t *d_list_216,
			  uint32_t *d_list_432, uint32_t *d_cnt, void *stream)
{
	if (!n)
		return 0;
	hipLaunchKernelGGL(k_lists2, dim3((n + 1024 * TG_MID_CHUNKS - 1) / (1024 * TG_MID_CHUNKS)), dim3(1024), 0, (hipStream_t)stream, d_cls, d_dbits, n, d_okbits, d_prevw,
			   d_word_chan, d_slot_entry, d_maskidx, d_list_216, d_list_432, d_cnt);
	return (int)hipGetLastError();
}
