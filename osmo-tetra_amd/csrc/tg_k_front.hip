/*
 * tg_k_front.hip -- the front ends: k_front, k_front_blocks, k_front_stream (+ _fix), k_float_to_bits(_afc), k_front_soft
 * (one of the four HIP units of the library: tg_dev.h has the map)
 */
#include "tg_dev.h"

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
#ifndef TGS_ST_AUX
#define TGS_ST_AUX 0
#endif
	__builtin_amdgcn_raw_buffer_store_b32(mo[lane], out, lane * 4, 0, TGS_ST_AUX);
	__builtin_amdgcn_raw_buffer_store_b32(mo[64 + lane], out, 256 + lane * 4, 0, TGS_ST_AUX);
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

#define STREAM_SLOT_TABLES(VIEWB) STREAM_SLOT_TABLES_W(VIEWB, 4)
#define STREAM_SLOT_TABLES_W(VIEWB, NW)									\
	constexpr int WINDW = (VIEWB) / 4 + 4;	/* the view's dwords + one zero pad row */			\
	__shared__ uint32_t s_slot[NW][WINDW];								\
	const uint32_t lane = threadIdx.x & 63;								\
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);				\
	const uint32_t wave = blockIdx.x * (NW) + wib;							\
	const uint32_t nwaves = gridDim.x * (NW);							\
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

/* second pass of the packed-bit front end: every slot the first pass deferred goes through the exact per-position search.
 * Round 5: the first pass keeps one list PER WAVE -- defer[w] = how many slots wave w of k_front_stream deferred, its
 * slots from defer[TG_DEFER_L0(fw) + w * capw] (fw = waves of that launch, capw = slots a wave can meet at most) -- and
 * appends with a counter of its own.  (Rounds 2-4 had one list and one counter for the launch: an atomicAdd with a
 * return value, at device scope -- on this part that is a trip to the memory side of the fabric, every wave's goes to the
 * same address, and the wait for its answer also waits for the wave's prefetched group: 15 of the kernel's 155 us at
 * 1 % deferred slots, tools/experiments/front_ablate.sh "-DTGS_ABLATE=64".)  A wave of k_front_stream takes groups wave, wave +
 * nwaves, ..., so every list is an even sample of the grid whatever the damage looks like; a wave of this kernel takes the
 * lists w = wave, wave + nwaves, ... one entry at a time. */
#define TG_DEFER_L0(fw) (((fw) + 15u) & ~15u)
#ifndef TG_FIX_LISTS
#define TG_FIX_LISTS 4
#endif
#ifndef TG_FIX_WAVES
#define TG_FIX_WAVES 4	/* waves per workgroup of k_front_stream_fix: they share the workgroup's lists */
#endif
template <bool PACKED, int VIEWT>
__global__ __launch_bounds__(64 * TG_FIX_WAVES)
void k_front_stream_fix(const uint8_t *__restrict__ stream, tg_stream_params prm,
			uint32_t *__restrict__ packed, uint32_t *__restrict__ cls, uint16_t *__restrict__ ysum,
			const uint32_t *__restrict__ defer, uint32_t fw, uint32_t capw)
{
	STREAM_SLOT_TABLES_W(VIEWT, TG_FIX_WAVES)
	(void)wave;
	(void)nwaves;
#if TGS_DEFER_ATOMIC	/* (A/B builds: the rounds 2-4 form, one list for the launch, its counter in the last word of the scratch) */
	for (uint32_t w = 0; w < 1; w++) {
		const uint32_t count = defer[TG_DEFER_L0(fw) + (size_t)fw * capw];
		const uint32_t *list = defer + TG_DEFER_L0(fw);
		for (uint32_t e = wave; e < count; e += nwaves) {
#else
	/* a workgroup takes TG_FIX_LISTS neighbouring lists at a time and deals their entries round its TG_FIX_WAVES waves: a list
	 * holds a handful of slots at most (0.8 on average at 1 % deferred) and the kernel is as long as its longest queue.  The
	 * single list of rounds 2-4 gave every wave one or two entries (18 us per 1 M slots with the round-4 front end).  Measured
	 * with this round's front end: 4 lists / 4 waves 26-28 us, 8 / 8 the same, 16 / 16 29-30, 2 / 4 28-29, 1 / 4 35-37 -- and the
	 * single atomic list 33: the kernel's time is no longer a matter of how its entries are dealt (open) */
	for (uint32_t w4 = TG_FIX_LISTS * blockIdx.x; w4 < fw; w4 += TG_FIX_LISTS * gridDim.x) {
		uint32_t cum[TG_FIX_LISTS + 1];
		cum[0] = 0;
#pragma unroll
		for (int q = 0; q < TG_FIX_LISTS; q++)
			cum[q + 1] = cum[q] + (w4 + q < fw ? defer[w4 + q] : 0u);
		for (uint32_t e4 = wib; e4 < cum[TG_FIX_LISTS]; e4 += TG_FIX_WAVES) {
			uint32_t q = 0;
#pragma unroll
			for (int k = 1; k < TG_FIX_LISTS; k++)
				q += e4 >= cum[k];
			uint32_t base = 0;
#pragma unroll
			for (int k = 1; k < TG_FIX_LISTS; k++)
				base = q == (uint32_t)k ? cum[k] : base;
			const uint32_t *list = defer + TG_DEFER_L0(fw) + (size_t)(w4 + q) * capw;
			const uint32_t e = e4 - base;
#endif
			const uint32_t slot = list[e];
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

/* the same in two halves (round 5, TGS_GPIPE): the eight reads of a slot are issued, and taken one slot later -- the next
 * slot's reads are in flight behind them, so a group pays two exposed LDS round trips for its four gathers, not four.
 * LDS answers in order: "my byte i is here" = at most NEWER + 7 - i younger reads outstanding, NEWER = the eight reads of
 * the slot issued in between (every slot issues exactly eight: one the kernel does not decode reads the zero word).  Reads
 * the compiler puts in between only make the waits longer than needed. */
template <int KOFF, int X>
__device__ __forceinline__ void front_gather_issue(const uint32_t (&a)[8], uint32_t (&t)[8])
{
	asm volatile("; gather issue %16\n\t"
		     "ds_read_u8 %0, %8 offset:%17\n\tds_read_u8 %1, %9 offset:%17\n\tds_read_u8 %2, %10 offset:%17\n\t"
		     "ds_read_u8 %3, %11 offset:%17\n\tds_read_u8 %4, %12 offset:%17\n\tds_read_u8 %5, %13 offset:%17\n\t"
		     "ds_read_u8 %6, %14 offset:%17\n\tds_read_u8 %7, %15 offset:%17"
		     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
		     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "n"(X), "n"(KOFF)
		     : "memory");
}

template <int NEWER>
__device__ __forceinline__ uint32_t front_gather_take(uint32_t (&t)[8])
{
	uint32_t acc;
	asm volatile("; gather take\n\t"
		     "s_waitcnt lgkmcnt(%9)\n\tv_lshlrev_b32 %0, 31, %1\n\t"
		     "s_waitcnt lgkmcnt(%10)\n\tv_alignbit_b32 %0, %2, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(%11)\n\tv_alignbit_b32 %0, %3, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(%12)\n\tv_alignbit_b32 %0, %4, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(%13)\n\tv_alignbit_b32 %0, %5, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(%14)\n\tv_alignbit_b32 %0, %6, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(%15)\n\tv_alignbit_b32 %0, %7, %0, 1\n\t"
		     "s_waitcnt lgkmcnt(%16)\n\tv_alignbit_b32 %0, %8, %0, 1\n\t"
		     "v_lshrrev_b32 %0, 24, %0"
		     : "=&v"(acc), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])
		     : "n"(NEWER + 7), "n"(NEWER + 6), "n"(NEWER + 5), "n"(NEWER + 4), "n"(NEWER + 3), "n"(NEWER + 2), "n"(NEWER + 1), "n"(NEWER + 0)
		     : "memory");
	return acc;
}

struct tg_group_data {
	uint4 a, b, c;	/* bytes 16 l .., 1024 + 16 l .., 2048 + 16 min(l, 7) .. of the group's aligned range */
	uint32_t a0;	/* the group starts a0 bytes into that range */
	uint32_t touch;	/* (TGS_TOUCH: one dword per 128-byte line of the group TGS_TOUCH rounds further on -- requested, never used) */
	bool fast;	/* all four windows of the group lie inside the stream */
};

#ifndef TGS_SYNC_LDS
#define TGS_SYNC_LDS 1	/* the SYNC burst's gather addresses wait in LDS, not in registers */
#endif
#ifndef TGS_LOAD_NT
#define TGS_LOAD_NT 0
#endif
#ifndef TGS_TOUCH
#define TGS_TOUCH 0	/* n > 0: every fetch also touches the lines of the group n rounds further on (one dword per 128-byte line) */
#endif
#ifndef TGS_GPIPE
#define TGS_GPIPE 0	/* 1: a slot's gather reads are issued one slot ahead of their use (front_gather_issue / _take) */
#endif
#ifndef TGS_DEFER_ATOMIC
#define TGS_DEFER_ATOMIC 0	/* A/B builds only: 1 = one deferred-slot list per launch, appended to with an atomicAdd (rounds 2-4) */
#endif
#ifndef TGS_PLAIN
#define TGS_PLAIN 1	/* 1: search and outcome verify "one sequence, at its place, nothing else" and hand everything else to the exact pass
			 * (round 5); 0: the round-3/4 form (first hit, SYNC summary and the rule's inputs for every slot) -- same outputs */
#endif
#ifndef TG_STREAM_WPE
#define TG_STREAM_WPE 5	/* waves per SIMD.  Round 5: five (96 VGPRs allowed, 86 used).  The kernel had sat exactly at the 80 VGPRs of six waves
			 * since round 3; taking the atomicAdd of the deferred-slot list out (above) let the scheduler reorder across that
			 * point and the same source needed 91: eleven spills to scratch, reloaded in front of every gather (164 us at six
			 * waves with spills, 135 at five without, 148 at four; tools/experiments/front_ablate.sh).  Rounds 3-4, with the grid at two
			 * rounds of resident workgroups (launch_stream_front): 4 -> 161-168 us per 1 M slots, 5 -> 156-161, 6 -> 155-159,
			 * 8 (64 VGPRs, spills) -> 195-200 */
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
#define TGS_ABLATE 0	/* measurement builds only (tools/experiments/front_ablate.sh): 1 no stores, 2 every group from one address, 4 no
			 * gathers, 8 no search and no classification, 16 no shifted copies, 32 no classification, 64 no atomic for the deferred slots,
			 * 128 classification kept but the gather always NORM_1's, 256 no classification but the gather's type varies -- the kernel's
			 * results are wrong with any of them */
#endif
#ifdef TGS_TIMING
/* measurement build: reference-clock ticks (s_memtime, 100 MHz) a wave spends between the marks of a group, summed over
 * all waves; tools/experiments/front_phases.py */
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
	/* per wave: the SYNC burst's eight gather addresses of every lane.  One slot in eight is a SYNC burst: its table waits
	 * here (two 16-byte reads in front of such a gather) instead of in eight of the 80 VGPRs six waves per SIMD allow */
	__shared__ __attribute__((aligned(16))) uint32_t s_sadr[TG_STREAM_WPB][TGS_SYNC_LDS ? 64 * 8 : 4];

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
#if TGS_SYNC_LDS
		*(uint4 *)&s_sadr[wib][8 * lane] = make_uint4(g_adr[2][0], g_adr[2][1], g_adr[2][2], g_adr[2][3]);
		*(uint4 *)&s_sadr[wib][8 * lane + 4] = make_uint4(g_adr[2][4], g_adr[2][5], g_adr[2][6], g_adr[2][7]);
#endif
#pragma unroll
		for (int x = 0; x < (TGS_SYNC_LDS ? 2 : 3); x++)
#pragma unroll
			for (int r = 0; r < 8; r++)
				asm volatile("" : "+v"(g_adr[x][r]));	/* the whole address in the register: the slot's offset is the immediate */
	}
	uint32_t zadr = ver0 + wib * (4 * TG_VER_SLOT * 4) + 64u;	/* the zero word of slot 0's window, as the asm blocks address LDS */
	asm volatile("" : "+v"(zadr));
	if (lane < 4)
		win[lane * TG_VER_SLOT + 16] = 0;	/* "no source" reads this */
	for (int i = lane; i < 128; i += 64)
		mo[i] = 0;				/* bytes of the staged slots that nobody owns stay zero */
	/* which positions of the lane's column count: main search 21..472, "early" 0..20, SYNC summary 0..509 */
	const uint32_t vmain = (col == 0) ? 0xffe00000u : (col == 14) ? 0x01ffffffu : (col == 15) ? 0u : 0xffffffffu;
	const uint32_t vearly = (col == 0) ? 0x001fffffu : 0u;
	const uint32_t vys = (col == 15) ? 0x3fffffffu : 0xffffffffu;
	const uint32_t pos0 = (lane >> 4) * TG_SLOT_BITS + 32 * col;	/* first bit of the column inside the group */
#if TGS_PLAIN
	/* round 5, the "plain slot" form of search and outcome.  What this kernel may settle on its own is a slot that holds
	 * exactly ONE training sequence, of a downlink type, at its nominal offset (y at 214 = column 6 bit 22, n / p at 244 =
	 * column 7 bit 20) -- 99 % of a recording.  So it only has to VERIFY that: the expected hit is there, and nothing else
	 * is: no n / p at any other position 0..472, no y anywhere in the slot.  "No y" is checked on y's first 22 bits (a
	 * necessary condition: the three sequences then share 21 shifted copies of the string instead of 37) and the one
	 * expected y is confirmed on its last 16.  Every other slot -- a damaged or misplaced sequence, a second hit, a payload
	 * coincidence (3e-4 of the slots), anything below offset 21 -- goes to k_front_stream_fix, which evaluates
	 * tetra_find_train_seq()'s rule position by position as before.  The words this kernel does write are the exact
	 * pass's words for the same slot (test_stream_front_packed_bits_equals_per_position). */
	const uint32_t m_enp = (col == 7) ? (1u << 20) : 0u;			/* the expected n / p hit */
	const uint32_t m_ey = (col == 6) ? (1u << 22) : 0u;			/* the expected y hit */
	const uint32_t c_np = (vmain | vearly) & ~m_enp;			/* n / p hits that are not the expected one */
	const uint32_t c_y = vys & ~m_ey;					/* y (prefix) hits that are not the expected one */
#endif

	const uint32_t ngroups = (prm.nslots + 3) >> 2;
	if (wave >= ngroups) {
		if (lane == 0)
			defer[wave] = 0;
		return;
	}
	/* this wave's list of slots for the exact pass (k_front_stream_fix) and how many are on it */
	/* (wave-uniform, and kept in scalar registers by hand: the kernel sits at the 80 VGPRs that six waves per SIMD allow) */
	uint32_t dpos = __builtin_amdgcn_readfirstlane(TG_DEFER_L0(nwaves) + wave * (4u * ((ngroups + nwaves - 1) / nwaves)));
	const uint32_t dpos0 = dpos;

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
#if TGS_TOUCH
		{	/* pull the lines of the group this wave takes TGS_TOUCH rounds after the one being fetched towards the L2 (a cold capture:
			 * DRAM page misses, translations), one dword per line, as long as that group lies in the same channel's bytes */
			const uint64_t adv = (uint64_t)TGS_TOUCH * nwaves * TG_GROUP_BYTES;
			const bool ahead = d.fast && (prm.nchan ? (gb - first) + adv + TG_GROUP_LOAD + 256 <= cspan : gb + adv + TG_GROUP_LOAD + 256 <= prm.len);
			d.touch = 0;
			if (ahead && lane < 18)
				d.touch = *(const volatile uint32_t *)(base16 + adv + 128 * lane);
		}
#endif
#if TGS_LOAD_NT	/* (A/B: the capture is read once -- non-temporal loads) */
		typedef uint32_t tgs_u4v __attribute__((ext_vector_type(4)));
		{
			const tgs_u4v va = __builtin_nontemporal_load((const tgs_u4v *)(base16 + 16 * lane));
			const tgs_u4v vb = __builtin_nontemporal_load((const tgs_u4v *)(base16 + 1024 + 16 * lane));
			const tgs_u4v vc = __builtin_nontemporal_load((const tgs_u4v *)(base16 + 2048 + 16 * (lane < 7 ? lane : 7)));
			d.a = make_uint4(va.x, va.y, va.z, va.w);
			d.b = make_uint4(vb.x, vb.y, vb.z, vb.w);
			d.c = make_uint4(vc.x, vc.y, vc.z, vc.w);
		}
#else
		d.a = *(const uint4 *)(base16 + 16 * lane);
		d.b = *(const uint4 *)(base16 + 1024 + 16 * lane);
		d.c = *(const uint4 *)(base16 + 2048 + 16 * (lane < 7 ? lane : 7));
#endif
	};

	auto work = [&](uint32_t g, const tg_group_data &cur) {
#if TGS_TOUCH
		asm volatile("" :: "v"(cur.touch));	/* (the touch load's register stays its own until the group's own bytes are here) */
#endif
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
#if TGS_PLAIN
		uint32_t my = 0xffffffffu, mn = 0xffffffffu, mp = 0xffffffffu;	/* (my: the first 22 bits of y only) */
#pragma unroll
		for (int j = 0; j < 22; j += 2) {
			const uint32_t t0 = (j == 0) ? W0 : __builtin_amdgcn_alignbit(W1, W0, j);
			const int k = j + 1;
			const uint32_t t1 = __builtin_amdgcn_alignbit(W1, W0, k);
#define TSQ_STEP(acc, P) acc = tsq_and2(acc, t0, t1, 2 * (int)(((P) >> j) & 1) + (int)(((P) >> k) & 1))
			TSQ_STEP(my, PY);
			TSQ_STEP(mn, PN);
			TSQ_STEP(mp, PP);
#undef TSQ_STEP
		}
		const uint32_t any = my | mn | mp;
		(void)W2;
#else
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

#endif
		TGS_MARK(3);	/* shifted copies stored, match masks, ballots */
#if !(TGS_ABLATE & (8 | 32 | 256))
		/* per slot (= 16-lane row): the first hit and the SYNC summary by reductions inside the row -- every lane makes a
		 * key of its own first hit ((position << 2 | type) in the high half, first y position in the low half: one
		 * v_pk_min_u16 reduces both) and a count word (hit below 21 in the high half, number of y hits in the low), four
		 * rotate-and-combine steps (DPP row_ror 8 4 2 1) leave the row's result in all of its lanes.  Vector
		 * instructions only: the form with ballots, per-lane shifts of them and the LDS crossbar cost 24 us per 1 M
		 * slots in round trips between the vector unit, scalar registers and LDS (TGS_ABLATE), this one */
#if TGS_PLAIN
		/* per lane: "something that is not the expected hit" (bit 23) and the expected hits it holds (y 22, n 21, p 20); OR over
		 * the slot's 16-lane row in four DPP steps; the row's four bits decide: exactly one expected hit and nothing else,
		 * or the slot is the exact pass's */
		(void)any;
		uint32_t rest = __builtin_amdgcn_bitop3_b32(mn, mp, c_np, 0xa8);		/* (mn | mp) & c_np */
		rest = __builtin_amdgcn_bitop3_b32(my, c_y, rest, 0xea);			/* (my & c_y) | rest */
		/* y's last 16 bits behind the expected prefix hit: positions 236..251 = bits 12..27 of column 6's second word */
		const bool ytail = ((W1 >> 12) & 0xffffu) == (uint32_t)((PY >> 22) & 0xffffu);
		uint32_t ex = ((mn & m_enp) << 1) | (mp & m_enp);
		ex = (my & (ytail ? m_ey : 0u)) | ex;
		uint32_t rowc = ((rest != 0u ? 1u : 0u) << 23) | ex;
#define ROW_STEP(CTRL) rowc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rowc, (CTRL), 0xf, 0xf, true);
		ROW_STEP(0x128)	/* row_ror:8 */
		ROW_STEP(0x124)
		ROW_STEP(0x122)
		ROW_STEP(0x121)
#undef ROW_STEP
		/* 0b0001 p alone -> NORM_2, 0b0010 n alone -> NORM_1, 0b0100 y alone -> SYNC; anything else: not this kernel's slot */
		const uint32_t kk = rowc >> 20;
		const uint32_t rc = (0xfff3f01fu >> (4u * (kk < 8u ? kk : 7u))) & 0xfu;
		const bool dfr = defer_all || rc == 0xfu;
		const uint32_t offs = rc == TG_BURST_SYNC ? (uint32_t)TG_SYNC_TRAIN_OFF : (uint32_t)TG_NORM_TRAIN_OFF;
		const uint32_t dtype = dfr ? (uint32_t)TG_BURST_NONE : rc;
		uint32_t ys = rc == TG_BURST_SYNC ? (uint32_t)TG_SYNC_TRAIN_OFF : (uint32_t)TG_YS_NONE;
#else
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
#endif
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
			else if (dt == TG_BURST_SYNC) {									\
				if (TGS_SYNC_LDS) {										\
					const uint4 a0_ = *(const uint4 *)&s_sadr[wib][8 * lane], a1_ = *(const uint4 *)&s_sadr[wib][8 * lane + 4];	\
					const uint32_t sa_[8] = { a0_.x, a0_.y, a0_.z, a0_.w, a1_.x, a1_.y, a1_.z, a1_.w };	\
					mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 2>(sa_);			\
				} else												\
					mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 2>(g_adr[2]);		\
			}													\
			((uint8_t *)mo)[(K) * TG_PACKED_WORDS * 4 + obyte] = (uint8_t)mybyte;				\
		}
		TGS_MARK(4);	/* classification of the four slots */
#if TGS_GPIPE && !TGS_ABLATE
#define GP_ISSUE(K, T)													\
		{													\
			const uint32_t dt = (uint32_t)__builtin_amdgcn_readlane(dtype, CLS_LANE_OF(K));		\
			if (dt == TG_BURST_NORM_1)										\
				front_gather_issue<4 * TG_VER_SLOT * (K), 0>(g_adr[0], T);				\
			else if (dt == TG_BURST_NORM_2)									\
				front_gather_issue<4 * TG_VER_SLOT * (K), 1>(g_adr[1], T);				\
			else if (dt == TG_BURST_SYNC)									\
				front_gather_issue<4 * TG_VER_SLOT * (K), 2>(g_adr[2], T);				\
			else		/* not this kernel's slot: eight reads of the zero word (the waits count on eight) */	\
				front_gather_issue<0, 3>(g_zero, T);							\
		}
#define GP_TAKE(K, T, NEWER) ((uint8_t *)mo)[(K) * TG_PACKED_WORDS * 4 + obyte] = (uint8_t)front_gather_take<NEWER>(T);
		{
			uint32_t tA[8], tB[8];
			const uint32_t g_zero[8] = { zadr, zadr, zadr, zadr, zadr, zadr, zadr, zadr };
			GP_ISSUE(0, tA)
			GP_ISSUE(1, tB)
			GP_TAKE(0, tA, 8)
			GP_ISSUE(2, tA)
			GP_TAKE(1, tB, 8)
			GP_ISSUE(3, tB)
			GP_TAKE(2, tA, 8)
			GP_TAKE(3, tB, 0)
		}
#undef GP_ISSUE
#undef GP_TAKE
#else
		STREAM_SLOT_K(0)
		STREAM_SLOT_K(1)
		STREAM_SLOT_K(2)
		STREAM_SLOT_K(3)
#endif
#undef STREAM_SLOT_K
		TGS_MARK(5);	/* the four gathers */
		if (CLS_OWNER) {
			mo[CLS_SLOT * TG_PACKED_WORDS + TG_PW_META] = meta;
			mo[80 + CLS_SLOT] = clsword;
			mo[84 + CLS_SLOT] = ys;
		}
		{	/* slots this pass could not settle: onto this wave's list for k_front_stream_fix */
#if TGS_SB & 1
			__builtin_amdgcn_sched_barrier(0);
#endif
#if TGS_SB & 2
			asm volatile("" ::: "memory");
#endif
			const bool mine = CLS_OWNER && CLS_SLOT < cnt && dfr;
			const unsigned long long dm = __ballot(mine);
			if (dm) {
#if TGS_DEFER_ATOMIC
				uint32_t pos = 0;
				if (lane == 0)
					pos = atomicAdd(defer + TG_DEFER_L0(nwaves) + (size_t)nwaves * capw, (uint32_t)__builtin_popcountll(dm));
				pos = __builtin_amdgcn_readfirstlane(pos);
				if (mine)
					defer[TG_DEFER_L0(nwaves) + pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u))] = first + CLS_SLOT;
#else
				if (mine)
					defer[dpos + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u))] = first + CLS_SLOT;
				if (!(TGS_ABLATE & 64))
					dpos = __builtin_amdgcn_readfirstlane(dpos + (uint32_t)__builtin_popcountll(dm));
#endif
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
#if TGS_SB & 4
#define TGS_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define TGS_FENCE do { } while (0)
#endif
	fetch(g, dA);
	for (;;) {
		const uint32_t gB = g + nwaves;
		TGS_FENCE;
		fetch(gB < ngroups ? gB : g, dB);
		TGS_FENCE;
		work(g, dA);
		if (gB >= ngroups)
			break;
		const uint32_t gA = gB + nwaves;
		TGS_FENCE;
		fetch(gA < ngroups ? gA : gB, dA);
		TGS_FENCE;
		work(gB, dB);
		if (gA >= ngroups)
			break;
		g = gA;
	}
#undef TGS_FENCE
	if (lane == 0)
		defer[wave] = dpos - dpos0;
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
/* host-side launch layer of this unit                                        */
/* ------------------------------------------------------------------------- */
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

extern "C" int tgk_upload_front(const tg_const_tables *host)
{
	static tg_soft_tables shost;
	build_soft_tables(&shost);
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_soft_tab), &shost, sizeof(shost)));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), host, sizeof(*host)));
	return 0;
}

#ifdef TG_TRACE
extern "C" int tgk_trace_read_front(void *out, unsigned int *n, int reset)
{
	return tg_trace_read_unit(out, n, reset);
}
#endif

extern "C" int tgk_front(const uint8_t *d_stream, const uint64_t *d_slot_desc,
			 uint32_t nslots, uint32_t *d_packed, uint8_t *d_rec, void *stream)
{
	if (!nslots)
		return 0;
	uint32_t blocks = (nslots + 3) / 4;
	uint32_t cap = 256 * 32;	/* measured on MI355X (tools/experiments/exp_front_grid.py): 2048 156 us, 4096 154 us, 8192 144 us */
	if (tgi_option(TGPU_OPT_FRONT_BLOCKS) > 0)
		cap = (uint32_t)tgi_option(TGPU_OPT_FRONT_BLOCKS);
	if (blocks > cap)
		blocks = cap;
	hipLaunchKernelGGL(k_front, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
			   d_stream, d_slot_desc, nslots, d_packed, d_rec);
	return (int)hipGetLastError();
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

/* both passes of the packed-bit front end; d_defer: scratch of TG_DEFER_WORDS(nslots) dwords (a count and a list per wave of the first pass) */
/* per-kernel timing: an event to be recorded right in front of the next k_front_stream launch of this thread */
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
#if TGS_DEFER_ATOMIC
	{
		const uint32_t fw_ = (blocks * (4 / TG_STREAM_WPB)) * TG_STREAM_WPB, fg_ = (nslots + 3) / 4, cw_ = 4u * ((fg_ + fw_ - 1) / fw_);
		HIPCHK(hipMemsetAsync(d_defer + TG_DEFER_L0(fw_) + (size_t)fw_ * cw_, 0, 4, s));
	}
#endif
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
#if !TGS_DEFER_ATOMIC
	fblocks = (fgrid.x * TG_STREAM_WPB + TG_FIX_LISTS - 1) / TG_FIX_LISTS;	/* a workgroup per TG_FIX_LISTS lists of the first pass */
	if (!fblocks)
		fblocks = 1;
#endif
	const uint32_t fw = fgrid.x * TG_STREAM_WPB, fgroups = (nslots + 3) / 4, capw = 4u * ((fgroups + fw - 1) / fw);	/* (as k_front_stream computes them) */
#define FIX_LAUNCH(P, V) hipLaunchKernelGGL((k_front_stream_fix<P, V>), dim3(fblocks), dim3(64 * TG_FIX_WAVES), 0, s, d_stream, prm, d_packed, d_cls, d_ysum, d_defer, fw, capw)
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

