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
#include "tg_dev_stream.h"	/* typedefs, front_flush, tg_stream_params, chan_of_slot, the packed-bit front end's pieces */


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


/* second pass of the packed-bit front end: every slot the first pass deferred goes through the exact per-position search.
 * Round 5: the first pass keeps one list PER WAVE -- defer[w] = how many slots wave w of k_front_stream deferred, its
 * slots from defer[TG_DEFER_L0(fw) + w * capw] (fw = waves of that launch, capw = slots a wave can meet at most) -- and
 * appends with a counter of its own.  (Rounds 2-4 had one list and one counter for the launch: an atomicAdd with a
 * return value, at device scope -- on this part that is a trip to the memory side of the fabric, every wave's goes to the
 * same address, and the wait for its answer also waits for the wave's prefetched group: 15 of the kernel's 155 us at
 * 1 % deferred slots, tools/experiments/front_ablate.sh "-DTGS_ABLATE=64".)  A wave of k_front_stream takes groups wave, wave +
 * nwaves, ..., so every list is an even sample of the grid whatever the damage looks like; a wave of this kernel takes the
 * lists w = wave, wave + nwaves, ... one entry at a time. */
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
#define TGS_FUSED 0
#include "tg_front_stream_body.h"

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
	if (int rc_ = tgk_front_stream_ev_fire(s))
		return rc_;
	const dim3 fgrid(blocks * (4 / TG_STREAM_WPB)), fblock(64 * TG_STREAM_WPB);
	if (packed_input)
		hipLaunchKernelGGL(k_front_stream<true>, fgrid, fblock, 0, s, d_stream, prm, d_packed, d_cls, d_ysum, d_defer);
	else
		hipLaunchKernelGGL(k_front_stream<false>, fgrid, fblock, 0, s, d_stream, prm, d_packed, d_cls, d_ysum, d_defer);
	if (ev_mid)
		HIPCHK(hipEventRecord((hipEvent_t)ev_mid, s));
	const uint32_t fw = fgrid.x * TG_STREAM_WPB, fgroups = (nslots + 3) / 4, capw = 4u * ((fgroups + fw - 1) / fw);	/* (as k_front_stream computes them) */
	return tgk_front_stream_fix(d_stream, &prm, d_packed, d_cls, d_ysum, d_defer, fw, capw, s, packed_input);
}

/* per-kernel timing: the event armed by tgk_front_stream_ev_start() goes onto the stream now (the front-end launch follows) */
int tgk_front_stream_ev_fire(hipStream_t s)
{
	if (tl_front_ev_start) {
		HIPCHK(hipEventRecord((hipEvent_t)tl_front_ev_start, s));
		tl_front_ev_start = nullptr;
	}
	return 0;
}

/* the exact pass over the lists a first pass left: fw lists (one per wave of k_front_stream / per task of k_slot) of up to capw slots */
int tgk_front_stream_fix(const uint8_t *d_stream, const tg_stream_params *prmp, uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum,
			 uint32_t *d_defer, uint32_t fw, uint32_t capw, hipStream_t s, bool packed_input)
{
	const tg_stream_params &prm = *prmp;
	const uint32_t nslots = prm.nslots;
	uint32_t fblocks = (nslots / 128 + 3) / 4 + 1;	/* about a wave per deferred slot at 1 % of them */
	if (fblocks > 256 * 16)
		fblocks = 256 * 16;
#if !TGS_DEFER_ATOMIC
	fblocks = (fw + TG_FIX_LISTS - 1) / TG_FIX_LISTS;	/* a workgroup per TG_FIX_LISTS lists of the first pass */
	if (!fblocks)
		fblocks = 1;
#endif
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

/* the parameter block of a multi-channel launch (tgk_front_stream_multi, tgk_slot_fused) */
void tgk_stream_params_multi(tg_stream_params *prm, const struct tg_chan_ent *d_chan, uint32_t nchan, uint32_t nslots, uint32_t chunk)
{
	memset(prm, 0, sizeof(*prm));
	prm->nslots = nslots;
	prm->chan = d_chan;
	prm->nchan = nchan;
	stream_patterns(*prm, chunk);
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
	tgk_stream_params_multi(&prm, d_chan, nchan, nslots, chunk);
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

