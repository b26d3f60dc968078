/*
 * tg_dev_stream.h -- what the packed-bit stream front end is made of, shared by its two users: k_front_stream (tg_k_front.hip) and the
 * front phase of the lane-per-slot kernel k_slot (tg_k_slot.hip, round 6).  Moved here from tg_k_front.hip as it was; the kernel body
 * itself is tg_front_stream_body.h, included once by each unit.
 */
#ifndef TG_DEV_STREAM_H
#define TG_DEV_STREAM_H

#include "tg_dev.h"

typedef uint32_t __attribute__((aligned(1))) tg_u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) tg_u16_unaligned;
typedef uint16_t __attribute__((may_alias)) tg_u16_alias;

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

/* classification word of a slot the packed-bit kernel leaves to k_front_stream_fix (never a valid word: offsets stay below 1088) */
#define TG_CLS_DEFER 0xffffffffu
#define TG_DEFER_L0(fw) (((fw) + 15u) & ~15u)

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
	uint32_t chan;	/* (the fused form, tg_k_slot.hip: the channel the group lies in) */
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

/* host-side pieces of the front unit's launch layer that the lane-per-slot unit uses as well (tg_k_front.hip) */
int tgk_front_stream_ev_fire(hipStream_t s);
int tgk_front_stream_fix(const uint8_t *d_stream, const tg_stream_params *prm, uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum,
			 uint32_t *d_defer, uint32_t fw, uint32_t capw, hipStream_t s, bool packed_input);
void tgk_stream_params_multi(tg_stream_params *prm, const struct tg_chan_ent *d_chan, uint32_t nchan, uint32_t nslots, uint32_t chunk);

#endif
