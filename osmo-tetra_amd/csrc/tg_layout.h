/*
 * tg_layout.h -- data layout shared by the host code and the HIP kernels.
 *
 * One TETRA downlink slot is 510 stream bytes (1 bit per byte, the tetra-rx input
 * format).  Field offsets follow phy/tetra_burst.c:31-47 of the reference, block
 * parameters lower_mac/tetra_lower_mac.c:55-102.
 *
 * Device pipeline:
 *   front kernel : 510 B slot  -> tg_packed_slot (80 B): code bits de-interleaved and
 *                  bit-packed, still scrambled, plus classification meta
 *   viterbi kernels: tg_packed_slot (+ scrambling masks) -> tg_slot_rec (320 B)
 */
#ifndef TG_LAYOUT_H
#define TG_LAYOUT_H

#include <stdint.h>

#define TG_SLOT_BITS      510

/* burst types = enum tetra_train_seq of the reference (phy/tetra_burst.h:30-36) */
#define TG_BURST_NORM_1   0
#define TG_BURST_NORM_2   1
#define TG_BURST_NORM_3   2
#define TG_BURST_SYNC     3
#define TG_BURST_NONE     0xff

/* block kinds handled by the trellis kernels */
#define TG_KIND_SB1   0	/* 120 bits, a=11,  80 type-2 bits, 60 type-1  */
#define TG_KIND_216   1	/* 216 bits, a=101, 144 type-2,     124 type-1 (NDB, SB2) */
#define TG_KIND_432   2	/* 432 bits, a=103, 288 type-2,     268 type-1 (SCH/F) */
#define TG_KIND_168   3	/* 168 bits, a=13,  112 type-2,     92 type-1  (SCH/HU; block mode only) */

/* slot field offsets (phy/tetra_burst.c:31-47) */
#define TG_SB_BLK1_OFF    94
#define TG_SB_BBK_OFF     252
#define TG_SB_BLK2_OFF    282
#define TG_NDB_BLK1_OFF   14
#define TG_NDB_BBK1_OFF   230
#define TG_NDB_BBK2_OFF   266
#define TG_NDB_BLK2_OFF   282
#define TG_SYNC_TRAIN_OFF 214	/* phy/tetra_burst_sync.c:123 */
#define TG_NORM_TRAIN_OFF 244	/* phy/tetra_burst_sync.c:133 */

/*
 * Packed slot: 20 dwords.
 *   w[0..17]  code words.  A code word carries 24 received type-3 bits (two 8-step
 *             trellis blocks of 12 bits); word 0 of every block additionally carries
 *             the first 6 type-3 bits (the 4 lead-in trellis steps) in bits 24..29.
 *             word d bit p (p<24) = type3[6 + 24*d + p];  word 0 bit 24+q = type3[q].
 *             NORM_1: SCH/F in w[0..17].  NORM_2: BLK1 w[0..8], BLK2 w[9..17].
 *             SYNC:   SB1 w[0..4], SB2 w[9..17].
 *   w[18]     the 30 BBK bits in stream order (bit p = bb(p+1))
 *   w[19]     meta: burst type | flags<<8 | train_offset<<16
 */
#define TG_PACKED_WORDS   20
#define TG_PW_BLK1        0
#define TG_PW_BLK2        9
#define TG_PW_BBK         18
#define TG_PW_META        19

/*
 * Classification word of the stream front-end (one per grid slot):
 *   bits 0..7   first training sequence found in the search window (enum tetra_train_seq) or 0xff
 *   bits 8..23  its offset
 *   bits 24..31 TG_CLS_* flags
 * Semantics = tetra_find_train_seq(slot, w, NORM_1|NORM_2|SYNC) of the reference for the steady-state window w,
 * including its rule for positions below 21, where the look-ahead filter is skewed (phy/tetra_burst.c:289-297: a
 * sequence there counts iff the filter -- the stream with in[20] missing -- equals the head of one of the five
 * training sequences); round 3: the kernels evaluate that rule themselves.  TG_CLS_EARLY21 is what a producer that does
 * not (the round-1/2 kernels, a simplified model) sets instead: "a full match exists below 21, ask the exact routine".
 */
#define TG_CLS_EARLY21    0x01	/* a full match exists at an offset < 21 and was NOT evaluated */
#define TG_CLS_NONBINARY  0x02
#define TG_CLS_CLIPPED    0x04	/* window longer than the kernel's view (TG_VIEW_OF) and nothing found in view */
#define TG_CLS_NOVIEW     0x08	/* nothing found, and nothing in the rest of the view (bounded by the stream's end)
				 * either: "nothing" is then also the answer for any longer window up to the view -- what the
				 * synchroniser searches while it works off a backlog (tg_walk_core.h) */
/* a word that says "nothing in the window" and does not carry TG_CLS_NOVIEW: bits 4..6 of the flags = 1 + type of the FIRST
 * sequence that starts inside the view and ends past the window, bits 8..23 of the word its offset -- what a longer window
 * (a slot handled some calls late, feeds of 128 / 256 bytes) finds first, if it reaches that far */
#define TG_CLS_VIEWHIT_SHIFT 4
#define TG_CLS_VIEWHIT(flags) (((flags) >> TG_CLS_VIEWHIT_SHIFT) & 7u)
/* SYNC-sequence summary of a grid slot (uint16): where the 38-bit y sequence starts inside the slot's own
 * 510 positions, regardless of any search window -- what an UNLOCKED synchroniser scans for */
#define TG_YS_NONE        0xffffu	/* no y sequence starts in this slot */
#define TG_YS_MULTI       0x8000u	/* more than one does; bits 0..8 hold the first */
#define TG_YS_FIRST(v)    ((v) & 0x1ffu)
#define TG_STREAM_VIEW    1088	/* most bytes of a slot's search window the kernel looks at (LDS rows, loop bounds) */
/* ... and what it looks at for a replay with feeds of 'chunk' bytes: a slot's own window ends at most chunk - 1 bytes past
 * the slot, the window of a slot that is handled a call late (tg_walk_core.h) another chunk further, and a 38-byte
 * sequence that starts inside must end inside: 510 + 63 + 64 = 637 -> 640, 510 + 127 + 128 = 765 (+ 38) -> 832,
 * 510 + 255 + 256 = 1021 (+ 38) -> 1088 */
#define TG_VIEW_OF(chunk) ((chunk) <= 64u ? 640u : (chunk) <= 128u ? 832u : 1088u)
#define TG_STREAM_SLACK   192	/* readable bytes the stream buffer must have after its last byte */

/* soft area of a slot (config 5): int8 values in trellis order, see k_front_soft */
#define TG_SOFT_SLOT_BYTES 512
#define TG_SOFT_AREA2      224	/* second block (BLK2 / SB2) */
#define TG_SOFT_BBK        448	/* 30 BBK values */

#define TG_FLAG_NONBINARY 0x01	/* a byte other than 0/1 was seen in the slot (block mode: in the block) */
/* set by the traffic stage (tg_traffic.hip, tgpu_plan_set_traffic / tgpu_plan_traffic) on bursts the caller marked as traffic: */
#define TG_FLAG_TRAFFIC   0x02	/* the burst's SCH/F block (NORM_1) or its second block (NORM_2, SYNC) went to the traffic dump and
				 * was not indicated (tetra_lower_mac.c:198-241): its crc_ok reads 0 */
#define TG_FLAG_BLK1_STOLEN 0x04	/* first half of a traffic NORM_2 burst: cur_burst.blk1_stolen (tetra_lower_mac.c:194-195) */

/* scrambling-mask table entry: 40 dwords, same bit layout as the code words */
#define TG_MASK_WORDS     40
#define TG_MW_432         0	/* 18 words */
#define TG_MW_216         18	/* 9 words  */
#define TG_MW_BBK         27
#define TG_MW_CODE        28	/* the scrambling code itself */
#define TG_MW_168         29	/* 7 words  */
#define TG_MW_ROUNDS      18	/* ballot rounds (two words each) that cover the words above */

/*
 * Output record, one per slot, fixed 320 bytes (the unit of the RCCL gather).
 *   @0   u8  burst_type
 *   @1   u8  flags
 *   @2   u8  crc_ok[2]      [0]: SB1 / BLK1 / SCH-F   [1]: SB2 / BLK2
 *   @4   u16 crc[2]
 *   @8   u32 scrambling code used for BBK/BLK/SB2 (SB1 always uses 3)
 *   @12  u32 slot id
 *   @16  u32 SYNC-PDU fields cc | tn<<8 | fn<<16 | mn<<24      (SYNC slots)
 *   @20  u32 SYNC-PDU fields mcc | mnc<<16
 *   @24  u32 scrambling code derived from the SYNC PDU
 *   @28  u8  BBK bit errors corrected by the optional RM(30,14) decoder
 *   @32  14 B  BBK type-1 bits (1 bit per byte)
 *   @48  SB1 (60 B) / BLK1 (124 B) / SCH-F (268 B) type-1 bits
 *   @176 SB2 / BLK2 (124 B) type-1 bits
 */
#define TG_REC_BYTES      320
#define TG_REC_TYPE       0
#define TG_REC_PENDING    0xee	/* never a burst type: a host polling mapped records for completion presets this (k_burst writes the type last) */
#define TG_REC_FLAGS      1
#define TG_REC_CRC_OK     2
#define TG_REC_CRC        4
#define TG_REC_CODE       8
#define TG_REC_SLOT       12
#define TG_REC_SBF0       16
#define TG_REC_SBF1       20
#define TG_REC_SBCODE     24
#define TG_REC_BBK_NERR   28	/* u8: bit errors the optional RM(30,14) decoder corrected in the BBK (0 when it is off) */
#define TG_REC_BBK        32
#define TG_REC_BITS1      48
#define TG_REC_BITS2      176

/*
 * Wire record, 40 bytes per slot = 10 dwords: the same decoded blocks bit-packed (LSB first) for transport over
 * xGMI (the unit of the RCCL gather when records leave the GPU that decoded them).
 *   w[0]     burst type (byte 0, 0xff = nothing) | flags (byte 1, TG_FLAG_*) | BBK type-1 bits << 16 (14 bits)
 *   w[1..9]  NORM_1: SCH/F 268 bits (w[9] bits 0..11 are its last 12), crc[0] in w[9] bits 12..27
 *            NORM_2: BLK1 124 bits in w[1..4], BLK2 in w[5..8], w[9] = crc[0] | crc[1] << 16
 *            SYNC  : SB1 60 bits in w[1..2], SB2 in w[5..8], w[9] = crc[0] | crc[1] << 16
 * 282 payload bits + flags + the CRC words; crc_ok is not carried, it is crc == 0x1d0f (lower_mac/crc_simple.h,
 * TETRA_CRC_OK).  (The 48-byte form of round 1 spent 8 bytes on byte-sized fields.)  The slot id and the
 * scrambling code are the receiver's knowledge (position in the gathered array, channel state):
 * tgpu_wire_unpack() takes them as arguments.  Slots the batch does not decode are not written: clear the buffer
 * to 0xff once.
 */
#define TG_WIRE_BYTES     40
#define TG_WIRE_WORDS     10
#define TG_WIRE_W_BITS1   1	/* first block from w[1] */
#define TG_WIRE_W_BITS2   5	/* second block (216-bit kinds) from w[5] */
#define TG_WIRE_W_CRC     9
#define TG_WIRE_SCHF_CRC_SHIFT 12
#define TG_CRC_OK         0x1d0f

#ifdef __cplusplus
extern "C" {
#endif

/* number of 8-step trellis blocks / type-1 bits / crc span per kind */
static inline int tg_kind_nblk(int kind)  { return kind == TG_KIND_SB1 ? 10 : kind == TG_KIND_216 ? 18 : kind == TG_KIND_168 ? 14 : 36; }
static inline int tg_kind_K(int kind)     { return kind == TG_KIND_SB1 ? 120 : kind == TG_KIND_216 ? 216 : kind == TG_KIND_168 ? 168 : 432; }
static inline int tg_kind_a(int kind)     { return kind == TG_KIND_SB1 ? 11 : kind == TG_KIND_216 ? 101 : kind == TG_KIND_168 ? 13 : 103; }
static inline int tg_kind_type1(int kind) { return kind == TG_KIND_SB1 ? 60 : kind == TG_KIND_216 ? 124 : kind == TG_KIND_168 ? 92 : 268; }

/*
 * type-4 (stream order inside a block) index feeding code-word bit (d, p) of a block
 * of kind 'kind'; -1 when the bit is unused.  Composition of the 2/3 puncturing order,
 * the block de-interleaver  type3[i] = type4[(a*(i+1)) mod K]
 * (lower_mac/tetra_interleave.c:36-39,51-59) and the code-word layout above.
 */
static inline int tg_codeword_src(int kind, int d, int p)
{
	int K = tg_kind_K(kind), a = tg_kind_a(kind), i;
	if (p < 24)
		i = 6 + 24 * d + p;
	else if (d == 0 && p < 30)
		i = p - 24;
	else
		return -1;
	if (i >= K)
		return -1;
	return (a * (i + 1)) % K;
}

/* slot byte offset of type-4 bit j of the block occupying code words starting at
 * 'wbase' (TG_PW_BLK1 / TG_PW_BLK2) in a burst of type 'btype' */
static inline int tg_block_stream_off(int btype, int wbase, int j)
{
	if (btype == TG_BURST_SYNC)
		return (wbase == TG_PW_BLK1) ? TG_SB_BLK1_OFF + j : TG_SB_BLK2_OFF + j;
	if (btype == TG_BURST_NORM_1)	/* SCH/F = BLK1 || BLK2, phy/tetra_burst.c:367-368 */
		return (j < 216) ? TG_NDB_BLK1_OFF + j : TG_NDB_BLK2_OFF + (j - 216);
	return (wbase == TG_PW_BLK1) ? TG_NDB_BLK1_OFF + j : TG_NDB_BLK2_OFF + j;
}

/* slot byte offset of BBK bit p (phy/tetra_burst.c:351,357-358) */
static inline int tg_bbk_stream_off(int btype, int p)
{
	if (btype == TG_BURST_SYNC)
		return TG_SB_BBK_OFF + p;
	return (p < 14) ? TG_NDB_BBK1_OFF + p : TG_NDB_BBK2_OFF + (p - 14);
}

/*
 * slot byte offset feeding bit p of packed word w for burst type btype; -1 = none.
 * This single function defines the front kernel's gather table and the scrambling
 * mask layout (via tg_packed_seqpos).
 */
static inline int tg_packed_src(int btype, int w, int p)
{
	int kind, wbase, j;
	if (w == TG_PW_BBK)
		return (p < 30) ? tg_bbk_stream_off(btype, p) : -1;
	if (w >= TG_PW_BBK)
		return -1;
	if (btype == TG_BURST_NORM_1) {
		kind = TG_KIND_432; wbase = TG_PW_BLK1;
	} else if (btype == TG_BURST_NORM_2) {
		kind = TG_KIND_216; wbase = (w >= TG_PW_BLK2) ? TG_PW_BLK2 : TG_PW_BLK1;
	} else if (btype == TG_BURST_SYNC) {
		if (w >= TG_PW_BLK2) { kind = TG_KIND_216; wbase = TG_PW_BLK2; }
		else if (w < 5)      { kind = TG_KIND_SB1; wbase = TG_PW_BLK1; }
		else return -1;
	} else
		return -1;
	j = tg_codeword_src(kind, w - wbase, p);
	if (j < 0)
		return -1;
	return tg_block_stream_off(btype, wbase, j);
}

#ifdef __cplusplus
}
#endif
#endif
