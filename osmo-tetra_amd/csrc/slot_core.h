/*
 * slot_core.h -- one lane = one SLOT: the whole lower-MAC decode of a downlink burst in one pass of one lane
 * (round 6; what k_slot / k_slot_t run, and what tests/host_emul compiles for the host).
 *
 * Replaces, for one burst, what the reference does in ONE call chain: phy/tetra_burst.c:341-379 (demux) ->
 * lower_mac/tetra_lower_mac.c:143-282 (descramble, de-interleave, de-puncture, Viterbi, CRC for every block of the burst).
 * The per-step arithmetic is vit_core.h's (difference form of a step pair, register-exchange history bytes, block-wise
 * traceback): this header only lays the blocks of a burst onto ONE 36-block schedule that every burst type shares, so
 * that the 64 lanes of a wave can hold 64 neighbouring slots of any mix of types and run the same instructions:
 *
 *   block slot   0 .. 7        8 .. 16      17                              18 .. 34     35
 *   NORM_1       SCH/F: lead-in, 35 full blocks, last block (4 steps + the K-1 flush steps)            292 steps
 *   NORM_2       BLK1: lead-in, 17 full blocks ..........| last | BLK2: lead-in, 17 full blocks, last  2 x 148
 *   SYNC         (idle)       | SB1: lead-in, 9 full ....| last | SB2:  lead-in, 17 full blocks, last  84 + 148
 *
 * Code word g (g = 0..17, tg_layout.h: 24 received bits = two block slots) is read by every lane at the same point; the
 * staging step in front (the kernel's) puts a SYNC burst's five SB1 words at g = 4..8 so that SB1 ends where BLK1 ends.
 * Lanes differ in exactly three places, all of them selects, none a branch:
 *   - in front of g = 4 a SYNC lane takes a fresh state and SB1's lead-in steps (tg_slot_sb_prologue);
 *   - block slot 17 (tg_slot_mid): after its first four steps a two-block lane (NORM_2, SYNC) runs the flush tree, keeps
 *     that history block, takes a fresh state and runs the second block's four lead-in steps where a NORM_1 lane runs the
 *     block's other four steps -- same instruction count, the table entries' addresses differ per lane;
 *   - the traceback re-starts in state 0 at block slot 17 for a two-block lane, and the CRC register re-starts there
 *     (and at byte 8 for SB1).
 * The record leaves as twenty 16-byte pieces (tg_slot_piece_*): the lane owns the whole 320-byte record, whatever the type.
 *
 * Bit-exactness: every decision is vit_core.h's (ties: libosmocore's rule); a common offset on all metrics of a lane
 * (the one normalisation after block slot 17, applied to all lanes) changes no decision.
 */
#ifndef SLOT_CORE_H
#define SLOT_CORE_H

#include "tg_layout.h"
#include "vit_core.h"

#define TG_SLOT_NBLK 36		/* block slots of the schedule */
#define TG_SLOT_NOD  9		/* dwords of decoded bits: 36 bytes */
#define TG_SLOT_MIDBLK 17	/* where the first block of a two-block burst ends */
#define TG_SLOT_SB1_G0 4	/* code-word position of SB1's first word in the schedule (five words: g = 4..8) */

TG_HD void tg_slot_state_init(tg_vit_state &v)
{
	v.Z[0] = tg_as_us2(TG_VIT_FLOOR | (TG_VIT_INF << 16));
#pragma unroll
	for (int k = 1; k < 8; k++)
		v.Z[k] = tg_as_us2(TG_VIT_INF | (TG_VIT_INF << 16));
}

/* v = sel ? a : v on all eight state registers */
TG_HD void tg_slot_state_sel(tg_vit_state &v, const tg_vit_state &a, bool sel)
{
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = tg_as_us2(sel ? tg_as_u32(a.Z[k]) : tg_as_u32(v.Z[k]));
}

/* the four lead-in steps on the six bits of 'six' (bits 0..5), history dropped.  bmdo(o, w): the ten dwords of the
 * table entry at byte offset o of the entry arrays (o = 128 pair + 16 triple: vit_core.h, TG_BMD_*) */
template <typename BmdO>
TG_HD void tg_slot_leadin(tg_vit_state &v, uint32_t six, BmdO bmdo)
{
	uint32_t w[10];
	bmdo((six << 4) & 0x70u, w);
	tg_step_pair_d(v, w);
	bmdo(((six << 1) & 0x70u) + 128u, w);
	tg_step_pair_d(v, w);
	tg_vit_clean(v);
}

/* in front of code word TG_SLOT_SB1_G0: a SYNC lane starts SB1 (fresh state, lead-in bits = the word's bits 24..29);
 * computed by all lanes, taken by the SYNC lanes */
template <typename BmdO>
TG_HD void tg_slot_sb_prologue(tg_vit_state &v, bool sb, uint32_t word, BmdO bmdo)
{
	tg_vit_state t;
	tg_slot_state_init(t);
	tg_slot_leadin(t, word >> 24, bmdo);
	tg_slot_state_sel(v, t, sb);
}

/* one full block (eight steps on the twelve bits 0..11 of tw), history out */
template <typename BmdO>
TG_HD void tg_slot_block(tg_vit_state &v, uint32_t tw, uint32_t h[4], BmdO bmdo)
{
#pragma unroll
	for (int p = 0; p < 4; p++) {
		uint32_t w[10];
		bmdo(((p < 2 ? tw << (4 - 3 * p) : tw >> (3 * p - 4)) & 0x70u) + 128u * p, w);
		tg_step_pair_d(v, w);
	}
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
	tg_vit_clean(v);
}

/* the last block of a trellis: four steps on bits 0..5 of tw, then the K-1 flush steps as a min tree */
template <typename BmdO>
TG_HD void tg_slot_block_last(tg_vit_state &v, uint32_t tw, uint32_t h[4], BmdO bmdo)
{
#pragma unroll
	for (int p = 0; p < 2; p++) {
		uint32_t w[10];
		bmdo(((tw << (4 - 3 * p)) & 0x70u) + 128u * p, w);
		tg_step_pair_d(v, w);
	}
	tg_flush4(v);
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
	tg_vit_clean(v);
}

/*
 * Block slot 17.  tw: its twelve bits (bits 12..23 of code word 8); lead: bits 24..29 of code word 9 (the second block's
 * lead-in bits; 0 for NORM_1).  A NORM_1 lane runs a full block.  A two-block lane runs its first block's LAST block (four
 * steps + flush tree: that is the history block kept), takes a fresh state and runs the second block's lead-in.  Both do two
 * step pairs, then two more whose table entries differ per lane (pair index 2, 3 on tw's bits 6..11 -- or 0, 1 on 'lead').
 */
template <typename BmdO>
TG_HD void tg_slot_mid(tg_vit_state &v, bool two, uint32_t tw, uint32_t lead, uint32_t h[4], BmdO bmdo)
{
	uint32_t w[10];
	bmdo((tw << 4) & 0x70u, w);
	tg_step_pair_d(v, w);
	bmdo(((tw << 1) & 0x70u) + 128u, w);
	tg_step_pair_d(v, w);
	/* the two-block lanes' flush, history block and fresh state (computed by all, selected) */
	tg_vit_state f;
#pragma unroll
	for (int k = 0; k < 8; k++)
		f.Z[k] = v.Z[k];
	tg_flush4(f);
	uint32_t hf[4];
#pragma unroll
	for (int d = 0; d < 4; d++)
		hf[d] = tg_pack_bytes02(tg_as_u32(f.Z[2 * d]), tg_as_u32(f.Z[2 * d + 1]));
	tg_vit_state ini;
	tg_slot_state_init(ini);
	tg_slot_state_sel(v, ini, two);
	const uint32_t six = two ? lead : (tw >> 6);
	const uint32_t psel = two ? 0u : 256u;		/* pairs 0, 1 (lead-in tie bits) or pairs 2, 3 */
	bmdo(((six << 4) & 0x70u) + psel, w);
	tg_step_pair_d(v, w);
	bmdo(((six << 1) & 0x70u) + psel + 128u, w);
	tg_step_pair_d(v, w);
#pragma unroll
	for (int d = 0; d < 4; d++) {
		const uint32_t hn = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
		h[d] = two ? hf[d] : hn;
	}
	tg_vit_clean(v);	/* (a two-block lane: the lead-in's decisions are dropped, as tg_vit_leadin does) */
}

/* byte 's' (0..15) of the 16 history bytes of a block: two v_perm_b32 + one select on the device */
TG_HD uint32_t tg_slot_hist_byte(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t sel = s & 7;
	const uint32_t lo = __builtin_amdgcn_perm(w1, w0, sel);
	const uint32_t hi = __builtin_amdgcn_perm(w3, w2, sel);
	return ((s & 8) ? hi : lo) & 0xff;
#else
	const uint32_t w[4] = { w0, w1, w2, w3 };
	return (w[(s >> 2) & 3] >> (8 * (s & 3))) & 0xff;
#endif
}

/* one traceback hop over block slot B (a constant): od gets the block's eight decoded bits, s moves to the block's start state */
template <int B>
TG_HD void tg_slot_hop(uint32_t (&od)[TG_SLOT_NOD + 1], uint32_t &s, bool two, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3)
{
	if (B == TG_SLOT_MIDBLK)
		s = two ? 0u : s;	/* the first block of a two-block burst ends in state 0 here */
	const uint32_t byte = tg_slot_hist_byte(h0, h1, h2, h3, s);
	od[B >> 2] |= byte << ((B & 3) * 8);
	s = tg_brev4(byte);
}

/*
 * CRC-16 of the burst's blocks over the decoded bytes (lower_mac/crc_simple.c:65-82, byte-table form; lower_mac/
 * tetra_lower_mac.c:258: over type-1 + 16 bits = all decoded bits but the four tail bits):
 *   NORM_1      bytes 0..34 + low nibble of 35                      -> crc[0]
 *   two-block   bytes 0..16 + nibble of 17 -> crc[0] (SYNC: SB1 = bytes 8..16 + nibble of 17); bytes 18..34 + nibble of 35 -> crc[1]
 * tl(x) / tm(x): the two 256-entry tables (input byte LSB first / the register's top byte).
 */
TG_HD uint32_t tg_slot_crc_nibble(uint32_t crc, uint32_t nib)
{
#pragma unroll
	for (int i = 0; i < 4; i++) {
		crc ^= ((nib >> i) & 1) << 15;
		crc = (crc & 0x8000) ? (((crc << 1) ^ 0x1021) & 0xffff) : ((crc << 1) & 0xffff);
	}
	return crc;
}

/* (I0: the first decoded byte to look at -- 8 where every lane holds a SYNC burst: SB1 starts there) */
template <int I0 = 0, typename TabL, typename TabM>
TG_HD void tg_slot_crc(const uint32_t (&od)[TG_SLOT_NOD + 1], bool two, bool sb, TabL tl, TabM tm, uint32_t &crc0, uint32_t &crc1)
{
	uint32_t crc = 0xffff;
	crc0 = crc1 = 0;
#pragma unroll
	for (int i = I0; i < TG_SLOT_NBLK; i++) {
		const uint32_t byte = (od[i >> 2] >> ((i & 3) * 8)) & 0xff;
		if (i == 2 * TG_SLOT_SB1_G0)
			crc = sb ? 0xffffu : crc;
		if (i == TG_SLOT_MIDBLK) {
			const uint32_t cn = tg_slot_crc_nibble(crc, byte & 15);
			const uint32_t cb = ((crc << 8) & 0xffff) ^ tm(crc >> 8) ^ tl(byte);
			crc0 = cn;
			crc = two ? 0xffffu : cb;
		} else if (i == TG_SLOT_NBLK - 1) {
			crc = tg_slot_crc_nibble(crc, byte & 15);
		} else
			crc = ((crc << 8) & 0xffff) ^ tm(crc >> 8) ^ tl(byte);
	}
	crc1 = two ? crc : 0u;
	crc0 = two ? crc0 : crc;
}

/*
 * The record as twenty 16-byte pieces (tg_layout.h "Output record"): piece 0 / 1 header, 2 BBK, 3 + P the type-1 bits at
 * record byte 48 + 16 P, P = 0..16.  tg_slot_piece_sel(P, ...) names the sixteen decoded bits a bits piece shows (half-word
 * index q into od[]: bits 16 q .. 16 q + 15), how many of its last four bytes are valid, and whether it is empty:
 *   NORM_1   P = 0..16: q = P (P = 16: 12 valid)                                  268 bits at byte 48
 *   NORM_2   P = 0..7:  q = P (P = 7: 12 valid)  P = 8..15: q = P + 1 (15: 12)    124 bits at 48, 124 at 176 (= decoded bit 144 on)
 *   SYNC     P = 0..3:  q = P + 4 (3: 12 valid)  P = 4..7: empty  P = 8..15: as NORM_2   SB1's 60 bits = decoded bits 64 on
 */
template <int P>
TG_HD uint32_t tg_slot_piece_bits(const uint32_t (&od)[TG_SLOT_NOD + 1], bool two, bool sb, bool &full, bool &empty)
{
	constexpr int qa = P, qb = (P < 8) ? P : P + 1, qc = (P < 4) ? P + 4 : P + 1;
	const uint32_t ha = (qa < 18) ? (od[qa >> 1] >> (16 * (qa & 1))) & 0xffffu : 0u;
	const uint32_t hb = (qb < 18) ? (od[qb >> 1] >> (16 * (qb & 1))) & 0xffffu : 0u;
	const uint32_t hc = (qc < 18) ? (od[qc >> 1] >> (16 * (qc & 1))) & 0xffffu : 0u;
	const bool sb1 = sb && P < 8;
	full = sb1 ? P != 3 : two ? (P != 7 && P != 15) : P != 16;
	empty = sb1 ? P >= 4 : two ? P == 16 : false;
	return sb1 ? hc : two ? hb : ha;
}

/* sixteen bits as sixteen bytes of 0 / 1 (bit i -> byte i): the plain statement; the kernels use a 16-entry LDS table per nibble */
TG_HD void tg_slot_spread16(uint32_t h16, uint32_t o[4])
{
#pragma unroll
	for (int k = 0; k < 4; k++)
		o[k] = (((h16 >> (4 * k)) & 15u) * 0x00204081u) & 0x01010101u;
}

/* the SYNC PDU's fields out of SB1's decoded bits (lower_mac/tetra_lower_mac.c:284-297); od2 = the decoded dwords from SB1's
 * first bit on (od + 2: SB1 starts at decoded byte 8) */
TG_HD uint32_t tg_slot_field_msb(uint32_t lo, uint32_t hi, int sh, int len)
{
	const unsigned long long two = (unsigned long long)lo | ((unsigned long long)hi << 32);
	const uint32_t f = (uint32_t)(two >> sh) & ((1u << len) - 1);
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_bitreverse32(f) >> (32 - len);
#else
	uint32_t r = 0;
	for (int i = 0; i < len; i++)
		r |= ((f >> i) & 1u) << (len - 1 - i);
	return r;
#endif
}
#define TG_SLOT_FIELD(od, n0, len) tg_slot_field_msb((od)[(n0) >> 5], (od)[((n0) >> 5) + 1], (n0) & 31, (len))

TG_HD void tg_slot_sync_fields(const uint32_t *od2, uint32_t &f0, uint32_t &f1, uint32_t &code)
{
	const uint32_t cc = TG_SLOT_FIELD(od2, 4, 6), tn = TG_SLOT_FIELD(od2, 10, 2) + 1;
	const uint32_t fn = TG_SLOT_FIELD(od2, 12, 5), mn = TG_SLOT_FIELD(od2, 17, 6);
	const uint32_t mcc = TG_SLOT_FIELD(od2, 31, 10), mnc = TG_SLOT_FIELD(od2, 41, 14);
	f0 = cc | (tn << 8) | (fn << 16) | (mn << 24);
	f1 = mcc | (mnc << 16);
	code = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
}

#endif
