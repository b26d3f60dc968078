/*
 * vit_core.h -- one lane = one trellis: the 16-state Viterbi decoder for the TETRA
 * rate-1/4 K=5 mother code under 2/3 puncturing, in packed 16-bit arithmetic.
 *
 * Replaces (bit-exactly, ties included) what the reference does in
 *   lower_mac/viterbi.c:6-25 -> lower_mac/viterbi_cch.c:58-66 -> libosmocore
 *   osmo_conv_decode() (start state 0, 'len' steps + 4 flush steps on erasures,
 *   traceback from state 0, tie -> predecessor whose oldest bit is 0).
 *
 * Design (MI355X-first; compiled for gfx950 by hipcc, and for the host by clang++
 * in tests/ so the very same code can be checked on a machine without a GPU):
 *
 *  - State s = last four input bits, newest in bit 0.  Butterfly j (0..7):
 *    predecessors j and j+8, successors 2j and 2j+1.  All four generator
 *    polynomials contain 1 and D^4, so out(j+8,b) = ~out(j,b) and out(j,1) = ~out(j,0):
 *    with m = Hamming distance of the received bits to out(j,0) and n received bits,
 *        new[2j]   = min(pm[j] + m,     pm[j+8] + n - m)
 *        new[2j+1] = min(pm[j] + n - m, pm[j+8] + m)
 *  - Eight VGPRs Z[0..7]; Z[k] holds states 2k (low half) and 2k+1 (high half).
 *    Each half: [15:8] path metric, [7:0] the last <=8 decisions of that state's
 *    survivor.  A butterfly is two v_pk_add_u16 (op_sel broadcasts the predecessor
 *    half) and one v_pk_min_u16.
 *  - Tie rule and decision recording in one go: the candidate coming from
 *    predecessor j+8 gets bit i of the low byte added (i = step index inside the
 *    current 8-step block).  Bits above i are still clear, bits below i are older
 *    decisions of the respective survivors, so on equal metrics the j candidate is
 *    smaller (tie -> j, i.e. oldest bit 0) and the winner's bit i records the choice.
 *    The low byte travels with the selected predecessor, i.e. it is that survivor's
 *    decision history (register exchange inside the metric word).
 *  - Decision d_k (step k) equals input bit u_(k-4).  After 4 lead-in steps every
 *    8-step block therefore leaves, in each state's low byte, 8 decoded bits of that
 *    state's survivor; they are extracted (v_perm_b32) to LDS, 16 bytes per block.
 *    Traceback hops block-wise: byte b of the output = hist[b][s],
 *    next s = bitreverse4(hist & 15).  len/8 dependent LDS reads instead of len+4.
 *  - Hamming metrics: spread <= 6 after the lead-in, growth <= 1.5/step, so 8 bits
 *    suffice for 148 steps; the 292-step SCH/F trellis subtracts the minimum once.
 *  - Round 5: what the kernels run is the DIFFERENCE FORM of a step pair (tg_step_pair_d
 *    below): the two-bit step compares pm[j] with pm[j+8] + (n - 2m) -- one add and one
 *    min per butterfly --, the one-bit step after it repays the bias inside its own
 *    increments.  Same decisions, ties and history bytes; 20 packed operations per step
 *    instead of 24.  tg_acs / tg_step_a / tg_step_b remain as the arithmetic statement
 *    the table forms are checked against (tests/host_emul).
 */
#ifndef VIT_CORE_H
#define VIT_CORE_H

#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define TG_HD __host__ __device__ __forceinline__
#else
#define TG_HD static inline __attribute__((always_inline))
#endif

typedef unsigned short tg_us2 __attribute__((ext_vector_type(2)));

TG_HD tg_us2 tg_as_us2(uint32_t x) { return __builtin_bit_cast(tg_us2, x); }
TG_HD uint32_t tg_as_u32(tg_us2 x) { return __builtin_bit_cast(uint32_t, x); }
TG_HD tg_us2 tg_min(tg_us2 a, tg_us2 b) { return __builtin_elementwise_min(a, b); }

/* (a[ALO] + b[BLO], a[AHI] + b[BHI]) and the same with min: one v_pk_add_u16 / v_pk_min_u16 whose op_sel bits pick the halves.
 * Written out for the device because hipcc's instruction selection folds most but not all half swaps into op_sel (it built
 * three of the difference form's ten entry dwords a second time with v_alignbit_b32, per step pair). */
template <int ALO, int AHI, int BLO, int BHI>
TG_HD tg_us2 tg_pk_add_sel(tg_us2 a, tg_us2 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	tg_us2 d;
	asm("v_pk_add_u16 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(d) : "v"(a), "v"(b), "n"(ALO), "n"(BLO), "n"(AHI), "n"(BHI));
	return d;
#else
	return tg_us2{ (unsigned short)(a[ALO] + b[BLO]), (unsigned short)(a[AHI] + b[BHI]) };
#endif
}

template <int ALO, int AHI, int BLO, int BHI>
TG_HD tg_us2 tg_pk_min_sel(tg_us2 a, tg_us2 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	tg_us2 d;
	asm("v_pk_min_u16 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(d) : "v"(a), "v"(b), "n"(ALO), "n"(BLO), "n"(AHI), "n"(BHI));
	return d;
#else
	const unsigned short x0 = a[ALO], y0 = b[BLO], x1 = a[AHI], y1 = b[BHI];
	return tg_us2{ x0 < y0 ? x0 : y0, x1 < y1 ? x1 : y1 };
#endif
}

TG_HD uint32_t tg_pack_bytes02(uint32_t z0, uint32_t z1)
{
	/* (z0.b0, z0.b2, z1.b0, z1.b2) -> one dword; a single v_perm_b32 on gfx950 */
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(z1, z0, 0x06040200u);
#else
	return (z0 & 0xff) | ((z0 >> 8) & 0xff00) | ((z1 & 0xff) << 16) | ((z1 << 8) & 0xff000000u);
#endif
}

TG_HD uint32_t tg_brev4(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_bitreverse32(x) >> 28;
#else
	x &= 15;
	return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3);
#endif
}

#define TG_VIT_INF   0x4200u	/* metric 64 above the floor: large enough to lose, small enough not to wrap */
/* every path metric is kept at or above 2: the difference form of a two-bit step (tg_acs_d2) adds n - 2m >= -2 to a
 * predecessor's metric before the compare, and the packed arithmetic is unsigned.  A common offset changes no decision. */
#define TG_VIT_FLOOR 0x0200u

struct tg_vit_state {
	tg_us2 Z[8];
	uint32_t LP, LQ;	/* branch-metric tables of the two-bit steps, kept in VGPRs (see tg_step_a) */
};

/* keep a value in a VGPR and hide how it was computed from the optimiser */
#if defined(__HIP_DEVICE_COMPILE__)
#define TG_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define TG_OPAQUE(x) ((void)0)
#endif

/* 2-bit entries x(f), f = 0..3, at bits 8 + 2f (low half) and 24 + 2f (high half) of a table word:
 * (table >> 2f) & 0x03000300 = (lo(f), hi(f)) << 8 */
#define TG_LUTB(x0, x1, x2, x3) ((uint32_t)((x0) | ((x1) << 2) | ((x2) << 4) | ((x3) << 6)))
#define TG_LUT(lo, hi) (((lo) << 8) | ((hi) << 24))
#define TG_KMASK 0x03000300u

TG_HD void tg_vit_init(tg_vit_state &v)
{
	v.Z[0] = tg_as_us2(TG_VIT_FLOOR | (TG_VIT_INF << 16));	/* state 0: the floor (= "metric 0"), state 1: INF */
#pragma unroll
	for (int k = 1; k < 8; k++)
		v.Z[k] = tg_as_us2(TG_VIT_INF | (TG_VIT_INF << 16));
	/* f = r1 | r2 << 1:  s = r1 + r2 = 0,1,1,2   u = 1 - r1 + r2 = 1,0,2,1 */
	v.LP = TG_LUT(TG_LUTB(0, 1, 1, 2), TG_LUTB(2, 1, 1, 0));	/* (s, 2 - s) */
	v.LQ = TG_LUT(TG_LUTB(1, 0, 2, 1), TG_LUTB(1, 2, 0, 1));	/* (u, 2 - u) */
	TG_OPAQUE(v.LP);
	TG_OPAQUE(v.LQ);
}

/*
 * One add-compare-select step over the 8 butterflies.
 *   P, Q   : (m, w) << 8 for the two base output classes;  Pt, Qt = P, Q + tie in both halves
 *   SWMASK : bit j set -> butterfly j uses the swapped pair (w, m)
 *   QMASK  : bit j set -> butterfly j uses Q/Qt instead of P/Pt
 */
/* (Pyx: P with its halves swapped, handed in by callers that have it for nothing -- the table entries of tg_bm_entry carry it;
 * hipcc folds most half swaps into the packed add's op_sel bits but builds P.yx with a v_alignbit_b32 per step otherwise) */
template <unsigned SWMASK, unsigned QMASK, bool HAVE_PYX = false, typename State>
TG_HD void tg_acs(State &v, tg_us2 P, tg_us2 Pt, tg_us2 Q, tg_us2 Qt, tg_us2 Pyx = tg_us2{0, 0})
{
	tg_us2 N[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const tg_us2 za = v.Z[j >> 1], zb = v.Z[4 + (j >> 1)];
		const tg_us2 a = (j & 1) ? za.yy : za.xx;
		const tg_us2 b = (j & 1) ? zb.yy : zb.xx;
		const tg_us2 inc = ((QMASK >> j) & 1) ? Q : P;
		const tg_us2 inct = ((QMASK >> j) & 1) ? Qt : Pt;
		const bool sw = (SWMASK >> j) & 1;
		const tg_us2 incs = (HAVE_PYX && !((QMASK >> j) & 1)) ? Pyx : inc.yx;
		const tg_us2 x = a + (sw ? incs : inc);
		const tg_us2 y = b + (sw ? inct : inct.yx);
		N[j] = tg_min(x, y);
	}
#pragma unroll
	for (int j = 0; j < 8; j++)
		v.Z[j] = N[j];
}

/* out(j,0) = {0,11,6,13,5,14,3,8} (lower_mac/viterbi_cch.c:35-40): (g1,g2) per butterfly
 *   j: 0:(0,0) 1:(1,0) 2:(0,1) 3:(1,1) 4:(0,1) 5:(1,1) 6:(0,0) 7:(1,0)
 * two received bits (g1,g2): base class P = (0,0) {j=0,6}, swapped (1,1) {3,5};
 *                            base class Q = (1,0) {j=1,7}, swapped (0,1) {2,4}
 * one received bit (g1):     P = g1=0 {0,2,4,6}, swapped g1=1 {1,3,5,7}             */
#define TG_SW_A 0x3cu	/* j = 2,3,4,5 */
#define TG_Q_A  0x96u	/* j = 1,2,4,7 */
#define TG_SW_B 0xaau	/* j = 1,3,5,7 */
#define TG_Q_B  0x00u

/* step with two received bits r1 (g1), r2 (g2); t2 = 2 f, f = r1 | r2<<1; tie2 = (1<<i) * 0x10001.
 * P = (s, 2-s) << 8 (mismatches vs (0,0)), Q = (u, 2-u) << 8 (vs (1,0)): two table shifts + two masks */
TG_HD void tg_step_a(tg_vit_state &v, uint32_t t2, uint32_t tie2)
{
	const uint32_t P = (v.LP >> t2) & TG_KMASK;
	const uint32_t Q = (v.LQ >> t2) & TG_KMASK;
	tg_acs<TG_SW_A, TG_Q_A>(v, tg_as_us2(P), tg_as_us2(P + tie2), tg_as_us2(Q), tg_as_us2(Q + tie2));
}

/* step with one received bit r (g1); rm = r ? ~0 : 0 (a one-bit signed field extract).
 * P = (r, 1-r) << 8 = r ? 0x00000100 : 0x01000000 */
TG_HD void tg_step_b(tg_vit_state &v, uint32_t rm, uint32_t tie2)
{
	const uint32_t P = (rm & 0x01000100u) ^ 0x01000000u;
	tg_acs<TG_SW_B, TG_Q_B>(v, tg_as_us2(P), tg_as_us2(P + tie2), tg_as_us2(P), tg_as_us2(P + tie2));
}

/* bit k of x as a mask (v_bfe_i32) */
TG_HD uint32_t tg_bitmask(uint32_t x, int k)
{
	return (uint32_t)((int32_t)(x << (31 - k)) >> 31);
}

/*
 * The K-1 = 4 flush steps (steps 4..7 of the last block; nothing received, all branch metrics 0) as a
 * min tree.  A flush step gives new[2j] = new[2j+1] = min(pm[j], pm[j+8] + tie), so after it only 8
 * distinct values exist, then 4, 2, 1: the traceback starts in state 0, and state 0 after the last
 * step is
 *     m1[j] = min(pm[j], pm[j+8] + T4)   m2[a] = min(m1[a], m1[a+4] + T5)   m3[b] = min(m2[b], m2[b+2] + T6)
 *     final = min(m3[0], m3[1] + T7)
 * with exactly the candidates, tie bits and order of the four full steps.  8 adds + 8 mins instead of
 * 4 x 24; only Z[0]'s low half (state 0) is meaningful afterwards.
 */
TG_HD void tg_flush4(tg_vit_state &v)
{
	const tg_us2 T4 = tg_as_us2(0x00100010u), T5 = tg_as_us2(0x00200020u);
	const tg_us2 T6 = tg_as_us2(0x00400040u), T7 = tg_as_us2(0x00800080u);
	tg_us2 M[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		M[k] = tg_min(v.Z[k], v.Z[k + 4] + T4);		/* (m1[2k], m1[2k+1]) */
	const tg_us2 N0 = tg_min(M[0], M[2] + T5);		/* (m2[0], m2[1]) */
	const tg_us2 N1 = tg_min(M[1], M[3] + T5);		/* (m2[2], m2[3]) */
	const tg_us2 R = tg_min(N0, N1 + T6);			/* (m3[0], m3[1]) */
	v.Z[0] = tg_min(R, (R + T7).yx);			/* low half: state 0 */
}

TG_HD void tg_vit_clean(tg_vit_state &v)
{
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = tg_as_us2(tg_as_u32(v.Z[k]) & 0xff00ff00u);
}

/* the 4 lead-in steps: 6 received bits in bits 0..5 of 'six' */
TG_HD void tg_vit_leadin(tg_vit_state &v, uint32_t six)
{
	tg_step_a(v, (six << 1) & 6, 0x00010001u);
	tg_step_b(v, tg_bitmask(six, 2), 0x00020002u);
	tg_step_a(v, (six >> 2) & 6, 0x00040004u);
	tg_step_b(v, tg_bitmask(six, 5), 0x00080008u);
	tg_vit_clean(v);
}

/* one 8-step block on 12 received bits (bits 0..11 of 'tw'); LAST: only the first
 * 4 steps receive bits (6), the other 4 are the K-1 flush steps. */
template <bool LAST>
TG_HD void tg_vit_block(tg_vit_state &v, uint32_t tw, uint32_t h[4])
{
	tg_step_a(v, (tw << 1) & 6, 0x00010001u);
	tg_step_b(v, tg_bitmask(tw, 2), 0x00020002u);
	tg_step_a(v, (tw >> 2) & 6, 0x00040004u);
	tg_step_b(v, tg_bitmask(tw, 5), 0x00080008u);
	if (LAST) {
		tg_flush4(v);
	} else {
		tg_step_a(v, (tw >> 5) & 6, 0x00100010u);
		tg_step_b(v, tg_bitmask(tw, 8), 0x00200020u);
		tg_step_a(v, (tw >> 8) & 6, 0x00400040u);
		tg_step_b(v, tg_bitmask(tw, 11), 0x00800080u);
	}
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
	tg_vit_clean(v);
}

/*
 * Branch metrics from a table instead of arithmetic.  The received bits enter the trellis in triples: (g1, g2)
 * of a two-bit step and g1 of the one-bit step after it, bits 3p .. 3p+2 of a 12-bit block word for step pair
 * p = 0..3.  Everything tg_step_a / tg_step_b derive from such a triple e and the pair's tie bits --
 * P, P+tie, Q, Q+tie for the first step, P', P'+tie' for the second -- is six dwords: 4 pairs x 8 triples x
 * 32 bytes = 1 KB, held in LDS by the kernels (entries of one pair lie in eight different bank groups, lanes
 * with equal triples read the same address: no bank conflicts).  Per step pair the vector unit then does one
 * field extract and one shift for the address and two LDS reads instead of twelve ALU operations.
 */
#define TG_BM_WORDS (4 * 8 * 8)

TG_HD void tg_bm_entry(int p, uint32_t e, uint32_t w[8])
{
	const uint32_t LP = TG_LUT(TG_LUTB(0, 1, 1, 2), TG_LUTB(2, 1, 1, 0));
	const uint32_t LQ = TG_LUT(TG_LUTB(1, 0, 2, 1), TG_LUTB(1, 2, 0, 1));
	const uint32_t t2 = 2 * (e & 3);
	const uint32_t tie_a = 0x00010001u << (2 * p), tie_b = tie_a << 1;
	w[0] = (LP >> t2) & TG_KMASK;
	w[1] = w[0] + tie_a;
	w[2] = (LQ >> t2) & TG_KMASK;
	w[3] = w[2] + tie_a;
	w[4] = (e & 4) ? 0x00000100u : 0x01000000u;
	w[5] = w[4] + tie_b;
	w[6] = (w[0] >> 16) | (w[0] << 16);	/* P and P' with their halves swapped (tg_step_pair8) */
	w[7] = (w[4] >> 16) | (w[4] << 16);
}

TG_HD void tg_bm_build(uint32_t *tab)
{
	for (int p = 0; p < 4; p++)
		for (uint32_t e = 0; e < 8; e++)
			tg_bm_entry(p, e, tab + (8 * p + e) * 8);
}

/* one step pair from its table entry */
TG_HD void tg_step_pair(tg_vit_state &v, const uint32_t w[6])
{
	tg_acs<TG_SW_A, TG_Q_A>(v, tg_as_us2(w[0]), tg_as_us2(w[1]), tg_as_us2(w[2]), tg_as_us2(w[3]));
	tg_acs<TG_SW_B, TG_Q_B>(v, tg_as_us2(w[4]), tg_as_us2(w[5]), tg_as_us2(w[4]), tg_as_us2(w[5]));
}

/* the same from all eight dwords of the entry: the swapped forms come from the table */
TG_HD void tg_step_pair8(tg_vit_state &v, const uint32_t w[8])
{
	tg_acs<TG_SW_A, TG_Q_A, true>(v, tg_as_us2(w[0]), tg_as_us2(w[1]), tg_as_us2(w[2]), tg_as_us2(w[3]), tg_as_us2(w[6]));
	tg_acs<TG_SW_B, TG_Q_B, true>(v, tg_as_us2(w[4]), tg_as_us2(w[5]), tg_as_us2(w[4]), tg_as_us2(w[5]), tg_as_us2(w[7]));
}

/* bm(p, e, w): fetch the dwords of pair p, triple e into w[8] (six of them, or all eight for BM8: then the swapped forms come
 * from the table -- one v_alignbit_b32 less per step, two more registers per pair in flight; k_vit takes it where it has the
 * registers).  Same results as tg_vit_leadin / tg_vit_block. */
template <bool BM8 = false, typename Bm>
TG_HD void tg_vit_leadin_bm(tg_vit_state &v, uint32_t six, Bm bm)
{
	uint32_t w[2][8];
	bm(0, six & 7, w[0]);
	bm(1, (six >> 3) & 7, w[1]);
	if (BM8) {
		tg_step_pair8(v, w[0]);
		tg_step_pair8(v, w[1]);
	} else {
		tg_step_pair(v, w[0]);
		tg_step_pair(v, w[1]);
	}
	tg_vit_clean(v);
}

template <bool LAST, bool BM8 = false, typename Bm>
TG_HD void tg_vit_block_bm(tg_vit_state &v, uint32_t tw, uint32_t h[4], Bm bm)
{
	constexpr int NP = LAST ? 2 : 4;
	uint32_t w[NP][8];
#pragma unroll
	for (int p = 0; p < NP; p++)
		bm(p, (tw >> (3 * p)) & 7, w[p]);
#pragma unroll
	for (int p = 0; p < NP; p++) {
		if (BM8)
			tg_step_pair8(v, w[p]);
		else
			tg_step_pair(v, w[p]);
	}
	if (LAST)
		tg_flush4(v);
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
	tg_vit_clean(v);
}

/*
 * Difference form of a step pair (round 5): 40 packed operations per pair instead of 48.
 *
 * Subtract the j candidate's increment from BOTH candidates of a butterfly and the comparison -- metrics, tie bit, history
 * byte -- is the same one, but the j candidate needs no add:
 *     S[2j]   = min(pm[j], pm[j+8] + (n - 2m) + tie)        = new[2j]   - m
 *     S[2j+1] = min(pm[j], pm[j+8] - (n - 2m) + tie)        = new[2j+1] - (n - m)
 * One v_pk_add_u16 + one v_pk_min_u16 per butterfly (tg_acs_d2), and the states are left short of their true metrics by a
 * bias beta(2j) = m_j, beta(2j+1) = n - m_j that depends on the received bits and on the state, not on the path.  The next
 * step repays it inside its own increments (tg_acs_d3: two adds and a min as before, the increments being
 * beta(k) + m'_k, beta(k) + 1 - m'_k for the k candidate and beta(k+8) + 1 - m'_k, beta(k+8) + m'_k for the k+8 candidate).
 * The two-bit step of a pair runs in the difference form, the one-bit step after it repays: its increments are functions of
 * the pair's received triple alone, every butterfly's pair of them is one of four dwords or its half swap (op_sel), and
 * after the pair the states hold their true metrics again -- blocks, history extraction, normalisation and the flush steps
 * see what they saw before.  (Letting the bias run on instead makes every later increment depend on the last four steps'
 * received bits: eight different dwords per step and lane, no static selection -- more work, not less.)
 * The unsigned packed add needs pm[j+8] + (n - 2m) >= 0: all metrics are kept at or above TG_VIT_FLOOR.
 *
 * With s = r1 + r2 (mismatches of a two-bit step's bits against (0,0)), u = 1 - r1 + r2 (against (1,0)), r the one-bit
 * step's bit, tA / tB the pair's two tie bits (in both halves):
 *   w[0] = U  = ( 2 - 2s,  2s - 2) + tA      butterflies 0, 6; swapped: 3, 5        (TG_SW_A / TG_Q_A as in tg_step_a)
 *   w[1] = V  = ( 2 - 2u,  2u - 2) + tA      butterflies 1, 7; swapped: 2, 4
 *   A1..A4 = (b + r, b + 1 - r) for b = s, 2 - s, u, 2 - u                          (b = the bias of the predecessor)
 *   w[2 + 2g] = A(g+1), w[3 + 2g] = A(4-g) + tB, g = 0..3: what butterflies g and 7 - g of the one-bit step take -- the
 *   k candidate w[2 + 2g] (half-swapped for odd k), the k+8 candidate w[3 + 2g] (half-swapped for even k) -- so a kernel
 *   short of registers can take the entry in as it goes (dwords 0..3, 4..7, 8..9)
 * all << 8 in each half.  Entry = ten dwords; in LDS as three arrays with a stride of 16 bytes per entry (the third holds 8) so
 * that the eight entries of a pair stay within 32 banks per read instruction and one address serves the three reads.
 */
#define TG_BMD_WORDS (3 * 32 * 4)
#define TG_BMD_A0 0		/* dwords 0..3 of entry q at 4 q       */
#define TG_BMD_A1 (32 * 4)	/* dwords 4..7 of entry q at 128 + 4 q  */
#define TG_BMD_A2 (32 * 8)	/* dwords 8..9 of entry q at 256 + 4 q (one 16-byte stride for all three: one address per entry) */

TG_HD void tg_bmd_entry(int p, uint32_t e, uint32_t w[10])
{
	const uint32_t r1 = e & 1, r2 = (e >> 1) & 1, r = (e >> 2) & 1;
	const int s = (int)(r1 + r2), u = (int)(1 - r1 + r2);
	const uint32_t tA = 0x00010001u << (2 * p), tB = tA << 1;
	auto pk = [](int lo, int hi) -> uint32_t { return (((uint32_t)lo << 8) & 0xffffu) | (((uint32_t)hi << 8) << 16); };
	w[0] = pk(2 - 2 * s, 2 * s - 2) + tA;
	w[1] = pk(2 - 2 * u, 2 * u - 2) + tA;
	const int b[4] = { s, 2 - s, u, 2 - u };
	for (int g = 0; g < 4; g++) {
		w[2 + 2 * g] = pk(b[g] + (int)r, b[g] + 1 - (int)r);
		w[3 + 2 * g] = pk(b[3 - g] + (int)r, b[3 - g] + 1 - (int)r) + tB;
	}
}

/* the table as the kernels hold it (TG_BMD_WORDS dwords) */
TG_HD void tg_bmd_store(uint32_t *tab, int q, const uint32_t w[10])
{
	for (int i = 0; i < 4; i++) {
		tab[TG_BMD_A0 + 4 * q + i] = w[i];
		tab[TG_BMD_A1 + 4 * q + i] = w[4 + i];
	}
	tab[TG_BMD_A2 + 4 * q] = w[8];
	tab[TG_BMD_A2 + 4 * q + 1] = w[9];
	tab[TG_BMD_A2 + 4 * q + 2] = 0;
	tab[TG_BMD_A2 + 4 * q + 3] = 0;
}

TG_HD void tg_bmd_build(uint32_t *tab)
{
	for (int p = 0; p < 4; p++)
		for (uint32_t e = 0; e < 8; e++) {
			uint32_t w[10];
			tg_bmd_entry(p, e, w);
			tg_bmd_store(tab, 8 * p + (int)e, w);
		}
}

/* the difference form's precondition, checked where this header is compiled for the host (tests/host_emul): every
 * metric a two-bit step adds n - 2m >= -2 to sits at or above TG_VIT_FLOOR -- a caller that normalised with tg_vit_normalize()
 * (minimum back to 0) instead of tg_vit_normalize_floor() would wrap the unsigned packed add without any other sign.  The
 * emulation counts violations (tg_vit_floor_violations, read by the CPU tests); device code pays nothing. */
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(__HIP__) && !defined(__HIPCC__)
static unsigned long tg_vit_floor_violations;
#define TG_VIT_CHECK_FLOOR(v) do { for (int k_ = 4; k_ < 8; k_++) for (int h_ = 0; h_ < 2; h_++) \
		if (((v).Z[k_][h_] & 0xff00u) < TG_VIT_FLOOR) tg_vit_floor_violations++; } while (0)
#else
#define TG_VIT_CHECK_FLOOR(v) do { } while (0)
#endif

/* the two-bit step in the difference form: U, V as above */
template <unsigned SWMASK, unsigned QMASK>
TG_HD void tg_acs_d2(tg_vit_state &v, tg_us2 U, tg_us2 V)
{
	TG_VIT_CHECK_FLOOR(v);
	tg_us2 N[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const tg_us2 za = v.Z[j >> 1], zb = v.Z[4 + (j >> 1)];
		const tg_us2 D = ((QMASK >> j) & 1) ? V : U;
		const bool sw = (SWMASK >> j) & 1;
		tg_us2 y;
		if (j & 1)
			y = sw ? tg_pk_add_sel<1, 1, 1, 0>(zb, D) : tg_pk_add_sel<1, 1, 0, 1>(zb, D);
		else
			y = sw ? tg_pk_add_sel<0, 0, 1, 0>(zb, D) : tg_pk_add_sel<0, 0, 0, 1>(zb, D);
		N[j] = (j & 1) ? tg_pk_min_sel<1, 1, 0, 1>(za, y) : tg_pk_min_sel<0, 0, 0, 1>(za, y);
	}
#pragma unroll
	for (int j = 0; j < 8; j++)
		v.Z[j] = N[j];
}

/* the one-bit step that repays the bias, from dwords 2..9 of the entry: butterflies in the order 0, 7, 1, 6, 2, 5, 3, 4 */
TG_HD void tg_acs_d3(tg_vit_state &v, const uint32_t *w)
{
	tg_us2 N[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const int g = i >> 1, k = (i & 1) ? 7 - g : g;
		const tg_us2 za = v.Z[k >> 1], zb = v.Z[4 + (k >> 1)];
		const tg_us2 A = tg_as_us2(w[2 + 2 * g]), Bt = tg_as_us2(w[3 + 2 * g]);
		tg_us2 x, y;
		if (k & 1) {
			x = tg_pk_add_sel<1, 1, 1, 0>(za, A);
			y = tg_pk_add_sel<1, 1, 0, 1>(zb, Bt);
		} else {
			x = tg_pk_add_sel<0, 0, 0, 1>(za, A);
			y = tg_pk_add_sel<0, 0, 1, 0>(zb, Bt);
		}
		N[k] = tg_min(x, y);
	}
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = N[k];
}

TG_HD void tg_step_pair_d(tg_vit_state &v, const uint32_t w[10])
{
	tg_acs_d2<TG_SW_A, TG_Q_A>(v, tg_as_us2(w[0]), tg_as_us2(w[1]));
	tg_acs_d3(v, w);
}

/* bmd(p, o, w): fetch the ten dwords of pair p, triple e, o = 16 e (the entry's byte offset inside its array's part for pair p:
 * a shift and a mask per pair, the rest of the address is the reads' immediate offset).  Same results as tg_vit_leadin /
 * tg_vit_block. */
template <typename Bm>
TG_HD void tg_vit_leadin_bmd(tg_vit_state &v, uint32_t six, Bm bmd)
{
	uint32_t w[2][10];
	bmd(0, (six << 4) & 0x70u, w[0]);
	bmd(1, (six << 1) & 0x70u, w[1]);
	tg_step_pair_d(v, w[0]);
	tg_step_pair_d(v, w[1]);
	tg_vit_clean(v);
}

template <bool LAST, bool JIT = false, typename Bm>
TG_HD void tg_vit_block_bmd(tg_vit_state &v, uint32_t tw, uint32_t h[4], Bm bmd)
{
	constexpr int NP = LAST ? 2 : 4;
	if (JIT) {	/* a kernel short of registers: one entry at a time in the source (the scheduler moves the reads up as far as it has room) */
#pragma unroll
		for (int p = 0; p < NP; p++) {
			uint32_t w[10];
			bmd(p, (p < 2 ? tw << (4 - 3 * p) : tw >> (3 * p - 4)) & 0x70u, w);
			tg_step_pair_d(v, w);
		}
	} else {
		uint32_t w[NP][10];
#pragma unroll
		for (int p = 0; p < NP; p++)
			bmd(p, (p < 2 ? tw << (4 - 3 * p) : tw >> (3 * p - 4)) & 0x70u, w[p]);
#pragma unroll
		for (int p = 0; p < NP; p++)
			tg_step_pair_d(v, w[p]);
	}
	if (LAST)
		tg_flush4(v);
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
	tg_vit_clean(v);
}

/* subtract the smallest path metric from all 16 (only between blocks: low bytes are clear) */
template <typename State>
TG_HD void tg_vit_normalize(State &v)
{
	tg_us2 m01 = tg_min(v.Z[0], v.Z[1]), m23 = tg_min(v.Z[2], v.Z[3]);
	tg_us2 m45 = tg_min(v.Z[4], v.Z[5]), m67 = tg_min(v.Z[6], v.Z[7]);
	tg_us2 m = tg_min(tg_min(m01, m23), tg_min(m45, m67));
	m = tg_min(m, m.yx);
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = v.Z[k] - m;
}

/* the same for the hard trellis: the smallest metric goes back to the floor (TG_VIT_FLOOR), not to 0 */
TG_HD void tg_vit_normalize_floor(tg_vit_state &v)
{
	tg_us2 m01 = tg_min(v.Z[0], v.Z[1]), m23 = tg_min(v.Z[2], v.Z[3]);
	tg_us2 m45 = tg_min(v.Z[4], v.Z[5]), m67 = tg_min(v.Z[6], v.Z[7]);
	tg_us2 m = tg_min(tg_min(m01, m23), tg_min(m45, m67));
	m = tg_min(m, m.yx) - tg_as_us2(TG_VIT_FLOOR * 0x10001u);
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = v.Z[k] - m;
}

/* =========================================================================================
 * Generic step (k_conv): up to three received bits g1, g2, g3 of one trellis step, each possibly absent
 * (punctured) or erased, on either mother code.  Same state layout, tie rule and history as above.
 *
 * T_k (k = 1..3) = (mismatch if generator k's expected bit is 1, mismatch if it is 0) << 8 in (low, high)
 * half = 0x00000100 for a received 0, 0x01000000 for a received 1, 0 for nothing.  For a butterfly whose
 * out(j,0) has generator bits (e1,e2,e3) the pair (m, n - m) is sum_k (e_k ? T_k : swap(T_k)); complementing
 * all e_k swaps the pair, so four sums (e3 = 0) serve the eight butterflies; the swaps are op_sel bits.
 * CODE 0: rate-1/4 code (lower_mac/viterbi_cch.c:35-40), g1..g3 of out(j,0) = {0,11,6,13,5,14,3,8} >> 1
 * CODE 1: rate-1/3 speech code (lower_mac/viterbi_tch.c:34-39), out(j,0) = {0,6,5,3,6,0,3,5}
 * Both have 1 and D^4 in every generator, which is all the butterfly symmetry needs.
 * ========================================================================================= */
template <int CODE, bool G3>
TG_HD void tg_step_gen(tg_vit_state &v, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t tie2)
{
	const tg_us2 T1 = tg_as_us2(t1), T2 = tg_as_us2(t2), T3s = tg_as_us2(t3).yx, tie = tg_as_us2(tie2);
	const tg_us2 A = T1.yx + T2.yx, B = T1 + T2.yx;
	tg_us2 Cn[4], Ct[4];
	if (G3) {
		Cn[0] = A + T3s;	/* (e1,e2) = (0,0) */
		Cn[1] = B.yx + T3s;	/* (0,1) */
		Cn[2] = B + T3s;	/* (1,0) */
		Cn[3] = A.yx + T3s;	/* (1,1) */
	} else {			/* a puncturer that never keeps g3 (2/3, 292/432, 8/12) */
		Cn[0] = A;
		Cn[1] = B.yx;
		Cn[2] = B;
		Cn[3] = A.yx;
	}
#pragma unroll
	for (int q = 0; q < 4; q++)
		Ct[q] = Cn[q] + tie;
	constexpr unsigned pat_cch[8] = { 0, 5, 3, 6, 2, 7, 1, 4 };
	constexpr unsigned pat_tch[8] = { 0, 6, 5, 3, 6, 0, 3, 5 };
	tg_us2 N[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const unsigned p = CODE ? pat_tch[j] : pat_cch[j];
		const bool sw = p & 1;
		const unsigned q = ((sw ? ~p : p) >> 1) & 3;
		const tg_us2 za = v.Z[j >> 1], zb = v.Z[4 + (j >> 1)];
		const tg_us2 a = (j & 1) ? za.yy : za.xx;
		const tg_us2 b = (j & 1) ? zb.yy : zb.xx;
		const tg_us2 x = a + (sw ? Cn[q].yx : Cn[q]);
		const tg_us2 y = b + (sw ? Ct[q] : Ct[q].yx);
		N[j] = tg_min(x, y);
	}
#pragma unroll
	for (int j = 0; j < 8; j++)
		v.Z[j] = N[j];
}

/*
 * Received bytes are held as 2-bit classes, four to a byte (byte k of a dword in bits 2k..2k+1):
 * 0 = erased (0xff), 1 = a 0 bit (0x00), 2 = a 1 bit (anything else) -- lower_mac/viterbi.c:12-22.
 * SWAR: bit 7 of each byte of nz / nf says "byte != 0" / "byte != 0xff".
 */
TG_HD uint32_t tg_conv_pack4(uint32_t x)
{
	const uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
	const uint32_t y = ~x;
	const uint32_t nf = (((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;
	const uint32_t q = ((nz ^ 0x80808080u) >> 7) | ((nz & nf) >> 6);
	return (q | (q >> 6) | (q >> 12) | (q >> 18)) & 0xffu;
}

/* T of a received position from the class byte that holds it and its descriptor dword (tg_conv.h): class bit 0
 * -> bit 8, class bit 1 -> bit 24 with one 24-bit multiply by 0x800100 >> shift (the two shifted copies of the
 * byte cannot overlap; the multiply ignores the descriptor's top byte) and one mask */
TG_HD uint32_t tg_conv_t(uint32_t classbyte, uint32_t desc)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul24(classbyte, desc) & 0x01000100u;
#else
	return (classbyte * (desc & 0xffffffu)) & 0x01000100u;
#endif
}

/* one history block of the generic trellis: NST (0: nst_rt, at run time) steps described by desc[] (three
 * dwords per step, tg_conv.h); fetch(q) returns class byte q (received positions 4q .. 4q+3) of this lane's
 * type-3 block.  The descriptors are wave-uniform (scalar loads); per generator and step the vector unit does one
 * address add, one LDS byte read, one multiply and one mask, whether or not that position was punctured.
 * Two phases, so that the block's reads are in flight before the first add-compare-select needs one. */
template <int CODE, bool G3, int NST, typename Fetch>
TG_HD void tg_conv_block(tg_vit_state &v, const uint32_t *desc, int nst_rt, Fetch fetch, uint32_t h[4])
{
	const int nst = NST ? NST : nst_rt;
	uint32_t t[8][3];
#pragma unroll
	for (int i = 0; i < 8; i++) {
#pragma unroll
		for (int k = 0; k < 3; k++) {
			t[i][k] = 0;
			if (k < 2 || G3) {
				const uint32_t d = (i < nst) ? desc[3 * i + k] : 0u;
				t[i][k] = tg_conv_t(fetch(d >> 24), d);
			}
		}
	}
#pragma unroll
	for (int i = 0; i < 8; i++) {
		if (i < nst)
			tg_step_gen<CODE, G3>(v, t[i][0], t[i][1], t[i][2], 0x00010001u << i);
	}
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = tg_pack_bytes02(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1]));
	tg_vit_clean(v);
}

/* =========================================================================================
 * Soft-input trellis (BASELINE config 5; the reference has no soft path, the definition is ours:
 * int8 soft values, positive = bit 0, correlation metrics, same tie rule -- what libosmocore's
 * accelerated decoder computes when it is handed soft values).
 *
 * One state per 32-bit word: metric * 256 + survivor byte.  Butterfly j with correlation
 * M = sum over received values of value * (+1 if out(j,0) bit is 0, -1 if 1):
 *     new[2j]   = max(pm[j] + M, pm[j+8] - M)        new[2j+1] = max(pm[j] - M, pm[j+8] + M)
 * The candidate from predecessor j gets bit i of the low byte added: on equal metrics it is larger,
 * so the tie goes to j (oldest bit 0) and the winner's bit i says "came from j".  The decoded bit is
 * the predecessor's oldest bit, i.e. the complement; the byte is inverted when it is extracted, which
 * makes the history format identical to the hard trellis (same traceback, same CRC, same outputs).
 * |metric| <= 292 * 254 < 2^17, so metric * 256 fits 32 bits with room for a -2^28 "unreachable".
 * ========================================================================================= */
struct tg_svit_state {
	int32_t Z[16];
};

#define TG_SVIT_NEG (-(1 << 28))

TG_HD void tg_svit_init(tg_svit_state &v)
{
	v.Z[0] = 0;
#pragma unroll
	for (int s = 1; s < 16; s++)
		v.Z[s] = TG_SVIT_NEG;
}

TG_HD int32_t tg_smax(int32_t a, int32_t b) { return a > b ? a : b; }

/* CLS[j]: 0 -> M = +X, 1 -> M = -X, 2 -> M = +Y, 3 -> M = -Y   (X, Y already scaled by 256) */
template <unsigned C0, unsigned C1, unsigned C2, unsigned C3, unsigned C4, unsigned C5, unsigned C6, unsigned C7>
TG_HD void tg_sacs(tg_svit_state &v, int32_t X, int32_t Y, int32_t tie)
{
	const unsigned cls[8] = { C0, C1, C2, C3, C4, C5, C6, C7 };
	const int32_t pos[2] = { X, Y }, neg[2] = { -X, -Y };
	const int32_t post[2] = { X + tie, Y + tie }, negt[2] = { -X + tie, -Y + tie };
	int32_t N[16];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const int k = cls[j] >> 1;
		const bool inv = cls[j] & 1;
		const int32_t a = v.Z[j], b = v.Z[j + 8];
		/* M = pos[k] (inv: neg[k]) */
		N[2 * j] = tg_smax(a + (inv ? negt[k] : post[k]), b + (inv ? pos[k] : neg[k]));
		N[2 * j + 1] = tg_smax(a + (inv ? post[k] : negt[k]), b + (inv ? neg[k] : pos[k]));
	}
#pragma unroll
	for (int s = 0; s < 16; s++)
		v.Z[s] = N[s];
}

/* two received values a (g1), b (g2): (g1,g2) per butterfly as in the hard trellis:
 * j: 0:(0,0) 1:(1,0) 2:(0,1) 3:(1,1) 4:(0,1) 5:(1,1) 6:(0,0) 7:(1,0);  X = a+b, Y = b-a */
TG_HD void tg_sstep_a(tg_svit_state &v, int32_t a, int32_t b, int32_t tie)
{
	tg_sacs<0, 2, 3, 1, 3, 1, 0, 2>(v, (a + b) * 256, (b - a) * 256, tie);
}

/* one received value a (g1): g1 of butterfly j is j & 1 */
TG_HD void tg_sstep_b(tg_svit_state &v, int32_t a, int32_t tie)
{
	tg_sacs<0, 1, 0, 1, 0, 1, 0, 1>(v, a * 256, 0, tie);
}

TG_HD void tg_sstep_flush(tg_svit_state &v, int32_t tie)
{
	tg_sacs<0, 0, 0, 0, 0, 0, 0, 0>(v, 0, 0, tie);
}

TG_HD void tg_svit_clean(tg_svit_state &v)
{
#pragma unroll
	for (int s = 0; s < 16; s++)
		v.Z[s] &= ~0xff;
}

/* signed byte k of a 3-dword group, sign flipped where the scrambling mask bit is set */
TG_HD int32_t tg_soft_val(const uint32_t w[3], int k, uint32_t maskbits)
{
	const int32_t s = (int32_t)(int8_t)(w[k >> 2] >> ((k & 3) * 8));
	return ((maskbits >> k) & 1) ? -s : s;
}

TG_HD void tg_svit_leadin(tg_svit_state &v, const uint32_t w[2], uint32_t mask6)
{
	const uint32_t ww[3] = { w[0], w[1], 0 };
	tg_sstep_a(v, tg_soft_val(ww, 0, mask6), tg_soft_val(ww, 1, mask6), 1);
	tg_sstep_b(v, tg_soft_val(ww, 2, mask6), 2);
	tg_sstep_a(v, tg_soft_val(ww, 3, mask6), tg_soft_val(ww, 4, mask6), 4);
	tg_sstep_b(v, tg_soft_val(ww, 5, mask6), 8);
	tg_svit_clean(v);
}

template <bool LAST>
TG_HD void tg_svit_block(tg_svit_state &v, const uint32_t w[3], uint32_t mask12, uint32_t h[4])
{
	tg_sstep_a(v, tg_soft_val(w, 0, mask12), tg_soft_val(w, 1, mask12), 1);
	tg_sstep_b(v, tg_soft_val(w, 2, mask12), 2);
	tg_sstep_a(v, tg_soft_val(w, 3, mask12), tg_soft_val(w, 4, mask12), 4);
	tg_sstep_b(v, tg_soft_val(w, 5, mask12), 8);
	if (LAST) {
		tg_sstep_flush(v, 16);
		tg_sstep_flush(v, 32);
		tg_sstep_flush(v, 64);
		tg_sstep_flush(v, 128);
	} else {
		tg_sstep_a(v, tg_soft_val(w, 6, mask12), tg_soft_val(w, 7, mask12), 16);
		tg_sstep_b(v, tg_soft_val(w, 8, mask12), 32);
		tg_sstep_a(v, tg_soft_val(w, 9, mask12), tg_soft_val(w, 10, mask12), 64);
		tg_sstep_b(v, tg_soft_val(w, 11, mask12), 128);
	}
	/* history bytes, inverted: bit = 1 then means "predecessor j+8", as in the hard trellis */
#pragma unroll
	for (int d = 0; d < 4; d++)
		h[d] = ~(((uint32_t)v.Z[4 * d] & 0xff) | (((uint32_t)v.Z[4 * d + 1] & 0xff) << 8) |
			 (((uint32_t)v.Z[4 * d + 2] & 0xff) << 16) | (((uint32_t)v.Z[4 * d + 3] & 0xff) << 24));
	tg_svit_clean(v);
}

/* =========================================================================================
 * The same soft-input trellis in packed 16-bit arithmetic (round 2; what the kernels run -- tg_svit_* above stays
 * as the 32-bit statement of the definition, and the host tests check one against the other and both against the
 * oracle).
 *
 * Correlation maximised = mismatch cost minimised: a received value x costs |x| on a branch whose expected bit
 * disagrees with its sign and 0 otherwise, i.e. cost = (sum |x| - correlation) / 2 with the same sum on every
 * branch of a step, so every comparison -- ties included -- comes out as in tg_sacs.  Costs are non-negative, which
 * gives the hard trellis' word again: per 16-bit half [15:4] path metric, [3:0] the last <= 4 decisions of the
 * survivor; tg_acs unchanged (two v_pk_add_u16 + one v_pk_min_u16 per butterfly), tie bit i of the current
 * FOUR-step block on the candidate from predecessor j + 8.
 *
 *  - branch metrics: T(x) = (cost against an expected 0, cost against an expected 1) << 4 in (low, high) half =
 *    (max(-x, 0), max(x, 0)) << 4, from a 512-entry table indexed by flip << 8 | byte (flip = the scrambling
 *    sequence's bit: the value changes sign; -(-128) = 128 is exact in the table).  Two-value step:
 *    P = Ta + Tb (class (0,0)), Q = swap(Ta) + Tb (class (1,0)); one-value step: P = T.
 *  - range: a state's metric exceeds the smallest one by at most the six values of the last four steps, 6 x 128 =
 *    768; sixteen steps add at most 24 x 128 = 3072; 3840 < 4096, so subtracting the minimum once per sixteen steps
 *    keeps twelve bits exact.  Unreachable start states: 1024 (> 768, the largest cost a real path has after the
 *    lead-in).
 *  - history: four decisions per state and four-step block = the block's start state bit-reversed (decision k is
 *    input bit k - 4).  Sixteen nibbles = two dwords per block, state s at nibble rotr4(brev4(s)) of the 64 bits
 *    (what one v_perm_b32 per register pair, one mask and one shift-or per dword produce); the traceback hops
 *    nibble -> 64-bit shift -> nibble.
 * ========================================================================================= */
struct tg_pvit_state {
	tg_us2 Z[8];
};

#define TG_PSOFT_TAB 512
#define TG_PVIT_INF 0x4000u	/* an unreachable start state of the soft trellis: 1024 << 4 (its own constant: the hard trellis' TG_VIT_INF carries the floor) */

TG_HD uint32_t tg_psoft_entry(uint32_t idx)
{
	int32_t x = (int32_t)(int8_t)(idx & 0xff);
	if (idx & 0x100)
		x = -x;
	const uint32_t c0 = x < 0 ? (uint32_t)-x : 0u, c1 = x > 0 ? (uint32_t)x : 0u;
	return (c0 << 4) | (c1 << 20);
}

TG_HD void tg_pvit_init(tg_pvit_state &v)
{
	v.Z[0] = tg_as_us2(TG_PVIT_INF << 16);	/* state 0: metric 0; the others: 1024 << 4 */
#pragma unroll
	for (int k = 1; k < 8; k++)
		v.Z[k] = tg_as_us2(TG_PVIT_INF | (TG_PVIT_INF << 16));
}

TG_HD void tg_pstep_a(tg_pvit_state &v, uint32_t ta, uint32_t tb, uint32_t tie2)
{
	const tg_us2 Ta = tg_as_us2(ta), Tb = tg_as_us2(tb), tie = tg_as_us2(tie2);
	const tg_us2 P = Ta + Tb, Q = Ta.yx + Tb;
	tg_acs<TG_SW_A, TG_Q_A>(v, P, P + tie, Q, Q + tie);
}

TG_HD void tg_pstep_b(tg_pvit_state &v, uint32_t t, uint32_t tie2)
{
	const tg_us2 P = tg_as_us2(t), Pt = P + tg_as_us2(tie2);
	tg_acs<TG_SW_B, TG_Q_B>(v, P, Pt, P, Pt);
}

/* table index of value k of a group of dwords */
TG_HD uint32_t tg_psoft_idx(const uint32_t *w, int k, uint32_t maskbits)
{
	return ((w[k >> 2] >> ((k & 3) * 8)) & 0xff) | (((maskbits >> k) & 1) << 8);
}

/* the table entries of values K0 .. K0 + N - 1 of the group w[] (flips: bits K0.. of maskbits) */
template <int K0, int N, typename Tab>
TG_HD void tg_psoft_fetch(const uint32_t *w, uint32_t maskbits, Tab tab, uint32_t *t)
{
#pragma unroll
	for (int i = 0; i < N; i++)
		t[i] = tab(tg_psoft_idx(w, K0 + i, maskbits));
}

/* four steps on six fetched table entries */
TG_HD void tg_pvit_quad(tg_pvit_state &v, const uint32_t t[6])
{
	tg_pstep_a(v, t[0], t[1], 0x00010001u);
	tg_pstep_b(v, t[2], 0x00020002u);
	tg_pstep_a(v, t[3], t[4], 0x00040004u);
	tg_pstep_b(v, t[5], 0x00080008u);
}

/* the four flush steps as a min tree (tg_flush4 with the tie bits of a four-step block) */
TG_HD void tg_pflush4(tg_pvit_state &v)
{
	const tg_us2 T0 = tg_as_us2(0x00010001u), T1 = tg_as_us2(0x00020002u);
	const tg_us2 T2 = tg_as_us2(0x00040004u), T3 = tg_as_us2(0x00080008u);
	tg_us2 M[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		M[k] = tg_min(v.Z[k], v.Z[k + 4] + T0);
	const tg_us2 N0 = tg_min(M[0], M[2] + T1);
	const tg_us2 N1 = tg_min(M[1], M[3] + T1);
	const tg_us2 R = tg_min(N0, N1 + T2);
	v.Z[0] = tg_min(R, (R + T3).yx);
}

TG_HD uint32_t tg_pack_bytes0022(uint32_t z0, uint32_t z1)
{
	/* (z0.b0, z1.b0, z0.b2, z1.b2) */
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(z1, z0, 0x06020400u);
#else
	return (z0 & 0xff) | ((z1 & 0xff) << 8) | (z0 & 0xff0000) | ((z1 << 8) & 0xff000000u);
#endif
}

/* history of a four-step block (two dwords) out, history bits cleared */
TG_HD void tg_pvit_hist(tg_pvit_state &v, uint32_t h[2])
{
	uint32_t g[4];
#pragma unroll
	for (int d = 0; d < 4; d++)
		g[d] = tg_pack_bytes0022(tg_as_u32(v.Z[2 * d]), tg_as_u32(v.Z[2 * d + 1])) & 0x0f0f0f0fu;
	h[0] = g[0] | (g[1] << 4);
	h[1] = g[2] | (g[3] << 4);
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = tg_as_us2(tg_as_u32(v.Z[k]) & 0xfff0fff0u);
}

/* one traceback hop: the nibble of the current state from a block's two history dwords; pos (bit offset of the
 * current state's nibble; 0 = state 0) moves on to the block's start state.  Only bits 0..5 of pos count. */
TG_HD uint32_t tg_ptrace_hop(uint32_t h0, uint32_t h1, uint32_t &pos)
{
	const uint64_t H = ((uint64_t)h1 << 32) | h0;
	const uint32_t X = (uint32_t)(H >> (pos & 63));
	pos = ((X << 4) | (X & 0xeu)) << 1;
	return X & 15u;
}

/* lead-in: values 0..5 of the lead-in group, history dropped */
template <typename Tab>
TG_HD void tg_pvit_leadin(tg_pvit_state &v, const uint32_t w[2], uint32_t mask6, Tab tab)
{
	uint32_t t[6];
	tg_psoft_fetch<0, 6>(w, mask6, tab, t);
	tg_pvit_quad(v, t);
#pragma unroll
	for (int k = 0; k < 8; k++)
		v.Z[k] = tg_as_us2(tg_as_u32(v.Z[k]) & 0xfff0fff0u);
}

/* eight steps = two four-step blocks on the 12 fetched table entries t[] (tg_psoft_fetch<0, 12> of the block's three
 * dwords); h[0..1], h[2..3]: their histories.  LAST: the second one is the four flush steps (t[6..11] unused). */
template <bool LAST>
TG_HD void tg_pvit_block(tg_pvit_state &v, const uint32_t t[12], uint32_t h[4])
{
	tg_pvit_quad(v, t);
	tg_pvit_hist(v, h);
	if (LAST)
		tg_pflush4(v);
	else
		tg_pvit_quad(v, t + 6);
	tg_pvit_hist(v, h + 2);
}

/* byte offset, inside a block's soft area, of the 12 values of trellis block b (after the 8-byte lead-in
 * group) -- the layout k_front_soft writes: [6 lead-in values, 2 pad][12 values] x NBLK, type-3 order */
#define TG_SOFT_LEADIN_BYTES 8
#define TG_SOFT_BLOCK_BYTES  12

/* ---- CRC-16/CCITT over the decoded bit string (lower_mac/crc_simple.c:65-82) ---- */
/* table[x] for x = 8 input bits given LSB-first (bit i of x is the (i+1)-th bit fed) */
static inline uint16_t tg_crc16_step_bits(uint16_t crc, uint32_t bits_lsb_first, int n)
{
	for (int i = 0; i < n; i++) {
		crc ^= (uint16_t)(((bits_lsb_first >> i) & 1) << 15);
		crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
	}
	return crc;
}

static inline void tg_crc16_make_table(uint16_t tab[256])
{
	for (int x = 0; x < 256; x++)
		tab[x] = tg_crc16_step_bits(0, (uint32_t)x, 8);
}

/* crc over nbits = 8*nfull + 4 bits held LSB-first in bytes[]; tab from tg_crc16_make_table.
 * CRC is linear: crc(state, byte) = crc(state, 0) ^ tab[byte]; crc(state,0) over 8 zero
 * bits = (state << 8) ^ tabz[state >> 8] where tabz is the classic MSB-first table, which
 * equals tab[bitrev8(.)]; we fold that by also passing tabm (MSB-first table).            */
template <typename TabFn>
TG_HD uint16_t tg_crc16_bytes(const uint8_t *bytes, int nfull, TabFn tab_lsb, TabFn tab_msb)
{
	uint16_t crc = 0xffff;
	for (int i = 0; i < nfull; i++)
		crc = (uint16_t)((crc << 8) ^ tab_msb(crc >> 8) ^ tab_lsb(bytes[i]));
	/* trailing 4 bits */
	uint32_t nib = bytes[nfull] & 15;
	for (int i = 0; i < 4; i++) {
		crc ^= (uint16_t)(((nib >> i) & 1) << 15);
		crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
	}
	return crc;
}

#endif
