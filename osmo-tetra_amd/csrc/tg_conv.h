/*
 * tg_conv.h -- step programs for the generic trellis kernel (k_conv): any of the reference's seven RCPC
 * puncturers on either mother code.
 *
 * Reference: lower_mac/tetra_conv_enc.c:96-198 (the puncturer parameter sets: P table, t, period, i_func),
 * :226-248 (tetra_rcpc_depunct: type-3 bit j -> mother position k = period*((i-1)/t) + P[i - t*((i-1)/t)],
 * i = i_func(j)).  Instead of scattering into a 0xff-filled mother buffer and decoding that, the scatter is
 * turned around once per shape on the host: for every trellis step, which type-3 bit (if any) carries g1, g2, g3.
 * Mother position k (1-based) of a rate-1/N code belongs to step (k-1)/N, generator (k-1)%N.
 */
#ifndef TG_CONV_H
#define TG_CONV_H

#include <stdint.h>

#define TG_CONV_NPUNCT   7
/* step descriptor: three dwords, one per generator g1, g2, g3.  For received position p (class byte p >> 2,
 * field shift 2 * (p & 3), vit_core.h): (p >> 2) << 24 | multiplier 0x800100 >> shift; 0 = nothing received
 * (the multiplier 0 makes the branch-metric term vanish).  The kernel uses the dword as it is as the 24-bit
 * multiplier and its top byte as the LDS byte index. */
#define TG_CONV_DESC_WORDS 3
#define TG_CONV_MAX_T3   1022u
#define TG_CONV_MAX_T2   504u		/* 8 * 63 history blocks */

/* 1-based mother positions kept per period (the P tables without their unused entry 0) */
static inline uint32_t tg_conv_mother_pos(int pu, uint32_t j)
{
	static const uint8_t p23[] = { 1, 2, 5 };			/* 8.2.3.1.3, also 292/432 */
	static const uint8_t p13[] = { 1, 2, 3, 5, 6, 7 };		/* 8.2.3.1.4, also 148/432 */
	static const uint8_t p812[] = { 1, 2, 4 };			/* EN 300 395-2 5.5.2.1 */
	static const uint8_t p818[] = { 1, 2, 3, 4, 5, 7, 8, 10, 11 };	/* 5.5.2.2 */
	static const uint8_t p817[] = { 1, 2, 3, 4, 5, 7, 8, 10, 11, 13, 14, 16, 17, 19, 20, 22, 23 };	/* 5.6.2.1 */
	static const struct { const uint8_t *p; uint8_t t, period, skip; } d[TG_CONV_NPUNCT] = {
		{ p23, 3, 8, 0 }, { p13, 6, 8, 0 }, { p23, 3, 8, 65 }, { p13, 6, 8, 35 },
		{ p812, 3, 6, 0 }, { p818, 9, 12, 0 }, { p817, 17, 24, 0 },
	};
	const uint32_t i = d[pu].skip ? j + (j - 1) / d[pu].skip : j;
	const uint32_t q = (i - 1) / d[pu].t;
	return d[pu].period * q + d[pu].p[i - 1 - d[pu].t * q];
}

/* steps[0 .. 3 * (type2_len + 4) - 1]: the last four steps are the flush steps (nothing received).  0 on success. */
static inline int tg_conv_build_steps(int pu, int mother_rate, uint32_t type3_len, uint32_t type2_len, uint32_t *steps)
{
	if (pu < 0 || pu >= TG_CONV_NPUNCT || (mother_rate != 3 && mother_rate != 4))
		return -1;
	if (type3_len < 1 || type3_len > TG_CONV_MAX_T3 || type2_len < 8 || type2_len > TG_CONV_MAX_T2)
		return -1;
	if ((type2_len & 7) != 0 && (type2_len & 7) < 4)
		return -1;	/* the last (partial) history block must hold the state it starts in */
	for (uint32_t s = 0; s < (type2_len + 4) * TG_CONV_DESC_WORDS; s++)
		steps[s] = 0;
	for (uint32_t j = 1; j <= type3_len; j++) {
		const uint32_t k = tg_conv_mother_pos(pu, j) - 1;
		const uint32_t s = k / (uint32_t)mother_rate, g = k % (uint32_t)mother_rate;
		if (s >= type2_len || g > 2)
			return -1;
		if (steps[s * TG_CONV_DESC_WORDS + g])
			return -1;
		steps[s * TG_CONV_DESC_WORDS + g] = (((j - 1) >> 2) << 24) | (0x800100u >> (2 * ((j - 1) & 3)));
	}
	return 0;
}

/* does any step of the program receive g3?  (selects the kernel variant) */
static inline int tg_conv_uses_g3(const uint32_t *steps, uint32_t type2_len)
{
	for (uint32_t s = 0; s < type2_len + 4; s++)
		if (steps[s * TG_CONV_DESC_WORDS + 2])
			return 1;
	return 0;
}

#endif
