/*
 * tetra_oracle.c -- CPU restatement of the TETRA lower-MAC receive path.
 *
 * TEST INFRASTRUCTURE ONLY (see tetra_oracle.h for the rules and the parity
 * status of every row).  Nothing in the product library may call this.
 *
 * Citations are file:line under /root/reference/src.
 */
#include "tetra_oracle.h"

#include <string.h>
#include <stdio.h>
#include <stdlib.h>

/* ======================================================================
 * row X -- scrambler.  lower_mac/tetra_scramb.c:34-50 (LFSR step),
 * :67-85 (get_bits / scramb_bits), :87-99 (get_init), tetra_scramb.h:14.
 * Fibonacci LFSR, 32 bit, taps 32 26 23 22 16 12 11 10 8 7 5 4 2 1 where
 * "tap y" looks at state bit (32-y).  The feedback bit is also the output.
 * ==================================================================== */
static const uint8_t lfsr_taps[14] = { 32, 26, 23, 22, 16, 12, 11, 10, 8, 7, 5, 4, 2, 1 };

static inline uint32_t lfsr_step(uint32_t *st)
{
	uint32_t s = *st, fb = 0;
	for (int i = 0; i < 14; i++)
		fb ^= s >> (32 - lfsr_taps[i]);
	fb &= 1;
	*st = (s >> 1) | (fb << 31);
	return fb;
}

uint32_t orc_scramb_get_init(uint16_t mcc, uint16_t mnc, uint8_t colour)
{
	uint32_t v = (uint32_t)(colour & 0x3f) | ((uint32_t)(mnc & 0x3fff) << 6) | ((uint32_t)(mcc & 0x3ff) << 20);
	return (v << 2) | 3;
}

void orc_scramb_get_bits(uint32_t lfsr_init, uint8_t *out, int len)
{
	for (int i = 0; i < len; i++)
		out[i] = (uint8_t)lfsr_step(&lfsr_init);
}

void orc_scramb_bits(uint32_t lfsr_init, uint8_t *io, int len)
{
	for (int i = 0; i < len; i++)
		io[i] ^= (uint8_t)lfsr_step(&lfsr_init);
}

/* ======================================================================
 * row I -- block interleaver.  lower_mac/tetra_interleave.c:36-59.
 * k = 1 + (a*i mod K), i = 1..K.  interleave: out[k-1] = in[i-1];
 * deinterleave: out[i-1] = in[k-1].
 * ==================================================================== */
void orc_block_interleave(uint32_t K, uint32_t a, const uint8_t *in, uint8_t *out)
{
	for (uint32_t i = 1; i <= K; i++)
		out[(a * i) % K] = in[i - 1];
}

void orc_block_deinterleave(uint32_t K, uint32_t a, const uint8_t *in, uint8_t *out)
{
	for (uint32_t i = 1; i <= K; i++)
		out[i - 1] = in[(a * i) % K];
}

/* ======================================================================
 * rows E/U -- rate-1/4 mother code and the puncturers.
 * lower_mac/tetra_conv_enc.c:43-86 (encoder: G1=1+D+D4, G2=1+D2+D3+D4,
 * G3=1+D+D2+D4, G4=1+D+D3+D4; output order g1,g2,g3,g4),
 * :96-198 (puncturer parameter sets), :201-248 (puncture / depuncture:
 * k = period*((i-1)/t) + P[i - t*((i-1)/t)], i = i_func(j)).
 * ==================================================================== */
void orc_conv_encode(const uint8_t *in, int len, uint8_t *out4)
{
	unsigned d1 = 0, d2 = 0, d3 = 0, d4 = 0;	/* D, D^2, D^3, D^4 */
	for (int n = 0; n < len; n++) {
		unsigned b = in[n];
		out4[4 * n + 0] = (b + d1 + d4) & 1;
		out4[4 * n + 1] = (b + d2 + d3 + d4) & 1;
		out4[4 * n + 2] = (b + d1 + d2 + d4) & 1;
		out4[4 * n + 3] = (b + d1 + d3 + d4) & 1;
		d4 = d3; d3 = d2; d2 = d1; d1 = b;
	}
}

struct punct_def {
	const uint8_t *P;
	uint8_t t, period;
	uint8_t ifunc;		/* 0: i=j, 1: j+(j-1)/65, 2: j+(j-1)/35 */
};
static const uint8_t P_2_3[]  = { 0, 1, 2, 5 };
static const uint8_t P_1_3[]  = { 0, 1, 2, 3, 5, 6, 7 };
static const uint8_t P_8_12[] = { 0, 1, 2, 4 };
static const uint8_t P_8_18[] = { 0, 1, 2, 3, 4, 5, 7, 8, 10, 11 };
static const uint8_t P_8_17[] = { 0, 1, 2, 3, 4, 5, 7, 8, 10, 11, 13, 14, 16, 17, 19, 20, 22, 23 };
static const struct punct_def punct_defs[7] = {
	[ORC_PUNCT_2_3]     = { P_2_3,  3,  8,  0 },
	[ORC_PUNCT_1_3]     = { P_1_3,  6,  8,  0 },
	[ORC_PUNCT_292_432] = { P_2_3,  3,  8,  1 },
	[ORC_PUNCT_148_432] = { P_1_3,  6,  8,  2 },
	[ORC_PUNCT_112_168] = { P_8_12, 3,  6,  0 },
	[ORC_PUNCT_72_162]  = { P_8_18, 9,  12, 0 },
	[ORC_PUNCT_38_80]   = { P_8_17, 17, 24, 0 },
};

static inline uint32_t punct_k(const struct punct_def *pd, uint32_t j)
{
	uint32_t i = j;
	if (pd->ifunc == 1)
		i = j + (j - 1) / 65;
	else if (pd->ifunc == 2)
		i = j + (j - 1) / 35;
	uint32_t q = (i - 1) / pd->t;
	return pd->period * q + pd->P[i - pd->t * q];
}

int orc_puncture(enum orc_punct pu, const uint8_t *mother, int len, uint8_t *out)
{
	if ((unsigned)pu >= 7)
		return -1;
	for (uint32_t j = 1; j <= (uint32_t)len; j++)
		out[j - 1] = mother[punct_k(&punct_defs[pu], j) - 1];
	return 0;
}

int orc_depuncture(enum orc_punct pu, const uint8_t *in, int len, uint8_t *mother)
{
	if ((unsigned)pu >= 7)
		return -1;
	for (uint32_t j = 1; j <= (uint32_t)len; j++)
		mother[punct_k(&punct_defs[pu], j) - 1] = in[j - 1];
	return 0;
}

/* the speech mother code as an encoder (EN 300 395-2 5.5.2; the reference only holds its trellis tables,
 * lower_mac/viterbi_tch.c:29-47) -- test generator for the rate-1/3 puncturers */
void orc_conv_encode_tch(const uint8_t *in, int len, uint8_t *out3)
{
	unsigned d1 = 0, d2 = 0, d3 = 0, d4 = 0;
	for (int n = 0; n < len; n++) {
		unsigned b = in[n] & 1;
		out3[3 * n + 0] = (b + d1 + d2 + d3 + d4) & 1;
		out3[3 * n + 1] = (b + d1 + d3 + d4) & 1;
		out3[3 * n + 2] = (b + d2 + d4) & 1;
		d4 = d3; d3 = d2; d2 = d1; d1 = b;
	}
}

/* ======================================================================
 * row V -- Viterbi.
 *
 * Trellis tables: lower_mac/viterbi_cch.c:35-47.  State = last four input
 * bits, newest in the LSB: next_state[s][b] = ((s<<1)|b)&15; next_output is
 * the g1..g4 nibble (g1 = MSB) of the encoder above.  We derive both from
 * the generator polynomials instead of tabulating them.
 *
 * Call convention: lower_mac/viterbi_cch.c:58-66 copies the code, sets
 * code.len = n and calls osmo_conv_decode(); .term is zero-initialised =
 * CONV_TERM_FLUSH.  lower_mac/viterbi.c:8 zero-initialises vit_inp[864*4],
 * so the K-1 = 4 flush steps read erasures.
 *
 * libosmocore is NOT in this container (un-vendored, version unpinned).
 * Two published algorithms are restated:
 *   orc_viterbi_generic -- libosmocore src/conv.c: osmo_conv_decode_init /
 *     _reset(start_state 0) / _scan / _flush / _get_output(has_flush=1,
 *     end_state=0).  Accumulated error ae[] (unsigned), MAX_AE 0x00ffffff,
 *     per-bit error ((in - ov)^2) >> 9 for non-zero in, ov = +-127;
 *     survivor replaced only if strictly better, states scanned ascending,
 *     b = 0 before b = 1.
 *   orc_viterbi_acc -- libosmocore src/conv_acc.c + conv_acc_generic.c
 *     (what upstream's osmo_conv_decode dispatches to for N<=4, K in {5,7}):
 *     int16 correlation metrics, sums[0] = 127*N*K, butterflies with
 *     "sum0 >= sum1 -> path from state 2i", min-subtraction every
 *     intrvl = INT16_MAX/(N*127) - K steps, all len+K-1 steps are full ACS
 *     steps, traceback from state 0.
 * For hard-decision input (+-127 / 0) both give the same decisions, ties
 * included: tie -> predecessor whose oldest bit is 0.
 * ==================================================================== */
static inline unsigned cch_output(unsigned s, unsigned b)
{
	/* s: bit0 = D (newest) ... bit3 = D^4 */
	unsigned d1 = s & 1, d2 = (s >> 1) & 1, d3 = (s >> 2) & 1, d4 = (s >> 3) & 1;
	unsigned g1 = (b + d1 + d4) & 1;
	unsigned g2 = (b + d2 + d3 + d4) & 1;
	unsigned g3 = (b + d1 + d2 + d4) & 1;
	unsigned g4 = (b + d1 + d3 + d4) & 1;
	return (g1 << 3) | (g2 << 2) | (g3 << 1) | g4;
}

/* lower_mac/viterbi_tch.c:29-47: the speech (TCH) mother code G1 = 1+D+D2+D3+D4, G2 = 1+D+D3+D4, G3 = 1+D2+D4.
 * The reference's struct says N = 4 with 3-bit table entries (:49-54), so libosmocore reads FOUR soft values
 * per step and the first one (output bit 3) is always expected to be a 0 bit; g1,g2,g3 are values 1..3. */
static inline unsigned tch_output(unsigned s, unsigned b)
{
	unsigned d1 = s & 1, d2 = (s >> 1) & 1, d3 = (s >> 2) & 1, d4 = (s >> 3) & 1;
	unsigned g1 = (b + d1 + d2 + d3 + d4) & 1;
	unsigned g2 = (b + d1 + d3 + d4) & 1;
	unsigned g3 = (b + d2 + d4) & 1;
	return (g1 << 2) | (g2 << 1) | g3;
}

static inline unsigned code_output(int code, unsigned s, unsigned b)
{
	return code == ORC_CODE_TCH ? tch_output(s, b) : cch_output(s, b);
}

unsigned orc_code_output(int code, unsigned s, unsigned b)
{
	return code_output(code, s & 15, b & 1);
}

#define VIT_MAX_STEPS (864 + 4)
#define MAX_AE 0x00ffffffu

int orc_viterbi_generic(const int8_t *in, uint8_t *out, int len)
{
	return orc_viterbi_generic_code(ORC_CODE_CCH, in, out, len);
}

int orc_viterbi_generic_code(int code, const int8_t *in, uint8_t *out, int len)
{
	static __thread uint8_t hist[VIT_MAX_STEPS][16];
	unsigned ae[16], ae_next[16];
	int total = len + 4;

	if (len < 1 || len > 864)
		return -1;
	for (int s = 0; s < 16; s++)
		ae[s] = s ? MAX_AE : 0;

	for (int i = 0; i < total; i++) {
		const int8_t *sym = in + 4 * i;
		int nb = (i < len) ? 2 : 1;	/* flush steps: input bit 0 only */
		for (int s = 0; s < 16; s++)
			ae_next[s] = MAX_AE;
		for (unsigned s = 0; s < 16; s++) {
			for (int b = 0; b < nb; b++) {
				unsigned o = code_output(code, s, b);
				unsigned t = ((s << 1) | b) & 15;
				unsigned nae = ae[s];
				unsigned m = 8;
				for (int j = 0; j < 4; j++, m >>= 1) {
					int is = sym[j];
					if (is) {
						int ov = (o & m) ? -127 : 127;
						int e = is - ov;
						nae += (unsigned)(e * e) >> 9;
					}
				}
				if (ae_next[t] > nae) {
					ae_next[t] = nae;
					hist[i][t] = (uint8_t)s;
				}
			}
		}
		memcpy(ae, ae_next, sizeof(ae));
	}

	unsigned cur = 0;
	int i = total - 1;
	for (int f = 0; f < 4; f++, i--)
		cur = hist[i][cur];
	for (; i >= 0; i--) {
		unsigned prev = hist[i][cur];
		out[i] = (((prev << 1) & 15) == cur) ? 0 : 1;
		cur = prev;
	}
	return (int)ae[0];
}

int orc_viterbi_acc(const int8_t *in, uint8_t *out, int len)
{
	return orc_viterbi_acc_code(ORC_CODE_CCH, in, out, len);
}

int orc_viterbi_acc_code(int code, const int8_t *in, uint8_t *out, int len)
{
	/* acc state convention: most recent input bit in bit 3 (K-2), so the
	 * predecessors of state r are 2*(r&7) and 2*(r&7)+1. */
	static __thread int16_t paths[VIT_MAX_STEPS][16];
	int16_t sums[16], outs[16][4];
	uint8_t vals[16];
	const int total = len + 4;
	const int intrvl = 32767 / (4 * 127) - 5;

	if (len < 1 || len > 864)
		return -1;

	for (unsigned r = 0; r < 16; r++) {
		unsigned val = (r >> 3) & 1;
		unsigned prev0 = (r << 1) & 0xe;	/* vstate_lshift(reg, 5, 0) */
		/* bit-swap into the API convention (newest bit in the LSB) */
		unsigned ps = ((prev0 & 1) << 3) | ((prev0 & 2) << 1) | ((prev0 & 4) >> 1) | ((prev0 & 8) >> 3);
		unsigned o = code_output(code, ps, val);
		for (int j = 0; j < 4; j++)
			outs[r][j] = ((o >> (3 - j)) & 1) ? -1 : 1;
		vals[r] = (uint8_t)val;
	}

	memset(sums, 0, sizeof(sums));
	sums[0] = 127 * 4 * 5;

	for (int i = 0; i < total; i++) {
		const int8_t *seq = in + 4 * i;
		int16_t ns[16];
		for (int b = 0; b < 8; b++) {
			int metric = seq[0] * outs[b][0] + seq[1] * outs[b][1] + seq[2] * outs[b][2] + seq[3] * outs[b][3];
			int s0 = sums[2 * b], s1 = sums[2 * b + 1];
			int sum0 = s0 + metric, sum1 = s1 - metric, sum2 = s0 - metric, sum3 = s1 + metric;
			if (sum0 >= sum1) { ns[b] = (int16_t)sum0; paths[i][b] = -1; }
			else              { ns[b] = (int16_t)sum1; paths[i][b] = 0; }
			if (sum2 >= sum3) { ns[b + 8] = (int16_t)sum2; paths[i][b + 8] = -1; }
			else              { ns[b + 8] = (int16_t)sum3; paths[i][b + 8] = 0; }
		}
		if (!(i % intrvl)) {
			int16_t mn = ns[0];
			for (int s = 1; s < 16; s++)
				if (ns[s] < mn)
					mn = ns[s];
			for (int s = 0; s < 16; s++)
				ns[s] = (int16_t)(ns[s] - mn);
		}
		memcpy(sums, ns, sizeof(sums));
	}

	unsigned state = 0;
	int i;
	for (i = total - 1; i >= len; i--) {
		unsigned path = (unsigned)(paths[i][state] + 1);
		state = ((state << 1) & 0xe) | path;
	}
	for (; i >= 0; i--) {
		unsigned path = (unsigned)(paths[i][state] + 1);
		out[i] = vals[state];
		state = ((state << 1) & 0xe) | path;
	}
	return 0;
}

void orc_viterbi_dec_wrapper(const uint8_t *in, uint8_t *out, unsigned sym_count, int use_acc)
{
	/* lower_mac/viterbi.c:6-25 */
	static __thread int8_t vit_inp[864 * 4 + 16];
	memset(vit_inp, 0, sizeof(vit_inp));
	for (unsigned i = 0; i < sym_count * 4; i++)
		vit_inp[i] = (in[i] == 0) ? 127 : (in[i] == 0xff ? 0 : -127);
	if (use_acc)
		orc_viterbi_acc(vit_inp, out, (int)sym_count);
	else
		orc_viterbi_generic(vit_inp, out, (int)sym_count);
}

/* (f)1: any puncturer on either mother code -- what a caller of the reference writes with its pieces:
 * memset(dp, 0xff), tetra_rcpc_depunct(pu, type3, type3_len, dp) (conv_enc_test.c:66-68, tetra_lower_mac.c:249-251),
 * the 0 -> +127 / 0xff -> 0 / else -> -127 map of lower_mac/viterbi.c:12-22, then conv_cch_decode()
 * (viterbi_cch.c:58-66) or, for the rate-1/3 speech code, conv_tch_decode() (viterbi_tch.c:56-64) with the three
 * de-punctured values of a step behind one erased value (that code's N = 4, see tch_output). */
int orc_conv_decode_block(int pu, int mother_rate, const uint8_t *type3, unsigned type3_len, unsigned type2_len,
			  int use_acc, uint8_t *type2)
{
	static __thread uint8_t dp[864 * 4];
	static __thread int8_t vit_inp[(864 + 4) * 4];
	const int code = (mother_rate == 3) ? ORC_CODE_TCH : ORC_CODE_CCH;

	if ((unsigned)pu >= 7 || (mother_rate != 3 && mother_rate != 4) || type2_len < 1 || type2_len > 864)
		return -1;
	for (uint32_t j = 1; j <= type3_len; j++)
		if (punct_k(&punct_defs[pu], j) > type2_len * (unsigned)mother_rate)
			return -1;
	memset(dp, 0xff, sizeof(dp));
	orc_depuncture((enum orc_punct)pu, type3, (int)type3_len, dp);
	memset(vit_inp, 0, sizeof(vit_inp));
	for (unsigned n = 0; n < type2_len; n++)
		for (int g = 0; g < mother_rate; g++) {
			const uint8_t v = dp[n * (unsigned)mother_rate + (unsigned)g];
			vit_inp[4 * n + (4 - (unsigned)mother_rate) + (unsigned)g] = (v == 0) ? 127 : (v == 0xff ? 0 : -127);
		}
	if (use_acc)
		return orc_viterbi_acc_code(code, vit_inp, type2, (int)type2_len);
	orc_viterbi_generic_code(code, vit_inp, type2, (int)type2_len);
	return 0;
}

void orc_viterbi_soft(const int8_t *sbits_mother, uint8_t *out, unsigned sym_count)
{
	static __thread int8_t vit_inp[864 * 4 + 16];
	memset(vit_inp, 0, sizeof(vit_inp));
	memcpy(vit_inp, sbits_mother, sym_count * 4);
	orc_viterbi_acc(vit_inp, out, (int)sym_count);
}

/* ======================================================================
 * row C -- CRC-16.  lower_mac/crc_simple.c:65-82 (bit loop), :103-106.
 * ==================================================================== */
uint16_t orc_crc16_itut_bits(uint16_t crc, const uint8_t *bits, int n)
{
	for (int i = 0; i < n; i++) {
		crc ^= (uint16_t)((bits[i] & 1) << 15);
		crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
	}
	return crc;
}

uint16_t orc_crc16_ccitt_bits(const uint8_t *bits, unsigned n)
{
	return orc_crc16_itut_bits(0xffff, bits, (int)n);
}

/* ======================================================================
 * row R -- shortened RM(30,14).  lower_mac/tetra_rm3014.c:28-72 (generator:
 * row i = identity bit 1<<(29-i) | 16 parity bits, first parity column in
 * bit 15), :74-86 (encode: in bit (13-i) selects row i).  The RX side does
 * no decoding (tetra_lower_mac.c:268-274).
 * ==================================================================== */
static const uint16_t rm_parity[14] = {
	0x9b60, 0x2de0, 0xfc20, 0xe03c, 0x983a, 0x5436, 0x2c2e,
	0xffdf, 0x8339, 0x42b5, 0x21ad, 0x1273, 0x096b, 0x04e7,
};

uint32_t orc_rm3014_row(int i)
{
	return (1u << (29 - i)) | rm_parity[i];
}

uint32_t orc_rm3014_compute(uint16_t in)
{
	uint32_t v = 0;
	for (int i = 0; i < 14; i++)
		if ((in >> (13 - i)) & 1)
			v ^= orc_rm3014_row(i);
	return v;
}

/* Minimum-distance decoding by exhaustive search (test reference for the product's optional syndrome
 * decoder; the reference itself has no decoder, tetra_rm3014.c:88-96 is a stub): the codeword closest to
 * rx30, ties -> the one whose error pattern rx30 ^ cw is numerically smallest. */
uint16_t orc_rm3014_decode_ml(uint32_t rx30, unsigned *nerr)
{
	static uint32_t cw[1 << 14];
	static int ready;
	if (!ready) {
		for (uint32_t d = 0; d < (1u << 14); d++)
			cw[d] = orc_rm3014_compute((uint16_t)d);
		ready = 1;
	}
	rx30 &= 0x3fffffffu;
	uint32_t best_e = 0xffffffffu, best_d = 0;
	int best_w = 99;
	for (uint32_t d = 0; d < (1u << 14); d++) {
		const uint32_t e = cw[d] ^ rx30;
		const int w = __builtin_popcount(e);
		if (w < best_w || (w == best_w && e < best_e)) {
			best_w = w;
			best_e = e;
			best_d = d;
		}
	}
	if (nerr)
		*nerr = (unsigned)best_w;
	return (uint16_t)best_d;
}

/* ======================================================================
 * row T -- TDMA time.  tetra_tdma.c:27-94.
 * ==================================================================== */
void orc_tdma_add_tn(struct orc_tdma_time *tm, uint32_t tn_count)
{
	tm->tn += tn_count;
	if (tm->tn > 4) {
		uint32_t d = tm->tn / 4;
		tm->tn %= 4;
		tm->fn += d;
	}
	if (tm->fn > 18) {
		uint32_t d = tm->fn / 18;
		tm->fn %= 18;
		tm->mn += d;
	}
	if (tm->mn > 60)
		tm->mn %= 60;
}

void orc_tdma_dump(const struct orc_tdma_time *tm, char *buf, size_t buflen)
{
	snprintf(buf, buflen, "%02u/%02u/%u/%03u", tm->mn, tm->fn, tm->tn, tm->sn);
}

/* ======================================================================
 * row F -- training sequences (EN 300 392-2 9.4.4.3; phy/tetra_burst.c:
 * 59-70) and the search (phy/tetra_burst.c:269-339) with its quirks:
 * the 22-bit look-ahead filter is preloaded with in[0..19] and then shifts
 * in in[cur+21], so for cur < 21 it holds a window that skips in[20].
 * ==================================================================== */
static const uint8_t seq_n[22] = { 1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0 };
static const uint8_t seq_p[22] = { 0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0 };
static const uint8_t seq_q[22] = { 1,0,1,1,0,1,1,1,0,0,0,0,0,1,1,0,1,0,1,1,0,1 };
static const uint8_t seq_x[30] = { 1,0,0,1,1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0,0,0,1,1 };
static const uint8_t seq_y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };
static const uint8_t seq_f[80] = { 1,1,1,1,1,1,1,1, [72] = 1,1,1,1,1,1,1,1 };

const uint8_t *orc_train_bits(enum orc_train_seq t, unsigned *len)
{
	switch (t) {
	case ORC_TRAIN_NORM_1: *len = 22; return seq_n;
	case ORC_TRAIN_NORM_2: *len = 22; return seq_p;
	case ORC_TRAIN_NORM_3: *len = 22; return seq_q;
	case ORC_TRAIN_SYNC:   *len = 38; return seq_y;
	case ORC_TRAIN_EXT:    *len = 30; return seq_x;
	}
	*len = 0;
	return NULL;
}

int orc_find_train_seq(const uint8_t *in, unsigned end_of_in, uint32_t mask, unsigned *offset)
{
	/* candidates in the reference's test order: SYNC, NORM_1, NORM_2, NORM_3, EXT */
	static const struct { int id; const uint8_t *s; unsigned len; } cand[5] = {
		{ ORC_TRAIN_SYNC, seq_y, 38 }, { ORC_TRAIN_NORM_1, seq_n, 22 }, { ORC_TRAIN_NORM_2, seq_p, 22 },
		{ ORC_TRAIN_NORM_3, seq_q, 22 }, { ORC_TRAIN_EXT, seq_x, 30 },
	};
	uint32_t pre[5] = { 0, 0, 0, 0, 0 };
	for (int c = 0; c < 5; c++)
		for (int i = 0; i < 22; i++)
			pre[c] = (pre[c] << 1) | cand[c].s[i];

	uint32_t filt = 0;
	for (int i = 0; i < 20; i++)
		filt = (filt << 1) | in[i];

	for (unsigned cur = 0; cur < end_of_in; cur++) {
		filt = ((filt << 1) | in[cur + 21]) & 0x3fffff;
		if (filt != pre[0] && filt != pre[1] && filt != pre[2] && filt != pre[3] && filt != pre[4])
			continue;
		unsigned remain = end_of_in - cur;
		for (int c = 0; c < 5; c++) {
			if (!(mask & (1u << cand[c].id)))
				continue;
			if (remain >= cand[c].len && !memcmp(in + cur, cand[c].s, cand[c].len)) {
				*offset = cur;
				return cand[c].id;
			}
		}
	}
	return -1;
}

/* ======================================================================
 * row E -- burst builders.  phy/tetra_burst.c:117-166 (phase adjustment),
 * :169-219 (sync continuous downlink burst), :222-267 (normal cont. DL).
 * NOTE: the reference indexes phase2bits[] with the raw adjustment value
 * instead of PHASE(value) (tetra_burst.c:160), i.e. reads outside/at the
 * wrong slots of the table; what it emits for the four phase-adjustment
 * bits is therefore not meaningful.  We emit the EN 300 392-2 values.  The
 * receive path never looks at those bits, and tests mask them (slot bits
 * 12,13 and 498,499) when comparing with oracle/_ref.
 * ==================================================================== */
static void phase_adj(const uint8_t *burst, unsigned n1, unsigned n2, uint8_t *out2)
{
	static const int8_t b2p[4] = { 1, -1, 3, -3 };
	int sum = 0;
	for (unsigned n = n1 - 1; n < n2; n++)
		sum += b2p[burst[2 * n] | (burst[2 * n + 1] << 1)];
	int adj = -(sum % 8);
	if (adj > 3)
		adj -= 8;
	else if (adj < -3)
		adj += 8;
	switch (adj) {
	case -3: out2[0] = 1; out2[1] = 1; break;
	case -1: out2[0] = 0; out2[1] = 1; break;
	case  1: out2[0] = 0; out2[1] = 0; break;
	case  3: out2[0] = 1; out2[1] = 0; break;
	default: out2[0] = 0; out2[1] = 0; break;
	}
}

int orc_build_sync_burst(uint8_t *buf, const uint8_t *sb, const uint8_t *bb, const uint8_t *bkn)
{
	uint8_t *c = buf;
	memcpy(c, seq_q + 10, 12); c += 12;
	uint8_t *hc = c; c[0] = c[1] = 0; c += 2;
	memcpy(c, seq_f, 80); c += 80;
	memcpy(c, sb, 120); c += 120;
	memcpy(c, seq_y, 38); c += 38;
	memcpy(c, bb, 30); c += 30;
	memcpy(c, bkn, 216); c += 216;
	uint8_t *hd = c; c[0] = c[1] = 0; c += 2;
	memcpy(c, seq_q, 10); c += 10;
	phase_adj(buf, 8, 108, hc);
	phase_adj(buf, 109, 249, hd);
	return (int)(c - buf);
}

int orc_build_norm_burst(uint8_t *buf, const uint8_t *bkn1, const uint8_t *bb, const uint8_t *bkn2, int two_log_chan)
{
	uint8_t *c = buf;
	memcpy(c, seq_q + 10, 12); c += 12;
	uint8_t *ha = c; c[0] = c[1] = 0; c += 2;
	memcpy(c, bkn1, 216); c += 216;
	memcpy(c, bb, 14); c += 14;
	memcpy(c, two_log_chan ? seq_p : seq_n, 22); c += 22;
	memcpy(c, bb + 14, 16); c += 16;
	memcpy(c, bkn2, 216); c += 216;
	uint8_t *hb = c; c[0] = c[1] = 0; c += 2;
	memcpy(c, seq_q, 10); c += 10;
	phase_adj(buf, 8, 122, ha);
	phase_adj(buf, 123, 249, hb);
	return (int)(c - buf);
}

/* ======================================================================
 * row P -- block parameters.  lower_mac/tetra_lower_mac.c:45-102.
 * ==================================================================== */
static const struct orc_blk_param blk_params[6] = {
	[ORC_T_SB1]    = { "SB1",    120,  80,  60,  11, 1 },
	[ORC_T_SB2]    = { "SB2",    216, 144, 124, 101, 1 },
	[ORC_T_NDB]    = { "NDB",    216, 144, 124, 101, 1 },
	[ORC_T_BBK]    = { "BBK",     30,  30,  14,   0, 0 },
	[ORC_T_SCH_HU] = { "SCH/HU", 168, 112,  92,  13, 1 },
	[ORC_T_SCH_F]  = { "SCH/F",  432, 288, 268, 103, 1 },
};

const struct orc_blk_param *orc_blk_param(enum orc_tpsap_type t)
{
	return ((unsigned)t < 6) ? &blk_params[t] : NULL;
}

/* ======================================================================
 * row E -- whole-block encoder, following conv_enc_test.c:88-134
 * (build_ndb_schf): type-1 | ~crc16 (MSB first) | 4 zero tail bits ->
 * mother code -> 2/3 puncture -> interleave -> scramble.
 * ==================================================================== */
void orc_encode_block(enum orc_tpsap_type type, const uint8_t *type1, uint32_t scramb_init, uint8_t *type5)
{
	const struct orc_blk_param *p = &blk_params[type];
	uint8_t type2[288], mother[288 * 4], type3[432], type4[432];

	memset(type2, 0, sizeof(type2));
	for (unsigned i = 0; i < p->type1_bits; i++)
		type2[i] = type1[i] & 1;
	uint16_t crc = (uint16_t)~orc_crc16_ccitt_bits(type2, p->type1_bits);
	for (int i = 0; i < 16; i++)
		type2[p->type1_bits + i] = (crc >> (15 - i)) & 1;
	orc_conv_encode(type2, p->type2_bits, mother);
	orc_puncture(ORC_PUNCT_2_3, mother, p->type345_bits, type3);
	orc_block_interleave(p->type345_bits, p->interleave_a, type3, type4);
	memcpy(type5, type4, p->type345_bits);
	orc_scramb_bits(scramb_init, type5, p->type345_bits);
}

void orc_encode_bbk(const uint8_t *type1_14, uint32_t scramb_init, uint8_t *type5_30)
{
	/* conv_enc_test.c:138-147: RM(30,14) word written MSB first as 30 bits */
	uint16_t v = 0;
	for (int i = 0; i < 14; i++)
		v = (uint16_t)((v << 1) | (type1_14[i] & 1));
	uint32_t cw = orc_rm3014_compute(v);
	for (int i = 0; i < 30; i++)
		type5_30[i] = (cw >> (29 - i)) & 1;
	orc_scramb_bits(scramb_init, type5_30, 30);
}

/* ======================================================================
 * row L (per-block part) -- lower_mac/tetra_lower_mac.c:179-186 (descramble)
 * :243-255 (deinterleave, depuncture into 0xff-filled buffer, Viterbi),
 * :257-274 (CRC / BBK copy).
 * ==================================================================== */
void orc_decode_block(enum orc_tpsap_type type, const uint8_t *type5, uint32_t scramb_init,
		      int use_acc, struct orc_block_result *res)
{
	const struct orc_blk_param *p = &blk_params[type];
	uint8_t type3[432];
	static __thread uint8_t type3dp[512 * 4];

	memcpy(res->type4, type5, p->type345_bits);
	orc_scramb_bits(scramb_init, res->type4, p->type345_bits);
	res->crc = 0;
	res->crc_ok = 0;
	memset(res->type2, 0, sizeof(res->type2));

	if (p->interleave_a) {
		orc_block_deinterleave(p->type345_bits, p->interleave_a, res->type4, type3);
		memset(type3dp, 0xff, sizeof(type3dp));
		orc_depuncture(ORC_PUNCT_2_3, type3, p->type345_bits, type3dp);
		orc_viterbi_dec_wrapper(type3dp, res->type2, p->type2_bits, use_acc);
	}
	if (p->have_crc16) {
		res->crc = orc_crc16_ccitt_bits(res->type2, p->type1_bits + 16u);
		res->crc_ok = (res->crc == ORC_CRC_OK);
	} else if (type == ORC_T_BBK) {
		res->crc_ok = 1;
		memcpy(res->type2, res->type4, p->type2_bits);
	}
	memcpy(res->type1, res->type2, p->type1_bits);
}

/* ======================================================================
 * rows L/D/S -- the streaming receiver.
 * ==================================================================== */
static uint32_t bits_to_uint(const uint8_t *bits, unsigned len)
{
	/* tetra_common.c:31-39 */
	uint32_t r = 0;
	while (len--)
		r = (r << 1) | (*bits++ & 1);
	return r;
}

static int is_bnch(const struct orc_tdma_time *tm)
{
	/* lower_mac/tetra_lower_mac.c:122-127 */
	return tm->fn == 18 && tm->tn == 4 - ((tm->mn + 3) % 4);
}

void orc_rx_init(struct orc_rx *rx, orc_upper_cb upper, orc_event_cb event, void *priv)
{
	memset(rx, 0, sizeof(*rx));
	rx->upper = upper;
	rx->event = event;
	rx->priv = priv;
}

/* lower_mac/tetra_lower_mac.c:143-357 */
void orc_tp_sap_udata_ind(struct orc_rx *rx, int type, int blk_num, const uint8_t *bits, unsigned len)
{
	const struct orc_blk_param *p = &blk_params[type];
	struct orc_record rec;
	struct orc_block_result res;
	(void)len;

	memset(&rec, 0, sizeof(rec));
	rec.burst_seq = rx->burst_seq;
	rec.burst_type = rx->cur_burst_type;
	rec.type = (uint8_t)type;
	rec.blk_num = (uint8_t)blk_num;
	rec.type1_len = p->type1_bits;

	rx->cell_time = rx->phy_time;					/* :167 */
	rec.time_str = rx->cell_time;					/* :168 */
	if (type == ORC_T_SB2 && is_bnch(&rx->cell_time))		/* :170-173 */
		rec.lchan = ORC_LC_BNCH;

	uint32_t code = (type == ORC_T_SB1) ? 3u : rx->scramb_init;	/* :179-186 */
	rec.scrambling_code = code;

	if (rx->is_traffic && type == ORC_T_NDB && blk_num == 1)	/* :194-195 */
		rx->blk1_stolen = 1;

	if (rx->is_traffic && (type == ORC_T_SCH_F || (blk_num == 2 && !rx->blk2_stolen))) {	/* :198-241 */
		memcpy(rec.type4, bits, p->type345_bits);
		orc_scramb_bits(code, rec.type4, p->type345_bits);
		rec.traffic_dumped = (uint8_t)rx->is_traffic;
		rec.time = rx->cell_time;
		if (rx->upper)
			rx->upper(rx, &rec, 0xffffffffu, rx->priv);	/* offset ~0: "dumped, not delivered" */
		return;
	}

	orc_decode_block((enum orc_tpsap_type)type, bits, code, rx->use_acc, &res);
	memcpy(rec.type4, res.type4, p->type345_bits);
	memcpy(rec.type1, res.type1, p->type1_bits);
	rec.crc = res.crc;
	rec.crc_ok = (uint8_t)res.crc_ok;

	switch (type) {
	case ORC_T_SB1:							/* :283-310 */
		if (rec.crc_ok) {
			rx->colour_code = (uint8_t)bits_to_uint(res.type2 + 4, 6);
			rx->cell_time.tn = bits_to_uint(res.type2 + 10, 2) + 1;
			rx->cell_time.fn = bits_to_uint(res.type2 + 12, 5);
			rx->cell_time.mn = bits_to_uint(res.type2 + 17, 6);
			rx->mcc = (uint16_t)bits_to_uint(res.type2 + 31, 10);
			rx->mnc = (uint16_t)bits_to_uint(res.type2 + 41, 14);
			rx->scramb_init = orc_scramb_get_init(rx->mcc, rx->mnc, rx->colour_code);
		}
		rx->phy_time = rx->cell_time;
		rec.lchan = ORC_LC_BSCH;
		break;
	case ORC_T_BBK:
		rec.lchan = ORC_LC_AACH;
		break;
	case ORC_T_SCH_F:
		rec.lchan = ORC_LC_SCH_F;
		break;
	default:
		break;
	}

	/* :326-352 -- upper-MAC loop.  The limit is (int)type1_bits - 16 compared
	 * as unsigned, so for the 14-bit BBK it is 0xfffffffe. */
	uint32_t offset = 0;
	uint32_t limit = (uint32_t)((int)p->type1_bits - 16);
	while (offset < limit) {
		rec.time = rx->cell_time;
		int n = rx->upper ? rx->upper(rx, &rec, offset, rx->priv) : -1;
		if (n <= 0)	/* the reference would spin forever on 0; we stop */
			break;
		offset += (uint32_t)n;
	}
}

/* lower_mac/tetra_lower_mac.c:213-231: the 690-word block appended to traffic_<usage>_<tsn>.out.
 * type4[] is a local array there, so for a 216-bit block the words made from bits 216..431 are
 * whatever the stack held; this restatement takes them as 0 bits (documented deviation from UB). */
void orc_traffic_block(const uint8_t *type4, unsigned len, int16_t *block)
{
	uint8_t t4[432];
	int i;
	memset(t4, 0, sizeof(t4));
	memcpy(t4, type4, len < 432 ? len : 432);
	memset(block, 0x00, sizeof(int16_t) * 690);
	for (i = 0; i < 6; i++)
		block[115 * i] = (int16_t)(0x6b21 + i);
	for (i = 0; i < 114; i++)
		block[1 + i] = t4[i] ? -127 : 127;
	for (i = 0; i < 114; i++)
		block[116 + i] = t4[114 + i] ? -127 : 127;
	for (i = 0; i < 114; i++)
		block[231 + i] = t4[228 + i] ? -127 : 127;
	for (i = 0; i < 90; i++)
		block[346 + i] = t4[342 + i] ? -127 : 127;
}

/*
 * lower_mac/tch_reordering.c:94-117 / :119-140 as an OPERATION: the class position tables (EN 300 395-2 Table 4,
 * 1-based positions inside a 137-bit codec frame) are the caller's data.  The type-2 bits of a full-rate speech
 * block hold the classes one after the other, every position twice (frame 0, frame 1).
 */
void orc_acelp_type2_to_codec(const uint8_t *in, uint8_t *out, const uint8_t *const cls[3], const unsigned ncls[3])
{
	const int nbits = (int)(ncls[0] + ncls[1] + ncls[2]);
	const uint8_t *cur = in;
	for (int c = 0; c < 3; c++) {
		for (unsigned bit = 0; bit < ncls[c]; bit++)
			for (int frame = 0; frame < 2; frame++) {
				/* a table entry 0 (the reference's class-0 table is one initialiser short) indexes
				 * out[-1] there: skipped here, the in-bounds index of frame 1 is kept */
				const int idx = frame * nbits + (int)cls[c][bit] - 1;
				if (idx >= 0)
					out[idx] = cur[2 * bit + frame];
			}
		cur += 2 * ncls[c];
	}
}

void orc_acelp_codec_to_acelp(const uint8_t *in, uint8_t *out, const uint8_t *const cls[3], const unsigned ncls[3])
{
	const int nbits = (int)(ncls[0] + ncls[1] + ncls[2]);
	uint8_t *cur = out;
	for (int c = 0; c < 3; c++) {
		for (unsigned bit = 0; bit < ncls[c]; bit++)
			for (int frame = 0; frame < 2; frame++) {
				const int idx = frame * nbits + (int)cls[c][bit] - 1;
				if (idx >= 0)	/* (the reference reads in[-1] for an entry 0: left as it was here) */
					cur[2 * bit + frame] = in[idx];
			}
		cur += 2 * ncls[c];
	}
}

/* phy/tetra_burst.c:341-379 with the offsets of :31-47 */
void orc_burst_rx_cb(struct orc_rx *rx, const uint8_t *burst, unsigned len, int type)
{
	uint8_t bbk[30], schf[432];
	(void)len;
	rx->cur_burst_type = (uint8_t)type;
	switch (type) {
	case ORC_TRAIN_SYNC:
		orc_tp_sap_udata_ind(rx, ORC_T_SB1, 1, burst + 94, 120);
		orc_tp_sap_udata_ind(rx, ORC_T_BBK, 0, burst + 252, 30);
		orc_tp_sap_udata_ind(rx, ORC_T_SB2, 2, burst + 282, 216);
		break;
	case ORC_TRAIN_NORM_2:
		memcpy(bbk, burst + 230, 14);
		memcpy(bbk + 14, burst + 266, 16);
		orc_tp_sap_udata_ind(rx, ORC_T_BBK, 0, bbk, 30);
		orc_tp_sap_udata_ind(rx, ORC_T_NDB, 1, burst + 14, 216);
		orc_tp_sap_udata_ind(rx, ORC_T_NDB, 2, burst + 282, 216);
		break;
	case ORC_TRAIN_NORM_1:
		memcpy(bbk, burst + 230, 14);
		memcpy(bbk + 14, burst + 266, 16);
		memcpy(schf, burst + 14, 216);
		memcpy(schf + 216, burst + 282, 216);
		orc_tp_sap_udata_ind(rx, ORC_T_BBK, 0, bbk, 30);
		orc_tp_sap_udata_ind(rx, ORC_T_SCH_F, 0, schf, 432);
		break;
	default:
		break;
	}
}

/* phy/tetra_burst_sync.c:38-154 */
enum { RXS_UNLOCKED = 0, RXS_KNOW_FSTART = 1, RXS_LOCKED = 2 };

int orc_burst_sync_in(struct orc_rx *rx, const uint8_t *bits, unsigned len)
{
	unsigned space = (unsigned)sizeof(rx->bitbuf) - rx->bits_in_buf;	/* :38-52 */
	if (space < len) {
		unsigned delta = len - space;
		memmove(rx->bitbuf, rx->bitbuf + delta, rx->bits_in_buf - delta);
		rx->bits_in_buf -= delta;
		rx->bitbuf_start_bitnum += delta;
	}
	memcpy(rx->bitbuf + rx->bits_in_buf, bits, len);			/* :62-64 */
	rx->bits_in_buf += len;

	unsigned offs = 0;
	int rc;

	switch (rx->state) {
	case RXS_UNLOCKED:							/* :67-90 */
		if (rx->bits_in_buf < 510 * 2)
			return (int)len;
		rc = orc_find_train_seq(rx->bitbuf, rx->bits_in_buf, 1u << ORC_TRAIN_SYNC, &offs);
		if (rc < 0)
			return rc;
		if (rx->event)
			rx->event(ORC_EV_FOUND_SYNC, rx->bitbuf_start_bitnum, offs, rx->priv);
		rx->state = RXS_KNOW_FSTART;
		rx->next_frame_start_bitnum = rx->bitbuf_start_bitnum + offs + 296;
		break;
	case RXS_KNOW_FSTART:							/* :91-105, falls through */
		if (rx->bitbuf_start_bitnum + rx->bits_in_buf < rx->next_frame_start_bitnum)
			return 0;
		{
			int offset = (int)(rx->next_frame_start_bitnum - rx->bitbuf_start_bitnum);
			int remaining = (int)rx->bits_in_buf - offset;
			memmove(rx->bitbuf, rx->bitbuf + offset, (size_t)remaining);
			rx->bits_in_buf = (unsigned)remaining;
			rx->bitbuf_start_bitnum += (unsigned)offset;
			rx->next_frame_start_bitnum += 510;
			rx->state = RXS_LOCKED;
		}
		/* fall through */
	case RXS_LOCKED:							/* :106-149 */
		if (rx->bits_in_buf < 510)
			return (int)len;
		orc_tdma_add_tn(&rx->phy_time, 1);
		rx->burst_seq++;
		if (rx->event)
			rx->event(ORC_EV_BURST, rx->bitbuf_start_bitnum, rx->bits_in_buf, rx->priv);
		rc = orc_find_train_seq(rx->bitbuf, rx->bits_in_buf,
					(1u << ORC_TRAIN_NORM_1) | (1u << ORC_TRAIN_NORM_2) | (1u << ORC_TRAIN_SYNC), &offs);
		switch (rc) {
		case ORC_TRAIN_SYNC:
			if (offs == 214)
				orc_burst_rx_cb(rx, rx->bitbuf, 510, rc);
			else {
				if (rx->event)
					rx->event(ORC_EV_SYNC_MISPLACED, rx->bitbuf_start_bitnum, offs, rx->priv);
				rx->state = RXS_UNLOCKED;
			}
			break;
		case ORC_TRAIN_NORM_1:
		case ORC_TRAIN_NORM_2:
		case ORC_TRAIN_NORM_3:
			if (offs == 244)
				orc_burst_rx_cb(rx, rx->bitbuf, 510, rc);
			else if (rx->event)
				rx->event(ORC_EV_NORM_MISPLACED, rx->bitbuf_start_bitnum, offs, rx->priv);
			break;
		default:
			if (rx->event)
				rx->event(ORC_EV_NO_TRAIN, rx->bitbuf_start_bitnum, 0, rx->priv);
			rx->state = RXS_UNLOCKED;
			break;
		}
		rx->bits_in_buf -= 510;
		memmove(rx->bitbuf, rx->bitbuf + 510, rx->bits_in_buf);
		rx->bitbuf_start_bitnum += 510;
		rx->next_frame_start_bitnum += 510;
		break;
	}
	return (int)len;
}

void orc_rx_feed(struct orc_rx *rx, const uint8_t *bits, size_t len, unsigned chunk)
{
	/* tetra-rx.c:82-95: read(fd, buf, 64) until EOF */
	while (len) {
		unsigned n = (len > chunk) ? chunk : (unsigned)len;
		orc_burst_sync_in(rx, bits, n);
		bits += n;
		len -= n;
	}
}

/* ======================================================================
 * row B -- float_to_bits.c:33-72 (slicer + symbol map), :128-164 (AFC IIR:
 * float state, double intermediate of (1.0 - filter_val)).
 * ==================================================================== */
void orc_float_to_bits(const float *in, size_t n, uint8_t *out, int afc,
		       float filter_val, float filter_goal, float *filter_state)
{
	float filter = filter_state ? *filter_state : 0.0f;
	for (size_t i = 0; i < n; i++) {
		float fl = in[i];
		if (afc) {
			if ((fl > -5.0) && (fl < 5.0))
				filter = filter * (1.0 - filter_val) + (fl - filter_goal) * filter_val;
			fl = fl - filter;
		}
		int sym = (fl > 2) ? 3 : (fl > 0) ? 1 : (fl < -2) ? -3 : -1;
		switch (sym) {
		case -3: out[2 * i] = 1; out[2 * i + 1] = 1; break;
		case  1: out[2 * i] = 0; out[2 * i + 1] = 0; break;
		case  3: out[2 * i] = 0; out[2 * i + 1] = 1; break;
		default: out[2 * i] = 1; out[2 * i + 1] = 0; break;
		}
	}
	if (filter_state)
		*filter_state = filter;
}

/* ======================================================================
 * CPU baseline helper: the per-slot work of burst_rx_cb + tp_sap_udata_ind
 * without callbacks / allocation / printing, for bench.py's cpu_baseline.
 * ==================================================================== */
uint64_t orc_bench_decode_slots(const uint8_t *slots, const uint8_t *types, size_t n,
				uint32_t scramb_init, int use_acc, uint8_t *type1_out, uint16_t *crc_out)
{
	uint64_t ok = 0;
	struct orc_block_result res;
	uint8_t bbk[30], schf[432];
	for (size_t i = 0; i < n; i++) {
		const uint8_t *b = slots + 510 * i;
		uint8_t *o = type1_out ? type1_out + 288 * i : NULL;
		uint16_t crcdummy[2], *c = crc_out ? crc_out + 2 * i : crcdummy;
		c[0] = c[1] = 0;
		switch (types[i]) {
		case ORC_TRAIN_SYNC:
			orc_decode_block(ORC_T_SB1, b + 94, 3, use_acc, &res); ok += (uint64_t)res.crc_ok; c[0] = res.crc;
			if (o) memcpy(o + 14, res.type1, 60);
			orc_decode_block(ORC_T_BBK, b + 252, scramb_init, use_acc, &res);
			if (o) memcpy(o, res.type1, 14);
			orc_decode_block(ORC_T_SB2, b + 282, scramb_init, use_acc, &res); ok += (uint64_t)res.crc_ok; c[1] = res.crc;
			if (o) memcpy(o + 14 + 124, res.type1, 124);
			break;
		case ORC_TRAIN_NORM_2:
			memcpy(bbk, b + 230, 14); memcpy(bbk + 14, b + 266, 16);
			orc_decode_block(ORC_T_BBK, bbk, scramb_init, use_acc, &res);
			if (o) memcpy(o, res.type1, 14);
			orc_decode_block(ORC_T_NDB, b + 14, scramb_init, use_acc, &res); ok += (uint64_t)res.crc_ok; c[0] = res.crc;
			if (o) memcpy(o + 14, res.type1, 124);
			orc_decode_block(ORC_T_NDB, b + 282, scramb_init, use_acc, &res); ok += (uint64_t)res.crc_ok; c[1] = res.crc;
			if (o) memcpy(o + 14 + 124, res.type1, 124);
			break;
		case ORC_TRAIN_NORM_1:
			memcpy(bbk, b + 230, 14); memcpy(bbk + 14, b + 266, 16);
			memcpy(schf, b + 14, 216); memcpy(schf + 216, b + 282, 216);
			orc_decode_block(ORC_T_BBK, bbk, scramb_init, use_acc, &res);
			if (o) memcpy(o, res.type1, 14);
			orc_decode_block(ORC_T_SCH_F, schf, scramb_init, use_acc, &res); ok += (uint64_t)res.crc_ok; c[0] = res.crc;
			if (o) memcpy(o + 14, res.type1, 268);
			break;
		default:
			break;
		}
	}
	return ok;
}

/* ======================================================================
 * Soft-input extension (BASELINE config 5).  The reference has NO soft path:
 * float_to_bits hard-slices (float_to_bits.c:33-72) and viterbi.c:11-23 maps
 * bits to +-127.  What follows is OUR definition,
 * restated here so the GPU path has something to be bit-exact against:
 *   per symbol phi (units of pi/4):  soft0 = sat(rint(64*phi)),
 *                                    soft1 = sat(rint(64*(2-|phi|))),  sat to [-127,127]
 *   sign convention of viterbi.c: positive = bit 0, negative = bit 1, 0 = erasure;
 *   descrambling flips the sign where the scrambling bit is 1;
 *   decoder = libosmocore's accelerated algorithm (correlation metrics), same tie rule.
 * Hard-slicing the soft values (bit = soft < 0) gives float_to_bits' output
 * except at phi = 0, +-2 exactly and NaN (float_to_bits has its own rules there).
 * ==================================================================== */
static inline int8_t sat127(float x)
{
	if (x != x)
		return 0;
	if (x > 127.0f)
		return 127;
	if (x < -127.0f)
		return -127;
	return (int8_t)__builtin_rintf(x);
}

void orc_float_to_soft(const float *in, size_t n, int8_t *out2n)
{
	for (size_t i = 0; i < n; i++) {
		float phi = in[i];
		out2n[2 * i] = sat127(64.0f * phi);
		out2n[2 * i + 1] = sat127(64.0f * (2.0f - __builtin_fabsf(phi)));
	}
}

void orc_decode_block_soft(enum orc_tpsap_type type, const int8_t *soft5, uint32_t scramb_init, struct orc_block_result *res)
{
	const struct orc_blk_param *p = &blk_params[type];
	int8_t t4[432], t3[432];
	static __thread int8_t mother[288 * 4 + 16];
	uint8_t seq[432];

	orc_scramb_get_bits(scramb_init, seq, p->type345_bits);
	for (unsigned i = 0; i < p->type345_bits; i++) {
		t4[i] = seq[i] ? (int8_t)-soft5[i] : soft5[i];
		res->type4[i] = (uint8_t)((soft5[i] < 0) ^ seq[i]);	/* hard decision on the received value, then descrambled */
	}
	res->crc = 0;
	res->crc_ok = 0;
	memset(res->type2, 0, sizeof(res->type2));
	if (p->interleave_a) {
		for (uint32_t i = 1; i <= p->type345_bits; i++)
			t3[i - 1] = t4[(p->interleave_a * i) % p->type345_bits];
		memset(mother, 0, sizeof(mother));
		for (uint32_t j = 1; j <= p->type345_bits; j++)
			mother[punct_k(&punct_defs[ORC_PUNCT_2_3], j) - 1] = t3[j - 1];
		orc_viterbi_soft(mother, res->type2, p->type2_bits);
	}
	if (p->have_crc16) {
		res->crc = orc_crc16_ccitt_bits(res->type2, p->type1_bits + 16u);
		res->crc_ok = (res->crc == ORC_CRC_OK);
	} else if (type == ORC_T_BBK) {
		res->crc_ok = 1;
		memcpy(res->type2, res->type4, p->type2_bits);
	}
	memcpy(res->type1, res->type2, p->type1_bits);
}

/* ======================================================================
 * (f)3 -- GSMTAP message of a decoded block: tetra_gsmtap.c:31-63 (header fields, osmo_ubit2pbit),
 * :19-29 (lchan -> GSMTAP_TETRA_* sub-type), tetra_tdma.c:96-99 (frame number).  struct gsmtap_hdr and the
 * constants are libosmocore's gsmtap.h (un-vendored): version 2, header 16 bytes = {version, hdr_len (words),
 * type, timeslot, arfcn be16, signal_dbm, snr_db, frame_number be32, sub_type, antenna_nr, sub_slot, res};
 * GSMTAP_TYPE_TETRA_I1 = 5; GSMTAP_TETRA_BSCH 1, AACH 2, SCH_HU 3, SCH_HD 4, SCH_F 5, BNCH 6, STCH 7, TCH_F 8.
 * ==================================================================== */
int orc_gsmtap_makemsg(const struct orc_tdma_time *tm, int lchan, uint8_t ts, uint8_t ss, int8_t signal_dbm,
		       uint8_t snr, const uint8_t *bits, unsigned bitlen, uint8_t *out)
{
	/* enum tetra_log_chan (tetra_common.h:22-39): UNKNOWN 0, SCH_F 1, SCH_HD 2, SCH_HU 3, STCH 4, P8 5..7,
	 * AACH 8, TCH 9, BSCH 10, BNCH 11 */
	uint8_t sub = 0;
	switch (lchan) {
	case 1: sub = 5; break;
	case 2: sub = 4; break;
	case 3: sub = 3; break;
	case 4: sub = 7; break;
	case 8: sub = 2; break;
	case 9: sub = 8; break;
	case 10: sub = 1; break;
	case 11: sub = 6; break;
	}
	uint32_t fn = (((uint32_t)tm->hn * 60u) + tm->mn) * 18u + tm->fn;
	unsigned nbytes = (bitlen + 7) / 8;
	memset(out, 0, 16 + nbytes);
	out[0] = 2;
	out[1] = 4;
	out[2] = 5;
	out[3] = ts;
	out[6] = (uint8_t)signal_dbm;
	out[7] = snr;
	out[8] = fn >> 24; out[9] = fn >> 16; out[10] = fn >> 8; out[11] = fn;
	out[12] = sub;
	out[14] = ss;
	for (unsigned i = 0; i < bitlen; i++) {
		unsigned byte = i / 8, bit = 7 - (i % 8);
		out[16 + byte] |= (uint8_t)((bits[i] & 1) << bit);
	}
	return (int)(16 + nbytes);
}

/* the soft chain on n aligned slots of int8 soft values (510 per slot, slot layout of the hard path): the batch form
 * of orc_decode_block_soft() for full-size parity runs */
uint64_t orc_bench_decode_slots_soft(const int8_t *slots, const uint8_t *types, size_t n, uint32_t scramb_init,
				     uint8_t *type1_out, uint16_t *crc_out)
{
	uint64_t ok = 0;
	struct orc_block_result res;
	int8_t bbk[30], schf[432];
	for (size_t i = 0; i < n; i++) {
		const int8_t *b = slots + 510 * i;
		uint8_t *o = type1_out ? type1_out + 288 * i : NULL;
		uint16_t crcdummy[2], *c = crc_out ? crc_out + 2 * i : crcdummy;
		c[0] = c[1] = 0;
		switch (types[i]) {
		case ORC_TRAIN_SYNC:
			orc_decode_block_soft(ORC_T_SB1, b + 94, 3, &res); ok += (uint64_t)res.crc_ok; c[0] = res.crc;
			if (o) memcpy(o + 14, res.type1, 60);
			orc_decode_block_soft(ORC_T_BBK, b + 252, scramb_init, &res);
			if (o) memcpy(o, res.type1, 14);
			orc_decode_block_soft(ORC_T_SB2, b + 282, scramb_init, &res); ok += (uint64_t)res.crc_ok; c[1] = res.crc;
			if (o) memcpy(o + 14 + 124, res.type1, 124);
			break;
		case ORC_TRAIN_NORM_2:
			memcpy(bbk, b + 230, 14); memcpy(bbk + 14, b + 266, 16);
			orc_decode_block_soft(ORC_T_BBK, bbk, scramb_init, &res);
			if (o) memcpy(o, res.type1, 14);
			orc_decode_block_soft(ORC_T_NDB, b + 14, scramb_init, &res); ok += (uint64_t)res.crc_ok; c[0] = res.crc;
			if (o) memcpy(o + 14, res.type1, 124);
			orc_decode_block_soft(ORC_T_NDB, b + 282, scramb_init, &res); ok += (uint64_t)res.crc_ok; c[1] = res.crc;
			if (o) memcpy(o + 14 + 124, res.type1, 124);
			break;
		case ORC_TRAIN_NORM_1:
			memcpy(bbk, b + 230, 14); memcpy(bbk + 14, b + 266, 16);
			memcpy(schf, b + 14, 216); memcpy(schf + 216, b + 282, 216);
			orc_decode_block_soft(ORC_T_BBK, bbk, scramb_init, &res);
			if (o) memcpy(o, res.type1, 14);
			orc_decode_block_soft(ORC_T_SCH_F, schf, scramb_init, &res); ok += (uint64_t)res.crc_ok; c[0] = res.crc;
			if (o) memcpy(o + 14, res.type1, 268);
			break;
		default:
			break;
		}
	}
	return ok;
}
