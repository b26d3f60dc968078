/*
 * tg_synth.c -- synthetic TETRA downlink slots (the TX side of the chain), used to feed
 * benchmarks and tests with valid SB / NDB bursts.  Host code, multi-threaded.
 *
 * Follows the reference's test encoder conv_enc_test.c:88-156 (type-1 -> CRC-16 -> tail ->
 * rate-1/4 mother code -> 2/3 puncturing -> block interleaver -> scrambler -> burst) and the
 * burst layouts of phy/tetra_burst.c:169-267 (EN 300 392-2 clause 9.4.4.2).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "tetra_gpu.h"
#include "tg_layout.h"

static inline uint64_t splitmix64(uint64_t *x)
{
	uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}

struct bitrng {
	uint64_t st, cur;
	int left;
};

static inline unsigned rng_bit(struct bitrng *r)
{
	if (!r->left) {
		r->cur = splitmix64(&r->st);
		r->left = 64;
	}
	unsigned b = (unsigned)(r->cur & 1);
	r->cur >>= 1;
	r->left--;
	return b;
}

static inline double rng_unit(struct bitrng *r)
{
	return (double)(splitmix64(&r->st) >> 11) * (1.0 / 9007199254740992.0);
}

static uint16_t crc16_bits(const uint8_t *b, int n)
{
	uint16_t crc = 0xffff;
	for (int i = 0; i < n; i++) {
		unsigned top = ((crc >> 15) ^ b[i]) & 1;
		crc = (uint16_t)(crc << 1);
		if (top)
			crc ^= 0x1021;
	}
	return crc;
}

static uint32_t lfsr_bit(uint32_t *st)
{
	const uint32_t s = *st;
	const uint32_t fb = __builtin_parity(s & 0xdb710641u);
	*st = (s >> 1) | (fb << 31);
	return fb;
}

/* type-1 bits -> type-5 bits of one block */
static void encode_block(int kind, const uint8_t *type1, uint32_t code, uint8_t *type5)
{
	const int K = tg_kind_K(kind), a = tg_kind_a(kind), n1 = tg_kind_type1(kind), n2 = n1 + 20;
	uint8_t t2[288], t3[432];

	memcpy(t2, type1, (size_t)n1);
	uint16_t crc = (uint16_t)~crc16_bits(t2, n1);
	for (int i = 0; i < 16; i++)
		t2[n1 + i] = (crc >> (15 - i)) & 1;
	memset(t2 + n1 + 16, 0, 4);

	/* mother code g1..g4 with 2/3 puncturing: per pair of input bits keep g1,g2 of the first
	 * and g1 of the second (P = {1,2,5} of every 8 mother bits, tetra_conv_enc.c:96,128-134) */
	unsigned sr = 0;	/* bit0 = D ... bit3 = D^4 */
	int o = 0;
	for (int n = 0; n < n2; n++) {
		unsigned b = t2[n];
		unsigned g1 = (b ^ sr ^ (sr >> 3)) & 1;				/* 1 + D + D^4 */
		unsigned g2 = (b ^ (sr >> 1) ^ (sr >> 2) ^ (sr >> 3)) & 1;	/* 1 + D^2 + D^3 + D^4 */
		t3[o++] = (uint8_t)g1;
		if (!(n & 1))
			t3[o++] = (uint8_t)g2;
		sr = ((sr << 1) | b) & 15;
	}
	/* interleave: type4[(a*i) mod K] = type3[i-1], i = 1..K (tetra_interleave.c:41-49) */
	uint32_t st = code;
	uint8_t t4[432];
	for (int i = 1; i <= K; i++)
		t4[(a * i) % K] = t3[i - 1];
	for (int i = 0; i < K; i++)
		type5[i] = t4[i] ^ (uint8_t)lfsr_bit(&st);
}

static const uint16_t rm_par[14] = { 0x9b60, 0x2de0, 0xfc20, 0xe03c, 0x983a, 0x5436, 0x2c2e,
				     0xffdf, 0x8339, 0x42b5, 0x21ad, 0x1273, 0x096b, 0x04e7 };

static void encode_bbk(const uint8_t *aach14, uint32_t code, uint8_t *type5)
{
	/* systematic (30,14) code, EN 300 392-2 8.2.3.2: 14 info bits then 16 parity bits */
	uint16_t par = 0;
	for (int i = 0; i < 14; i++) {
		type5[i] = aach14[i] & 1;
		if (type5[i])
			par ^= rm_par[i];
	}
	for (int i = 0; i < 16; i++)
		type5[14 + i] = (par >> (15 - i)) & 1;
	uint32_t st = code;
	for (int i = 0; i < 30; i++)
		type5[i] ^= (uint8_t)lfsr_bit(&st);
}

static const uint8_t seq_n[22] = { 1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0 };
static const uint8_t seq_p[22] = { 0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0 };
static const uint8_t seq_q[22] = { 1,0,1,1,0,1,1,1,0,0,0,0,0,1,1,0,1,0,1,1,0,1 };
static const uint8_t seq_y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };

/* phase adjustment bits, EN 300 392-2 9.4.4.3.6: make the phase accumulated over symbols
 * n1..n2 a multiple of 2*pi */
static void phase_adjust(const uint8_t *slot, int n1, int n2, uint8_t *out2)
{
	static const int8_t dphi[4] = { 1, -1, 3, -3 };	/* index = first bit | second bit << 1, units of pi/4 */
	int sum = 0;
	for (int n = n1 - 1; n < n2; n++)
		sum += dphi[slot[2 * n] | (slot[2 * n + 1] << 1)];
	int adj = -(sum % 8);
	if (adj > 3) adj -= 8;
	if (adj < -3) adj += 8;
	out2[0] = (adj == -3 || adj == 3);
	out2[1] = (adj == -3 || adj == -1);
}

static void sync_pdu(const struct tgpu_synth_cfg *c, uint32_t tn, uint32_t fn, uint32_t mn, uint8_t *t1)
{
	/* SYNC PDU fields as read back in lower_mac/tetra_lower_mac.c:284-297 (testpdu.c:43-58) */
	memset(t1, 0, 60);
#define PUT(off, len, v) do { for (int i_ = 0; i_ < (len); i_++) t1[(off) + i_] = ((v) >> ((len) - 1 - i_)) & 1; } while (0)
	PUT(4, 6, c->cc);
	PUT(10, 2, (tn - 1) & 3);
	PUT(12, 5, fn);
	PUT(17, 6, mn);
	PUT(31, 10, c->mcc);
	PUT(41, 14, c->mnc);
#undef PUT
}

static void payload(struct bitrng *r, int n, int hdr, uint8_t *out)
{
	for (int i = 0; i < n; i++)
		out[i] = (uint8_t)rng_bit(r);
	if (hdr) {	/* MAC-RESOURCE, fill 0, pos-of-grant 0, enc 0, random access 0, length 2, address type 0 */
		memset(out, 0, 16);
		out[10] = 1;
	}
}

static void make_slot(const struct tgpu_synth_cfg *c, uint64_t idx, int type, uint8_t *slot, uint8_t *t1out)
{
	struct bitrng r = { c->seed + idx * 0x632be59bd9b4e019ull, 0, 0 };
	uint8_t t1[268], b1[432], b2[216], bb[30], aach[14] = { 0 };
	memset(slot, 0, TG_SLOT_BITS);
	if (t1out)
		memset(t1out, 0, 288);

	/* common frame: q11..q22, 2 phase bits, ..., 2 phase bits, q1..q10 */
	memcpy(slot, seq_q + 10, 12);
	memcpy(slot + 500, seq_q, 10);

	if (type == TETRA_TRAIN_SYNC) {
		/* time fields derived from the slot index so that consecutive slots look like a live cell */
		uint32_t tn = (uint32_t)(idx % 4) + 1, fn = (uint32_t)((idx / 4) % 18) + 1, mn = (uint32_t)((idx / 72) % 60) + 1;
		sync_pdu(c, tn, fn, mn, t1);
		if (t1out)
			memcpy(t1out + 14, t1, 60);
		encode_block(TG_KIND_SB1, t1, SCRAMB_INIT, b1);
		payload(&r, 124, 0, t1);
		t1[0] = 1; t1[1] = 0;		/* MAC PDU type: broadcast */
		if (t1out)
			memcpy(t1out + 14 + 124, t1, 124);
		encode_block(TG_KIND_216, t1, c->scramb_init, b2);
		encode_bbk(aach, c->scramb_init, bb);
		memset(slot + 14, 1, 8);	/* frequency correction field f1..f8, f73..f80 = 1 */
		memset(slot + 14 + 72, 1, 8);
		memcpy(slot + TG_SB_BLK1_OFF, b1, 120);
		memcpy(slot + TG_SYNC_TRAIN_OFF, seq_y, 38);
		memcpy(slot + TG_SB_BBK_OFF, bb, 30);
		memcpy(slot + TG_SB_BLK2_OFF, b2, 216);
		phase_adjust(slot, 8, 108, slot + 12);
		phase_adjust(slot, 109, 249, slot + 498);
	} else {
		encode_bbk(aach, c->scramb_init, bb);
		if (type == TETRA_TRAIN_NORM_1) {
			payload(&r, 268, c->null_pdu_header, t1);
			if (t1out)
				memcpy(t1out + 14, t1, 268);
			encode_block(TG_KIND_432, t1, c->scramb_init, b1);
			memcpy(slot + TG_NDB_BLK1_OFF, b1, 216);
			memcpy(slot + TG_NDB_BLK2_OFF, b1 + 216, 216);
			memcpy(slot + TG_NORM_TRAIN_OFF, seq_n, 22);
		} else {
			payload(&r, 124, c->null_pdu_header, t1);
			if (t1out)
				memcpy(t1out + 14, t1, 124);
			encode_block(TG_KIND_216, t1, c->scramb_init, b1);
			payload(&r, 124, c->null_pdu_header, t1);
			if (t1out)
				memcpy(t1out + 14 + 124, t1, 124);
			encode_block(TG_KIND_216, t1, c->scramb_init, b2);
			memcpy(slot + TG_NDB_BLK1_OFF, b1, 216);
			memcpy(slot + TG_NDB_BLK2_OFF, b2, 216);
			memcpy(slot + TG_NORM_TRAIN_OFF, seq_p, 22);
		}
		memcpy(slot + TG_NDB_BBK1_OFF, bb, 14);
		memcpy(slot + TG_NDB_BBK2_OFF, bb + 14, 16);
		phase_adjust(slot, 8, 122, slot + 12);
		phase_adjust(slot, 123, 249, slot + 498);
	}

	if (c->ber > 0.0) {
		/* flips only inside the coded fields: training sequences are matched exactly */
		static const int f_sb[3][2] = { { 94, 214 }, { 252, 282 }, { 282, 498 } };
		static const int f_nb[4][2] = { { 14, 230 }, { 230, 244 }, { 266, 282 }, { 282, 498 } };
		const int (*f)[2] = (type == TETRA_TRAIN_SYNC) ? f_sb : f_nb;
		const int nf = (type == TETRA_TRAIN_SYNC) ? 3 : 4;
		for (int k = 0; k < nf; k++)
			for (int i = f[k][0]; i < f[k][1]; i++)
				if (rng_unit(&r) < c->ber)
					slot[i] ^= 1;
	}
}

struct job {
	const struct tgpu_synth_cfg *cfg;
	const uint8_t *types;
	size_t lo, hi;
	uint8_t *out, *t1;
};

static void *worker(void *arg)
{
	struct job *j = arg;
	for (size_t i = j->lo; i < j->hi; i++)
		make_slot(j->cfg, i, j->types[i], j->out + i * TG_SLOT_BITS, j->t1 ? j->t1 + i * 288 : NULL);
	return NULL;
}

int tgpu_synth_slots(const struct tgpu_synth_cfg *cfg, const uint8_t *types, size_t n, uint8_t *out, uint8_t *type1_out)
{
	if (!cfg || !types || !out)
		return TGPU_EINVAL;
	long nc = sysconf(_SC_NPROCESSORS_ONLN);
	int nt = (int)(nc < 1 ? 1 : nc > 64 ? 64 : nc);
	if (n < 4096)
		nt = 1;
	pthread_t th[64];
	struct job jobs[64];
	for (int t = 0; t < nt; t++) {
		jobs[t] = (struct job){ cfg, types, n * (size_t)t / (size_t)nt, n * (size_t)(t + 1) / (size_t)nt, out, type1_out };
		if (nt == 1)
			worker(&jobs[t]);
		else
			pthread_create(&th[t], NULL, worker, &jobs[t]);
	}
	if (nt > 1)
		for (int t = 0; t < nt; t++)
			pthread_join(th[t], NULL);
	return TGPU_OK;
}
