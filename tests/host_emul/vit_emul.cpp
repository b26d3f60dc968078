/*
 * vit_emul.cpp -- compiles the product's per-lane trellis code (osmo-tetra_amd/csrc/vit_core.h)
 * and its bit layout (tg_layout.h) for the HOST, so that the exact arithmetic the HIP kernels
 * run can be checked against the oracle on a machine without a GPU.
 *
 * TEST INFRASTRUCTURE: lives under tests/, is never loaded by the product library.
 */
#include <stdint.h>
#include <string.h>

#include "tg_layout.h"
#include "tg_conv.h"
#include "vit_core.h"
#include "slot_core.h"

static uint16_t crc_lsb[256], crc_msb[256];
static bool crc_ready;

static void crc_init()
{
	if (crc_ready)
		return;
	tg_crc16_make_table(crc_lsb);
	for (int x = 0; x < 256; x++) {
		int rv = 0;
		for (int i = 0; i < 8; i++)
			if (x & (1 << i))
				rv |= 0x80 >> i;
		crc_msb[x] = crc_lsb[rv];
	}
	crc_ready = true;
}

/* pack the type-4 bits (already descrambled, stream order inside the block) of one block */
extern "C" void emul_pack_block(int kind, const uint8_t *type4, uint32_t *words)
{
	const int nw = tg_kind_nblk(kind) / 2;
	for (int d = 0; d < nw; d++) {
		uint32_t w = 0;
		for (int p = 0; p < 32; p++) {
			int j = tg_codeword_src(kind, d, p);
			if (j >= 0 && (type4[j] & 1))
				w |= 1u << p;
		}
		words[d] = w;
	}
}

/* decode one block from packed words; out_bits: type-2 bits (8*nblk of them), returns crc */
extern "C" unsigned emul_decode_words(int kind, const uint32_t *words, uint8_t *out_bits)
{
	crc_init();
	const int nblk = tg_kind_nblk(kind), nw = nblk / 2;
	static uint8_t hist[36][16];
	/* the kernels' branch-metric table (LDS there): the difference form's ten dwords per pair and triple, three arrays */
	static uint32_t bmdtab[TG_BMD_WORDS];
	tg_bmd_build(bmdtab);
	auto bmd = [&](int p, uint32_t o, uint32_t w[10]) {
		const int q = 8 * p + (int)(o >> 4);
		memcpy(w, bmdtab + TG_BMD_A0 + 4 * q, 16);
		memcpy(w + 4, bmdtab + TG_BMD_A1 + 4 * q, 16);
		memcpy(w + 8, bmdtab + TG_BMD_A2 + 4 * q, 8);
	};
	tg_vit_state v;
	tg_vit_init(v);
	tg_vit_leadin_bmd(v, words[0] >> 24, bmd);
	for (int it = 0; it < nw; it++) {
		uint32_t h[4];
		tg_vit_block_bmd<false>(v, words[it], h, bmd);
		memcpy(hist[2 * it], h, 16);
		if (it == nw - 1)
			tg_vit_block_bmd<true>(v, words[it] >> 12, h, bmd);
		else
			tg_vit_block_bmd<false>(v, words[it] >> 12, h, bmd);
		memcpy(hist[2 * it + 1], h, 16);
		if (kind == TG_KIND_432 && it == 8)
			tg_vit_normalize_floor(v);
	}
	uint8_t bytes[37] = { 0 };
	uint32_t s = 0;
	for (int b = nblk - 1; b >= 0; b--) {
		uint8_t byte = hist[b][s];
		bytes[b] = byte;
		s = tg_brev4(byte);
	}
	for (int i = 0; i < 8 * nblk; i++)
		out_bits[i] = (bytes[i >> 3] >> (i & 7)) & 1;
	auto tl = [](uint32_t x) -> uint16_t { return crc_lsb[x & 255]; };
	auto tm = [](uint32_t x) -> uint16_t { return crc_msb[x & 255]; };
	(void)tm;
	/* same recurrence as the kernel */
	uint32_t crc = 0xffff;
	for (int i = 0; i < nblk - 1; i++)
		crc = ((crc << 8) & 0xffff) ^ crc_msb[crc >> 8] ^ tl(bytes[i]);
	uint32_t nib = bytes[nblk - 1] & 15;
	for (int i = 0; i < 4; i++) {
		crc ^= ((nib >> i) & 1) << 15;
		crc = (crc & 0x8000) ? (((crc << 1) ^ 0x1021) & 0xffff) : ((crc << 1) & 0xffff);
	}
	return crc;
}

/* how often a two-bit step of the difference form met a metric below TG_VIT_FLOOR since the last call (vit_core.h: must stay 0) */
extern "C" unsigned long emul_floor_violations(void)
{
	const unsigned long n = tg_vit_floor_violations;
	tg_vit_floor_violations = 0;
	return n;
}

/* the slot-level gather the front kernel performs (same table function) */
extern "C" void emul_pack_slot(int btype, const uint8_t *slot, uint32_t *words20)
{
	for (int w = 0; w < TG_PACKED_WORDS; w++) {
		uint32_t x = 0;
		for (int p = 0; p < 32; p++) {
			int o = tg_packed_src(btype, w, p);
			if (o >= 0 && (slot[o] & 1))
				x |= 1u << p;
		}
		words20[w] = x;
	}
}

/* ---- soft trellis: type-4 soft values (descrambled = mask 0) -> decoded bits ---- */
extern "C" void emul_soft_layout(int kind, const int8_t *soft4, int8_t *area /* 8 + 12*nblk */)
{
	const int K = tg_kind_K(kind), a = tg_kind_a(kind), nblk = tg_kind_nblk(kind);
	memset(area, 0, 8 + 12 * nblk);
	for (int i = 0; i < K; i++) {
		const int8_t v = soft4[(a * (i + 1)) % K];	/* type3[i] */
		if (i < 6)
			area[i] = v;
		else
			area[8 + (i - 6)] = v;
	}
}

extern "C" unsigned emul_decode_soft(int kind, const int8_t *area, const uint32_t *maskwords, uint8_t *out_bits)
{
	crc_init();
	const int nblk = tg_kind_nblk(kind);
	static uint8_t hist[36][16];
	const uint32_t *aw = (const uint32_t *)area;
	tg_svit_state v;
	tg_svit_init(v);
	tg_svit_leadin(v, aw, maskwords ? (maskwords[0] >> 24) & 0x3f : 0);
	for (int b = 0; b < nblk; b++) {
		uint32_t h[4];
		const uint32_t m12 = maskwords ? (maskwords[b >> 1] >> (12 * (b & 1))) & 0xfff : 0;
		if (b == nblk - 1)
			tg_svit_block<true>(v, aw + 2 + 3 * b, m12, h);
		else
			tg_svit_block<false>(v, aw + 2 + 3 * b, m12, h);
		memcpy(hist[b], h, 16);
	}
	uint8_t bytes[37] = { 0 };
	uint32_t s = 0;
	for (int b = nblk - 1; b >= 0; b--) {
		uint8_t byte = hist[b][s];
		bytes[b] = byte;
		s = tg_brev4(byte);
	}
	for (int i = 0; i < 8 * nblk; i++)
		out_bits[i] = (bytes[i >> 3] >> (i & 7)) & 1;
	uint32_t crc = 0xffff;
	for (int i = 0; i < nblk - 1; i++)
		crc = ((crc << 8) & 0xffff) ^ crc_msb[crc >> 8] ^ crc_lsb[bytes[i]];
	uint32_t nib = bytes[nblk - 1] & 15;
	for (int i = 0; i < 4; i++) {
		crc ^= ((nib >> i) & 1) << 15;
		crc = (crc & 0x8000) ? (((crc << 1) ^ 0x1021) & 0xffff) : ((crc << 1) & 0xffff);
	}
	return crc;
}

/* the packed 16-bit soft trellis (tg_pvit_*, what k_vit<., 2> runs): same area and mask words in, same outputs.
 * *maxmetric (optional) receives the largest 12-bit metric seen before a normalisation (must stay < 4096). */
extern "C" unsigned emul_decode_psoft(int kind, const int8_t *area, const uint32_t *maskwords, uint8_t *out_bits, unsigned *maxmetric)
{
	crc_init();
	const int nblk = tg_kind_nblk(kind);
	static uint32_t hist[36][4];
	static uint32_t T[TG_PSOFT_TAB];
	for (uint32_t i = 0; i < TG_PSOFT_TAB; i++)
		T[i] = tg_psoft_entry(i);
	auto tab = [&](uint32_t idx) { return T[idx]; };
	const uint32_t *aw = (const uint32_t *)area;
	unsigned mx = 0;
	auto track = [&](const tg_pvit_state &v) {
		for (int k = 0; k < 8; k++) {
			if ((unsigned)(v.Z[k].x >> 4) > mx) mx = v.Z[k].x >> 4;
			if ((unsigned)(v.Z[k].y >> 4) > mx) mx = v.Z[k].y >> 4;
		}
	};
	tg_pvit_state v;
	tg_pvit_init(v);
	tg_pvit_leadin(v, aw, maskwords ? (maskwords[0] >> 24) & 0x3f : 0, tab);
	track(v);
	for (int b = 0; b < nblk; b++) {
		const uint32_t m12 = maskwords ? (maskwords[b >> 1] >> (12 * (b & 1))) & 0xfff : 0;
		uint32_t t[12];
		tg_psoft_fetch<0, 12>(aw + 2 + 3 * b, m12, tab, t);
		if (b == nblk - 1) {
			tg_pvit_block<true>(v, t, hist[b]);
		} else {
			tg_pvit_block<false>(v, t, hist[b]);
			track(v);
		}
		if ((b & 1) && b != nblk - 1)
			tg_vit_normalize(v);	/* the kernel's schedule: after every second block */
	}
	if (maxmetric)
		*maxmetric = mx;
	uint8_t bytes[37] = { 0 };
	uint32_t pos = 0;
	for (int b = nblk - 1; b >= 0; b--) {
		const uint32_t hi = tg_ptrace_hop(hist[b][2], hist[b][3], pos);
		const uint32_t lo = tg_ptrace_hop(hist[b][0], hist[b][1], pos);
		bytes[b] = (uint8_t)(lo | (hi << 4));
	}
	for (int i = 0; i < 8 * nblk; i++)
		out_bits[i] = (bytes[i >> 3] >> (i & 7)) & 1;
	uint32_t crc = 0xffff;
	for (int i = 0; i < nblk - 1; i++)
		crc = ((crc << 8) & 0xffff) ^ crc_msb[crc >> 8] ^ crc_lsb[bytes[i]];
	uint32_t nib = bytes[nblk - 1] & 15;
	for (int i = 0; i < 4; i++) {
		crc ^= ((nib >> i) & 1) << 15;
		crc = (crc & 0x8000) ? (((crc << 1) ^ 0x1021) & 0xffff) : ((crc << 1) & 0xffff);
	}
	return crc;
}

/* the generic trellis (k_conv) for one block: same step programs, step function, normalisation schedule and
 * block-wise traceback as the kernel.  returns 0, or -1 for a shape the product rejects */
template <int CODE, bool G3>
static void conv_decode(const uint32_t *steps, unsigned t3len, unsigned L, const uint8_t *type3, uint8_t *type2)
{
	static uint8_t hist[64][16];
	static uint8_t cls[(TG_CONV_MAX_T3 + 3) / 4];
	const unsigned nblk = (L + 7) / 8;
	/* the kernel's staging: four received bytes -> one class byte (positions past the block: erased) */
	for (unsigned q = 0; q < (t3len + 3) / 4; q++) {
		uint32_t x = 0xffffffffu;
		for (unsigned k = 0; k < 4 && 4 * q + k < t3len; k++)
			x = (x & ~(0xffu << (8 * k))) | ((uint32_t)type3[4 * q + k] << (8 * k));
		cls[q] = (uint8_t)tg_conv_pack4(x);
	}
	auto fetch = [&](uint32_t q) -> uint32_t { return cls[q]; };
	tg_vit_state v;
	uint32_t h[4];
	tg_vit_init(v);
	tg_conv_block<CODE, G3, 4>(v, steps, 4, fetch, h);
	for (unsigned b = 0; b < nblk; b++) {
		if (b && !(b & 7))
			tg_vit_normalize(v);
		const unsigned left = L + 4 - (4 + 8 * b);
		if (left >= 8)
			tg_conv_block<CODE, G3, 8>(v, steps + 3 * (4 + 8 * b), 8, fetch, h);
		else
			tg_conv_block<CODE, G3, 0>(v, steps + 3 * (4 + 8 * b), (int)left, fetch, h);
		memcpy(hist[b], h, 16);
	}
	uint32_t s = 0;
	for (int b = (int)nblk - 1; b >= 0; b--) {
		const uint32_t byte = hist[b][s];
		for (unsigned i = 0; i < 8 && 8 * b + i < L; i++)
			type2[8 * b + i] = (byte >> i) & 1;
		s = tg_brev4(byte);
	}
}

extern "C" int emul_conv_decode(int pu, int mother, unsigned t3len, unsigned L, const uint8_t *type3, uint8_t *type2)
{
	static uint32_t steps[(TG_CONV_MAX_T2 + 4) * TG_CONV_DESC_WORDS];
	if (tg_conv_build_steps(pu, mother, t3len, L, steps))
		return -1;
	const bool g3 = tg_conv_uses_g3(steps, L);
	if (mother == 3)
		g3 ? conv_decode<1, true>(steps, t3len, L, type3, type2) : conv_decode<1, false>(steps, t3len, L, type3, type2);
	else
		g3 ? conv_decode<0, true>(steps, t3len, L, type3, type2) : conv_decode<0, false>(steps, t3len, L, type3, type2);
	return 0;
}

/* arithmetic form (tg_vit_block) vs table form (tg_vit_block_bm) on one random block sequence: 0 if equal */
extern "C" int emul_bm_selfcheck(uint32_t seed, int nblocks)
{
	static uint32_t bmtab[TG_BM_WORDS];
	tg_bm_build(bmtab);
	auto bm = [&](int p, uint32_t e, uint32_t w[8]) { memcpy(w, bmtab + (8 * p + e) * 8, 32); };
	static uint32_t bmdtab[TG_BMD_WORDS];
	tg_bmd_build(bmdtab);
	auto bmd = [&](int p, uint32_t o, uint32_t w[10]) {	/* (the kernels' three-array layout; o = 16 x triple) */
		const int q = 8 * p + (int)(o >> 4);
		memcpy(w, bmdtab + TG_BMD_A0 + 4 * q, 16);
		memcpy(w + 4, bmdtab + TG_BMD_A1 + 4 * q, 16);
		memcpy(w + 8, bmdtab + TG_BMD_A2 + 4 * q, 8);
	};
	tg_vit_state a, b, c, d;	/* arithmetic / table entries of six dwords / of eight (swapped forms from the table) / difference form */
	tg_vit_init(a);
	tg_vit_init(b);
	tg_vit_init(c);
	tg_vit_init(d);
	uint32_t x = seed;
	tg_vit_leadin(a, x & 63);
	tg_vit_leadin_bm<false>(b, x & 63, bm);
	tg_vit_leadin_bm<true>(c, x & 63, bm);
	tg_vit_leadin_bmd(d, x & 63, bmd);
	if (memcmp(&a.Z, &d.Z, sizeof(a.Z)))
		return -1;
	for (int i = 0; i < nblocks; i++) {
		uint32_t ha[4], hb[4], hc[4], hd[4];
		x = x * 1664525u + 1013904223u;
		if (i == nblocks - 1) {
			tg_vit_block<true>(a, x >> 8, ha);
			tg_vit_block_bm<true, false>(b, x >> 8, hb, bm);
			tg_vit_block_bm<true, true>(c, x >> 8, hc, bm);
			tg_vit_block_bmd<true>(d, x >> 8, hd, bmd);
		} else {
			tg_vit_block<false>(a, x >> 8, ha);
			tg_vit_block_bm<false, false>(b, x >> 8, hb, bm);
			tg_vit_block_bm<false, true>(c, x >> 8, hc, bm);
			tg_vit_block_bmd<false>(d, x >> 8, hd, bmd);
		}
		if (memcmp(ha, hb, 16) || memcmp(&a.Z, &b.Z, sizeof(a.Z)) || memcmp(ha, hc, 16) || memcmp(&a.Z, &c.Z, sizeof(a.Z)))
			return i + 1;
		if (memcmp(ha, hd, 16) || memcmp(&a.Z, &d.Z, sizeof(a.Z)))
			return 1000 + i + 1;
		if ((i & 7) == 7) {
			tg_vit_normalize_floor(a);
			tg_vit_normalize_floor(b);
			tg_vit_normalize_floor(c);
			tg_vit_normalize_floor(d);
		}
	}
	return 0;
}

/* ---- one lane = one slot (slot_core.h): the schedule k_slot / k_slot_t run, for the host ----
 * words20: a packed slot (emul_pack_slot) whose blocks are already descrambled.  out: od[36] decoded bytes, crc[2],
 * bits[17][16] = the record's bytes 48..319 as the pieces say, sync[3] = the SYNC PDU words (SYNC bursts).  Returns 0. */
extern "C" int emul_decode_slot(int btype, const uint32_t *words20, uint8_t *od_bytes, uint32_t *crc, uint8_t *bits272, uint32_t *sync3)
{
	crc_init();
	if (btype != TG_BURST_NORM_1 && btype != TG_BURST_NORM_2 && btype != TG_BURST_SYNC)
		return -1;
	const bool sb = btype == TG_BURST_SYNC, two = btype != TG_BURST_NORM_1;
	/* the staging step of the kernels: code word g of the schedule */
	uint32_t col[18];
	for (int g = 0; g < 18; g++) {
		if (!sb)
			col[g] = words20[g];
		else
			col[g] = (g >= 9) ? words20[g] : (g >= TG_SLOT_SB1_G0) ? words20[g - TG_SLOT_SB1_G0] : 0u;
	}
	static uint32_t bmdtab[TG_BMD_WORDS];
	tg_bmd_build(bmdtab);
	auto bmdo = [&](uint32_t o, uint32_t w[10]) {
		const int q = (int)(o >> 4);
		memcpy(w, bmdtab + TG_BMD_A0 + 4 * q, 16);
		memcpy(w + 4, bmdtab + TG_BMD_A1 + 4 * q, 16);
		memcpy(w + 8, bmdtab + TG_BMD_A2 + 4 * q, 8);
	};
	uint32_t H[TG_SLOT_NBLK][4];
	tg_vit_state v;
	tg_slot_state_init(v);
	uint32_t cur = col[0];
	tg_slot_leadin(v, cur >> 24, bmdo);
	for (int g = 0; g < 18; g++) {
		const uint32_t nxt = g < 17 ? col[g + 1] : 0u;
		if (g == TG_SLOT_SB1_G0)
			tg_slot_sb_prologue(v, sb, cur, bmdo);
		tg_slot_block(v, cur, H[2 * g], bmdo);
		if (g == 8) {
			tg_slot_mid(v, two, cur >> 12, nxt >> 24, H[2 * g + 1], bmdo);
			tg_vit_normalize_floor(v);
		} else if (g == 17)
			tg_slot_block_last(v, cur >> 12, H[2 * g + 1], bmdo);
		else
			tg_slot_block(v, cur >> 12, H[2 * g + 1], bmdo);
		cur = nxt;
	}
	uint32_t od[TG_SLOT_NOD + 1] = { 0 };
	uint32_t s = 0;
#define HOP(B) tg_slot_hop<B>(od, s, two, H[B][0], H[B][1], H[B][2], H[B][3]);
	HOP(35) HOP(34) HOP(33) HOP(32) HOP(31) HOP(30) HOP(29) HOP(28) HOP(27) HOP(26) HOP(25) HOP(24) HOP(23) HOP(22) HOP(21) HOP(20) HOP(19) HOP(18)
	HOP(17) HOP(16) HOP(15) HOP(14) HOP(13) HOP(12) HOP(11) HOP(10) HOP(9) HOP(8) HOP(7) HOP(6) HOP(5) HOP(4) HOP(3) HOP(2) HOP(1) HOP(0)
#undef HOP
	memcpy(od_bytes, od, 36);
	auto tl = [](uint32_t x) -> uint32_t { return crc_lsb[x & 255]; };
	auto tm = [](uint32_t x) -> uint32_t { return crc_msb[x & 255]; };
	tg_slot_crc(od, two, sb, tl, tm, crc[0], crc[1]);
#define PIECE(P) { bool full, empty; const uint32_t h16 = tg_slot_piece_bits<P>(od, two, sb, full, empty); uint32_t o[4]; \
		   tg_slot_spread16(h16, o); if (!full) o[3] = 0; if (empty) o[0] = o[1] = o[2] = o[3] = 0; memcpy(bits272 + 16 * (P), o, 16); }
	PIECE(0) PIECE(1) PIECE(2) PIECE(3) PIECE(4) PIECE(5) PIECE(6) PIECE(7) PIECE(8) PIECE(9) PIECE(10) PIECE(11) PIECE(12) PIECE(13)
	PIECE(14) PIECE(15) PIECE(16)
#undef PIECE
	sync3[0] = sync3[1] = sync3[2] = 0;
	if (sb)
		tg_slot_sync_fields(od + 2, sync3[0], sync3[1], sync3[2]);
	return 0;
}
