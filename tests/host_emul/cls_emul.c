/*
 * cls_emul.c -- test infrastructure: a plain-C statement of what k_front_stream (+ fix) leaves per grid slot on a
 * stream of 0 / 1 bytes -- classification word, SYNC-sequence summary -- so that the CPU test-suite can drive the
 * host walk and the device form of the walk (tg_walk_core.h) on streams of bench size without a GPU.  Same semantics
 * as emul_cls() / emul_ysum() in tests/test_stream_sync_cpu.py (which it is checked against there); never part of
 * the product.
 */
#include <stdint.h>
#include <string.h>

static const uint8_t Y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };
static const uint8_t N[22] = { 1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0 };
static const uint8_t P[22] = { 0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0 };
static const uint8_t Q[22] = { 1,0,1,1,0,1,1,1,0,0,0,0,0,1,1,0,1,0,1,1,0,1 };
static const uint8_t X[22] = { 1,0,0,1,1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1 };	/* first 22 of the 30-bit extended sequence */

/* phy/tetra_burst.c:289-297 for a position c < 21: the look-ahead window holds the stream with in[20] missing */
static int skewed_gate(const uint8_t *in, uint32_t c)
{
	uint8_t e[22];
	if (c == 0) {
		e[0] = 0;
		memcpy(e + 1, in, 20);
		e[21] = in[21];
	} else {
		memcpy(e, in + c - 1, 21 - c);
		memcpy(e + 21 - c, in + 21, c + 1);
	}
	return !memcmp(e, Y, 22) || !memcmp(e, N, 22) || !memcmp(e, P, 22) || !memcmp(e, Q, 22) || !memcmp(e, X, 22);
}

void emul_cls_ysum(const uint8_t *s, uint64_t L, uint64_t anchor, uint32_t chunk, uint32_t view, uint32_t *cls, uint16_t *ysum)
{
	const uint64_t n = L >= anchor + 510 ? (L - anchor) / 510 : 0;
	uint8_t buf[1280];
	for (uint64_t i = 0; i < n; i++) {
		const uint64_t bs = anchor + 510 * i;
		uint64_t f = (bs + 510 + chunk - 1) / chunk * chunk;
		if (f > L)
			f = L;
		const uint32_t w = (uint32_t)(f - bs), wv = w < view ? w : view;
		memset(buf, 0, sizeof(buf));
		const uint64_t have = L - bs < 1200 ? L - bs : 1200;
		memcpy(buf, s + bs, have < wv ? have : wv);
		uint32_t rc = 0xff, off = 0;
		for (uint32_t c = 0; c < wv; c++) {
			int t = -1;
			if (buf[c] == 1 && buf[c + 1] == 1) {
				if (c + 38 <= w && !memcmp(buf + c, Y, 38))
					t = 3;
				else if (c + 22 <= w && !memcmp(buf + c, N, 22))
					t = 0;
			} else if (buf[c] == 0 && c + 22 <= w && !memcmp(buf + c, P, 22))
				t = 1;
			if (t < 0)
				continue;
			if (c < 21 && !skewed_gate(buf, c))
				continue;
			rc = (uint32_t)t;
			off = c;
			break;
		}
		uint32_t flags = (rc == 0xff && w > view) ? 4u : 0u;
		if (rc == 0xff) {	/* TG_CLS_NOVIEW: nothing in the rest of the view (up to the stream's end) either */
			uint8_t full[1280];
			const uint32_t vis = (uint32_t)(L - bs < view ? L - bs : view);
			memset(full, 0, sizeof(full));
			memcpy(full, s + bs, have);
			int anyv = 0;
			for (uint32_t c = 21; c < vis && !anyv; c++) {
				/* the first sequence that ends inside the view: 1 + its type into bits 4..6 of the flags, its offset into the word */
				const int t = (c + 38 <= vis && !memcmp(full + c, Y, 38)) ? 3 :
					      (c + 22 <= vis && !memcmp(full + c, N, 22)) ? 0 :
					      (c + 22 <= vis && !memcmp(full + c, P, 22)) ? 1 : -1;
				if (t >= 0) {
					anyv = 1;
					off = c;
					flags |= (uint32_t)(t + 1) << 4;
				}
			}
			if (!anyv)
				flags |= 8u;
		}
		cls[i] = rc | (off << 8) | (flags << 24);
		ysum[i] = 0xffff;
	}
	if (L < 38)
		return;
	for (uint64_t p = anchor; p + 38 <= L; p++) {
		if (s[p] != 1 || s[p + 1] != 1 || s[p + 2] != 0 || memcmp(s + p, Y, 38))
			continue;
		const uint64_t g = (p - anchor) / 510, o = (p - anchor) % 510;
		if (g >= n)
			continue;
		uint64_t vis = L - (anchor + 510 * g);
		if (vis > 640)
			vis = 640;	/* (any bound past 510 + 38 will do: the summary is about the slot's own positions) */
		if (o + 38 > vis)
			continue;
		if (ysum[g] == 0xffff)
			ysum[g] = (uint16_t)o;
		else
			ysum[g] |= 0x8000;
	}
}
