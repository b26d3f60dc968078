/*
 * tg_rm.c -- minimum-distance decoder for the shortened (30,14) Reed-Muller code of the AACH.
 *
 * The reference only encodes (lower_mac/tetra_rm3014.c:28-86: systematic, codeword bit 29 = first data bit,
 * 16 parity bits in bits 15..0) and its receiver keeps the first 14 received bits
 * (lower_mac/tetra_lower_mac.c:268-274; tetra_rm3014.c:88-96 is a stub).  This is the optional real decoder
 * of SURVEY.md 8(f) item 4, off by default: syndrome decoding with a table of coset leaders.
 *   syndrome(rx) = parity(rx >> 16) ^ (rx & 0xffff), 16 bits;
 *   leader[s]    = the error pattern of minimum weight with that syndrome, ties -> the numerically smallest
 *                  30-bit pattern (bit 29 = first transmitted bit), so the decoder is a deterministic
 *                  maximum-likelihood decoder for hard decisions;
 *   corrected    = rx ^ leader[syndrome(rx)].
 * d_min = 8: up to 3 bit errors are always corrected.  The table (65536 x 4 bytes) is built once on first use
 * by enumerating error patterns in order of weight, then of value.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "tetra_gpu.h"
#include "tg_internal.h"

static const uint16_t rm_parity[14] = {	/* parity half of the generator rows, tetra_rm3014.c:28-43 */
	0x9b60, 0x2de0, 0xfc20, 0xe03c, 0x983a, 0x5436, 0x2c2e,
	0xffdf, 0x8339, 0x42b5, 0x21ad, 0x1273, 0x096b, 0x04e7,
};

static uint32_t *rm_leader;
static pthread_once_t rm_once = PTHREAD_ONCE_INIT;

const uint16_t *tgi_rm_parity(void)
{
	return rm_parity;
}

static void rm_build(void)
{
	uint32_t *t = malloc(65536 * sizeof(*t));
	if (!t)
		return;
	memset(t, 0xff, 65536 * sizeof(*t));
	uint16_t colsyn[30];	/* syndrome of the unit error in codeword bit b */
	for (int b = 0; b < 30; b++)
		colsyn[b] = b >= 16 ? rm_parity[29 - b] : (uint16_t)(1u << b);
	uint32_t left = 65536;
	t[0] = 0;
	left--;
	for (int w = 1; w <= 30 && left; w++) {
		/* all 30-bit patterns of weight w in ascending numeric order (Gosper) */
		uint32_t v = (1u << w) - 1;
		while (v < (1u << 30)) {
			uint16_t s = 0;
			for (uint32_t x = v; x; x &= x - 1)
				s ^= colsyn[__builtin_ctz(x)];
			if (t[s] == 0xffffffffu) {
				t[s] = v;
				if (!--left)
					break;
			}
			const uint32_t c = v & -v, r = v + c;
			v = (((r ^ v) >> 2) / c) | r;
		}
	}
	rm_leader = t;
}

const uint32_t *tgi_rm_leader_table(void)
{
	pthread_once(&rm_once, rm_build);
	return rm_leader;
}

int tgpu_rm3014_decode(uint32_t rx30, uint16_t *data14, unsigned int *nerr)
{
	const uint32_t *t = tgi_rm_leader_table();
	if (!t)
		return TGPU_ENOMEM;
	rx30 &= 0x3fffffffu;
	uint16_t s = (uint16_t)rx30;
	for (int i = 0; i < 14; i++)
		if ((rx30 >> (29 - i)) & 1)
			s ^= rm_parity[i];
	const uint32_t e = t[s];
	if (data14)
		*data14 = (uint16_t)((rx30 ^ e) >> 16);
	if (nerr)
		*nerr = (unsigned int)__builtin_popcount(e);
	return TGPU_OK;
}
