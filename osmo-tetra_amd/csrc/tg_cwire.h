/*
 * tg_cwire.h -- the compact transport form of a decoded batch ("cwire"), shared by the host code and the kernels.
 *
 * The 40-byte wire record (tg_layout.h) exists for EVERY grid slot of a batch; what leaves the GPU for the collecting
 * rank needs only the DELIVERED bursts (the reference's upper MAC sees nothing else: tetra_burst_rx_cb() is called for
 * delivered bursts only, phy/tetra_burst_sync.c:113-150, and consumes CRC-good blocks, tetra_upper_mac.c:480-488), and
 * of those only what is not implied: a block whose CRC is good carries the CRC word 0x1d0f by definition.
 *
 * One buffer per batch, little endian, everything at dword-aligned offsets:
 *
 *   header, 8 dwords   [0] TG_CW_MAGIC  [1] nchan  [2] ngrid  [3] total bytes of the buffer  [4] delivered bursts
 *                      [5] byte offset of the delivered bitmap  [6] of the block table  [7] of the records
 *   channel table      nchan x { gbase, ncls, first ordinal, first record's byte offset (relative to the records) }:
 *                      channel c owns grid slots gbase .. gbase + ncls - 1 (gbase a multiple of 32); its delivered
 *                      bursts are the ordinals first .. first(c + 1) - 1 (the last channel's end at [4])
 *   delivered bitmap   one bit per grid slot, (ngrid + 31) / 32 dwords
 *   block table        per 1024 grid slots { byte offset of the block's first record, ordinal of its first delivered
 *                      burst }, plus one closing entry { bytes of all records, delivered bursts }: the way in for a
 *                      reader that does not start at the front
 *   records            delivered bursts in grid order.  The records of one bitmap word (32 grid slots) are contiguous
 *                      and start on a dword boundary (0..3 zero bytes behind the word's last record).
 *
 * A record starts with a 16-bit header: bits 0..1 = burst type (enum tetra_train_seq: 0 NORM_1, 1 NORM_2, 3 SYNC;
 * 2 = escape), bits 2..15 = the 14 BBK type-1 bits.  Then the type-1 bits, LSB first, no gaps:
 *   NORM_1  268 SCH/F bits                      -> 36 bytes (4 spare bits, zero)
 *   NORM_2  124 + 124 bits (BLK1, BLK2)         -> 33 bytes
 *   SYNC    60 + 124 bits  (SB1, SB2)           -> 25 bytes
 * This short form stands for a burst whose flags are 0 and whose blocks all passed their CRC.  Anything else -- a CRC
 * failure (the CRC word travels then), the non-binary flag, a slot the batch delivered without decoding it -- is an
 * escape record: one byte 0x02 followed by the slot's 40-byte wire record as it is, 41 bytes.
 * SB+NDB mix (SB, 4 NORM_1, 3 NORM_2 per frame): 33.5 bytes per delivered burst against 40 per grid slot.
 *
 * tg_cw_encode() / tg_cw_decode() are exact inverses on ANY 40-byte input: a wire record that is not in the canonical
 * form the trellis kernels write (unused bits zero) takes the escape form.
 */
#ifndef TG_CWIRE_H
#define TG_CWIRE_H

#include <stdint.h>
#include <string.h>
#include "tg_layout.h"

#define TG_CW_MAGIC      0x57434754u	/* "TGCW" */
#define TG_CW_HDR_WORDS  8
#define TG_CW_BLOCK      1024u		/* grid slots per block-table entry */
#define TG_CW_ESC        2u		/* record kind "escape" (never a decoded burst type: NORM_3) */
#define TG_CW_ESC_BYTES  41u
#define TG_CW_MAX_WORDS  11		/* dwords an encoded record occupies at most */
#define TG_CW_OK2        ((uint32_t)TG_CRC_OK | ((uint32_t)TG_CRC_OK << 16))

#if defined(__HIPCC__)
#define TG_CW_FN __host__ __device__ static inline
#else
#define TG_CW_FN static inline
#endif

struct tg_cw_layout {
	uint32_t o_chan, o_bits, o_blk, o_rec;	/* byte offsets */
	uint32_t nwords, nblk;
};

TG_CW_FN void tg_cw_offsets(uint32_t nchan, uint32_t ngrid, struct tg_cw_layout *L)
{
	L->nwords = (ngrid + 31) / 32;
	L->nblk = (ngrid + TG_CW_BLOCK - 1) / TG_CW_BLOCK;
	L->o_chan = TG_CW_HDR_WORDS * 4;
	L->o_bits = L->o_chan + nchan * 16;
	L->o_blk = L->o_bits + L->nwords * 4;
	L->o_rec = (L->o_blk + (L->nblk + 1) * 8 + 15) & ~15u;
}

/* bytes a batch of ngrid slots can need at most: every slot delivered as an escape record, a pad per bitmap word */
TG_CW_FN uint64_t tg_cw_bound(uint32_t nchan, uint32_t ngrid)
{
	struct tg_cw_layout L;
	tg_cw_offsets(nchan, ngrid, &L);
	return (uint64_t)L.o_rec + (uint64_t)ngrid * TG_CW_ESC_BYTES + (uint64_t)L.nwords * 3 + 16;
}

/* size of the compact record of the 40-byte wire record w[] */
TG_CW_FN uint32_t tg_cw_size(const uint32_t w[TG_WIRE_WORDS])
{
	const uint32_t type = w[0] & 0xff;
	if ((w[0] & 0xc000ff00u) != 0)		/* a flag, or something above the 14 BBK bits */
		return TG_CW_ESC_BYTES;
	if (type == TG_BURST_NORM_1)
		return (w[TG_WIRE_W_CRC] >> TG_WIRE_SCHF_CRC_SHIFT) == TG_CRC_OK ? 36u : TG_CW_ESC_BYTES;
	if (type == TG_BURST_NORM_2)
		return (w[TG_WIRE_W_CRC] == TG_CW_OK2 && ((w[4] | w[8]) >> 28) == 0) ? 33u : TG_CW_ESC_BYTES;
	if (type == TG_BURST_SYNC)
		return (w[TG_WIRE_W_CRC] == TG_CW_OK2 && ((w[2] | w[8]) >> 28) == 0 && (w[3] | w[4]) == 0) ? 25u : TG_CW_ESC_BYTES;
	return TG_CW_ESC_BYTES;
}

/* compact record of w[] into c[] (dwords, the bytes behind the record are zero); returns its size in bytes */
TG_CW_FN uint32_t tg_cw_encode(const uint32_t w[TG_WIRE_WORDS], uint32_t c[TG_CW_MAX_WORDS])
{
	const uint32_t size = tg_cw_size(w);
	const uint32_t hdr = (w[0] & 3u) | ((w[0] >> 16) << 2);
	for (int k = 0; k < TG_CW_MAX_WORDS; k++)
		c[k] = 0;
	if (size == 36u) {
		c[0] = hdr | (w[1] << 16);
		for (int k = 1; k < 8; k++)
			c[k] = (w[k] >> 16) | (w[k + 1] << 16);
		c[8] = (w[8] >> 16) | ((w[9] & 0xfffu) << 16);
	} else if (size == 33u) {
		c[0] = hdr | (w[1] << 16);
		c[1] = (w[1] >> 16) | (w[2] << 16);
		c[2] = (w[2] >> 16) | (w[3] << 16);
		c[3] = (w[3] >> 16) | (w[4] << 16);
		c[4] = (w[4] >> 16) | (w[5] << 12);
		c[5] = (w[5] >> 20) | (w[6] << 12);
		c[6] = (w[6] >> 20) | (w[7] << 12);
		c[7] = (w[7] >> 20) | (w[8] << 12);
		c[8] = w[8] >> 20;
	} else if (size == 25u) {
		c[0] = hdr | (w[1] << 16);
		c[1] = (w[1] >> 16) | (w[2] << 16);
		c[2] = (w[2] >> 16) | (w[5] << 12);
		c[3] = (w[5] >> 20) | (w[6] << 12);
		c[4] = (w[6] >> 20) | (w[7] << 12);
		c[5] = (w[7] >> 20) | (w[8] << 12);
		c[6] = w[8] >> 20;
	} else {
		c[0] = TG_CW_ESC | (w[0] << 8);
		for (int k = 1; k < TG_WIRE_WORDS; k++)
			c[k] = (w[k - 1] >> 24) | (w[k] << 8);
		c[TG_WIRE_WORDS] = w[TG_WIRE_WORDS - 1] >> 24;
	}
	return size;
}

/* the record at p (avail readable bytes) back into its 40-byte wire record; returns the record's size, 0 if it does
 * not fit into avail */
TG_CW_FN uint32_t tg_cw_decode(const uint8_t *p, size_t avail, uint32_t w[TG_WIRE_WORDS])
{
	if (avail < 1)
		return 0;
	const uint32_t kind = p[0] & 3u;
	const uint32_t size = kind == TG_BURST_NORM_1 ? 36u : kind == TG_BURST_NORM_2 ? 33u : kind == TG_BURST_SYNC ? 25u : TG_CW_ESC_BYTES;
	if (avail < size)
		return 0;
	uint32_t c[TG_CW_MAX_WORDS];
	for (int k = 0; k < TG_CW_MAX_WORDS; k++)
		c[k] = 0;
	for (uint32_t i = 0; i < size; i++)
		c[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
	if (kind == TG_CW_ESC) {
		for (int k = 0; k < TG_WIRE_WORDS; k++)
			w[k] = (c[k] >> 8) | (c[k + 1] << 24);
		return size;
	}
	w[0] = kind | ((c[0] & 0xfffcu) << 14);
	w[1] = (c[0] >> 16) | (c[1] << 16);
	if (kind == TG_BURST_NORM_1) {
		for (int k = 2; k <= 8; k++)
			w[k] = (c[k - 1] >> 16) | (c[k] << 16);
		w[9] = ((c[8] >> 16) & 0xfffu) | ((uint32_t)TG_CRC_OK << TG_WIRE_SCHF_CRC_SHIFT);
		return size;
	}
	if (kind == TG_BURST_NORM_2) {
		w[2] = (c[1] >> 16) | (c[2] << 16);
		w[3] = (c[2] >> 16) | (c[3] << 16);
		w[4] = ((c[3] >> 16) | (c[4] << 16)) & 0x0fffffffu;
		w[5] = (c[4] >> 12) | (c[5] << 20);
		w[6] = (c[5] >> 12) | (c[6] << 20);
		w[7] = (c[6] >> 12) | (c[7] << 20);
		w[8] = ((c[7] >> 12) | (c[8] << 20)) & 0x0fffffffu;
	} else {
		w[2] = ((c[1] >> 16) | (c[2] << 16)) & 0x0fffffffu;
		w[3] = 0;
		w[4] = 0;
		w[5] = (c[2] >> 12) | (c[3] << 20);
		w[6] = (c[3] >> 12) | (c[4] << 20);
		w[7] = (c[4] >> 12) | (c[5] << 20);
		w[8] = ((c[5] >> 12) | (c[6] << 20)) & 0x0fffffffu;
	}
	w[9] = TG_CW_OK2;
	return size;
}

#endif
