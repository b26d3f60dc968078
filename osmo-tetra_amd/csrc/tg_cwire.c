/*
 * tg_cwire.c -- host side of the compact transport form (tg_cwire.h): the packer in plain C (what k_cw_* do on the
 * device, and the definition the kernels are tested against), the reader the collecting rank runs on a gathered
 * buffer, and the launch of the device form.
 *
 * Reference: the consumer of a gathered batch is the upper MAC of the collecting process, which is handed delivered
 * bursts' blocks only (phy/tetra_burst_sync.c:113-150 -> tetra_burst_rx_cb()) and acts on the CRC-good ones
 * (tetra_upper_mac.c:480-488); one receiver process per channel in the reference (src/receiver1:1-10).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdlib.h>
#include <string.h>

#include "tetra_gpu.h"
#include "tg_layout.h"
#include "tg_internal.h"
#include "tg_cwire.h"

uint64_t tgpu_cwire_bound(uint32_t ngrid, uint32_t nchan)
{
	return tg_cw_bound(nchan, ngrid);
}

static int chans_ok(uint32_t ngrid, uint32_t nchan, const uint32_t *gbase, const uint32_t *ncls)
{
	if (!nchan || nchan > 64 || !gbase || !ncls)
		return 0;
	uint64_t end = 0;
	for (uint32_t c = 0; c < nchan; c++) {
		if ((gbase[c] & 31u) || gbase[c] < end)
			return 0;
		end = (uint64_t)gbase[c] + ncls[c];
		if (end > ngrid)
			return 0;
	}
	return 1;
}

int64_t tgpu_cwire_pack(const uint8_t *wire, const uint32_t *grid_bits, uint32_t ngrid, uint32_t nchan, const uint32_t *gbase,
			const uint32_t *ncls, uint8_t *out, size_t cap)
{
	if (!wire || !grid_bits || !out || ngrid > 0x03ffffffu || !chans_ok(ngrid, nchan, gbase, ncls))
		return TGPU_EINVAL;
	struct tg_cw_layout L;
	tg_cw_offsets(nchan, ngrid, &L);
	if (cap < L.o_rec)
		return TGPU_ECAPACITY;
	memset(out, 0, L.o_rec);
	uint32_t *hdr = (uint32_t *)out, *chan = (uint32_t *)(out + L.o_chan), *bits = (uint32_t *)(out + L.o_bits);
	uint32_t *blk = (uint32_t *)(out + L.o_blk);
	size_t o = 0;		/* relative to the records */
	uint32_t ord = 0, c = 0;
	for (uint32_t wd = 0; wd < L.nwords; wd++) {
		uint32_t z = grid_bits[wd];
		if (32 * wd + 32 > ngrid && (ngrid & 31))
			z &= (1u << (ngrid & 31)) - 1u;
		bits[wd] = z;
		if (!(wd & (TG_CW_BLOCK / 32 - 1))) {
			blk[2 * (wd / (TG_CW_BLOCK / 32))] = (uint32_t)o;
			blk[2 * (wd / (TG_CW_BLOCK / 32)) + 1] = ord;
		}
		while (c < nchan && gbase[c] == 32 * wd) {	/* (empty channels share a start) */
			chan[4 * c] = gbase[c];
			chan[4 * c + 1] = ncls[c];
			chan[4 * c + 2] = ord;
			chan[4 * c + 3] = (uint32_t)o;
			c++;
		}
		for (; z; z &= z - 1) {
			const uint32_t g = 32 * wd + (uint32_t)__builtin_ctz(z);
			uint32_t w[TG_WIRE_WORDS], e[TG_CW_MAX_WORDS];
			memcpy(w, wire + (size_t)g * TG_WIRE_BYTES, sizeof(w));
			const uint32_t size = tg_cw_encode(w, e);
			if ((size_t)L.o_rec + o + size + 4 > cap)
				return TGPU_ECAPACITY;
			memcpy(out + L.o_rec + o, e, size);
			o += size;
			ord++;
		}
		while (o & 3)
			out[L.o_rec + o++] = 0;
	}
	for (; c < nchan; c++) {	/* channels that start behind the last word: gbase == ngrid, nothing in them */
		chan[4 * c] = gbase[c];
		chan[4 * c + 1] = ncls[c];
		chan[4 * c + 2] = ord;
		chan[4 * c + 3] = (uint32_t)o;
	}
	blk[2 * L.nblk] = (uint32_t)o;
	blk[2 * L.nblk + 1] = ord;
	hdr[0] = TG_CW_MAGIC;
	hdr[1] = nchan;
	hdr[2] = ngrid;
	hdr[3] = (uint32_t)(L.o_rec + o);
	hdr[4] = ord;
	hdr[5] = L.o_bits;
	hdr[6] = L.o_blk;
	hdr[7] = L.o_rec;
	return (int64_t)(L.o_rec + o);
}

int tgpu_cwire_info(const uint8_t *cw, size_t nbytes, struct tgpu_cwire_info *out)
{
	if (!cw || !out || nbytes < TG_CW_HDR_WORDS * 4)
		return TGPU_EINVAL;
	uint32_t hdr[TG_CW_HDR_WORDS];
	memcpy(hdr, cw, sizeof(hdr));
	struct tg_cw_layout L;
	if (hdr[0] != TG_CW_MAGIC || !hdr[1] || hdr[1] > 64 || hdr[2] > 0x03ffffffu)	/* (2^26 grid slots: sizes and offsets stay inside 32 bits) */
		return TGPU_EINVAL;
	tg_cw_offsets(hdr[1], hdr[2], &L);
	if (hdr[5] != L.o_bits || hdr[6] != L.o_blk || hdr[7] != L.o_rec || hdr[3] < L.o_rec || hdr[3] > nbytes)
		return TGPU_EINVAL;
	/* this reader checks a buffer that came over a link: nothing in it is trusted.  Records are padded to dwords per
	 * bitmap word, so the record area is a whole number of dwords; no delivered bit at or above ngrid; the channel table
	 * names slot ranges inside the grid, in order, with ordinals and record offsets that do not run backwards */
	if ((hdr[3] - L.o_rec) & 3u)
		return TGPU_EINVAL;
	if ((hdr[2] & 31u) && L.nwords) {
		uint32_t last;
		memcpy(&last, cw + L.o_bits + 4 * (size_t)(L.nwords - 1), 4);
		if (last >> (hdr[2] & 31u))
			return TGPU_EINVAL;
	}
	uint64_t end = 0;
	uint32_t ord = 0, off = 0;
	for (uint32_t c = 0; c < hdr[1]; c++) {
		uint32_t e[4];
		memcpy(e, cw + TG_CW_HDR_WORDS * 4 + 16 * (size_t)c, sizeof(e));
		if ((e[0] & 31u) || e[0] < end || (uint64_t)e[0] + e[1] > hdr[2] || e[2] < ord || e[2] > hdr[4] || e[3] < off ||
		    e[3] > hdr[3] - L.o_rec)
			return TGPU_EINVAL;
		end = (uint64_t)e[0] + e[1];
		ord = e[2];
		off = e[3];
	}
	out->nchan = hdr[1];
	out->ngrid = hdr[2];
	out->total_bytes = hdr[3];
	out->ndelivered = hdr[4];
	return TGPU_OK;
}

int tgpu_cwire_chan(const uint8_t *cw, size_t nbytes, uint32_t c, uint32_t *gbase, uint32_t *ncls, uint32_t *ndelivered)
{
	struct tgpu_cwire_info in;
	int rc = tgpu_cwire_info(cw, nbytes, &in);
	if (rc)
		return rc;
	if (c >= in.nchan)
		return TGPU_EINVAL;
	uint32_t e[4], next = in.ndelivered;
	memcpy(e, cw + TG_CW_HDR_WORDS * 4 + 16 * (size_t)c, sizeof(e));
	if (c + 1 < in.nchan)
		memcpy(&next, cw + TG_CW_HDR_WORDS * 4 + 16 * (size_t)(c + 1) + 8, 4);
	if (gbase)
		*gbase = e[0];
	if (ncls)
		*ncls = e[1];
	if (ndelivered)
		*ndelivered = next - e[2];
	return TGPU_OK;
}

/* every delivered burst in grid order, as its 40-byte wire record (rebuilt in a local the callback may read during the
 * call).  Returns the number handed over, or a negative TGPU_E* code when the buffer does not parse. */
int64_t tgpu_cwire_foreach(const uint8_t *cw, size_t nbytes, tgpu_wire_cb cb, void *priv)
{
	struct tgpu_cwire_info in;
	int rc = tgpu_cwire_info(cw, nbytes, &in);
	if (rc)
		return rc;
	struct tg_cw_layout L;
	tg_cw_offsets(in.nchan, in.ngrid, &L);
	const uint8_t *rec = cw + L.o_rec;
	const size_t rbytes = (size_t)in.total_bytes - L.o_rec;
	size_t o = 0;
	int64_t n = 0;
	for (uint32_t wd = 0; wd < L.nwords; wd++) {
		uint32_t z;
		memcpy(&z, cw + L.o_bits + 4 * (size_t)wd, 4);
		for (; z; z &= z - 1) {
			uint32_t w[TG_WIRE_WORDS];
			if (o >= rbytes)	/* (more delivered bits than records) */
				return TGPU_EINVAL;
			const uint32_t size = tg_cw_decode(rec + o, rbytes - o, w);
			if (!size)
				return TGPU_EINVAL;
			if (cb)
				cb((const uint8_t *)w, 32 * wd + (uint32_t)__builtin_ctz(z), priv);
			o += size;
			n++;
		}
		o = (o + 3) & ~(size_t)3;
		if (o > rbytes)
			return TGPU_EINVAL;
	}
	if (o != rbytes || n != (int64_t)in.ndelivered)
		return TGPU_EINVAL;
	return n;
}

struct expand_ctx {
	uint8_t *wire;
};

static void expand_cb(const uint8_t *w, uint32_t g, void *priv)
{
	memcpy(((struct expand_ctx *)priv)->wire + (size_t)g * TG_WIRE_BYTES, w, TG_WIRE_BYTES);
}

int tgpu_cwire_expand(const uint8_t *cw, size_t nbytes, uint8_t *wire, uint32_t *grid_bits)
{
	struct tgpu_cwire_info in;
	int rc = tgpu_cwire_info(cw, nbytes, &in);
	if (rc)
		return rc;
	if (!wire)
		return TGPU_EINVAL;
	struct tg_cw_layout L;
	tg_cw_offsets(in.nchan, in.ngrid, &L);
	memset(wire, 0xff, (size_t)in.ngrid * TG_WIRE_BYTES);
	if (grid_bits)
		memcpy(grid_bits, cw + L.o_bits, (size_t)L.nwords * 4);
	struct expand_ctx cx = { wire };
	const int64_t n = tgpu_cwire_foreach(cw, nbytes, expand_cb, &cx);
	return n < 0 ? (int)n : TGPU_OK;
}

int tgpu_wire_compact(struct tgpu_engine *eng, const uint8_t *d_wire, const uint32_t *d_grid_bits, uint32_t ngrid, uint32_t nchan,
		      const uint32_t *gbase, const uint32_t *ncls, uint8_t *d_cwire, size_t cap, uint32_t *d_total, void *hip_stream)
{
	if (!eng || !d_wire || !d_grid_bits || !d_cwire || !ngrid || ngrid > 0x03ffffffu || !chans_ok(ngrid, nchan, gbase, ncls))
		return TGPU_EINVAL;
	if (((uintptr_t)d_cwire & 15) || ((uintptr_t)d_wire & 7))
		return TGPU_EINVAL;
	int rc = tgpi_engine_bind(eng);
	if (rc)
		return rc;
	struct tg_cw_layout L;
	tg_cw_offsets(nchan, ngrid, &L);
	if (cap < (size_t)L.o_rec + 16 || cap > 0xfffffff0u)
		return TGPU_ECAPACITY;
	struct tg_cw_chans ch;
	memset(&ch, 0, sizeof(ch));
	ch.n = nchan;
	for (uint32_t c = 0; c < nchan; c++) {
		ch.gbase[c] = gbase[c];
		ch.ncls[c] = ncls[c];
	}
	return tgk_cwire(d_wire, d_grid_bits, ngrid, &ch, d_cwire, (uint32_t)cap, d_total, hip_stream);
}
