/*
 * tg_cwire.hip -- the compact transport form of a decoded batch on the device (format and per-record arithmetic:
 * tg_cwire.h; host packer / reader: tg_cwire.c).
 *
 * Input: the batch's 40-byte wire records (one per grid slot, as the trellis kernels leave them) and its delivered
 * bitmap.  Output: ONE buffer -- header, channel table, bitmap, block table, the delivered bursts' records back to back
 * (25 / 33 / 36 bytes, 41 for the exceptions) -- i.e. what a rank hands to the gather.  HBM-bound byte shuffling:
 *
 *   k_cw_sizes   a thread per grid slot, a workgroup per 1024 slots: record sizes of the delivered slots, the bytes of
 *                every 32-slot word rounded up to dwords, the block's bytes and bursts
 *   k_cw_scan    one workgroup: exclusive scan of the block totals in place (they ARE the block table), header, totals
 *   k_cw_emit    same shape as the first: sizes again, offsets inside the block by a scan over its 32 words, every lane
 *                encodes its record (funnel shifts) and ORs it into the block's staging area in LDS at its byte offset
 *                (ds_or_b32 on a cleared area: records straddle dwords), then the workgroup copies the area out as
 *                whole dwords, coalesced.  Also the bitmap copy and the channel table.
 *
 * 40 MB in (twice: the sizes must be known before anything can be placed) + 32 MB out per 1 M grid slots.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tg_layout.h"
#include "tg_internal.h"
#include "tg_cwire.h"

#define CW_THREADS TG_CW_BLOCK
#define CW_STAGE_WORDS ((TG_CW_BLOCK * TG_CW_ESC_BYTES + 32 * 3 + 3) / 4 + 4)

/* the slot's wire record (five 8-byte loads: records are 40 bytes apart) */
__device__ __forceinline__ void cw_load(const uint8_t *__restrict__ wire, uint32_t g, uint32_t (&w)[TG_WIRE_WORDS])
{
	const uint2 *p = (const uint2 *)(wire + (size_t)g * TG_WIRE_BYTES);
#pragma unroll
	for (int k = 0; k < TG_WIRE_WORDS / 2; k++) {
		const uint2 v = p[k];
		w[2 * k] = v.x;
		w[2 * k + 1] = v.y;
	}
}

/* inclusive prefix sum over the 32 lanes of a half wave */
__device__ __forceinline__ uint32_t cw_half_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
	for (int d = 1; d < 32; d <<= 1) {
		const uint32_t o = __shfl_up(v, d, 32);
		if ((lane & 31) >= (uint32_t)d)
			v += o;
	}
	return v;
}

/* what both passes agree on: is grid slot g delivered, and how long is its record */
__device__ __forceinline__ bool cw_slot(const uint8_t *__restrict__ wire, const uint32_t *__restrict__ bits, uint32_t ngrid, uint32_t g,
					uint32_t (&w)[TG_WIRE_WORDS], uint32_t &size)
{
	size = 0;
	const bool del = g < ngrid && ((bits[g >> 5] >> (g & 31)) & 1);
	if (del) {
		cw_load(wire, g, w);
		size = tg_cw_size(w);
	}
	return del;
}

__global__ __launch_bounds__(CW_THREADS)
void k_cw_sizes(const uint8_t *__restrict__ wire, const uint32_t *__restrict__ bits, uint32_t ngrid, uint32_t *__restrict__ blk)
{
	__shared__ uint32_t s_bytes[32], s_cnt[32];
	const uint32_t g = blockIdx.x * CW_THREADS + threadIdx.x, lane = threadIdx.x & 63, hw = threadIdx.x >> 5;
	uint32_t w[TG_WIRE_WORDS], size;
	const bool del = cw_slot(wire, bits, ngrid, g, w, size);
	const uint32_t incl = cw_half_scan(size, lane);
	const unsigned long long bal = __ballot(del);
	if ((lane & 31) == 31) {
		s_bytes[hw] = (incl + 3) & ~3u;
		s_cnt[hw] = (uint32_t)__builtin_popcount((uint32_t)(bal >> (lane & 32)));
	}
	__syncthreads();
	if (threadIdx.x < 32) {
		uint32_t b = s_bytes[threadIdx.x], c = s_cnt[threadIdx.x];
#pragma unroll
		for (int d = 16; d; d >>= 1) {
			b += __shfl_xor(b, d, 32);
			c += __shfl_xor(c, d, 32);
		}
		if (threadIdx.x == 0) {
			blk[2 * blockIdx.x] = b;
			blk[2 * blockIdx.x + 1] = c;
		}
	}
}

__global__ __launch_bounds__(1024)
void k_cw_scan(uint32_t *__restrict__ out, tg_cw_chans ch, uint32_t ngrid, uint32_t cap, uint32_t *__restrict__ total_out)
{
	/* exclusive scan of (bytes, bursts) per block, in place, 1024 blocks per pass */
	__shared__ uint32_t s_b[16], s_c[16];
	tg_cw_layout L;
	tg_cw_offsets(ch.n, ngrid, &L);
	uint32_t *blk = out + L.o_blk / 4;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t cb = 0, cc = 0;
	for (uint32_t base = 0; base < L.nblk; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t vb = i < L.nblk ? blk[2 * i] : 0u, vc = i < L.nblk ? blk[2 * i + 1] : 0u;
		uint32_t ib = vb, ic = vc;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t ob = __shfl_up(ib, d), oc = __shfl_up(ic, d);
			if (lane >= (uint32_t)d) {
				ib += ob;
				ic += oc;
			}
		}
		__syncthreads();
		if (lane == 63) {
			s_b[wv] = ib;
			s_c[wv] = ic;
		}
		__syncthreads();
		uint32_t pb = cb, pc = cc, tb = cb, tc = cc;
		for (uint32_t q = 0; q < 16; q++) {
			if (q < wv) {
				pb += s_b[q];
				pc += s_c[q];
			}
			tb += s_b[q];
			tc += s_c[q];
		}
		if (i < L.nblk) {
			blk[2 * i] = pb + ib - vb;
			blk[2 * i + 1] = pc + ic - vc;
		}
		cb = tb;
		cc = tc;
	}
	if (threadIdx.x == 0) {
		const uint64_t total = (uint64_t)L.o_rec + cb;
		const bool fits = total <= cap;
		blk[2 * L.nblk] = cb;
		blk[2 * L.nblk + 1] = cc;
		out[0] = fits ? TG_CW_MAGIC : 0u;	/* a buffer that is too small carries no records: k_cw_emit checks this word */
		out[1] = ch.n;
		out[2] = ngrid;
		out[3] = (uint32_t)total;
		out[4] = cc;
		out[5] = L.o_bits;
		out[6] = L.o_blk;
		out[7] = L.o_rec;
		for (uint32_t i = L.o_blk / 4 + 2 * (L.nblk + 1); i < L.o_rec / 4; i++)
			out[i] = 0;			/* (the pad in front of the records) */
		if (total_out) {
			total_out[0] = (uint32_t)total;	/* the bytes the batch needs, whether they fit or not */
			total_out[1] = fits ? cc : 0xffffffffu;
		}
	}
	/* channels that start behind the last block (empty ones at the grid's end) */
	if (threadIdx.x < ch.n && ch.gbase[threadIdx.x] >= L.nblk * TG_CW_BLOCK) {
		uint32_t *e = out + L.o_chan / 4 + 4 * threadIdx.x;
		e[0] = ch.gbase[threadIdx.x];
		e[1] = ch.ncls[threadIdx.x];
		e[2] = cc;
		e[3] = cb;
	}
}

__global__ __launch_bounds__(CW_THREADS)
void k_cw_emit(const uint8_t *__restrict__ wire, const uint32_t *__restrict__ bits, uint32_t ngrid, tg_cw_chans ch,
	       uint32_t *__restrict__ out)
{
	__shared__ uint32_t s_stage[CW_STAGE_WORDS];
	__shared__ uint32_t s_bytes[32], s_cnt[32];
	if (out[0] != TG_CW_MAGIC)
		return;
	tg_cw_layout L;
	tg_cw_offsets(ch.n, ngrid, &L);
	const uint32_t g = blockIdx.x * CW_THREADS + threadIdx.x, lane = threadIdx.x & 63, hw = threadIdx.x >> 5;
	uint32_t w[TG_WIRE_WORDS], size;
	const bool del = cw_slot(wire, bits, ngrid, g, w, size);
	const uint32_t incl = cw_half_scan(size, lane);
	const uint32_t half = (uint32_t)(__ballot(del) >> (lane & 32));
	if ((lane & 31) == 31) {
		s_bytes[hw] = (incl + 3) & ~3u;
		s_cnt[hw] = (uint32_t)__builtin_popcount(half);
	}
	for (uint32_t i = threadIdx.x; i < CW_STAGE_WORDS; i += CW_THREADS)
		s_stage[i] = 0;
	__syncthreads();
	/* where this block's words start: every thread of the first half wave scans the 32 word totals */
	if (threadIdx.x < 32) {
		const uint32_t vb = s_bytes[threadIdx.x], vc = s_cnt[threadIdx.x];
		uint32_t ib = vb, ic = vc;
#pragma unroll
		for (int d = 1; d < 32; d <<= 1) {
			const uint32_t ob = __shfl_up(ib, d, 32), oc = __shfl_up(ic, d, 32);
			if (threadIdx.x >= (uint32_t)d) {
				ib += ob;
				ic += oc;
			}
		}
		s_bytes[threadIdx.x] = ib - vb;
		s_cnt[threadIdx.x] = ic - vc;
		if (threadIdx.x == 31)
			s_stage[CW_STAGE_WORDS - 1] = ib;	/* the block's bytes (the last staging word is never a record's) */
	}
	__syncthreads();
	const uint32_t *blk = out + L.o_blk / 4;
	const uint32_t blk_off = blk[2 * blockIdx.x], blk_ord = blk[2 * blockIdx.x + 1];
	const uint32_t blk_bytes = s_stage[CW_STAGE_WORDS - 1];
	const uint32_t woff = s_bytes[hw];
	/* bitmap word and the channels that start here */
	if ((lane & 31) == 0) {
		if (g < 32 * L.nwords)
			out[L.o_bits / 4 + (g >> 5)] = g + 32 <= ngrid ? half : half & ((1u << (ngrid & 31)) - 1u);
		for (uint32_t c = 0; c < ch.n; c++)
			if (ch.gbase[c] == g) {
				uint32_t *e = out + L.o_chan / 4 + 4 * c;
				e[0] = g;
				e[1] = ch.ncls[c];
				e[2] = blk_ord + s_cnt[hw];
				e[3] = blk_off + woff;
			}
	}
	if (del) {
		uint32_t c[TG_CW_MAX_WORDS];
		tg_cw_encode(w, c);
		const uint32_t o = woff + incl - size;		/* byte offset inside the block */
		const uint32_t sh = 8 * (o & 3), nd = ((o & 3) + size + 3) >> 2;
		uint32_t *dst = s_stage + (o >> 2);
		uint32_t prev = 0;
#pragma unroll
		for (int k = 0; k <= TG_CW_MAX_WORDS; k++) {
			const uint32_t cur = k < TG_CW_MAX_WORDS ? c[k] : 0u;
			const uint32_t v = (uint32_t)((((unsigned long long)cur << 32) | prev) >> (32 - sh));
			if ((uint32_t)k < nd && v)
				atomicOr(dst + k, v);
			prev = cur;
		}
	}
	__syncthreads();
	uint32_t *rec = out + (L.o_rec + blk_off) / 4;
	for (uint32_t i = threadIdx.x; i < blk_bytes / 4; i += CW_THREADS)
		rec[i] = s_stage[i];
}

extern "C" int tgk_cwire(const uint8_t *d_wire, const uint32_t *d_bits, uint32_t ngrid, const struct tg_cw_chans *ch, uint8_t *d_out,
			 uint32_t cap, uint32_t *d_total, void *stream)
{
	tg_cw_layout L;
	tg_cw_offsets(ch->n, ngrid, &L);
	hipStream_t s = (hipStream_t)stream;
	uint32_t *out = (uint32_t *)d_out;
	hipLaunchKernelGGL(k_cw_sizes, dim3(L.nblk), dim3(CW_THREADS), 0, s, d_wire, d_bits, ngrid, out + L.o_blk / 4);
	hipLaunchKernelGGL(k_cw_scan, dim3(1), dim3(1024), 0, s, out, *ch, ngrid, cap, d_total);
	hipLaunchKernelGGL(k_cw_emit, dim3(L.nblk), dim3(CW_THREADS), 0, s, d_wire, d_bits, ngrid, *ch, out);
	return (int)hipGetLastError();
}
