/*
 * tg_traffic.hip -- traffic blocks of a decoded batch on the device.
 *
 * Reference: lower_mac/tetra_lower_mac.c:194-241.  When the upper MAC has marked the burst as a traffic burst
 * (tms->cur_burst.is_traffic, from the ACCESS-ASSIGN PDU of the burst's own AACH), tp_sap_udata_ind()
 *   :194-195  notes "block 1 stolen" for the first half of a NORM_2 burst (which is then decoded as signalling),
 *   :198      does NOT decode an SCH/F block, nor a second block (blk_num == BLK_2: the BLK2 of a NORM_2 burst -- and, by
 *             the same test, the SB2 of a SYNC burst) unless the upper MAC said that block 2 is stolen too,
 *   :213-231  but turns the block's descrambled type-4 bits into the codec tools' input block: 690 int16 = six frames of
 *             a marker 0x6b21 + i and 114 soft bits (bit 1 -> -127, bit 0 -> +127), 432 bits in all, the rest 0, appended
 *             to traffic_<usage>_<tsn>.out.
 * k_traffic does that for every flagged burst of a batch the plan has decoded: the caller's byte per slot has the shape
 * tgpu_gsmtap_batch() takes (bit 0 = traffic burst, bit 1 = its second block was stolen), the outputs are the descrambled
 * type-4 bits (what struct tgpu_unitdata.type4 is on the callback path), the 690-word block and the bit count per slot.
 * The file I/O stays the caller's (the reference's fopen / fwrite per block is its control plane).
 *
 * Source of the bits: the plan's packed slots.  A packed slot (tg_layout.h) holds every received bit of the two code blocks
 * in code-word order, still scrambled, and the plan's mask table holds the scrambling sequence of the slot's code in the
 * same order -- word XOR word is the descrambled block, and the inverse of tg_codeword_src() (de-interleaver and 2/3
 * puncturing order composed) puts a bit back at its type-4 position.  80 + 72 bytes read per block instead of the
 * 432 stream bytes and a 432-step LFSR; no slot offsets needed, so the same kernel serves load-mode and stream-mode
 * batches.  Driven by the batch's own item lists (the slots this batch decoded, whatever walked the stream): one wave per
 * item, no LDS traffic besides the inverse table (code words travel by ds_bpermute), dword stores.
 *
 * For a 216-bit block the reference builds the block from a 432-entry local array of which it wrote 216: words made from
 * bits 216..431 are whatever its stack held.  Here they are bit 0 (+127), as tgpu_traffic_block() and the oracle have it.
 */
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <stdint.h>

#include "tg_layout.h"
#include "tg_internal.h"

#define TGT_WAVES 4

/* type-4 position -> (code word << 5 | bit) of the block's code words: [0..431] SCH/F, [432..647] a 216-bit block */
__constant__ uint16_t c_t4pos[432 + 216];

__global__ __launch_bounds__(64 * TGT_WAVES)
void k_traffic(const uint32_t *__restrict__ items432, uint32_t n432, const uint32_t *__restrict__ items216, uint32_t n216,
	       const uint32_t *__restrict__ counts, const uint8_t *__restrict__ traffic, const uint32_t *__restrict__ packed,
	       const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec,
	       uint8_t *__restrict__ wire, uint8_t *__restrict__ type4, int16_t *__restrict__ blocks, uint16_t *__restrict__ lens)
{
	__shared__ uint16_t s_pos[432 + 216];
	for (uint32_t i = threadIdx.x; i < 432 + 216; i += 64 * TGT_WAVES)
		s_pos[i] = c_t4pos[i];
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t id = blockIdx.x * TGT_WAVES + (threadIdx.x >> 6);	/* wave-uniform from here on */
	uint32_t slot;
	bool full;
	if (id < n432) {
		if (counts && id >= counts[2])		/* (device-walk batches: the lists' lengths live on the device) */
			return;
		slot = items432[id];
		full = true;
	} else {
		const uint32_t k = id - n432;
		if (k >= n216 || (counts && k >= counts[1]))
			return;
		const uint32_t it = items216[k];
		if (!(it & 1))				/* a first half (BLK1 / nothing of a SYNC burst is listed here): decoded, never dumped */
			return;
		slot = it >> 1;
		full = false;
	}
	const uint32_t t = traffic[slot];
	if (!(t & 1))
		return;
	const uint32_t btype = rec ? rec[(size_t)slot * TG_REC_BYTES + TG_REC_TYPE] : wire[(size_t)slot * TG_WIRE_BYTES];
	const bool dump = full || !(t & 2);		/* :198 */
	if (lane == 0) {
		/* :194-195 (is_traffic, NDB, BLK_1 -> blk1_stolen) and the mark of a block that went to the dump, not to the decoder */
		const uint32_t fl = (!full && btype == TG_BURST_NORM_2 ? TG_FLAG_BLK1_STOLEN : 0u) | (dump ? TG_FLAG_TRAFFIC : 0u);
		if (rec) {
			rec[(size_t)slot * TG_REC_BYTES + TG_REC_FLAGS] |= (uint8_t)fl;
			if (dump)
				rec[(size_t)slot * TG_REC_BYTES + TG_REC_CRC_OK + (full ? 0 : 1)] = 0;	/* nothing was indicated */
		}
		if (wire)
			wire[(size_t)slot * TG_WIRE_BYTES + 1] |= (uint8_t)fl;
		if (dump)
			lens[slot] = full ? 432 : 216;
	}
	if (!dump)
		return;

	const uint32_t n = full ? 432u : 216u, nw = full ? 18u : 9u, tb = full ? 0u : 432u;
	uint32_t cw = 0;
	if (lane < nw)
		cw = packed[(size_t)slot * TG_PACKED_WORDS + (full ? TG_PW_BLK1 : TG_PW_BLK2) + lane] ^
		     masks[(size_t)maskidx[slot] * TG_MASK_WORDS + (full ? TG_MW_432 : TG_MW_216) + lane];
	auto bit = [&](uint32_t j) -> uint32_t {	/* descrambled type-4 bit j (j < n; every lane of the wave takes part) */
		const uint32_t pos = s_pos[tb + j];
		return ((uint32_t)__shfl((int)cw, (int)(pos >> 5)) >> (pos & 31)) & 1u;
	};

	if (type4) {
		uint32_t *o = (uint32_t *)(type4 + (size_t)slot * 432);
#pragma unroll
		for (uint32_t r = 0; r < 2; r++) {
			const uint32_t q = lane + 64 * r, j = 4 * q;
			uint32_t v = 0;
#pragma unroll
			for (uint32_t b = 0; b < 4; b++) {
				const uint32_t x = bit(j + b < n ? j + b : n - 1);
				v |= (j + b < n ? x : 0u) << (8 * b);
			}
			if (q < 108)
				o[q] = v;		/* (a 216-bit block: bytes 216..431 of its row are 0) */
		}
	}
	if (blocks) {
		uint32_t *o = (uint32_t *)(blocks + (size_t)slot * 690);
		auto word = [&](uint32_t e) -> uint32_t {	/* element e of the 690-word block, as 16 bits */
			const uint32_t f = e / 115u, k = e - 115u * f;
			const uint32_t b = 114u * f + k - 1u;		/* (k == 0: unused) */
			const bool inblk = k != 0 && b < 432u && f < 4u;	/* frames 0..2 hold 114 bits, frame 3 holds 90, frames 4 and 5 none */
			const uint32_t x = bit(inblk && b < n ? b : 0u);
			if (k == 0)
				return 0x6b21u + f;
			if (!inblk)
				return 0u;
			return (b < n && x) ? (uint32_t)(uint16_t)(int16_t)-127 : 127u;
		};
#pragma unroll
		for (uint32_t r = 0; r < 6; r++) {
			const uint32_t q = lane + 64 * r, e = 2 * (q < 345 ? q : 344);
			const uint32_t lo = word(e), hi = word(e + 1);
			if (q < 345)
				o[q] = lo | (hi << 16);
		}
	}
}

/* the inverse of tg_codeword_src() for the two kinds, uploaded once per device */
static pthread_mutex_t tab_lock = PTHREAD_MUTEX_INITIALIZER;
static unsigned long long tab_ready;	/* bit per device */

static int traffic_tables(void)
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return (int)e;
	if (dev < 0 || dev > 63)
		return -1;
	pthread_mutex_lock(&tab_lock);
	if (!((tab_ready >> dev) & 1)) {
		uint16_t h[432 + 216];
		for (int i = 0; i < 432 + 216; i++)
			h[i] = 0xffff;
		for (int x = 0; x < 2; x++) {
			const int kind = x ? TG_KIND_216 : TG_KIND_432, base = x ? 432 : 0, nwords = x ? 9 : 18;
			for (int d = 0; d < nwords; d++)
				for (int p = 0; p < 30; p++) {
					const int j = tg_codeword_src(kind, d, p);
					if (j >= 0)
						h[base + j] = (uint16_t)(d << 5 | p);
				}
		}
		for (int i = 0; i < 432 + 216 && e == hipSuccess; i++)
			if (h[i] == 0xffff)
				e = hipErrorUnknown;	/* (every type-4 bit sits in exactly one code-word bit) */
		if (e == hipSuccess)
			e = hipMemcpyToSymbol(HIP_SYMBOL(c_t4pos), h, sizeof(h));
		if (e == hipSuccess)
			tab_ready |= 1ull << dev;
	}
	pthread_mutex_unlock(&tab_lock);
	return (int)e;
}

extern "C" int tgk_traffic(const uint32_t *d_items432, uint32_t n432, const uint32_t *d_items216, uint32_t n216, const uint32_t *d_counts,
			   const uint8_t *d_traffic, const uint32_t *d_packed, const uint32_t *d_masks, const uint32_t *d_maskidx,
			   uint8_t *d_rec, uint8_t *d_wire, uint32_t nslots, uint8_t *d_type4, int16_t *d_blocks, uint16_t *d_lens,
			   void *stream)
{
	int rc = traffic_tables();
	if (rc)
		return rc;
	hipStream_t s = (hipStream_t)stream;
	hipError_t e = hipMemsetAsync(d_lens, 0, (size_t)nslots * sizeof(uint16_t), s);
	if (e != hipSuccess)
		return (int)e;
	const uint32_t items = n432 + n216;
	if (!items)
		return 0;
	hipLaunchKernelGGL(k_traffic, dim3((items + TGT_WAVES - 1) / TGT_WAVES), dim3(64 * TGT_WAVES), 0, s, d_items432, n432, d_items216,
			   n216, d_counts, d_traffic, d_packed, d_masks, d_maskidx, d_rec, d_wire, d_type4, d_blocks, d_lens);
	return (int)hipGetLastError();
}
