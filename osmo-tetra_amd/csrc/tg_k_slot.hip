/*
 * tg_k_slot.hip -- one lane = one slot (round 6): the whole lower-MAC decode of a burst -- descramble, both blocks' trellises (a SYNC
 * burst's SB1 included), CRC-16, BBK, the 320-byte record and the wire record -- in one pass of one lane, whatever the burst type.
 *
 *   k_slot_t   the trellis side alone: a wave takes 64 items of ONE list (the batch's delivered slots, any mix of NORM_1 / NORM_2 /
 *              SYNC) and replaces k_vit<216> + k_vit<432> of a batch -- every line of the packed-slot area is touched by one
 *              launch instead of two (the burst types alternate inside a line), one tail instead of two, every record leaves as
 *              whole 64-byte segments.
 *
 * The schedule that lets 64 slots of mixed types run the same instructions is slot_core.h's (compiled for the host and checked against
 * the oracle in tests/test_host_logic.py); the per-step arithmetic is vit_core.h's (difference form, ties as libosmocore).  This file
 * holds what is wave-level: staging the code words as LDS columns, the survivor history in VGPRs, the record's way out through LDS.
 *
 * Reference: lower_mac/tetra_lower_mac.c:143-282 (tp_sap_udata_ind: one block's chain), phy/tetra_burst.c:341-379 (the blocks of a burst).
 */
#include "tg_dev_vit.h"
#include "slot_core.h"

#ifndef TG_SLOT_WAVES
#define TG_SLOT_WAVES 3		/* waves per SIMD the kernel is compiled for: 36 history blocks x 4 dwords = 144 VGPRs of a lane's 168 */
#endif

/* per-launch flags of the slot kernels (beside TGK_F_*) */
#define TGS_F_LOOKBACK TGK_F_LOOKBACK	/* SYNC lanes with a good SB1 take a code-table entry and set their okbit (as k_vit<SB1> does) */

struct tg_slot_lds {
	uint16_t crc[TG_CRC_WORDS + 32];			/* the two CRC tables + the 16-dword spread table (tg_bits16) */
	uint32_t bm[TG_BMD_WORDS];				/* branch-metric entries of the difference form (vit_core.h) */
	union {
		uint32_t cw[18 * 64];				/* code word g of lane l at g * 64 + l (descrambled) */
		uint32_t stage[64 * TG_STAGE_PITCH + 64];	/* the record on its way out (after the trellis: the columns are dead) */
	} u;
};

/* fill the wave's tables (one wave per workgroup) */
__device__ __forceinline__ void slot_tables(tg_slot_lds &L, uint32_t lane)
{
	for (int i = lane; i < 256; i += 64) {
		L.crc[i] = c_tab.crc_lsb[i];
		L.crc[256 + i] = c_tab.crc_msb[i];
	}
	tg_sp_fill(L.crc, lane);
	if (lane < 32) {
		uint32_t w[10];
		tg_bmd_entry(lane >> 3, lane & 7, w);
		tg_bmd_store(L.bm, (int)lane, w);
	}
}

/*
 * The trellis side of a wave's 64 slots.  In: the lanes' code words as LDS columns (L.u.cw, descrambled, a SYNC burst's SB1 words
 * at g = 4..8: slot_core.h), the burst type per lane.  Out: od[] (36 decoded bytes), the two CRC words.
 */
__device__ __forceinline__ void slot_trellis(tg_slot_lds &L, uint32_t lane, bool two, bool sb, uint32_t (&od)[TG_SLOT_NOD + 1],
					     uint32_t &crc0, uint32_t &crc1)
{
	auto bmdo = [&](uint32_t o, uint32_t w[10]) {
		const uint8_t *q = (const uint8_t *)L.bm + o;
		const uint4 a = *(const uint4 *)(q + 4 * TG_BMD_A0);
		const uint4 b = *(const uint4 *)(q + 4 * TG_BMD_A1);
		const uint2 c = *(const uint2 *)(q + 4 * TG_BMD_A2);
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
		w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
		w[8] = c.x; w[9] = c.y;
	};
	const uint32_t *col = L.u.cw + lane;
	tg_v32 H[5];
	tg_vit_state v;
	tg_slot_state_init(v);
	uint32_t cur = col[0];
	tg_slot_leadin(v, cur >> 24, bmdo);
	tg_static_for<5>([&](auto cc) __attribute__((always_inline)) {
		constexpr int c = decltype(cc)::value;
		constexpr int it0 = (c == 2) ? 1 : 0;			/* chunk 2's first iteration (code word 8) is written out: block slot 17 */
		constexpr int nloop = (c == 4) ? 1 : 4;			/* chunk 4: code words 16 and 17, the second one written out (the last block) */
		if (c == 1)
			tg_slot_sb_prologue(v, sb, cur, bmdo);		/* in front of code word 4: a SYNC lane starts SB1 */
		if (c == 2) {
			const uint32_t nxt = col[9 * 64];
			uint32_t h[4];
			tg_slot_block(v, cur, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H[c][d] = h[d];
			tg_slot_mid(v, two, cur >> 12, nxt >> 24, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H[c][4 + d] = h[d];
			tg_vit_normalize_floor(v);
			cur = nxt;
		}
#pragma unroll 1
		for (int it = it0; it < nloop; it++) {
			const int g = 4 * c + it;
			const uint32_t nxt = col[(g + 1) * 64];
			uint32_t h[4];
			tg_slot_block(v, cur, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H[c][8 * it + d] = h[d];
			tg_slot_block(v, cur >> 12, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H[c][8 * it + 4 + d] = h[d];
			cur = nxt;
		}
		if (c == 4) {
			uint32_t h[4];
			tg_slot_block(v, cur, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H[c][8 + d] = h[d];
			tg_slot_block_last(v, cur >> 12, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H[c][12 + d] = h[d];
		}
	});
	/* block-wise traceback from state 0, all register indices static */
#pragma unroll
	for (int i = 0; i <= TG_SLOT_NOD; i++)
		od[i] = 0;
	uint32_t s = 0;
	tg_static_for<TG_SLOT_NBLK>([&](auto ii) __attribute__((always_inline)) {
		constexpr int b = TG_SLOT_NBLK - 1 - decltype(ii)::value;
		constexpr int c = b >> 3, o = 4 * (b & 7);
		tg_slot_hop<b>(od, s, two, H[c][o], H[c][o + 1], H[c][o + 2], H[c][o + 3]);
	});
	auto tl = [&](uint32_t x) -> uint32_t { return L.crc[x]; };
	auto tm = [&](uint32_t x) -> uint32_t { return L.crc[256 + x]; };
	tg_slot_crc(od, two, sb, tl, tm, crc0, crc1);
}

/*
 * What follows the decoded bits: the record as five 64-byte pieces through LDS (each lane parks its four dwordx4 of the piece, then
 * lane l stores quarter l & 3 of the pieces of records (l >> 2) + 16 i: four lanes = one whole segment, sixteen records per store
 * instruction, non-temporal: records are output only), the wire record, and for SYNC lanes what the code look-back wants.
 * Lanes past the end of the list hold a copy of the last item and store the same bytes again.
 */
__device__ __forceinline__ void slot_finish(tg_slot_lds &L, uint32_t lane, bool valid, bool two, bool sb, uint32_t slot, uint32_t meta,
					    uint32_t bb, uint32_t code, const uint32_t (&od)[TG_SLOT_NOD + 1], uint32_t crc0, uint32_t crc1,
					    uint8_t *__restrict__ rec, uint8_t *__restrict__ wire, uint32_t *__restrict__ tbl,
					    uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_entry, int kflags)
{
	const uint32_t ok0 = crc0 == TG_CRC_OK, ok1 = two && crc1 == TG_CRC_OK;
	uint32_t f0 = 0, f1 = 0, sbcode = 0;
	if (sb)
		tg_slot_sync_fields(od + 2, f0, f1, sbcode);
	const uint32_t *lut = TG_SP_LUT(L.crc);
	if (!(kflags & TGK_F_WIREONLY)) {
		uint32_t *stage = L.u.stage;
		uint32_t *st_slot = stage + 64 * TG_STAGE_PITCH;
		st_slot[lane] = slot;
		uint4 *mine = (uint4 *)(stage + lane * TG_STAGE_PITCH);
		tg_static_for<5>([&](auto cc) __attribute__((always_inline)) {
			constexpr int c = decltype(cc)::value;
			tg_static_for<4>([&](auto ii) __attribute__((always_inline)) {
				constexpr int i = decltype(ii)::value, u = 4 * c + i;
				if constexpr (u == 0)
					mine[i] = make_uint4((meta & 0xffffu) | (ok0 << 16) | (ok1 << 24), crc0 | (crc1 << 16), code, slot);
				else if constexpr (u == 1)
					mine[i] = make_uint4(f0, f1, sbcode, 0u);
				else if constexpr (u == 2) {
					uint4 b4 = tg_bits16(lut, bb, 0);
					b4.w &= 0x0000ffffu;	/* 14 type-1 bits (tetra_lower_mac.c:268-274) */
					mine[i] = b4;
				} else {
					bool full, empty;
					const uint32_t h16 = tg_slot_piece_bits<u - 3>(od, two, sb, full, empty);
					uint4 o = tg_bits16(lut, h16, 0);
					o.w = full ? o.w : 0u;
					if (empty)
						o = make_uint4(0u, 0u, 0u, 0u);
					mine[i] = o;
				}
			});
			__builtin_amdgcn_s_waitcnt(0xc07f);	/* lgkmcnt(0): single wave, LDS visible */
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t rr = (lane >> 2) + 16 * i;
				const uint4 vv = *(const uint4 *)(stage + rr * TG_STAGE_PITCH + 4 * (lane & 3));
				uint4 *dst = (uint4 *)(rec + (size_t)st_slot[rr] * TG_REC_BYTES + 64 * c + 16 * (lane & 3));
				TG_REC_STORE_SEG(dst, vv);
			}
			__builtin_amdgcn_wave_barrier();
		});
	}
	if (wire && valid) {		/* tg_layout.h "Wire record": ten dwords, the lane owns all of them */
		uint32_t *wr = (uint32_t *)(wire + (size_t)slot * TG_WIRE_BYTES);
		uint32_t w[TG_WIRE_WORDS];
		w[0] = (meta & 0xffffu) | ((bb & 0x3fff) << 16);
		if (!two) {
#pragma unroll
			for (int q = 0; q < 8; q++)
				w[1 + q] = od[q];
			w[9] = (od[8] & 0xfffu) | (crc0 << TG_WIRE_SCHF_CRC_SHIFT);
		} else {
			/* first block: 124 bits from decoded bit 0 (SYNC: SB1's 60 bits from decoded bit 64), second: 124 bits from bit 144 */
			w[1] = sb ? od[2] : od[0];
			w[2] = sb ? (od[3] & 0x0fffffffu) : od[1];
			w[3] = sb ? 0u : od[2];
			w[4] = sb ? 0u : (od[3] & 0x0fffffffu);
#pragma unroll
			for (int q = 0; q < 4; q++)
				w[5 + q] = (od[4 + q] >> 16) | (od[5 + q] << 16);
			w[8] &= 0x0fffffffu;
			w[9] = crc0 | (crc1 << 16);
		}
#pragma unroll
		for (int q = 0; q < TG_WIRE_WORDS; q += 2)		/* (40-byte records: 8-byte aligned) */
			*(uint2 *)(wr + q) = make_uint2(w[q], w[q + 1]);
	}
	if ((kflags & TGS_F_LOOKBACK) && tbl) {
		/* device-walk batches (k_lists2): sb_ok = one bit per grid slot "SB1 passed its CRC", sb_entry = the slot's mask-table entry,
		 * tbl = the batch's code table (open addressing, 0 = free: a code ends in binary 11); one table access per DISTINCT code of
		 * the wave (vit_finish<SB1> has the reasoning) */
		const bool live = valid && sb && ok0;
		uint32_t myh = 0;
		unsigned long long todo = __ballot(live);
		while (todo) {
			const uint32_t l0 = (uint32_t)__builtin_ctzll(todo);
			const uint32_t c0 = __builtin_amdgcn_readlane(sbcode, l0);
			uint32_t h = (c0 * 2654435761u) >> 20, probe = 0;
			if (lane == l0) {
				for (; probe < TG_LB_TBL; probe++) {
					uint32_t old = __hip_atomic_load(&tbl[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (old == 0u)
						old = atomicCAS(&tbl[h], 0u, c0);
					if (old == 0u || old == c0)
						break;
					h = (h + 1) & (TG_LB_TBL - 1);
				}
				if (probe == TG_LB_TBL) {
					atomicOr(&tbl[TG_LB_TBL], 1u);
					h = 0;
				}
			}
			h = __builtin_amdgcn_readlane(h, l0);
			const bool minec = live && sbcode == c0;
			if (minec)
				myh = h;
			todo &= ~__ballot(minec);
		}
		if (live) {
			sb_entry[slot] = 1u + ((uint32_t)kflags >> 8) + myh;
			atomicOr(&sb_ok[slot >> 5], 1u << (slot & 31));
		}
	}
}

/*
 * k_slot_t: items[] = grid slots of the batch's delivered bursts (any order, any mix of types; the count on the device).
 * packed / masks / maskidx as k_vit takes them.
 */
__global__ __launch_bounds__(64, TG_SLOT_WAVES)
void k_slot_t(const uint32_t *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ nitems_dev,
	      const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx,
	      uint8_t *__restrict__ rec, uint8_t *__restrict__ wire, int kflags)
{
	TG_TRACE_BEGIN;
	__shared__ __attribute__((aligned(16))) tg_slot_lds L;
	if (nitems_dev) {
		nitems = *nitems_dev;
		if (blockIdx.x * 64 >= nitems)
			return;
	}
	const uint32_t lane = threadIdx.x;
	slot_tables(L, lane);
	uint32_t idx = blockIdx.x * 64 + lane;
	const bool valid = idx < nitems;
	if (!valid)
		idx = nitems - 1;
	const uint32_t slot = items[idx];
	const uint4 *pw = (const uint4 *)(packed + (size_t)slot * TG_PACKED_WORDS);
	const uint4 p0 = pw[0], p1 = pw[1], p2 = pw[2], p3 = pw[3], p4 = pw[4];
	const uint32_t w[TG_PACKED_WORDS] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w,
					      p3.x, p3.y, p3.z, p3.w, p4.x, p4.y, p4.z, p4.w };
	const uint32_t meta = w[TG_PW_META], btype = meta & 0xff;
	const bool sb = btype == TG_BURST_SYNC, two = btype != TG_BURST_NORM_1;
	const uint32_t *mk = masks + (size_t)maskidx[slot] * TG_MASK_WORDS;
	/* the code words as columns, descrambled: NORM_1 its 18 words under the 432-bit mask; a two-block burst words 0..8 and 9..17
	 * under the 216-bit mask each; a SYNC burst's SB1 words (fixed code) at g = 4..8 (slot_core.h) */
#pragma unroll
	for (int g = 0; g < 18; g++) {
		const uint32_t m432 = mk[TG_MW_432 + g], m216 = mk[TG_MW_216 + (g < 9 ? g : g - 9)];
		uint32_t x = w[g] ^ (two ? m216 : m432);
		if (g < 9) {
			const uint32_t sbw = (g >= TG_SLOT_SB1_G0) ? (w[g - TG_SLOT_SB1_G0] ^ c_tab.sb1_mask[g - TG_SLOT_SB1_G0]) : 0u;
			x = sb ? sbw : x;
		}
		L.u.cw[g * 64 + lane] = x;
	}
	const uint32_t bb = w[TG_PW_BBK] ^ mk[TG_MW_BBK];
	const uint32_t code = mk[TG_MW_CODE];
	__syncthreads();
	uint32_t od[TG_SLOT_NOD + 1], crc0, crc1;
	slot_trellis(L, lane, two, sb, od, crc0, crc1);
	__syncthreads();		/* (single wave: the columns are dead, the staging area takes their place) */
	slot_finish(L, lane, valid, two, sb, slot, meta, bb, code, od, crc0, crc1, rec, wire, nullptr, nullptr, nullptr, kflags & ~TGS_F_LOOKBACK);
	TG_TRACE_END(5u, 8u);
}

/* ------------------------------------------------------------------------- */
/* host-side launch layer of this unit                                        */
/* ------------------------------------------------------------------------- */
extern "C" int tgk_upload_slot(const tg_const_tables *host)
{
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), host, sizeof(*host)));
	return 0;
}

#ifdef TG_TRACE
extern "C" int tgk_trace_read_slot(void *out, unsigned int *n, int reset)
{
	return tg_trace_read_unit(out, n, reset);
}
#endif

extern "C" int tgk_slot_t(const uint32_t *d_items, uint32_t nitems, const uint32_t *d_nitems, const uint32_t *d_packed,
			  const uint32_t *d_masks, const uint32_t *d_maskidx, uint8_t *d_rec, uint8_t *d_wire, int flags, void *stream)
{
	if (!nitems)
		return 0;
	hipLaunchKernelGGL(k_slot_t, dim3((nitems + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_items, nitems, d_nitems, d_packed, d_masks,
			   d_maskidx, d_rec, d_wire, flags);
	return (int)hipGetLastError();
}
