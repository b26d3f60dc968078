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
#include "tg_dev_stream.h"
#include "slot_core.h"

/* the front phase of k_slot: the packed-bit stream front end's own text, in its fused form (tg_front_stream_body.h) */
#define TGS_FUSED 1
#ifndef TGS_FUSED_DEPTH
#define TGS_FUSED_DEPTH 4	/* groups' loads in flight per wave in k_slot's front phase (2: as k_front_stream) */
#endif
#define TGS_MARK(i) do { } while (0)
#include "tg_front_stream_body.h"

#ifndef TG_SLOT_WAVES
#define TG_SLOT_WAVES 3		/* waves per SIMD the kernel is compiled for: 36 history blocks x 4 dwords = 144 VGPRs of a lane's 168 */
#endif

/* per-launch flags of the slot kernels (beside TGK_F_*) */
#define TGS_F_LOOKBACK TGK_F_LOOKBACK	/* SYNC lanes with a good SB1 take a code-table entry and set their okbit (as k_vit<SB1> does) */

struct tg_slot_tabs {
	uint16_t crc[TG_CRC_WORDS + 32];			/* the two CRC tables + the 16-dword spread table (tg_bits16) */
	uint32_t bm[TG_BMD_WORDS];				/* branch-metric entries of the difference form (vit_core.h) */
	/* what a lane needs again behind the trellis (slot, meta word, BBK bits, scrambling code) waits HERE, not in registers: the
	 * forward pass holds 144 history registers of the 168 three waves per SIMD allow, and what does not fit goes to scratch memory --
	 * 350 MB of traffic per launch when it did, and kernels with a scratch segment share the chip badly with each other */
	uint32_t keep[4][64];
};
#define TG_SLOT_KEEP(L, lane, a, b, c, d) do { (L).keep[0][lane] = (a); (L).keep[1][lane] = (b); (L).keep[2][lane] = (c); (L).keep[3][lane] = (d); } while (0)
#define TG_SLOT_FORGET() asm volatile("" ::: "memory")	/* (what was parked is read back, not carried) */
#define TG_SLOT_STAGE_WORDS (64 * TG_STAGE_PITCH + 64)		/* the record on its way out: 64 lanes x four dwordx4 at a pitch of 20 dwords + 64 slot numbers */

/* k_slot_t's LDS: tables, and the code words as columns -- word g of lane l at g * 64 + l, descrambled -- whose place the record's
 * staging area takes after the trellis */
struct tg_slot_lds {
	tg_slot_tabs t;
	union {
		uint32_t cw[18 * 64];
		uint32_t stage[TG_SLOT_STAGE_WORDS];
	} u;
};

/* k_slot's LDS: tables, the task's sixteen groups of packed slots as the front phase leaves them (word g of lane l at 20 l + g: the
 * trellis phase's columns; the staging area afterwards), the front phase's own areas */
struct tg_kslot_lds {
	tg_slot_tabs t;
	union {
		uint32_t out[1][16 * 80];
		uint32_t stage[TG_SLOT_STAGE_WORDS];
	} u;
	tg_slot_front_lds F;
};

/* fill the wave's tables (one wave per workgroup) */
__device__ __forceinline__ void slot_tables(tg_slot_tabs &L, uint32_t lane)
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
 * The trellis side of a wave's 64 slots.  In: the lanes' code words as LDS columns, descrambled -- word g of this lane at col[g * GS];
 * colA: the same for g <= 8 (a SYNC burst's SB1 words are read at g = 4..8, slot_core.h: k_slot_t stages them there, colA = col; k_slot
 * keeps the packed slot's layout and hands a SYNC lane col - 4 * GS) --, the burst type per lane.  Out: od[] (36 decoded bytes), the
 * two CRC words.
 */
/* SBMODE: 0 = the lanes say what they hold (two, sb); 1 = no SYNC burst among them (k_slot_t over the batch's NORM_1 / NORM_2 list: the SYNC
 * lanes' prologue and selects fall away); 2 = SYNC bursts only (k_slot_t over the SYNC list: the schedule starts at block slot 8 -- SB1 84
 * steps + SB2 148 instead of 296 for lanes that idle through the first eight block slots in a mixed wave) */
template <int GS, int SBMODE = 0>
__device__ __forceinline__ void slot_trellis(const tg_slot_tabs &L, const uint32_t *col, const uint32_t *colA, bool two_, bool sb_,
					     uint32_t (&od)[TG_SLOT_NOD + 1], uint32_t &crc0, uint32_t &crc1)
{
	const bool sb = SBMODE == 1 ? false : SBMODE == 2 ? true : sb_;
	const bool two = SBMODE == 2 ? true : two_;
	auto bmdo = [&](uint32_t o, uint32_t w[10]) {
		const uint8_t *q = (const uint8_t *)L.bm + o;
		const uint4 a = *(const uint4 *)(q + 4 * TG_BMD_A0);
		const uint4 b = *(const uint4 *)(q + 4 * TG_BMD_A1);
		const uint2 c = *(const uint2 *)(q + 4 * TG_BMD_A2);
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
		w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
		w[8] = c.x; w[9] = c.y;
	};
	/* survivor history: 36 block slots x 4 dwords.  Chunks of eight block slots = 32 registers written through the VGPR index mode; the
	 * last chunk holds four block slots and is a 16-register vector of its own (as a fifth tg_v32 it cost sixteen registers nobody used,
	 * and the compiler parked a whole chunk in scratch memory for the length of the trellis to make room) */
	tg_v32 H[4];
	tg_v16 H4;
	tg_vit_state v;
	tg_slot_state_init(v);
	uint32_t cur = 0;
	if (SBMODE != 2) {
		cur = col[0];
		tg_slot_leadin(v, cur >> 24, bmdo);
	}
	/* (one block slot per loop iteration, a loop per history chunk of eight: 1.7 KB of code each -- the kernel's waves are NOT in step
	 * with each other, by design, and what they execute between them should fit the instruction cache) */
	uint32_t nxt = 0;
	tg_static_for<5>([&](auto cc) __attribute__((always_inline)) {
		constexpr int c = decltype(cc)::value;
		constexpr int b0 = (c == 2) ? 2 : 0;			/* chunk 2's first two block slots (code word 8: slots 16, 17) are written out */
		constexpr int nb = (c == 4) ? 3 : 8;			/* chunk 4: block slots 32..35, the last one written out */
		if constexpr (SBMODE == 2 && c == 0)
			return;						/* (SYNC bursts only: nothing in front of block slot 8) */
		if (c == 1 && SBMODE == 2)
			cur = colA[TG_SLOT_SB1_G0 * GS];
		if (c == 1 && SBMODE != 1)
			tg_slot_sb_prologue(v, sb, cur, bmdo);		/* in front of code word 4: a SYNC lane starts SB1 */
		if (c == 2) {
			nxt = col[9 * GS];
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
		for (int b = b0; b < nb; b++) {
			const int g = 4 * c + (b >> 1);
			uint32_t h[4];
			if (!(b & 1) && g < 17)		/* (wave-uniform) */
				nxt = (c < 2 ? colA : col)[(g + 1) * GS];	/* (c < 2: g + 1 <= 8) */
			tg_slot_block(v, cur >> (12 * (b & 1)), h, bmdo);
			if constexpr (c == 4) {
#pragma unroll
				for (int d = 0; d < 4; d++)
					H4[4 * b + d] = h[d];
			} else {
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][4 * b + d] = h[d];
			}
			if (b & 1)
				cur = nxt;
		}
		if (c == 4) {
			uint32_t h[4];
			tg_slot_block_last(v, cur >> 12, h, bmdo);
#pragma unroll
			for (int d = 0; d < 4; d++)
				H4[12 + d] = h[d];
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
		if constexpr (SBMODE == 2 && b < 2 * TG_SLOT_SB1_G0)
			return;
		else if constexpr (c == 4)
			tg_slot_hop<b>(od, s, two, H4[o], H4[o + 1], H4[o + 2], H4[o + 3]);
		else
			tg_slot_hop<b>(od, s, two, H[c][o], H[c][o + 1], H[c][o + 2], H[c][o + 3]);
	});
	auto tl = [&](uint32_t x) -> uint32_t { return L.crc[x]; };
	auto tm = [&](uint32_t x) -> uint32_t { return L.crc[256 + x]; };
	tg_slot_crc<SBMODE == 2 ? 2 * TG_SLOT_SB1_G0 : 0>(od, two, sb, tl, tm, crc0, crc1);
}

/*
 * What follows the decoded bits: the record as five 64-byte pieces through LDS (each lane parks its four dwordx4 of the piece, then
 * lane l stores quarter l & 3 of the pieces of records (l >> 2) + 16 i: four lanes = one whole segment, sixteen records per store
 * instruction, non-temporal: records are output only), the wire record, and for SYNC lanes what the code look-back wants.
 * Lanes past the end of the list hold a copy of the last item and store the same bytes again.
 */
__device__ __forceinline__ void slot_finish(const tg_slot_tabs &L, uint32_t *stage, uint32_t lane, bool valid, bool two, bool sb, uint32_t slot, uint32_t meta,
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
		uint32_t *st_slot = stage + 64 * TG_STAGE_PITCH;
		st_slot[lane] = valid ? slot : 0xffffffffu;	/* (k_slot_t's lanes past the end of the list come in as copies of the last item, "valid"
								 * for this purpose: they store the same bytes again; a lane of k_slot that decodes nothing stores nothing) */
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
				const uint32_t rs = st_slot[rr];
				if (rs != 0xffffffffu) {
					uint4 *dst = (uint4 *)(rec + (size_t)rs * TG_REC_BYTES + 64 * c + 16 * (lane & 3));
					TG_REC_STORE_SEG(dst, vv);
				}
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
/* one task of k_slot_t: 64 items of a list */
template <int SBMODE>
__device__ __forceinline__ void slot_t_task(tg_slot_lds &L, uint32_t lane, uint32_t task, const uint32_t *__restrict__ items, uint32_t nitems,
					    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks,
					    const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec, uint8_t *__restrict__ wire, int kflags)
{
	uint32_t idx = task * 64 + lane;
	const bool valid = idx < nitems;
	if (!valid)
		idx = nitems - 1;
	const uint32_t slot = items[idx];
	const uint4 *pw = (const uint4 *)(packed + (size_t)slot * TG_PACKED_WORDS);
	const uint4 p0 = pw[0], p1 = pw[1], p2 = pw[2], p3 = pw[3], p4 = pw[4];
	const uint32_t w[TG_PACKED_WORDS] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w,
					      p3.x, p3.y, p3.z, p3.w, p4.x, p4.y, p4.z, p4.w };
	const uint32_t meta = w[TG_PW_META], btype = meta & 0xff;
	const bool sb = SBMODE == 1 ? false : SBMODE == 2 ? true : btype == TG_BURST_SYNC;
	const bool two = SBMODE == 2 ? true : btype != TG_BURST_NORM_1;
	const uint32_t *mk = masks + (size_t)maskidx[slot] * TG_MASK_WORDS;
	/* the code words as columns, descrambled: NORM_1 its 18 words under the 432-bit mask; a two-block burst words 0..8 and 9..17
	 * under the 216-bit mask each; a SYNC burst's SB1 words (fixed code) at g = 4..8 (slot_core.h) */
#pragma unroll
	for (int g = (SBMODE == 2 ? TG_SLOT_SB1_G0 : 0); g < 18; g++) {
		const uint32_t m432 = mk[TG_MW_432 + g], m216 = mk[TG_MW_216 + (g < 9 ? g : g - 9)];
		uint32_t x = w[g] ^ (two ? m216 : m432);
		if (g < 9 && SBMODE != 1) {
			const uint32_t sbw = (g >= TG_SLOT_SB1_G0) ? (w[g - TG_SLOT_SB1_G0] ^ c_tab.sb1_mask[g - TG_SLOT_SB1_G0]) : 0u;
			x = sb ? sbw : x;
		}
		L.u.cw[g * 64 + lane] = x;
	}
	TG_SLOT_KEEP(L.t, lane, slot, meta, w[TG_PW_BBK] ^ mk[TG_MW_BBK], mk[TG_MW_CODE]);
	__syncthreads();
	uint32_t od[TG_SLOT_NOD + 1], crc0, crc1;
	slot_trellis<64, SBMODE>(L.t, L.u.cw + lane, L.u.cw + lane, two, sb, od, crc0, crc1);
	__syncthreads();		/* (single wave: the columns are dead, the staging area takes their place) */
	TG_SLOT_FORGET();
	/* (lanes past the end of the list are copies of the last item: they store its bytes again) */
	slot_finish(L.t, L.u.stage, lane, true, two, sb, L.t.keep[0][lane], L.t.keep[1][lane], L.t.keep[2][lane], L.t.keep[3][lane], od, crc0, crc1, rec,
		    valid ? wire : nullptr, nullptr, nullptr, nullptr, kflags & ~TGS_F_LOOKBACK);
}

/*
 * k_slot_t: the trellises of a batch, one lane per slot.  Two lists of grid slots in one launch (any order inside a list; the counts on
 * the device when the lists were built there): items = the batch's NORM_1 / NORM_2 slots, items2 = its SYNC slots; packed / masks /
 * maskidx as k_vit takes them.  A workgroup takes one task (64 items) of the first list, or -- behind them -- of the second, and runs the
 * schedule compiled for that kind (slot_trellis<., 1> without the SYNC lanes' selects, <., 2> from block slot 8).
 */
__global__ __launch_bounds__(64, TG_SLOT_WAVES)
void k_slot_t(const uint32_t *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ nitems_dev,
	      const uint32_t *__restrict__ items2, uint32_t nitems2, const uint32_t *__restrict__ nitems2_dev,
	      const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx,
	      uint8_t *__restrict__ rec, uint8_t *__restrict__ wire, int kflags)
{
	TG_TRACE_BEGIN;
	__shared__ __attribute__((aligned(16))) tg_slot_lds L;
	if (nitems_dev)
		nitems = *nitems_dev;
	if (nitems2_dev)
		nitems2 = *nitems2_dev;
	const uint32_t t1 = (nitems + 63) / 64, t2 = (nitems2 + 63) / 64;
	if (blockIdx.x >= t1 + t2)
		return;
	const uint32_t lane = threadIdx.x;
	slot_tables(L.t, lane);
	/* (one task per workgroup: a loop over tasks here made the compiler carry enough across the trellis to park a history chunk in
	 * scratch memory again) */
	const uint32_t t = blockIdx.x;
	if (t < t1)
		slot_t_task<1>(L, lane, t, items, nitems, packed, masks, maskidx, rec, wire, kflags);
	else
		slot_t_task<2>(L, lane, t - t1, items2, nitems2, packed, masks, maskidx, rec, wire, kflags);
	TG_TRACE_END(5u, 8u);
}

/*
 * k_slot: front end and trellis of a task's 64 neighbouring grid slots in ONE wave (round 6; VERDICT r5 "next" 1).
 *
 *   front phase    the packed-bit stream front end as it is (tg_front_stream_body.h, fused form): bytes -> bits, the training-sequence
 *                  search of every slot, the de-interleaving / de-puncturing gather; packed slots, classification words and SYNC
 *                  summaries go out to memory as k_front_stream writes them (the exact pass, the walk and the traffic stage read them),
 *                  and the packed slots STAY in LDS;
 *   trellis phase  every lane takes the slot it holds -- if the front phase settled it as a plain NORM_1 / NORM_2 / SYNC burst -- through
 *                  slot_core.h's schedule: no packed-slot read from memory, no item list, no second launch; a wave that waits for
 *                  its next group's bytes in the front phase shares its SIMD with waves that issue add-compare-selects.
 *
 * The trellis phase runs BEFORE the synchroniser's walk and the code look-back, so it decodes on a HINT: the scrambling code the
 * caller's channel table carries in (or, without one, the code the plan's last batch of that channel ended with) -- hint.code[c], its
 * mask in entry hint_base + c.  Nothing is taken on trust: a SYNC lane's own SB1 (fixed code) tells the batch's code table what it
 * decoded, as k_vit<SB1> does; after the walk k_lists2 compares, slot by slot, the code in force (the latest delivered good SB1 at or
 * before the slot, else the carry-in: lower_mac/tetra_lower_mac.c:179-186, 291-300) with the hint's, and every delivered slot that
 * was decoded under another code, or not at all (the exact pass's slots), goes through k_slot_t behind it.  A recording has one cell:
 * after a channel's first batch that list is the handful of delivered slots the exact pass settled.  Records of slots the walk does not
 * deliver are written all the same (their bits are never read: the delivered bitmap says what counts).
 * specbits: one bit per grid slot, "decoded here under the hint".
 */
struct tg_slot_hints {
	uint32_t code[64];	/* per channel; 0 = no hint (the channel's slots are left to k_slot_t) */
};

#ifdef TG_SLOT_TIMING	/* measurement build (tools/experiments/slot_phases.py): per task HW_ID and the 100 MHz clock at start / front phase done /
			 * trellis done / end -- which waves of a SIMD are in which phase when */
__device__ unsigned long long g_slot_stamp[5 * 16384];
extern "C" int tgk_slot_stamps(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slot_stamp), sizeof(g_slot_stamp));
}
#define TGSL_STAMP(k) do { if (lane == 0 && task < 16384u) g_slot_stamp[5 * task + (k)] = (k) ? wall_clock64() : (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)); } while (0)
#else
#define TGSL_STAMP(k) do { } while (0)
#endif

template <bool PACKED>
__global__ __launch_bounds__(64, TG_SLOT_WAVES)
void k_slot(const uint8_t *__restrict__ stream, tg_stream_params prm, uint32_t *__restrict__ packed, uint32_t *__restrict__ cls,
	    uint16_t *__restrict__ ysum, uint32_t *__restrict__ defer, const uint32_t *__restrict__ masks, uint32_t hint_base,
	    tg_slot_hints hint, uint32_t *__restrict__ specbits, uint8_t *__restrict__ rec, uint8_t *__restrict__ wire,
	    uint32_t *__restrict__ tbl, uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_entry, int kflags)
{
	TG_TRACE_BEGIN;
	__shared__ __attribute__((aligned(16))) tg_kslot_lds S;
	const uint32_t lane = threadIdx.x;
	const uint32_t task = blockIdx.x, ntasks = gridDim.x;
#ifndef TG_SLOT_STAGGER
#define TG_SLOT_STAGGER 0
#endif
#if TG_SLOT_STAGGER
	/* (experiment) the first generation of workgroups starts the three waves of a SIMD a third of a task's duration apart, so that
	 * one wave's front phase -- waiting for memory -- lies beside the others' trellis phases; later workgroups take over a slot when
	 * it falls free and inherit its phase */
	if (blockIdx.x < 3072u) {
		const uint32_t wid = __builtin_amdgcn_s_getreg(4 | (3 << 11)) % 3u;	/* HW_ID bits 0..3: the wave's slot on its SIMD */
		const unsigned long long t0 = wall_clock64();
		while (wall_clock64() - t0 < (unsigned long long)wid * TG_SLOT_STAGGER)
			__builtin_amdgcn_s_sleep(64);
	}
#endif
	TGSL_STAMP(0);
	TGSL_STAMP(1);
	slot_tables(S.t, lane);
	slot_front_phase<PACKED>(stream, prm, packed, cls, ysum, defer, S.F, S.u.out, task, ntasks);
	__syncthreads();
	TGSL_STAMP(2);
	const uint32_t slot = 64u * task + lane;
	const uint32_t dt = S.F.s_dtype[lane], ch = S.F.s_chan[lane];
	uint32_t hc = 0;
#pragma unroll 1
	for (uint32_t c = 0; c < 64; c++)	/* (the hints are kernel arguments: scalar registers, picked by a wave-uniform loop) */
		hc = (c == ch) ? hint.code[c] : hc;
	const bool spec = dt != TG_BURST_NONE && hc != 0u && slot < prm.nslots;
	const unsigned long long sm = __ballot(spec);
	if (lane == 0 && 64u * task < prm.nslots)
		specbits[2 * task] = (uint32_t)sm;
	if (lane == 32 && 64u * task + 32u < prm.nslots)
		specbits[2 * task + 1] = (uint32_t)(sm >> 32);
	if (!sm)
		return;		/* (wave-uniform: nothing to decode here -- a channel without a hint, a stretch of damaged slots) */
	const bool sb = dt == TG_BURST_SYNC, two = dt != TG_BURST_NORM_1;
	uint32_t *col = S.u.out[0] + lane * TG_PACKED_WORDS;
	const uint32_t *mk = masks + (size_t)(hint_base + ch) * TG_MASK_WORDS;
	/* descramble in place: NORM_1 its 18 words under the 432-bit mask, a two-block burst words 0..8 and 9..17 under the 216-bit mask
	 * each, a SYNC burst's SB1 words 0..4 under the fixed code's (what went out to memory is the scrambled slot, as ever) */
#pragma unroll
	for (int g = 0; g < 18; g++) {
		const uint32_t m432 = mk[TG_MW_432 + g], m216 = mk[TG_MW_216 + (g < 9 ? g : g - 9)];
		uint32_t m = two ? m216 : m432;
		if (g < 5)
			m = sb ? c_tab.sb1_mask[g] : m;
		col[g] ^= m;
	}
	TG_SLOT_KEEP(S.t, lane, slot, col[TG_PW_META] | (spec ? 0x80000000u : 0u), col[TG_PW_BBK] ^ mk[TG_MW_BBK], mk[TG_MW_CODE]);	/* (meta: type, flags, offset < 2^25) */
	__syncthreads();
	uint32_t od[TG_SLOT_NOD + 1], crc0, crc1;
	slot_trellis<1>(S.t, col, sb ? col - TG_SLOT_SB1_G0 : col, two, sb, od, crc0, crc1);
	__syncthreads();		/* (single wave: the columns are dead, the staging area takes their place) */
	TGSL_STAMP(3);
	TG_SLOT_FORGET();
	/* (the lane number afresh: carried across the trellis -- as 16 x lane, the front phase's load offset -- it was the one thing left in
	 * scratch memory) */
	const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	const uint32_t km = S.t.keep[1][ln];	/* (bit 31: "this lane decodes its slot" -- parked with the rest, nothing carried per lane) */
	slot_finish(S.t, S.u.stage, ln, (km >> 31) != 0u, two, sb, S.t.keep[0][ln], km & 0x7fffffffu, S.t.keep[2][ln], S.t.keep[3][ln], od, crc0, crc1, rec,
		    wire, tbl, sb_ok, sb_entry, kflags);
	TGSL_STAMP(4);
	TG_TRACE_END(6u, 8u);
}

/*
 * k_slot_e: the trellis phase of k_slot on its own, EARLY -- over the grid slots the front end (both passes) left classified as plain
 * bursts, 64 neighbouring ones per wave, decoding on the channels' hinted codes as k_slot does, but reading the packed slots from
 * memory.  It needs nothing of the walk, so a caller who waits for every batch runs it on a stream of its own BESIDE the walk and the
 * code look-back (TGPU_OPT_SLOT 3: a batch's latency, not its rate -- it decodes the plain slots the walk drops as well, and its waves
 * hold bursts of all types); k_lists2 then lists what was decoded under another code than the one in force, or not at all, for
 * k_slot_t.  SB1 blocks are decoded here too but the look-back does not wait for them: k_vit<SB1> runs in front of it as ever.
 */
#ifndef TG_SLOT_E_WAVES
#define TG_SLOT_E_WAVES 2	/* waves per SIMD k_slot_e may hold: TWO -- at three it fills every SIMD's registers (3 x 168 of 512), the walk's and the
				 * look-back's small workgroups find no place until it has drained, and "beside the walk" becomes "in front of it"
				 * (one batch at a time 0.596 ms against 0.598 without the kernel; at two 0.603 against 0.620; at one 0.73:
				 * tools/experiments/one_batch.py) */
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TG_SLOT_E_WAVES, TG_SLOT_E_WAVES)))
void k_slot_e(const uint32_t *__restrict__ cls, uint32_t nslots, const tg_chan_ent *__restrict__ chan, uint32_t nchan,
	      const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks, uint32_t hint_base, tg_slot_hints hint,
	      uint8_t *__restrict__ rec, uint8_t *__restrict__ wire, int kflags)
{
	TG_TRACE_BEGIN;
	__shared__ __attribute__((aligned(16))) tg_slot_lds L;
	const uint32_t lane = threadIdx.x;
	const uint32_t slot = 64u * blockIdx.x + lane;
	const uint32_t v = slot < nslots ? cls[slot] & 0x03ffffffu : 0xffu;
	const bool sbq = v == (TG_BURST_SYNC | TG_SYNC_TRAIN_OFF << 8), n1q = v == (TG_BURST_NORM_1 | TG_NORM_TRAIN_OFF << 8);
	const bool plain = sbq || n1q || v == (TG_BURST_NORM_2 | TG_NORM_TRAIN_OFF << 8);
	uint32_t ch = 0;
	for (uint32_t q = 1; q < nchan; q++)	/* (grids start at multiples of 32: a wave's 64 slots lie in one or two channels, mostly) */
		ch += chan[q].gbase <= slot;
	uint32_t hc = 0;
#pragma unroll 1
	for (uint32_t c = 0; c < 64; c++)
		hc = (c == ch) ? hint.code[c] : hc;
	const bool spec = plain && hc != 0u;
	if (!__ballot(spec))
		return;
	slot_tables(L.t, lane);
	const bool sb = sbq, two = !n1q;
	const uint4 *pw = (const uint4 *)(packed + (size_t)(spec ? slot : 64u * blockIdx.x) * TG_PACKED_WORDS);
	const uint4 p0 = pw[0], p1 = pw[1], p2 = pw[2], p3 = pw[3], p4 = pw[4];
	const uint32_t w[TG_PACKED_WORDS] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w,
					      p3.x, p3.y, p3.z, p3.w, p4.x, p4.y, p4.z, p4.w };
	const uint32_t *mk = masks + (size_t)(hint_base + ch) * TG_MASK_WORDS;
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
	TG_SLOT_KEEP(L.t, lane, slot, w[TG_PW_META] | (spec ? 0x80000000u : 0u), w[TG_PW_BBK] ^ mk[TG_MW_BBK], mk[TG_MW_CODE]);
	__syncthreads();
	uint32_t od[TG_SLOT_NOD + 1], crc0, crc1;
	slot_trellis<64, 0>(L.t, L.u.cw + lane, L.u.cw + lane, two, sb, od, crc0, crc1);
	__syncthreads();
	TG_SLOT_FORGET();
	const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	const uint32_t km = L.t.keep[1][ln];
	slot_finish(L.t, L.u.stage, ln, (km >> 31) != 0u, two, sb, L.t.keep[0][ln], km & 0x7fffffffu, L.t.keep[2][ln], L.t.keep[3][ln], od, crc0, crc1, rec,
		    wire, nullptr, nullptr, nullptr, kflags & ~TGS_F_LOOKBACK);
	TG_TRACE_END(7u, 8u);
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

/* two lists in one launch: d_items the NORM_1 / NORM_2 slots, d_items2 the SYNC slots.  nitems / nitems2: the counts, or -- with
 * d_nitems / d_nitems2 -- upper bounds of counts that are on the device; ntotal (0: the sum): an upper bound of BOTH together where the
 * caller has one (a grid slot is on one list at most: the launch is sized for it) */
extern "C" int tgk_slot_t(const uint32_t *d_items, uint32_t nitems, const uint32_t *d_nitems, const uint32_t *d_items2, uint32_t nitems2,
			  const uint32_t *d_nitems2, uint32_t ntotal, const uint32_t *d_packed, const uint32_t *d_masks, const uint32_t *d_maskidx,
			  uint8_t *d_rec, uint8_t *d_wire, int flags, void *stream)
{
	if (!d_items || !d_items2)
		return -1;
	if (!nitems && !nitems2)
		return 0;
	uint32_t tasks = (nitems + 63) / 64 + (nitems2 + 63) / 64;
	if (ntotal && (ntotal + 63) / 64 + 1 < tasks)
		tasks = (ntotal + 63) / 64 + 1;		/* (each list ends in one partly filled task) */
	const dim3 grid(tasks), block(64);
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(k_slot_t, grid, block, 0, s, d_items, nitems, d_nitems, d_items2, nitems2, d_nitems2, d_packed, d_masks, d_maskidx, d_rec, d_wire,
			   flags);
	return (int)hipGetLastError();
}

/* front end + trellis of a device-walk batch in one launch (k_slot), then the exact pass over what the front phase deferred.  Arguments
 * as tgk_front_stream_multi takes them, plus: the mask table and the entry of channel 0's hint, the hints (nchan of them; 0 = none), the
 * "decoded here" bitmap, records / wire records, and the code look-back's table, okbits and per-slot entries (cleared by the caller) */
extern "C" int tgk_slot_fused(const uint8_t *d_base, const struct tg_chan_ent *d_chan, uint32_t nchan, uint32_t nslots, uint32_t chunk,
			      uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum, uint32_t *d_defer, const uint32_t *d_masks,
			      uint32_t hint_base, const uint32_t *hints, uint32_t *d_specbits, uint8_t *d_rec, uint8_t *d_wire,
			      uint32_t *d_tbl, uint32_t *d_sb_ok, uint32_t *d_sb_entry, int flags, void *stream, void *ev_mid, int packed_input)
{
	if (!nslots)
		return 0;
	if (!chunk || !nchan || nchan > 64 || (nslots & 31) || !hints)
		return -1;
	tg_stream_params prm;
	tgk_stream_params_multi(&prm, d_chan, nchan, nslots, chunk);
	tg_slot_hints h;
	memset(&h, 0, sizeof(h));
	memcpy(h.code, hints, (size_t)nchan * 4);
	const uint32_t ngroups = (nslots + 3) / 4, ntasks = (ngroups + 15) / 16;
	hipStream_t s = (hipStream_t)stream;
	tgk_front_stream_ev_fire(s);
	if (packed_input)
		hipLaunchKernelGGL(k_slot<true>, dim3(ntasks), dim3(64), 0, s, d_base, prm, d_packed, d_cls, d_ysum, d_defer, d_masks, hint_base, h,
				   d_specbits, d_rec, d_wire, d_tbl, d_sb_ok, d_sb_entry, flags);
	else
		hipLaunchKernelGGL(k_slot<false>, dim3(ntasks), dim3(64), 0, s, d_base, prm, d_packed, d_cls, d_ysum, d_defer, d_masks, hint_base, h,
				   d_specbits, d_rec, d_wire, d_tbl, d_sb_ok, d_sb_entry, flags);
	int rc = (int)hipGetLastError();
	if (rc)
		return rc;
	if (ev_mid)
		HIPCHK(hipEventRecord((hipEvent_t)ev_mid, s));
	/* the exact pass: a list per task, as many entries as the task's groups can hold (tg_front_stream_body.h) */
	return tgk_front_stream_fix(d_base, &prm, d_packed, d_cls, d_ysum, d_defer, ntasks, 4u * ((ngroups + ntasks - 1) / ntasks), s, packed_input);
}

/* the early trellis of a device-walk batch (k_slot_e): every plain grid slot on its channel's hinted code, beside the walk */
extern "C" int tgk_slot_early(const uint32_t *d_cls, uint32_t nslots, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_packed,
			      const uint32_t *d_masks, uint32_t hint_base, const uint32_t *hints, uint8_t *d_rec, uint8_t *d_wire, int flags, void *stream)
{
	if (!nslots)
		return 0;
	if (!nchan || nchan > 64 || !hints)
		return -1;
	tg_slot_hints h;
	memset(&h, 0, sizeof(h));
	memcpy(h.code, hints, (size_t)nchan * 4);
	hipLaunchKernelGGL(k_slot_e, dim3((nslots + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_cls, nslots, d_chan, nchan, d_packed, d_masks, hint_base, h,
			   d_rec, d_wire, flags);
	return (int)hipGetLastError();
}
