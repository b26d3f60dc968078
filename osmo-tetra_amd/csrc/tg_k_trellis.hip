/*
 * tg_k_trellis.hip -- the trellis kernels: k_vit<KIND,HMODE>, k_clean, k_bbk_blocks, k_burst / k_burst_ring, k_conv
 * (one of the four HIP units of the library: tg_dev.h has the map)
 */
#include "tg_dev.h"
#include <utility>


/* clean-block fast path (k_clean): [0..4095] 12 received bits of an 8-step block -> g1 bits | g2 bits << 8;
 * [4096..8191] (state << 8 | g1 bits) -> input bits | expected g2 bits << 8 | next state << 12 */
__device__ uint16_t g_clean_lut[8192];

/* optional RM(30,14) decoder of the BBK (tg_rm.c): coset leaders by syndrome, generator parity rows */
__device__ const uint32_t *g_rm_leader;
__constant__ uint16_t c_rm_parity[14];

/* bb: bit p = p-th received BBK bit (descrambled).  Returns the corrected word in the same order. */
__device__ __forceinline__ uint32_t rm3014_correct(uint32_t bb, uint32_t &nerr)
{
	const uint32_t rx = __builtin_bitreverse32(bb & 0x3fffffffu) >> 2;	/* codeword bit 29 = first received bit */
	uint32_t syn = rx & 0xffff;
#pragma unroll
	for (int i = 0; i < 14; i++)
		syn ^= ((rx >> (29 - i)) & 1) ? c_rm_parity[i] : 0u;
	const uint32_t e = g_rm_leader[syn];
	nerr = __builtin_popcount(e);
	return __builtin_bitreverse32(rx ^ e) >> 2;
}

__global__ __launch_bounds__(256)
void k_bbk_blocks(const uint32_t *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ packed,
		  const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec, int kflags)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nitems)
		return;
	const uint32_t b = items[i];
	const uint32_t midx = maskidx[b];
	const uint32_t meta = packed[(size_t)b * TG_PACKED_WORDS + TG_PW_META];
	uint32_t bb = packed[(size_t)b * TG_PACKED_WORDS + TG_PW_BBK] ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
	uint8_t *r = rec + (size_t)b * TG_REC_BYTES;
	uint32_t nerr = 0;
	if (kflags & TGK_F_RM)
		bb = rm3014_correct(bb, nerr);
	r[TG_REC_BBK_NERR] = (uint8_t)nerr;
	uint4 o;
	o.x = spread4(bb);
	o.y = spread4(bb >> 4);
	o.z = spread4(bb >> 8);
	o.w = spread4(bb >> 12) & 0x0000ffffu;
	*(uint4 *)(r + TG_REC_BBK) = o;
	r[TG_REC_TYPE] = (uint8_t)meta;
	r[TG_REC_FLAGS] = (uint8_t)(meta >> 8);
	r[TG_REC_CRC_OK] = 1;
	*(uint32_t *)(r + TG_REC_CODE) = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
	*(uint32_t *)(r + TG_REC_SLOT) = b;
}

/* ------------------------------------------------------------------------- */
/* k_vit<KIND>                                                               */
/* ------------------------------------------------------------------------- */
template <int KIND> struct vit_cfg;
template <> struct vit_cfg<TG_KIND_SB1> { enum { NBLK = 10, TYPE1 = 60, MW = 0 }; };
template <> struct vit_cfg<TG_KIND_216> { enum { NBLK = 18, TYPE1 = 124, MW = TG_MW_216 }; };
template <> struct vit_cfg<TG_KIND_432> { enum { NBLK = 36, TYPE1 = 268, MW = TG_MW_432 }; };
template <> struct vit_cfg<TG_KIND_168> { enum { NBLK = 14, TYPE1 = 92, MW = TG_MW_168 }; };

#include "tg_dev_vit.h"	/* field_msb, tg_static_for, tg_bits16 / TG_SP_LUT, hist_byte: shared with tg_k_slot.hip */

/*
 * What follows the decoded bits of a block, shared by the trellis kernels and the clean-block fast path:
 * CRC-16, type-1 bits at one byte per bit, record header / BBK / SYNC-PDU fields, optional wire record.
 * od[]: decoded type-2 bits, LSB first (bit i = input bit i of the encoder).
 */
template <int KIND, int HMODE>
__device__ __forceinline__ void vit_finish(const uint32_t (&od)[(vit_cfg<KIND>::NBLK + 3) / 4 + 1], const uint16_t *s_crc, bool valid,
					    uint32_t slot, uint32_t which, uint32_t idx, uint32_t midx,
					    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks,
					    uint8_t *__restrict__ rec, uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code,
					    uint8_t *__restrict__ wire, const uint32_t *__restrict__ softarea, int kflags,
					    uint32_t *stage = nullptr)
{
	constexpr int NBLK = vit_cfg<KIND>::NBLK;
	constexpr int TYPE1 = vit_cfg<KIND>::TYPE1;
	const bool block_mode = kflags & TGK_F_BLOCK;
	/* CRC-16 over type1 + 16 bits = (NBLK-1) bytes + 4 bits (lower_mac/tetra_lower_mac.c:258) */
	uint32_t crc = 0xffff;
#pragma unroll
	for (int i = 0; i < NBLK - 1; i++) {
		const uint32_t byte = (od[i >> 2] >> ((i & 3) * 8)) & 0xff;
		crc = ((crc << 8) & 0xffff) ^ s_crc[256 + (crc >> 8)] ^ s_crc[byte];
	}
	{
		const uint32_t nib = (od[(NBLK - 1) >> 2] >> (((NBLK - 1) & 3) * 8)) & 15;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			crc ^= ((nib >> i) & 1) << 15;
			crc = (crc & 0x8000) ? (((crc << 1) ^ 0x1021) & 0xffff) : ((crc << 1) & 0xffff);
		}
	}
	const uint32_t crc_ok = (crc == 0x1d0f);

	const bool wire_only = kflags & TGK_F_WIREONLY;
	if (KIND == TG_KIND_432 && stage && !block_mode && !wire_only) {
		/* SCH/F: the lane owns the whole 320-byte record.  Written 16 bytes at a time per lane, every store
		 * instruction touches 64 cache lines and every line is filled from memory before it is complete
		 * (FETCH_SIZE 3x the input).  Instead the record goes out in five 64-byte pieces through LDS: each lane parks
		 * its four dwordx4 of the piece, then lane l stores quarter (l & 3) of the pieces of records (l >> 2) + 16 i --
		 * four lanes = one complete 64-byte segment, sixteen records per store instruction.  Lanes past the end
		 * of the list hold a copy of the last item and store the same bytes again. */
		const uint32_t lane = threadIdx.x & 63;
		uint32_t *st_slot = stage + 64 * TG_STAGE_PITCH;	/* the 64 slot numbers */
		st_slot[lane] = slot;
		const uint32_t meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
		uint32_t bbraw;
		if (HMODE == 2) {
			const uint32_t *sb = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + TG_SOFT_BBK / 4;
			bbraw = 0;
#pragma unroll
			for (int q = 0; q < 4; q++)
				bbraw |= ((((sb[q] >> 7) & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
		} else
			bbraw = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_BBK];
		uint32_t bb = bbraw ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
		uint32_t nerr = 0;
		if (kflags & TGK_F_RM)
			bb = rm3014_correct(bb, nerr);
		const uint32_t code = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
		auto bits16 = [&](int q) {	/* type-1 bits 16 q .. 16 q + 15, one per byte */
			return tg_bits16(TG_SP_LUT(s_crc), od[q >> 1], q & 1, q * 16 + 12 < TYPE1);
		};
		uint4 *mine = (uint4 *)(stage + lane * TG_STAGE_PITCH);
#pragma unroll
		for (int c = 0; c < 5; c++) {
			if (c == 0) {
				/* bytes 0..15: type, flags, crc_ok[2], crc[2], code, slot; 16..31: SYNC fields (none), BBK errors */
				mine[0] = make_uint4((meta & 0xffffu) | (crc_ok << 16), crc, code, slot);
				mine[1] = make_uint4(0u, 0u, 0u, nerr);
				{ uint4 b4 = tg_bits16(TG_SP_LUT(s_crc), bb, 0); b4.w &= 0x0000ffffu; mine[2] = b4; }
				mine[3] = bits16(0);
			} else {
#pragma unroll
				for (int i = 0; i < 4; i++)
					mine[i] = bits16(4 * c - 3 + i);
			}
			__builtin_amdgcn_s_waitcnt(0xc07f);	/* lgkmcnt(0): single wave, LDS visible */
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t rr = (lane >> 2) + 16 * i;
				const uint4 v = *(const uint4 *)(stage + rr * TG_STAGE_PITCH + 4 * (lane & 3));
				uint4 *dst = (uint4 *)(rec + (size_t)st_slot[rr] * TG_REC_BYTES + 64 * c + 16 * (lane & 3));
				TG_REC_STORE_SEG(dst, v);
			}
			__builtin_amdgcn_wave_barrier();
		}
		if (!valid)
			return;
		if (wire) {
			uint32_t *wr = (uint32_t *)(wire + (size_t)slot * TG_WIRE_BYTES);
			uint32_t *wb = wr + TG_WIRE_W_BITS1;
			constexpr int NWD = (TYPE1 + 31) / 32;
#pragma unroll
			for (int q = 0; q < NWD - 1; q++)
				wb[q] = od[q];
			wb[NWD - 1] = (od[NWD - 1] & ((1u << (TYPE1 & 31)) - 1)) | (crc << TG_WIRE_SCHF_CRC_SHIFT);
			wr[0] = (meta & 0xff) | (((meta >> 8) & 0xff) << 8) | ((bb & 0x3fff) << 16);
		}
		return;
	}

	/* NORM_2 records of the 216 kernel (round 5, TG_REC_PAIR): the two blocks of a slot are neighbouring items -- lanes L and L + 1 of
	 * a wave, almost always -- and between them the two lanes own the whole 320-byte record.  Such a pair sends it out the way the
	 * SCH/F kernel does: five 64-byte segments through LDS, four lanes per segment, stored non-temporally (output-only lines that
	 * the next batch's front end would otherwise have to push out of the caches).  Lanes without their partner in the wave (the
	 * SB2 of a SYNC burst, a pair cut by a wave boundary, a block k_clean took) keep the 16-byte stores below. */
	bool staged = false;
	uint32_t st_bb = 0, st_nerr = 0;
	if (KIND == TG_KIND_216 && TG_REC_PAIR && stage && !block_mode && !wire_only) {
		const uint32_t lane = threadIdx.x & 63;
		const uint32_t v_ = valid ? 1u : 0u;
		/* (every lane takes part in every exchange: a lane that sits out a ds_bpermute hands its reader nothing defined) */
		const uint32_t key = slot << 2 | which << 1 | v_;
		const uint32_t key_n = __shfl_down(key, 1), key_p = __shfl_up(key, 1);
		const bool nxt_ok = lane < 63 && key_n == (slot << 2 | 3u);
		const bool prv_ok = lane > 0 && key_p == (slot << 2 | 1u);
		const bool isA = valid && which == 0 && nxt_ok, isB = valid && which == 1 && prv_ok;
		const unsigned long long mA = __ballot(isA);
		staged = isA || isB;
		if (mA) {	/* wave-uniform */
			const uint32_t npairs = (uint32_t)__builtin_popcountll(mA);
			const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mA, 0u));
			const uint32_t row = isA ? below : below - 1u;		/* (a B lane's partner is the lane below it) */
			uint32_t *st_slot = stage + 32 * TG_STAGE_PITCH;	/* the pairs' slot numbers */
			uint4 *mine = (uint4 *)(stage + (staged ? row : 0u) * TG_STAGE_PITCH);
			/* what the partner knows: the header needs both CRC words */
			const uint32_t crc_n = __shfl_down(crc, 1), ok_n = __shfl_down(crc_ok, 1);
			uint32_t meta = 0, code = 0;
			if (isA) {
				st_slot[row] = slot;
				meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
				uint32_t bbraw;
				if (HMODE == 2) {
					const uint32_t *sb = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + TG_SOFT_BBK / 4;
					bbraw = 0;
#pragma unroll
					for (int q = 0; q < 4; q++)
						bbraw |= ((((sb[q] >> 7) & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
				} else
					bbraw = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_BBK];
				st_bb = bbraw ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
				if (kflags & TGK_F_RM)
					st_bb = rm3014_correct(st_bb, st_nerr);
				code = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
			}
			auto bits16 = [&](int q) {	/* type-1 bits 16 q .. 16 q + 15, one per byte */
				return tg_bits16(TG_SP_LUT(s_crc), od[q >> 1], q & 1, q * 16 + 12 < TYPE1);
			};
#pragma unroll
			for (int c = 0; c < 5; c++) {
				if (isA) {
					if (c == 0) {
						mine[0] = make_uint4((meta & 0xffffu) | (crc_ok << 16) | (ok_n << 24), (crc & 0xffffu) | (crc_n << 16), code, slot);
						mine[1] = make_uint4(0u, 0u, 0u, st_nerr);
						{ uint4 b4 = tg_bits16(TG_SP_LUT(s_crc), st_bb, 0); b4.w &= 0x0000ffffu; mine[2] = b4; }
						mine[3] = bits16(0);
					} else if (c == 1) {
						mine[0] = bits16(1); mine[1] = bits16(2); mine[2] = bits16(3); mine[3] = bits16(4);
					} else if (c == 2) {
						mine[0] = bits16(5); mine[1] = bits16(6); mine[2] = bits16(7);
					}
				}
				if (isB) {
					if (c == 2) {
						mine[3] = bits16(0);
					} else if (c == 3) {
						mine[0] = bits16(1); mine[1] = bits16(2); mine[2] = bits16(3); mine[3] = bits16(4);
					} else if (c == 4) {
						mine[0] = bits16(5); mine[1] = bits16(6); mine[2] = bits16(7); mine[3] = make_uint4(0u, 0u, 0u, 0u);
					}
				}
				__builtin_amdgcn_s_waitcnt(0xc07f);	/* lgkmcnt(0): single wave, LDS visible */
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (int i = 0; i < 2; i++) {
					const uint32_t rr = (lane >> 2) + 16 * i;
					if (rr < npairs) {
						const uint4 v = *(const uint4 *)(stage + rr * TG_STAGE_PITCH + 4 * (lane & 3));
						uint4 *dst = (uint4 *)(rec + (size_t)st_slot[rr] * TG_REC_BYTES + 64 * c + 16 * (lane & 3));
						TG_REC_STORE_SEG(dst, v);
					}
				}
				__builtin_amdgcn_wave_barrier();
			}
		}
	}

	if (!valid)
		return;

	/* ---- outputs ---- */
	uint8_t *r = rec + (size_t)slot * TG_REC_BYTES;
	if (!wire_only && !staged) {
		uint4 *dst = (uint4 *)(r + (which ? TG_REC_BITS2 : TG_REC_BITS1));
		constexpr int NST = (TYPE1 + 15) / 16;
#pragma unroll
		for (int q = 0; q < NST; q++) {
			const uint32_t hw = od[q >> 1];
			const uint4 o = tg_bits16(TG_SP_LUT(s_crc), hw, q & 1, q * 16 + 12 < TYPE1);	/* TYPE1 = 12 mod 16 */
#ifdef TG_EXP_NOSTORE
			if (q == 0 || hw == 0x12345u)
#endif
			TG_REC_STORE(dst + q, o);
		}
	}
	if (!wire_only && !staged) {
		r[TG_REC_CRC_OK + which] = (uint8_t)crc_ok;
		*(uint16_t *)(r + TG_REC_CRC + 2 * which) = (uint16_t)crc;
	}

	/* optional bit-packed copy for transport (wave-uniform branch): tg_layout.h "Wire record".  The lanes of a slot
	 * write disjoint bytes: each block its payload words and its half (SCH/F: its field) of w[9], the primary lane
	 * the header word */
	uint32_t *wr = wire ? (uint32_t *)(wire + (size_t)slot * TG_WIRE_BYTES) : nullptr;
	if (wr) {
		uint32_t *wb = wr + (which ? TG_WIRE_W_BITS2 : TG_WIRE_W_BITS1);
		constexpr int NWD = (TYPE1 + 31) / 32;
		if (KIND == TG_KIND_432) {
#pragma unroll
			for (int q = 0; q < NWD - 1; q++)
				wb[q] = od[q];
			wb[NWD - 1] = (od[NWD - 1] & ((1u << (TYPE1 & 31)) - 1)) | (crc << TG_WIRE_SCHF_CRC_SHIFT);
		} else {
#pragma unroll
			for (int q = 0; q < NWD; q++)
				wb[q] = (q == NWD - 1) ? (od[q] & ((1u << (TYPE1 & 31)) - 1)) : od[q];
			((uint16_t *)(wr + TG_WIRE_W_CRC))[which] = (uint16_t)crc;
			if (KIND == TG_KIND_SB1) {	/* SB1 fills w[1..2]; w[3..4] are nobody else's */
				wr[3] = 0;
				wr[4] = 0;
			}
		}
	}

	if (KIND == TG_KIND_SB1) {
		/* SYNC PDU fields, lower_mac/tetra_lower_mac.c:284-297 */
		const uint32_t cc = FIELD_MSB(od, 4, 6), tn = FIELD_MSB(od, 10, 2) + 1;
		const uint32_t fn = FIELD_MSB(od, 12, 5), mn = FIELD_MSB(od, 17, 6);
		const uint32_t mcc = FIELD_MSB(od, 31, 10), mnc = FIELD_MSB(od, 41, 14);
		const uint32_t code = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
		if (!wire_only) {
			*(uint32_t *)(r + TG_REC_SBF0) = cc | (tn << 8) | (fn << 16) | (mn << 24);
			*(uint32_t *)(r + TG_REC_SBF1) = mcc | (mnc << 16);
			*(uint32_t *)(r + TG_REC_SBCODE) = code;
		}
		if (kflags & TGK_F_LOOKBACK) {
			/* device-walk batches (see k_lists2): sb_ok = one bit per grid slot "SB1 passed its CRC", sb_code = the slot's
			 * mask-table entry, masks = the batch's code table (open addressing, 0 = free: a code ends in binary 11) */
			/* one table access per DISTINCT code of the wave (a recording has one cell: every lane brings the same code, and
			 * a hundred thousand compare-and-swaps on one word would serialise): the first lane of each group looks its
			 * code up -- a plain read first, the atomic only while the slot reads free -- and hands the slot to the others */
			uint32_t *tbl = const_cast<uint32_t *>(masks);
			const bool live = valid && crc_ok;
			uint32_t myh = 0;
			unsigned long long todo = __ballot(live);
			while (todo) {
				const uint32_t l0 = (uint32_t)__builtin_ctzll(todo);
				const uint32_t c0 = __builtin_amdgcn_readlane(code, l0);
				uint32_t h = (c0 * 2654435761u) >> 20, probe = 0;
				if ((threadIdx.x & 63) == l0) {
					for (; probe < TG_LB_TBL; probe++) {
						uint32_t old = __hip_atomic_load(&tbl[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						if (old == 0u)
							old = atomicCAS(&tbl[h], 0u, c0);
						if (old == 0u || old == c0)
							break;
						h = (h + 1) & (TG_LB_TBL - 1);
					}
					if (probe == TG_LB_TBL) {	/* more codes than the table holds: the batch is handed to the host path */
						atomicOr(&tbl[TG_LB_TBL], 1u);
						h = 0;
					}
				}
				h = __builtin_amdgcn_readlane(h, l0);
				const bool mine = live && code == c0;
				if (mine)
					myh = h;
				todo &= ~__ballot(mine);
			}
			if (live) {
				sb_code[slot] = 1u + ((uint32_t)kflags >> 8) + myh;
				atomicOr(&sb_ok[slot >> 5], 1u << (slot & 31));
			}
		} else {
			sb_ok[idx] = crc_ok;
			sb_code[idx] = code;
		}
		if (block_mode && !wire_only) {	/* a block on its own: this lane also writes the header */
			r[TG_REC_TYPE] = (uint8_t)packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
			*(uint32_t *)(r + TG_REC_CODE) = 3u;
			*(uint32_t *)(r + TG_REC_SLOT) = slot;
		}
	} else if (block_mode) {
		if (wire_only)
			return;
		/* block mode (tgpu_plan_load_blocks): one block per record, no burst around it */
		const uint32_t meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
		r[TG_REC_TYPE] = (uint8_t)meta;
		r[TG_REC_FLAGS] = (uint8_t)(meta >> 8);
		*(uint32_t *)(r + TG_REC_CODE) = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
		*(uint32_t *)(r + TG_REC_SLOT) = slot;
	} else {
		/* BBK + header are written by the lane that owns the slot's "primary" block:
		 * SCH/F for NORM_1, BLK1 for NORM_2, SB2 for SYNC */
		const uint32_t meta = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_META];
		const uint32_t btype = meta & 0xff;
		const bool primary = (KIND == TG_KIND_432) || (btype == TG_BURST_SYNC ? which == 1 : which == 0);
		if (primary && staged) {	/* (the record went out in segments above: the wire record's header word is left) */
			if (wr)
				wr[0] = btype | (((meta >> 8) & 0xff) << 8) | ((st_bb & 0x3fff) << 16);
		} else if (primary) {
			uint32_t bbraw;
			if (HMODE == 2) {
				/* hard decision of the first 16 BBK soft values: bit = (value < 0) */
				const uint32_t *sb = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + TG_SOFT_BBK / 4;
				bbraw = 0;
#pragma unroll
				for (int q = 0; q < 4; q++)
					bbraw |= ((((sb[q] >> 7) & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
			} else
				bbraw = packed[(size_t)slot * TG_PACKED_WORDS + TG_PW_BBK];
			uint32_t bb = bbraw ^ masks[(size_t)midx * TG_MASK_WORDS + TG_MW_BBK];
			uint32_t nerr = 0;
			if (kflags & TGK_F_RM)		/* non-default: minimum-distance decoding of the (30,14) word first */
				bb = rm3014_correct(bb, nerr);
			if (!wire_only) {
				r[TG_REC_BBK_NERR] = (uint8_t)nerr;
				uint4 o;
				o.x = spread4(bb);
				o.y = spread4(bb >> 4);
				o.z = spread4(bb >> 8);
				o.w = spread4(bb >> 12) & 0x0000ffffu;	/* 14 type-1 bits (tetra_lower_mac.c:268-274) */
				*(uint4 *)(r + TG_REC_BBK) = o;
				r[TG_REC_TYPE] = (uint8_t)btype;
				r[TG_REC_FLAGS] = (uint8_t)(meta >> 8);
				*(uint32_t *)(r + TG_REC_CODE) = masks[(size_t)midx * TG_MASK_WORDS + TG_MW_CODE];
				*(uint32_t *)(r + TG_REC_SLOT) = slot;
			}
			if (wr)
				wr[0] = btype | (((meta >> 8) & 0xff) << 8) | ((bb & 0x3fff) << 16);
		}
	}
}

/*
 * k_clean<KIND>: optional pre-pass (tgpu_plan_set_fastpath).  A block whose received bits are exactly a code word
 * needs no trellis search: all four generators contain the term 1 and g1 is received at every step, so any other
 * path differs from the received word at the first step where its input differs -- the zero-distance path is the
 * unique minimum whatever the tie rule, and the decoder's answer is that path.  Its input bits follow from the g1
 * stream alone (G1 = 1 + D + D^4: u_k = r1_k ^ u_(k-1) ^ u_(k-4), start state 0), and it is the received word iff
 * the g2 bits it implies (G2 = 1 + D^2 + D^3 + D^4) equal the received ones.  Per 8-step block two table
 * look-ups in LDS do both (g_clean_lut).  Clean blocks are finished here (same vit_finish as the trellis kernels);
 * the others are appended to a list for k_vit, which then reads its item count from the device.
 * Results are identical with or without this pass.
 */
template <int KIND>
__global__ __launch_bounds__(256)
void k_clean(const uint32_t *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ packed,
	     const uint32_t *__restrict__ masks, const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec,
	     uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code, uint8_t *__restrict__ wire,
	     uint32_t *__restrict__ dirty_items, uint32_t *__restrict__ dirty_count, int kflags)
{
	constexpr int NBLK = vit_cfg<KIND>::NBLK;
	constexpr int NW = NBLK / 2;
	constexpr int NOD = (NBLK + 3) / 4;
	__shared__ uint16_t s_lut[8192];
	__shared__ __attribute__((aligned(16))) uint16_t s_crc[TG_CRC_WORDS + 32];	/* + the 16-dword spread table (tg_bits16) */
	for (int i = threadIdx.x; i < 8192 / 8; i += 256)
		((uint4 *)s_lut)[i] = ((const uint4 *)g_clean_lut)[i];
	for (int i = threadIdx.x; i < 256; i += 256) {
		s_crc[i] = c_tab.crc_lsb[i];
		s_crc[256 + i] = c_tab.crc_msb[i];
	}
	tg_sp_fill(s_crc, threadIdx.x);
	__syncthreads();
	const uint16_t *lutA = s_lut, *lutB = s_lut + 4096;
	const uint32_t lane = threadIdx.x & 63;

	for (uint32_t base = blockIdx.x * 256; base < nitems; base += gridDim.x * 256) {
		uint32_t idx = base + threadIdx.x;
		const bool valid = idx < nitems;
		if (!valid)
			idx = nitems - 1;
		uint32_t slot, which, item = items[idx];
		if (KIND == TG_KIND_216) {
			slot = item >> 1;
			which = item & 1;
		} else {
			slot = item;
			which = 0;
		}
		const uint32_t *pw = packed + (size_t)slot * TG_PACKED_WORDS + (which ? TG_PW_BLK2 : TG_PW_BLK1);
		const uint32_t midx = maskidx[slot];
		const uint32_t *mw = masks + (size_t)midx * TG_MASK_WORDS + vit_cfg<KIND>::MW;

		uint32_t od[NOD + 1];
#pragma unroll
		for (int i = 0; i <= NOD; i++)
			od[i] = 0;
		uint32_t state = 0, dirty = 0;
		uint32_t cur = pw[0] ^ mw[0];
		{	/* the four lead-in steps: six received bits, two g2 checks, four input bits */
			const uint32_t a = lutA[(cur >> 24) & 63];
			const uint32_t b = lutB[a & 0xff];
			dirty |= ((b >> 8) ^ (a >> 8)) & 3;
			od[0] = b & 15;
			state = tg_brev4(b & 15);
		}
#pragma unroll
		for (int d = 0; d < NW; d++) {
			if (d)
				cur = pw[d] ^ mw[d];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const bool last = (d == NW - 1) && h;		/* four steps + the flush steps */
				const uint32_t x = (h ? cur >> 12 : cur) & (last ? 0x3fu : 0xfffu);
				const uint32_t a = lutA[x];
				const uint32_t b = lutB[(state << 8) | (a & 0xff)];
				dirty |= ((b >> 8) ^ (a >> 8)) & (last ? 3u : 15u);
				const uint32_t u = b & (last ? 15u : 255u);
				constexpr int dummy = 0;
				(void)dummy;
				const int upos = 4 + 8 * (2 * d + h);
				od[upos >> 5] |= u << (upos & 31);
				if ((upos & 31) > 24 && !last)
					od[(upos >> 5) + 1] |= u >> (32 - (upos & 31));
				state = b >> 12;
			}
		}
		/* not a code word: hand the item to the trellis kernel (one atomic per wave) */
		const bool isdirty = valid && dirty != 0;
		const unsigned long long dm = __ballot(isdirty);
		if (dm) {
			uint32_t pos = 0;
			if (lane == (uint32_t)__builtin_ctzll(dm))
				pos = atomicAdd(dirty_count, (uint32_t)__builtin_popcountll(dm));
			pos = __shfl(pos, __builtin_ctzll(dm));
			if (isdirty)
				dirty_items[pos + __builtin_popcountll(dm & ((1ull << lane) - 1))] = item;
		}
		vit_finish<KIND, 1>(od, s_crc, valid && dirty == 0, slot, which, idx, midx, packed, masks, rec, sb_ok, sb_code, wire,
				    nullptr, kflags);
	}
}

/*
 * HMODE 1: survivor history in VGPRs -- chunks of 32 registers (8 blocks) written through
 *          the VGPR index mode (s_set_gpr_idx_on) with a wave-uniform block index, read back
 *          with static indices by the fully unrolled traceback.  No LDS for the trellis at
 *          all, so occupancy is set by registers: 2 waves/SIMD for SCH/F, 4 for the 216 blocks.
 * HMODE 2: soft input (BASELINE config 5): int8 soft values from k_front_soft's per-slot area instead of
 *          packed bits, 32-bit correlation metrics (tg_svit_*), history in VGPRs as in mode 1.
 */
template <int KIND, int HMODE>
/* (the SCH/F kernel at two waves per SIMD -- 189 VGPRs, no spills, the swapped table forms too -- runs 143-146 us against 138 at three) */
#ifndef TG_VIT_WAVES_216
#define TG_VIT_WAVES_216 4	/* waves per SIMD the half-slot kernels are compiled for */
#endif
__global__ __launch_bounds__(64, (HMODE == 2) ? (KIND == TG_KIND_SB1 ? 4 : 2) : (KIND == TG_KIND_432 ? 3 : TG_VIT_WAVES_216))
void k_vit(const uint32_t *__restrict__ items, uint32_t nitems,
	   const uint32_t *__restrict__ packed, const uint32_t *__restrict__ masks,
	   const uint32_t *__restrict__ maskidx, uint8_t *__restrict__ rec,
	   uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code, uint8_t *__restrict__ wire,
	   const uint32_t *__restrict__ softarea, int kflags, const uint32_t *__restrict__ nitems_dev)
{
	TG_TRACE_BEGIN;
	constexpr int NBLK = vit_cfg<KIND>::NBLK;
	constexpr int NW = NBLK / 2;			/* code words */
	if (nitems_dev) {	/* after k_clean: the list of blocks that still need the trellis was counted on the device */
		nitems = *nitems_dev;
		if (blockIdx.x * 64 >= nitems)
			return;
	}
	constexpr int NOD = (NBLK + 3) / 4;		/* dwords of decoded bits */
	constexpr int NCH = (NBLK + 7) / 8;		/* history chunks of 8 blocks */

	__shared__ __attribute__((aligned(16))) uint16_t s_crc[TG_CRC_WORDS + 32];	/* + the 16-dword spread table (tg_bits16) */
	/* record staging of the SCH/F kernel (vit_finish): 64 lanes x four dwordx4 at a pitch of 20 dwords + 64 slot numbers */
	__shared__ __attribute__((aligned(16))) uint32_t s_stage[(KIND == TG_KIND_432) ? 64 * TG_STAGE_PITCH + 64 : (KIND == TG_KIND_216 && TG_REC_PAIR) ? 32 * TG_STAGE_PITCH + 32 : 4];

	const uint32_t lane = threadIdx.x;
	for (int i = lane; i < 256; i += 64) {
		s_crc[i] = c_tab.crc_lsb[i];
		s_crc[256 + i] = c_tab.crc_msb[i];
	}
	tg_sp_fill(s_crc, lane);

	uint32_t idx = blockIdx.x * 64 + lane;
	const bool valid = idx < nitems;
	if (!valid)
		idx = nitems - 1;

	/* item: SB1 kernel -> position in the SYNC-slot list (items[] = slot ids);
	 *       216 kernel -> slot<<1 | which;  432 kernel -> slot id */
	uint32_t slot, which;
	if (KIND == TG_KIND_216) {
		const uint32_t it = items[idx];
		slot = it >> 1;
		which = it & 1;
	} else {
		slot = items[idx];
		which = 0;
	}

	const uint32_t *pw = packed + (size_t)slot * TG_PACKED_WORDS + (which ? TG_PW_BLK2 : TG_PW_BLK1);
	const uint32_t *mw;
	uint32_t midx = 0;
	if (KIND == TG_KIND_SB1) {
		mw = c_tab.sb1_mask;
	} else {
		midx = maskidx[slot];
		mw = masks + (size_t)midx * TG_MASK_WORDS + vit_cfg<KIND>::MW;
	}

	uint32_t od[NOD + 1];
#pragma unroll
	for (int i = 0; i <= NOD; i++)
		od[i] = 0;

	/* hard input: the block's descrambled code words are fetched in one burst (the 80-byte pitch means every lane
	 * touches its own cache lines; back to back they are fetched from HBM once) and parked in LDS, one column per
	 * lane; the trellis loop then has no global loads.  Reading them one per 16 steps instead refetched the same
	 * lines ~4x (FETCH_SIZE 183 MB for 42 MB of input on 500 k SCH/F blocks). */
	/* round 5, -DTG_CW_STAGED=1 (off): VERDICT r4 read the trellis kernels' FETCH_SIZE (2 x 70 MB each per million-slot batch) as
	 * packed slots coming in from memory more than once and asked for one coalesced fetch per wave.  Built here: the wave fetches its
	 * items' 16-byte pieces side by side -- five lanes cover an SCH/F slot's 80 bytes, three a half-slot block's 36 (pieces 0..2 or
	 * 2..4 of the slot) -- in 5 / 3 load instructions that use every byte they touch, parks them row-wise (a row per item), each
	 * lane folds its row's mask words in and the trellis reads its row (pitch 20 / 12 dwords).  Measured (tools/experiments/
	 * fetch_quick.sh, one box, both builds): FETCH_SIZE 70.28 -> 67.74 MB (SCH/F), 70.62 -> 70.62 MB (half slots), the kernels 2 and
	 * 4 us slower.  So the slots are NOT re-fetched; what the counter holds is (a) the whole 80 MB packed area per kernel -- SCH/F
	 * and half-slot bursts alternate, each kernel uses every other 80-byte slot and so every 128-byte line -- and (b) ~60 MB of line
	 * fills under the record stores: a 320-byte record is 2.5 lines, the half line it shares with a neighbour written by the OTHER
	 * kernel is filled before the 64-byte segment lands (500 k x 128 B = 64 MB).  Neither is this fetch's to fix (DESIGN.md s. 5). */
#ifndef TG_CW_STAGED
#define TG_CW_STAGED 0
#endif
	constexpr bool CWS = (HMODE != 2) && TG_CW_STAGED && (KIND == TG_KIND_432 || KIND == TG_KIND_216);
	constexpr int CW_P = (KIND == TG_KIND_432) ? 5 : 3;	/* 16-byte pieces per item */
	constexpr int CW_PITCH = 4 * CW_P;
	__shared__ __attribute__((aligned(16))) uint32_t s_cw[(HMODE != 2) ? (CWS ? 64 * CW_PITCH : NW * 64) : 4];
	const uint32_t cw_row = CWS ? lane * CW_PITCH + ((KIND == TG_KIND_216) ? which : 0u) : lane;	/* (BLK2 = words 9..17: one word into piece 2) */
	constexpr int CW_STEP = CWS ? 1 : 64;
	/* branch-metric table (vit_core.h, tg_bm_entry): six dwords per step pair and received triple */
	/* round 5, TG_ACS_D (default): the difference form of vit_core.h (tg_step_pair_d) -- the two-bit step of a pair with one add and
	 * one min per butterfly, the one-bit step repaying the bias: 40 packed operations per pair instead of 48; ten dwords per entry in
	 * three arrays.  -DTG_ACS_D=0 is the form of rounds 1-5 (A/B: tools/ab_lib.sh "" "-DTG_ACS_D=0"). */
#ifndef TG_ACS_D
#define TG_ACS_D 1
#endif
	constexpr bool ACSD = (HMODE != 2) && TG_ACS_D;
#ifndef TG_BMJ_MASK
#define TG_BMJ_MASK 0x4u	/* kinds whose blocks take their table entries one pair at a time: 432 */
#endif
	constexpr bool BMJ = (TG_BMJ_MASK >> KIND) & 1u;
	__shared__ __attribute__((aligned(16))) uint32_t s_bm[(HMODE != 2) ? (ACSD ? TG_BMD_WORDS : TG_BM_WORDS) : TG_PSOFT_TAB];
	/* (the entry's last two dwords -- P and P' with their halves swapped -- are taken where the kernel has registers to spare:
	 * 16 v_alignbit_b32 fewer per 16 steps; the SCH/F kernel sits at its 168 and would spill 41 of them) */
#ifndef TG_BM8_MASK
#define TG_BM8_MASK 0xbu	/* kinds SB1, 216, 168 */
#endif
	constexpr bool BM8 = (HMODE != 2) && ((TG_BM8_MASK >> KIND) & 1u);
	auto bm = [&](int p, uint32_t e, uint32_t w[8]) {
		const uint32_t *q = s_bm + (8 * p + e) * 8;
		const uint4 a = *(const uint4 *)q;
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
		if (BM8) {
			const uint4 b = *(const uint4 *)(q + 4);
			w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
		} else {
			const uint2 b = *(const uint2 *)(q + 4);
			w[4] = b.x; w[5] = b.y;
		}
	};
	auto bmd = [&](int p, uint32_t o, uint32_t w[10]) {	/* o = 16 x triple: the three reads differ in their immediate offsets only */
		const uint8_t *q = (const uint8_t *)s_bm + o;
		const uint4 a = *(const uint4 *)(q + 4 * TG_BMD_A0 + 128 * p);
		const uint4 b = *(const uint4 *)(q + 4 * TG_BMD_A1 + 128 * p);
		const uint2 c = *(const uint2 *)(q + 4 * TG_BMD_A2 + 128 * p);
		w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
		w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
		w[8] = c.x; w[9] = c.y;
	};
	tg_vit_state v;
	uint32_t cur = 0;
	if (HMODE != 2) {
		if (ACSD) {
			if (lane < 32) {
				uint32_t w[10];
				tg_bmd_entry(lane >> 3, lane & 7, w);
				tg_bmd_store(s_bm, (int)lane, w);
			}
		} else if (lane < 32)
			tg_bm_entry(lane >> 3, lane & 7, s_bm + 8 * lane);
		if (CWS) {
			const uint32_t myoff = slot * TG_PACKED_WORDS + ((KIND == TG_KIND_216 && which) ? 8u : 0u);
			uint4 t[CW_P];
#pragma unroll
			for (int k = 0; k < CW_P; k++) {
				const uint32_t i = lane + 64u * k, r = i / CW_P, pc = i - CW_P * r;
				const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * r), (int)myoff);
				t[k] = *(const uint4 *)(packed + (size_t)off + 4 * pc);
			}
#pragma unroll
			for (int k = 0; k < CW_P; k++)
				*(uint4 *)(s_cw + 4 * (lane + 64u * k)) = t[k];
			__syncthreads();
#pragma unroll
			for (int g = 0; g < NW; g++)
				s_cw[cw_row + g] ^= mw[g];
		} else {
#pragma unroll
			for (int g = 0; g < NW; g++)
				s_cw[g * 64 + lane] = pw[g] ^ mw[g];
		}
		__syncthreads();
		tg_vit_init(v);
		cur = s_cw[cw_row];
		if (ACSD)
			tg_vit_leadin_bmd(v, cur >> 24, bmd);
		else
			tg_vit_leadin_bm<BM8>(v, cur >> 24, bm);
	}

	if (HMODE == 2) {
		/* soft input: 6 dwords (2 x 12 int8 = 16 trellis steps) per iteration from this block's soft area; the packed
		 * 16-bit soft trellis of vit_core.h (tg_pvit_*), branch metrics from the 512-entry table in LDS */
		for (int i = lane; i < TG_PSOFT_TAB; i += 64)
			s_bm[i] = tg_psoft_entry(i);
		__syncthreads();
		auto tab = [&](uint32_t idx) { return s_bm[idx]; };
		const uint32_t *sw = softarea + (size_t)slot * (TG_SOFT_SLOT_BYTES / 4) + (which ? TG_SOFT_AREA2 / 4 : 0);
		/* the half-slot blocks: the wave's 64 soft areas (224 B each) come in ONCE, as 14 direct-to-LDS loads of 1 KB in
		 * which 14 neighbouring lanes cover one area (whole lines, each fetched one time), and the trellis reads its
		 * values from LDS.  Walking the areas 24 bytes per 16 steps per lane touched every line five times over ~20 us
		 * with a working set of the L2's size: 738 MB of traffic for 400 MB.  14 KB per wave: two waves per SIMD (what
		 * the SCH/F kernel runs with as well; the trellis has the instruction-level parallelism for it). */
		constexpr bool STAGED = (KIND == TG_KIND_216);
		constexpr int AREA_DW = (TG_SOFT_LEADIN_BYTES + TG_SOFT_BLOCK_BYTES * NBLK) / 4;	/* 56 */
		__shared__ __attribute__((aligned(16))) uint32_t s_soft[STAGED ? 64 * AREA_DW : 4];
		if (STAGED) {
			static_assert(!STAGED || AREA_DW % 4 == 0, "whole 16-byte pieces");
			constexpr int PIECES = AREA_DW / 4;	/* 14 */
			const uint32_t myoff = (uint32_t)(sw - softarea);
#pragma unroll
			for (int k = 0; k < PIECES; k++) {
				const uint32_t i = lane + 64u * k, r = i / PIECES, pc = i - PIECES * r;
				const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * r), (int)myoff);
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(softarea + (size_t)off + 4 * pc),
								 (__attribute__((address_space(3))) void *)(s_soft + 256 * k), 16, 0, 0);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		auto rd = [&](int i) { return STAGED ? s_soft[lane * AREA_DW + i] : sw[i]; };
		tg_pvit_state sv;
		tg_pvit_init(sv);
		{
			const uint32_t lw[2] = { rd(0), rd(1) };
			tg_pvit_leadin(sv, lw, (mw[0] >> 24) & 0x3f, tab);
		}
		/* software pipeline: the values (and mask word) of iteration g + 2 are loaded from global memory during
		 * iteration g; the table entries of a block are fetched from LDS while the block before it runs (the
		 * scheduling barriers keep the compiler from sinking the LDS reads to their first use) */
		uint32_t cw[6], nx[6], m = mw[0], nm = mw[NW > 1 ? 1 : 0];
#pragma unroll
		for (int q = 0; q < 6; q++) {
			cw[q] = rd(2 + q);
			nx[q] = rd(2 + (NW > 1 ? 6 : 0) + q);
		}
		uint32_t ta[12], tb[12];
		tg_psoft_fetch<0, 12>(cw, m & 0xfff, tab, ta);
		tg_v32 H[NCH];
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			const int nblk_c = (NBLK - 8 * c >= 8) ? 8 : (NBLK - 8 * c);
			const int nit = nblk_c / 2;
			const bool lastchunk = (c == NCH - 1);
			const int nloop = lastchunk ? nit - 1 : nit;
#pragma unroll 1
			for (int it = 0; it < nloop; it++) {
				const int g = 4 * c + it;
				const int g2 = (g + 2 < NW) ? g + 2 : NW - 1;
				uint32_t nn[6];
#pragma unroll
				for (int q = 0; q < 6; q++)
					nn[q] = rd(2 + 6 * g2 + q);
				const uint32_t nnm = mw[g2];
				tg_psoft_fetch<0, 12>(cw + 3, (m >> 12) & 0xfff, tab, tb);
				__builtin_amdgcn_sched_barrier(0);
				uint32_t h[4];
				tg_pvit_block<false>(sv, ta, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + d] = h[d];
				__builtin_amdgcn_sched_barrier(0);
				tg_psoft_fetch<0, 12>(nx, nm & 0xfff, tab, ta);
				__builtin_amdgcn_sched_barrier(0);
				tg_pvit_block<false>(sv, tb, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + 4 + d] = h[d];
				tg_vit_normalize(sv);	/* once per 16 steps: twelve metric bits stay exact (vit_core.h) */
#pragma unroll
				for (int q = 0; q < 6; q++) {
					cw[q] = nx[q];
					nx[q] = nn[q];
				}
				m = nm;
				nm = nnm;
			}
			if (lastchunk) {
				uint32_t h[4];
				tg_psoft_fetch<0, 6>(cw + 3, (m >> 12) & 0xfff, tab, tb);
				tg_pvit_block<false>(sv, ta, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + d] = h[d];
				tg_pvit_block<true>(sv, tb, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + 4 + d] = h[d];
			}
		}
		/* traceback from state 0, one nibble per four-step block */
		uint32_t pos = 0;
#pragma unroll
		for (int b = NBLK - 1; b >= 0; b--) {
			const int c = b >> 3, o = 4 * (b & 7);
			const uint32_t hi = tg_ptrace_hop(H[c][o + 2], H[c][o + 3], pos);
			const uint32_t lo = tg_ptrace_hop(H[c][o], H[c][o + 1], pos);
			od[b >> 2] |= (lo | (hi << 4)) << ((b & 3) * 8);
		}
	} else {
		tg_v32 H[NCH];
		tg_static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
			constexpr int c = decltype(cc)::value;
			constexpr int nblk_c = (NBLK - 8 * c >= 8) ? 8 : (NBLK - 8 * c);
			constexpr int nit = nblk_c / 2;
			constexpr bool lastchunk = (c == NCH - 1);
			constexpr int nloop = lastchunk ? nit - 1 : nit;
			/* TG_IT_UNROLL_MASK: kinds whose iteration loops are unrolled in full -- the history dwords then go to registers named at
			 * compile time (no VGPR index mode: 8 v_mov_b32 and 24 scalar instructions fewer per 16 steps) at 3.7 KB of code per iteration */
#ifndef TG_IT_UNROLL_MASK
#define TG_IT_UNROLL_MASK 0x0u
#endif
			constexpr int UNR = (((TG_IT_UNROLL_MASK >> KIND) & 1u) && nloop > 0) ? nloop : 1;
#pragma unroll UNR
			for (int it = 0; it < nloop; it++) {
				const int g = 4 * c + it;
				const uint32_t nxt = s_cw[cw_row + (g + 1) * CW_STEP];
				uint32_t h[4];
				if (ACSD)
					tg_vit_block_bmd<false, BMJ>(v, cur, h, bmd);
				else
					tg_vit_block_bm<false, BM8>(v, cur, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + d] = h[d];
				if (ACSD)
					tg_vit_block_bmd<false, BMJ>(v, cur >> 12, h, bmd);
				else
					tg_vit_block_bm<false, BM8>(v, cur >> 12, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * it + 4 + d] = h[d];
				if (KIND == TG_KIND_432 && g == 8)
					tg_vit_normalize_floor(v);
				cur = nxt;
			}
			if (lastchunk) {
				uint32_t h[4];
				if (ACSD)
					tg_vit_block_bmd<false, BMJ>(v, cur, h, bmd);
				else
					tg_vit_block_bm<false, BM8>(v, cur, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + d] = h[d];
				if (ACSD)
					tg_vit_block_bmd<true, BMJ>(v, cur >> 12, h, bmd);
				else
					tg_vit_block_bm<true, BM8>(v, cur >> 12, h, bm);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][8 * (nit - 1) + 4 + d] = h[d];
			}
		});
		/* block-wise traceback from state 0, all register indices static */
		uint32_t s = 0;
#pragma unroll
		for (int b = NBLK - 1; b >= 0; b--) {
			const int c = b >> 3, o = 4 * (b & 7);
			const uint32_t byte = hist_byte(H[c][o], H[c][o + 1], H[c][o + 2], H[c][o + 3], s);
			od[b >> 2] |= byte << ((b & 3) * 8);
			s = tg_brev4(byte);
		}
	}

	__syncthreads();	/* s_crc visible (single wave, but keep the compiler honest) */
	vit_finish<KIND, HMODE>(od, s_crc, valid, slot, which, idx, midx, packed, masks, rec, sb_ok, sb_code, wire, softarea, kflags,
				(KIND == TG_KIND_432 || (KIND == TG_KIND_216 && TG_REC_PAIR)) ? s_stage : nullptr);
	TG_TRACE_END(1u + (uint32_t)KIND, 8u);
}

/* ------------------------------------------------------------------------- */
/* k_burst: one workgroup per burst, the 16 states of a trellis across 16 lanes (small batches) */
/* ------------------------------------------------------------------------- */
/*
 * The kernel shape BASELINE.json's north star names: one workgroup per burst, its type-5 bits staged in LDS, the
 * 16-state add-compare-select as a butterfly ACROSS LANES.  It exists for small batches (the drop-in channel API
 * with a handful of bursts per flush), where the lane-per-trellis kernels leave 63 of 64 lanes idle and a flush is a
 * chain of eight launches: here a flush is two (k_burst<true> for the SB1 blocks of the SYNC slots, then k_burst<false>
 * for everything).
 *
 *   - descriptors and channels of the slot and the 255 before it + the channels' carry-in codes -> LDS in one parallel
 *     load (small batches keep them in mapped host memory: every dependent read would be a PCIe round trip), then the
 *     slot's 510 bytes -> LDS; every thread de-interleaves, de-punctures (2/3: the order of the bits is the type-3
 *     order) and descrambles its share of a block: r[i] = byte[(a (i + 1)) mod K] != 0, XOR bit (same position) of
 *     the scrambling sequence in its linear form (parity(code & lfsr_lin[pos]), lower_mac/tetra_scramb.c:34-50);
 *   - the scrambling code of the slot = the SYNC PDU of the latest SYNC slot at or before it (same channel) whose SB1
 *     passed its CRC, else the channel's carry-in (lower_mac/tetra_lower_mac.c:179-186, 291-300): pass 1 leaves
 *     (crc_ok, code) per SYNC slot, pass 2's workgroups look backwards through them -- no forward-fill launches;
 *   - trellis: a lane of a 16-lane row holds one state as metric << 8 | survivor byte (the lane-per-trellis word, one
 *     state per lane): same tie rule and register-exchange history as vit_core.h, so the 8-step blocks, the 16
 *     history bytes per block (one ds_write_b8 per lane) and the block-wise traceback are the same too.  The
 *     butterflies run in place with DPP partner exchanges and increments prepared by all threads (comment at s_inc
 *     below).  Row 0 of wave 0 decodes the slot's first block, row 1 the second, side by side;
 *   - CRC-16 as the linear map it is (c_tab.crc_lin), type-1 bits at one byte per bit, BBK, header: the record of the
 *     lane-per-trellis path, byte for byte (tests/test_gpu_parity.py::test_burst_kernel_equals_batch_kernels); the
 *     burst type is written last -- behind a system-wide fence when the owner polls it (marks).
 */
/* block parameters per kind on the device (tg_layout.h's host inlines: lower_mac/tetra_lower_mac.c:55-102) */
__device__ __forceinline__ uint32_t tgb_K(int kind)    { return kind == TG_KIND_SB1 ? 120u : kind == TG_KIND_216 ? 216u : 432u; }
__device__ __forceinline__ uint32_t tgb_a(int kind)    { return kind == TG_KIND_SB1 ? 11u : kind == TG_KIND_216 ? 101u : 103u; }
__device__ __forceinline__ uint32_t tgb_nblk(int kind) { return kind == TG_KIND_SB1 ? 10u : kind == TG_KIND_216 ? 18u : 36u; }
__device__ __forceinline__ uint32_t tgb_t1(int kind)   { return kind == TG_KIND_SB1 ? 60u : kind == TG_KIND_216 ? 124u : 268u; }

__device__ __forceinline__ uint32_t tgb_out_g12(uint32_t p, uint32_t u)
{
	/* (g1, g2) of the transition from state p with input u: out(j, 0) = {0,11,6,13,5,14,3,8} (lower_mac/viterbi_cch.c:35-40),
	 * g1 = bit 3, g2 = bit 2; complemented for u = 1 and for p >= 8 (every generator holds 1 and D^4) */
	const uint32_t tab = 0x83e5d6b0u;			/* nibble j = out(j, 0) */
	uint32_t o = (tab >> (4 * (p & 7))) & 15u;
	if (u)
		o ^= 15u;
	if (p & 8)
		o ^= 15u;
	return o >> 2;						/* bit 1 = g1, bit 0 = g2 */
}

#define TGB_MAX_STEPS (4 + 8 * 36)	/* the SCH/F trellis: 292 steps */

#ifdef TGB_TIMING	/* experiment build: phase time stamps of workgroup 0 (tools/flush_lat.c prints them) */
__device__ unsigned long long g_tgb_stamp[16];
#define TGB_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tgb_stamp[k] = wall_clock64(); } while (0)
extern "C" int tgk_burst_stamps(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tgb_stamp), sizeof(g_tgb_stamp));
}
#else
#define TGB_STAMP(k) do { } while (0)
#endif

/* the body: slot i of the batch by the calling workgroup (256 threads).  SB1_PASS is a constant at k_burst's two call sites;
 * the ring kernel below calls it with both values in turn (one copy of the LDS areas: the function's own) */
__device__ __forceinline__
void burst_body(const bool SB1_PASS, const uint32_t i, const uint8_t *__restrict__ stream, const uint64_t *__restrict__ slot_desc,
		const uint32_t *__restrict__ slot_chan, const uint32_t *__restrict__ chan_code, uint32_t nslots, uint32_t nchan,
		uint32_t *__restrict__ sb_ok, uint32_t *__restrict__ sb_code, uint8_t *__restrict__ rec, uint32_t *__restrict__ maskidx,
		uint32_t *__restrict__ masks, int marks)
{
	__shared__ uint32_t s_slot32[128];
	/* the descriptors and channels of this slot and the 255 before it, the channels' carry-in codes: small batches keep
	 * them in mapped host memory, where every dependent read is a PCIe round trip -- fetch them in one */
	__shared__ uint64_t s_desc[256];
	__shared__ uint32_t s_chan[256], s_ccode[64];
	__shared__ uint8_t s_r[2][432 + 16];		/* received type-3 bits per block, descrambled */
	__shared__ uint8_t s_hist[2][37][16];		/* [.][36]: spare row for a block that only runs along */
	__shared__ uint8_t s_od[2][40];			/* decoded bytes (8 bits per trellis block) */
	__shared__ uint32_t s_code, s_nonbin;
	uint8_t *s_slot = (uint8_t *)s_slot32;
	const uint32_t tid = threadIdx.x;
	const uint32_t back = i < 255u ? i : 255u;
	TGB_STAMP(0);
	if (tid <= back) {
		s_desc[tid] = slot_desc[i - tid];
		s_chan[tid] = slot_chan[i - tid];
	}
	if (tid < 64 && tid < nchan)
		s_ccode[tid] = chan_code[tid];
	__syncthreads();
	const uint64_t d = s_desc[0];
	TGB_STAMP(1);
	const uint32_t type = TG_DESC_TYPE(d);
	const uint8_t *base = stream + TG_DESC_OFF(d);
	uint8_t *r = rec + (size_t)i * TG_REC_BYTES;
	if (SB1_PASS && type != TG_BURST_SYNC)
		return;
	/* a burst type this path does not decode: no record, but the slot still has a code in force (the batch kernels'
	 * forward fill gives every slot one, and tgpu_plan_final_codes() reads the channel's last slot whatever its type) */
	const bool ignored = type != TG_BURST_SYNC && type != TG_BURST_NORM_1 && type != TG_BURST_NORM_2;
	if (tid == 0)
		s_nonbin = 0;
	__syncthreads();
	if (!ignored) {	/* the slot -> LDS (byte loads: any alignment), non-binary test */
		const uint32_t b0 = base[2 * tid < 510 ? 2 * tid : 509], b1 = base[2 * tid + 1 < 510 ? 2 * tid + 1 : 509];
		if (2 * tid < 510)
			s_slot[2 * tid] = (uint8_t)b0;
		if (2 * tid + 1 < 510)
			s_slot[2 * tid + 1] = (uint8_t)b1;
		if ((b0 | b1) > 1)
			s_nonbin = 1;
	}
	/* the code in force for this slot: look backwards through the SYNC slots of the batch (pass 1 left their results) */
	if (tid == 0) {
		const uint32_t ch = s_chan[0];
		uint32_t code = ch < 64 ? s_ccode[ch] : chan_code[ch];
		if (!SB1_PASS) {
			bool open = true;		/* still inside the channel's run and no good SYNC slot seen */
			for (uint32_t t = 0; t <= back && open; t++) {
				if (s_chan[t] != ch)
					open = false;
				else if (TG_DESC_TYPE(s_desc[t]) == TG_BURST_SYNC && sb_ok[i - t]) {
					code = sb_code[i - t];
					open = false;
				}
			}
			if (open)			/* (a run longer than the window: the rest from memory) */
				for (int j = (int)i - 256; j >= 0 && slot_chan[j] == ch; j--)
					if (TG_DESC_TYPE(slot_desc[j]) == TG_BURST_SYNC && sb_ok[j]) {
						code = sb_code[j];
						break;
					}
		}
		s_code = code;
		if (ignored) {
			r[TG_REC_TYPE] = TG_BURST_NONE;
			maskidx[i] = i;
			masks[(size_t)i * TG_MASK_WORDS + TG_MW_CODE] = code;
		}
	}
	if (ignored)		/* (workgroup-uniform; SB1_PASS never gets here with one) */
		return;
	__syncthreads();
	const uint32_t code = s_code;
	TGB_STAMP(2);

	/* blocks of this burst: kind and where its type-4 bits sit in the slot (phy/tetra_burst.c:31-47) */
	int kind[2] = { -1, -1 };
	uint32_t o1[2] = { 0, 0 }, o2[2] = { 0, 0 }, bcode[2] = { code, code };
	if (type == TG_BURST_SYNC) {
		kind[0] = TG_KIND_SB1; o1[0] = TG_SB_BLK1_OFF; bcode[0] = 3;		/* lower_mac/tetra_scramb.h:14 */
		if (!SB1_PASS) { kind[1] = TG_KIND_216; o1[1] = TG_SB_BLK2_OFF; }
	} else if (type == TG_BURST_NORM_2) {
		kind[0] = TG_KIND_216; o1[0] = TG_NDB_BLK1_OFF;
		kind[1] = TG_KIND_216; o1[1] = TG_NDB_BLK2_OFF;
	} else {
		kind[0] = TG_KIND_432; o1[0] = TG_NDB_BLK1_OFF; o2[0] = TG_NDB_BLK2_OFF;
	}
#pragma unroll
	for (int b = 0; b < 2; b++) {
		if (kind[b] < 0)
			continue;
		const uint32_t K = tgb_K(kind[b]), a = tgb_a(kind[b]);
		for (uint32_t t3 = tid; t3 < K; t3 += 256) {
			const uint32_t j = (a * (t3 + 1)) % K;		/* type3[i] = type4[(a (i + 1)) mod K] */
			const uint32_t byte = s_slot[j < 216 ? o1[b] + j : o2[b] + j - 216];
			s_r[b][t3] = (uint8_t)((byte != 0) ^ (__popc(bcode[b] & c_tab.lfsr_lin[j]) & 1));
		}
	}
	__syncthreads();

	/*
	 * The trellis: in-place butterflies across a 16-lane row, partner exchange with DPP row operations.
	 * Predecessors j and j + 8 (they differ in the oldest state bit) produce 2 j and 2 j + 1 (which differ in the
	 * newest): if the two lanes of such a pair swap words and each keeps the better candidate of its successor, no
	 * word ever has to travel further -- the lane <-> state map rotates by one bit per step instead (state of lane L
	 * before step k = rotl4(L, k mod 4); the pair's lanes differ in physical bit 3 - k mod 4) and is the identity
	 * again every four steps, in particular wherever history bytes are extracted.  The partner's word arrives with
	 * one DPP move (row_ror:8, quad permutes for bits 1 and 0) or two (row_shl:4 / row_shr:4 under bank masks for
	 * bit 2).  What a lane adds to its own and to its partner's word at step k -- branch metric of its successor from
	 * either predecessor, the tie / decision bit on the candidate from j + 8 -- does not depend on the metrics: all 256
	 * threads prepare these increments for the whole block up front (one 32-bit word per step and lane in LDS), and the
	 * serial loop is read, exchange, two adds, one min per step.
	 */
	__shared__ uint32_t s_inc[2][TGB_MAX_STEPS][16];
	TGB_STAMP(3);
	{
		const uint32_t L = tid & 15, kq = tid >> 4;		/* lane of the row; step index modulo 16 */
		const uint32_t c = kq & 3;				/* = k mod 4 for every step this thread prepares */
		const uint32_t sig = ((L << c) | (L >> (4 - c))) & 15;	/* the lane's state before such a step */
		const uint32_t pv = (L >> (3 - c)) & 1;		/* 0: it holds predecessor j and becomes 2 j; 1: j + 8 -> 2 j + 1 */
		const uint32_t e0 = tgb_out_g12(sig & 7, pv);		/* expected (g1, g2) from predecessor j; from j + 8: the complement */
		const bool odd = kq & 1;				/* one received bit (g1) instead of two */
#pragma unroll
		for (int b = 0; b < 2; b++) {
			if (kind[b] < 0)
				continue;
			const uint32_t nst = 4 + 8 * tgb_nblk(kind[b]);
			const uint8_t *rr = s_r[b];
			constexpr int NIT = (TGB_MAX_STEPS + 15) / 16;
			uint32_t ra[NIT], rb[NIT];		/* all the received bits first: one LDS latency, not one per step */
#pragma unroll
			for (int it = 0; it < NIT; it++) {
				const uint32_t k = kq + 16 * it;
				const uint32_t p3 = (k + 4 < nst) ? 3 * (k >> 1) : 0;	/* (the last four steps are the flush: nothing received) */
				ra[it] = rr[odd ? p3 + 2 : p3];
				rb[it] = rr[p3 + 1];
			}
#pragma unroll
			for (int it = 0; it < NIT; it++) {
				const uint32_t k = kq + 16 * it;
				if (k >= nst)
					break;
				const uint32_t tie = 1u << (k < 4 ? k : (k - 4) & 7);
				uint32_t d0 = 0, d1 = 0;
				if (k + 4 < nst) {
					if (odd) {
						d0 = ra[it] ^ (e0 >> 1);
						d1 = 1 - d0;
					} else {
						const uint32_t x0 = ((ra[it] << 1) | rb[it]) ^ e0;
						d0 = (x0 & 1) + (x0 >> 1);
						d1 = 2 - d0;
					}
				}
				/* low half: what the lane adds to its own word for its own successor; high half: what it adds
				 * to its own word for the PARTNER's successor (the other input bit: expected bits complemented, so
				 * the distances swap).  A word from predecessor j + 8 carries the tie / decision bit. */
				const uint32_t ca = d0 << 8, cb = d1 << 8;
				s_inc[b][k][L] = pv ? ((cb + tie) | ((ca + tie) << 16)) : (ca | (cb << 16));
			}
		}
	}
	__syncthreads();

	if (tid < 64) {		/* wave 0: row 0 = first block, row 1 = second */
		const uint32_t lane = tid;
		TGB_STAMP(4);
		const uint32_t row = lane >> 4, st = lane & 15;
		const int mykind = row < 2 ? kind[row] : -1;
		const uint32_t nblk = mykind >= 0 ? tgb_nblk(mykind) : 0;
		const uint32_t nblk_max = max(kind[0] >= 0 ? tgb_nblk(kind[0]) : 0u, kind[1] >= 0 ? tgb_nblk(kind[1]) : 0u);
		const uint32_t *inc = &s_inc[row & 1][0][st];
		uint32_t W = (st == 0) ? 0u : (1000u << 8);
#define TGB_ACS(C, w)												\
		{												\
			/* own candidate; the candidate for the partner, which the partner picks up with a DPP move */	\
			uint32_t x, g, P;									\
			asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"	\
			    : "=v"(x) : "v"(W), "v"(w));							\
			asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"	\
			    : "=v"(g) : "v"(W), "v"(w));							\
			if ((C) == 0)										\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x128, 0xf, 0xf, false);	/* row_ror:8 */	\
			else if ((C) == 1) {									\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x104, 0xf, 0x5, false);	/* row_shl:4, banks 0, 2 */	\
				P = (uint32_t)__builtin_amdgcn_update_dpp((int)P, (int)g, 0x114, 0xf, 0xa, false);	/* row_shr:4, banks 1, 3 */	\
			} else if ((C) == 2)									\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x4e, 0xf, 0xf, false);	/* quad_perm:[2,3,0,1] */	\
			else											\
				P = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0xb1, 0xf, 0xf, false);	/* quad_perm:[1,0,3,2] */	\
			W = x < P ? x : P;									\
		}
		{	/* four lead-in steps (type-3 bits 0..5) */
			const uint32_t w0 = inc[0], w1 = inc[16], w2 = inc[32], w3 = inc[48];
			TGB_ACS(0, w0) TGB_ACS(1, w1) TGB_ACS(2, w2) TGB_ACS(3, w3)
			W &= ~0xffu;
		}
		/* blocks of eight steps on twelve bits (vit_core.h), two per iteration with alternating register sets so that
		 * a block's increments are read from LDS while the block before it runs; a row with fewer blocks than the
		 * other runs along, its history bytes going to the spare row of s_hist */
		uint32_t wa[8], wb[8];
#define TGB_FETCH(dst, blk)											\
		{												\
			const uint32_t *q = inc + 16 * (4 + 8 * ((blk) < nblk ? (blk) : 0));			\
			_Pragma("unroll") for (int k = 0; k < 8; k++)						\
				dst[k] = q[16 * k];								\
		}
#define TGB_BLOCK(wv, blk)											\
		{												\
			TGB_ACS(0, wv[0]) TGB_ACS(1, wv[1]) TGB_ACS(2, wv[2]) TGB_ACS(3, wv[3])			\
			TGB_ACS(0, wv[4]) TGB_ACS(1, wv[5]) TGB_ACS(2, wv[6]) TGB_ACS(3, wv[7])			\
			s_hist[row & 1][(blk) < nblk ? (blk) : 36][st] = (uint8_t)W;				\
			W &= ~0xffu;										\
		}
		TGB_FETCH(wa, 0u)
		for (uint32_t b = 0; b < nblk_max; b += 2) {
			TGB_FETCH(wb, b + 1)
			TGB_BLOCK(wa, b)
			if (b + 1 >= nblk_max)
				break;
			TGB_FETCH(wa, b + 2)
			TGB_BLOCK(wb, b + 1)
		}
#undef TGB_FETCH
#undef TGB_BLOCK
#undef TGB_ACS
		TGB_STAMP(5);
		/* block-wise traceback from state 0 (row leaders) */
		if (st == 0 && mykind >= 0) {
			uint32_t sidx = 0;
			for (int b = (int)nblk - 1; b >= 0; b--) {
				const uint32_t byte = s_hist[row][b][sidx];
				s_od[row][b] = (uint8_t)byte;
				sidx = tg_brev4(byte);
			}
		}
	}
	__syncthreads();

	TGB_STAMP(6);
	/* CRC-16 per block (thread 0 / 1), then the record */
	__shared__ uint32_t s_crcv[2], s_okv[2];
	/* the CRC is linear in the decoded bits: every thread takes the bits i = tid and tid + 256 of both blocks, XORs
	 * their table vectors, the waves fold theirs (one memory latency + a reduction instead of 36 dependent look-ups) */
	__shared__ uint32_t s_cpart[4];
	{
		uint32_t v = 0;			/* block 0 in the low half, block 1 in the high half */
#pragma unroll
		for (int b = 0; b < 2; b++) {
			if (kind[b] < 0)
				continue;
			const uint32_t nbits = 8 * (tgb_nblk(kind[b]) - 1) + 4;
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const uint32_t ib = tid + 256 * h;
				if (ib < nbits && ((s_od[b][ib >> 3] >> (ib & 7)) & 1))
					v ^= (uint32_t)c_tab.crc_lin[kind[b]][ib] << (16 * b);
			}
		}
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1)
			v ^= (uint32_t)__shfl_xor((int)v, m, 64);
		if ((tid & 63) == 0)
			s_cpart[tid >> 6] = v;
	}
	__syncthreads();
	if (tid < 2 && kind[tid] >= 0) {
		const uint32_t all = s_cpart[0] ^ s_cpart[1] ^ s_cpart[2] ^ s_cpart[3];
		const uint32_t crc = ((all >> (16 * tid)) & 0xffff) ^ c_tab.crc_aff[kind[tid]];
		s_crcv[tid] = crc;
		s_okv[tid] = (crc == 0x1d0f);
	}
	__syncthreads();
	TGB_STAMP(7);
	if (SB1_PASS) {
		if (tid == 0) {
			const uint8_t *od = s_od[0];
			uint32_t w0 = od[0] | (od[1] << 8) | (od[2] << 16) | ((uint32_t)od[3] << 24);
			uint32_t w1 = od[4] | (od[5] << 8) | (od[6] << 16) | ((uint32_t)od[7] << 24);
			const uint32_t ow[3] = { w0, w1, 0 };
			const uint32_t cc = FIELD_MSB(ow, 4, 6), mcc = FIELD_MSB(ow, 31, 10), mnc = FIELD_MSB(ow, 41, 14);
			sb_ok[i] = s_okv[0];
			sb_code[i] = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
		}
		return;
	}
	/* type-1 bits at one byte per bit: block 0 at @48, block 1 at @176 */
#pragma unroll
	for (int b = 0; b < 2; b++) {
		if (kind[b] < 0)
			continue;
		const uint32_t n1 = tgb_t1(kind[b]);
		const uint32_t span = (n1 + 15) & ~15u;			/* the batch kernels store whole 16-byte groups, zero padded */
		uint8_t *dst = r + (b ? TG_REC_BITS2 : TG_REC_BITS1);
		for (uint32_t k = tid; k < span; k += 256)
			dst[k] = k < n1 ? (uint8_t)((s_od[b][k >> 3] >> (k & 7)) & 1) : 0;
	}
	if (tid < 32) {		/* BBK: 30 bits in stream order, descrambled, the first 14 kept (tetra_lower_mac.c:268-274) */
		uint8_t bit = 0;
		if (tid < 14) {
			const uint32_t pos = (type == TG_BURST_SYNC) ? TG_SB_BBK_OFF + tid : TG_NDB_BBK1_OFF + tid;
			bit = (uint8_t)((s_slot[pos] != 0) ^ (__popc(code & c_tab.lfsr_lin[tid]) & 1));
		}
		if (tid < 16)
			r[TG_REC_BBK + tid] = bit;
	}
	if (tid == 0) {
		r[TG_REC_FLAGS] = s_nonbin ? TG_FLAG_NONBINARY : 0;
		r[TG_REC_CRC_OK] = (uint8_t)s_okv[0];
		r[TG_REC_CRC_OK + 1] = kind[1] >= 0 ? (uint8_t)s_okv[1] : 0;
		*(uint16_t *)(r + TG_REC_CRC) = (uint16_t)s_crcv[0];
		*(uint16_t *)(r + TG_REC_CRC + 2) = kind[1] >= 0 ? (uint16_t)s_crcv[1] : 0;
		*(uint32_t *)(r + TG_REC_CODE) = code;
		*(uint32_t *)(r + TG_REC_SLOT) = i;
		r[TG_REC_BBK_NERR] = 0;
		maskidx[i] = i;		/* what tgpu_plan_final_codes() reads: the code in force at this slot */
		masks[(size_t)i * TG_MASK_WORDS + TG_MW_CODE] = code;
		if (type == TG_BURST_SYNC) {
			const uint8_t *od = s_od[0];
			uint32_t w0 = od[0] | (od[1] << 8) | (od[2] << 16) | ((uint32_t)od[3] << 24);
			uint32_t w1 = od[4] | (od[5] << 8) | (od[6] << 16) | ((uint32_t)od[7] << 24);
			const uint32_t ow[3] = { w0, w1, 0 };
			const uint32_t cc = FIELD_MSB(ow, 4, 6), tn = FIELD_MSB(ow, 10, 2) + 1;
			const uint32_t fn = FIELD_MSB(ow, 12, 5), mn = FIELD_MSB(ow, 17, 6);
			const uint32_t mcc = FIELD_MSB(ow, 31, 10), mnc = FIELD_MSB(ow, 41, 14);
			*(uint32_t *)(r + TG_REC_SBF0) = cc | (tn << 8) | (fn << 16) | (mn << 24);
			*(uint32_t *)(r + TG_REC_SBF1) = mcc | (mnc << 16);
			*(uint32_t *)(r + TG_REC_SBCODE) = (((mcc & 0x3ff) << 20) | ((mnc & 0x3fff) << 6) | (cc & 0x3f)) << 2 | 3u;
		}
	}
	/* the burst type is the record's completion mark: written last, after every thread's stores are visible system
	 * wide.  A host that holds the records in mapped memory (the channel API's small batches) presets the byte to
	 * TG_REC_PENDING and polls it instead of paying for a stream synchronise */
	if (marks)		/* (wave-uniform; on records in device memory the system-wide fence would only cost cache write-backs) */
		__threadfence_system();
	__syncthreads();
	if (tid == 0)
		*(volatile uint8_t *)(r + TG_REC_TYPE) = (uint8_t)type;
	TGB_STAMP(8);
}

template <bool SB1_PASS>
__global__ __launch_bounds__(256)
void k_burst(const uint8_t *__restrict__ stream, const uint64_t *__restrict__ slot_desc, const uint32_t *__restrict__ slot_chan,
	     const uint32_t *__restrict__ chan_code, uint32_t nslots, uint32_t nchan, uint32_t *__restrict__ sb_ok,
	     uint32_t *__restrict__ sb_code, uint8_t *__restrict__ rec, uint32_t *__restrict__ maskidx, uint32_t *__restrict__ masks,
	     int marks)
{
	burst_body(SB1_PASS, blockIdx.x, stream, slot_desc, slot_chan, chan_code, nslots, nchan, sb_ok, sb_code, rec, maskidx, masks, marks);
}

/*
 * k_burst_ring: the same decode by workgroups that STAY (TGPU_OPT_RING; the channel API's flushes of up to TG_RING_MAX
 * bursts).  A flush through k_burst is a kernel launch (7.5 us from the host's call to the first instruction's result back
 * on the host, tools/ubench/persist_rtt.hip) and two dependent reads over PCIe (descriptors, then the slot); a kernel that is
 * already running sees a request in mapped host memory after one PCIe read and answers in 2.9 us.
 *   - the request = one cache line the host fills (struct tg_ring_msg: descriptors, carry-in code, count) and numbers last;
 *     workgroup 0 polls it, takes the line in one 16-lane load (the number stands in both 32-byte halves: a half that shows
 *     the new number shows its new fields), hands it to the other workgroups through a box in device memory and all decode
 *     slot blockIdx.x: pass 1 (SB1 of SYNC slots), a barrier across the workgroups when more than one slot may need the
 *     result, pass 2; records and completion marks as k_burst writes them (the host polls the marks);
 *   - the workgroups leave on the host's stop request or when no request has come for idle_ticks of the 100 MHz clock
 *     (workgroup 0 decides and tells the others through the box; it clears `alive` last): a channel that falls silent frees
 *     its compute units, and the next flush starts the kernel again -- from `served`, so a request that was posted while the
 *     workgroups were leaving is not lost.
 */
__device__ __forceinline__ void ring_barrier(uint32_t *bar, uint32_t &epoch, uint32_t G)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		epoch++;
		__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * G)
			__builtin_amdgcn_s_sleep(1);
	}
	__syncthreads();
}

__global__ __launch_bounds__(256)
void k_burst_ring(tg_ring_msg *ring, tg_ring_box *box, const uint8_t *__restrict__ stream, uint32_t *__restrict__ sb_ok,
		  uint32_t *__restrict__ sb_code, uint8_t *__restrict__ rec, uint32_t *__restrict__ maskidx,
		  uint32_t *__restrict__ masks, uint32_t start_seq, uint32_t launch_no, unsigned long long idle_ticks)
{
	__shared__ uint32_t s_cmd[4];
	const uint32_t tid = threadIdx.x, w = blockIdx.x, G = gridDim.x;
	uint32_t last = start_seq, epoch = box->bar / G;	/* (the counter only grows: a new launch goes on where the last one stopped) */
	for (;;) {
		if (tid < 64) {
			uint32_t r = last, n = 0, hs = 0;
			if (w == 0) {
				const unsigned long long t0 = wall_clock64();
				for (;;) {
					/* the request line: lane l takes dword l */
					const uint32_t v = tid < 16 ? __hip_atomic_load((uint32_t *)ring + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
					const uint32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 15);
					if (r0 == r1 && r0 != last) {
						r = r0;
						n = __builtin_amdgcn_readlane(v, 1);
						hs = __builtin_amdgcn_readlane(v, 2);
						if (r != TG_RING_STOP) {
							if (tid >= 3 && tid < 12)	/* code, four descriptors */
								((uint32_t *)box)[tid] = v;
							if (tid == 1 || tid == 2)
								((uint32_t *)box)[tid] = v;
						}
						break;
					}
					if (wall_clock64() - t0 > idle_ticks) {
						r = TG_RING_STOP;
						break;
					}
				}
				TGB_STAMP(9);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				if (tid == 0) {		/* (leaving is told in a word of its own: the next launch must not find a stale "stop" in seq) */
					if (r == TG_RING_STOP)
						__hip_atomic_store(&box->stop, launch_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
					else
						__hip_atomic_store(&box->seq, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
				}
			} else {
				if (tid == 0)
					for (;;) {
						r = __hip_atomic_load(&box->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
						if (r != last)
							break;
						if (__hip_atomic_load(&box->stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == launch_no) {
							r = TG_RING_STOP;
							break;
						}
						__builtin_amdgcn_s_sleep(2);
					}
				r = __builtin_amdgcn_readfirstlane(r);
				n = box->n;
				hs = box->have_sync;
			}
			if (tid == 0) {
				s_cmd[0] = r;
				s_cmd[1] = n;
				s_cmd[2] = hs;
			}
		}
		__syncthreads();
		const uint32_t r = s_cmd[0], n = s_cmd[1], hs = s_cmd[2];
		if (r == TG_RING_STOP)
			break;
		/* (what the host and workgroup 0 wrote is read past this compute unit's vector cache) */
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		if (hs) {
			if (w < n)
				burst_body(true, w, stream, box->desc, box->chan, &box->code, n, 1u, sb_ok, sb_code, rec, maskidx, masks, 0);
			if (n > 1) {
				__threadfence();
				ring_barrier(&box->bar, epoch, G);
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			} else
				__syncthreads();
		}
		if (w < n)
			burst_body(false, w, stream, box->desc, box->chan, &box->code, n, 1u, sb_ok, sb_code, rec, maskidx, masks, 1);
		__syncthreads();
		last = r;
		if (w == 0 && tid == 0)
			__hip_atomic_store(&ring->served, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (w == 0 && tid == 0)
		__hip_atomic_store(&ring->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int tgk_burst_ring(struct tg_ring_msg *d_ring, struct tg_ring_box *d_box, const uint8_t *d_stream, uint32_t nwg,
			      uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_rec, uint32_t *d_maskidx, uint32_t *d_masks,
			      uint32_t start_seq, uint32_t launch_no, unsigned long long idle_ticks, void *stream)
{
	if (!nwg || nwg > TG_RING_MAX || !launch_no)
		return -1;
	hipLaunchKernelGGL(k_burst_ring, dim3(nwg), dim3(256), 0, (hipStream_t)stream, d_ring, d_box, d_stream, d_sb_ok, d_sb_code, d_rec,
			   d_maskidx, d_masks, start_seq, launch_no, idle_ticks);
	return (int)hipGetLastError();
}
extern "C" int tgk_burst(const uint8_t *d_stream, const uint64_t *d_slot_desc, const uint32_t *d_slot_chan,
			 const uint32_t *d_chan_code, uint32_t nslots, uint32_t nchan, int have_sync, uint32_t *d_sb_ok,
			 uint32_t *d_sb_code, uint8_t *d_rec, uint32_t *d_maskidx, uint32_t *d_masks, int marks, void *stream)
{
	if (!nslots)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	if (have_sync)
		hipLaunchKernelGGL((k_burst<true>), dim3(nslots), dim3(256), 0, s, d_stream, d_slot_desc, d_slot_chan, d_chan_code, nslots,
				   nchan, d_sb_ok, d_sb_code, d_rec, d_maskidx, d_masks, 0);
	hipLaunchKernelGGL((k_burst<false>), dim3(nslots), dim3(256), 0, s, d_stream, d_slot_desc, d_slot_chan, d_chan_code, nslots,
			   nchan, d_sb_ok, d_sb_code, d_rec, d_maskidx, d_masks, marks);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* generic trellis: any RCPC puncturer on either mother code  */
/* ------------------------------------------------------------------------- */
/*
 * k_conv<CODE, NCH, G3>: tetra_rcpc_depunct() + conv_cch_decode() / conv_tch_decode() for a batch of equally
 * shaped blocks (lower_mac/tetra_conv_enc.c:226-248, viterbi_cch.c:58-66, viterbi_tch.c:56-64).  One lane per
 * block, 64 blocks per wave.  The wave's 64 * type3_len received bytes (1 bit per byte, 0xff = erased, the
 * reference's depunct buffer convention) are one contiguous range: read with coalesced dwords, reduced to 2-bit
 * classes (0 bit / 1 bit / erased), four to an LDS byte.  The step program (tg_conv.h: which type-3 byte carries
 * g1 / g2 / g3 of each step) is uniform: it is fetched with scalar loads and all the index / shift / presence
 * arithmetic runs on the scalar unit.  Per step the vector unit does 3 x (address add, LDS byte read, 24-bit
 * multiply, mask), 10 packed adds for the four branch-metric pairs and their tie variants and 24 for the
 * add-compare-select (tg_step_gen).  History: 16 bytes per 8 steps in
 * VGPRs (NCH chunks of 32 registers, as k_vit), block-wise traceback, decoded bits transposed through the same
 * LDS range and written out as one contiguous range.  Metrics are renormalised every 64 steps.
 */
template <int CODE, int NCH, bool G3>
__global__ __launch_bounds__(64)
void k_conv(const uint8_t *__restrict__ type3, unsigned long long nblocks, uint32_t t3len, uint32_t L,
	    const uint32_t *__restrict__ steps, uint8_t *__restrict__ type2, uint32_t rawoff)
{
	extern __shared__ uint32_t s_dyn[];
	uint8_t *s_in = (uint8_t *)s_dyn;
	const uint32_t lane = threadIdx.x;
	const unsigned long long blk0 = (unsigned long long)blockIdx.x * 64ull;
	const uint32_t nvalid = (nblocks - blk0 < 64ull) ? (uint32_t)(nblocks - blk0) : 64u;
	const uint32_t nq = (t3len + 3) >> 2;		/* class bytes per block (LDS row pitch) */

	{	/* stage: four received bytes -> one class byte (tg_conv_pack4) */
		const uint8_t *src = type3 + blk0 * t3len;
		if ((t3len & 3) == 0 && ((uintptr_t)type3 & 3) == 0) {
			/* rows are whole dwords and blk0 * t3len is a multiple of 64: one flat, coalesced dword range */
			const uint32_t ndw = nvalid * nq;
			for (uint32_t w = lane; w < ndw; w += 64)
				s_in[w] = (uint8_t)tg_conv_pack4(((const uint32_t *)src)[w]);
		} else {
			/* odd row length or base: the raw bytes go to a second LDS range first (flat, dwords when the base
			 * allows), then (row, quad) pairs flat over the wave are reduced from there; the row index by a
			 * corrected float division (all values < 2^15) */
			uint8_t *raw = s_in + rawoff;
			const uint32_t nbytes = nvalid * t3len;
			uint32_t done = 0;
			if (((uintptr_t)type3 & 3) == 0) {
				const uint32_t ndw = nbytes >> 2;
				for (uint32_t w = lane; w < ndw; w += 64)
					((uint32_t *)raw)[w] = ((const uint32_t *)src)[w];
				done = ndw << 2;
			}
			for (uint32_t i = done + lane; i < nbytes; i += 64)
				raw[i] = src[i];
			__syncthreads();
			const uint32_t nitem = nvalid * nq;
			const float inv = 1.0f / (float)nq;
			for (uint32_t w = lane; w < nitem; w += 64) {
				uint32_t r = (uint32_t)(((float)w + 0.5f) * inv);
				r -= (r * nq > w);
				r += ((r + 1) * nq <= w);
				const uint32_t q = w - r * nq;
				const uint8_t *row = raw + r * t3len;
				/* one (unaligned) LDS dword read; bytes past the row (the raw range has 4 spare bytes) -> erased */
				uint32_t x;
				__builtin_memcpy(&x, row + 4 * q, 4);
				const uint32_t nv = t3len - 4 * q;
				x |= (nv < 4) ? (0xffffffffu << (8 * nv)) : 0u;
				s_in[w] = (uint8_t)tg_conv_pack4(x);
			}
		}
	}
	__syncthreads();

	const uint8_t *mine = s_in + (lane < nvalid ? lane : 0) * nq;
	auto fetch = [&](uint32_t q) -> uint32_t { return mine[q]; };

	const uint32_t nblk = (L + 7) >> 3;
	tg_vit_state v;
	uint32_t h[4];
	tg_vit_init(v);
	tg_conv_block<CODE, G3, 4>(v, steps, 4, fetch, h);
	tg_v32 H[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		if (8u * c < nblk) {
			if (c)
				tg_vit_normalize(v);
			const uint32_t nb = (nblk - 8u * c < 8u) ? nblk - 8u * c : 8u;
#pragma unroll 1
			for (uint32_t it = 0; it < nb; it++) {
				const uint32_t b = 8u * c + it;
				const uint32_t left = L - 8u * b;
				if (left >= 8)
					tg_conv_block<CODE, G3, 8>(v, steps + 3 * (4 + 8 * b), 8, fetch, h);
				else
					tg_conv_block<CODE, G3, 0>(v, steps + 3 * (4 + 8 * b), (int)left, fetch, h);
#pragma unroll
				for (int d = 0; d < 4; d++)
					H[c][4 * it + d] = h[d];
			}
		}
	}

	__syncthreads();	/* the received bytes are dead: the same LDS range now takes the decoded bits */
	uint8_t *outl = s_in + lane * L;
	const bool al = (L & 3) == 0;
	uint32_t s = 0;
#pragma unroll
	for (int b = 8 * NCH - 1; b >= 0; b--) {
		if ((uint32_t)b < nblk) {
			const int c = b >> 3, o = 4 * (b & 7);
			const uint32_t byte = hist_byte(H[c][o], H[c][o + 1], H[c][o + 2], H[c][o + 3], s);
			const uint32_t nbits = (L - 8u * b < 8u) ? L - 8u * b : 8u;
			if (al) {	/* L = 0 mod 4: a block holds 8 or 4 bits */
				*(uint32_t *)(outl + 8 * b) = spread4(byte);
				if (nbits > 4)
					*(uint32_t *)(outl + 8 * b + 4) = spread4(byte >> 4);
			} else {
				for (uint32_t i = 0; i < nbits; i++)
					outl[8 * b + i] = (uint8_t)((byte >> i) & 1);
			}
			s = tg_brev4(byte);
		}
	}
	__syncthreads();
	{
		uint8_t *dst = type2 + blk0 * L;
		const uint32_t nbytes = nvalid * L;
		uint32_t done = 0;
		if (((uintptr_t)type2 & 3) == 0) {
			const uint32_t ndw = nbytes >> 2;
			for (uint32_t w = lane; w < ndw; w += 64)
				((uint32_t *)dst)[w] = s_dyn[w];
			done = ndw << 2;
		}
		for (uint32_t q = done + lane; q < nbytes; q += 64)
			dst[q] = s_in[q];
	}
}


/* ------------------------------------------------------------------------- */
/* host-side launch layer of this unit                                        */
/* ------------------------------------------------------------------------- */
static void build_clean_lut(uint16_t *t)
{
	/* received bits of an 8-step block: a(g1,g2) b(g1) a b a b a b -> bits 0,1 | 2 | 3,4 | 5 | 6,7 | 8 | 9,10 | 11 */
	static const int g1pos[8] = { 0, 2, 3, 5, 6, 8, 9, 11 }, g2pos[4] = { 1, 4, 7, 10 };
	for (uint32_t x = 0; x < 4096; x++) {
		uint32_t g1 = 0, g2 = 0;
		for (int i = 0; i < 8; i++)
			g1 |= ((x >> g1pos[i]) & 1) << i;
		for (int i = 0; i < 4; i++)
			g2 |= ((x >> g2pos[i]) & 1) << i;
		t[x] = (uint16_t)(g1 | (g2 << 8));
	}
	for (uint32_t st = 0; st < 16; st++)
		for (uint32_t g1 = 0; g1 < 256; g1++) {
			uint32_t h = st;	/* last four input bits, newest in bit 0 */
			uint32_t u8 = 0, g2 = 0;
			for (int k = 0; k < 8; k++) {
				const uint32_t u = ((g1 >> k) ^ h ^ (h >> 3)) & 1;			/* G1 = 1 + D + D^4 */
				if (!(k & 1))
					g2 |= ((u ^ (h >> 1) ^ (h >> 2) ^ (h >> 3)) & 1) << (k >> 1);	/* G2 = 1 + D^2 + D^3 + D^4 */
				u8 |= u << k;
				h = ((h << 1) | u) & 15;
			}
			t[4096 + (st << 8 | g1)] = (uint16_t)(u8 | (g2 << 8) | (h << 12));
		}
}

extern "C" int tgk_upload_trellis(const tg_const_tables *host)
{
	static uint16_t lut[8192];
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), host, sizeof(*host)));
	build_clean_lut(lut);
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_clean_lut), lut, sizeof(lut)));
	return 0;
}

#ifdef TG_TRACE
extern "C" int tgk_trace_read_trellis(void *out, unsigned int *n, int reset)
{
	return tg_trace_read_unit(out, n, reset);
}
#endif

extern "C" int tgk_clean(int kind, const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed, const uint32_t *d_masks,
			 const uint32_t *d_maskidx, uint8_t *d_rec, uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_wire,
			 uint32_t *d_dirty_items, uint32_t *d_dirty_count, int flags, void *stream)
{
	if (!nitems)
		return 0;
	uint32_t blocks = (nitems + 255) / 256;
	if (blocks > 256 * 8)
		blocks = 256 * 8;
	hipStream_t s = (hipStream_t)stream;
	if (kind == TG_KIND_216)
		hipLaunchKernelGGL((k_clean<TG_KIND_216>), dim3(blocks), dim3(256), 0, s, d_items, nitems, d_packed, d_masks, d_maskidx, d_rec,
				   d_sb_ok, d_sb_code, d_wire, d_dirty_items, d_dirty_count, flags);
	else if (kind == TG_KIND_432)
		hipLaunchKernelGGL((k_clean<TG_KIND_432>), dim3(blocks), dim3(256), 0, s, d_items, nitems, d_packed, d_masks, d_maskidx, d_rec,
				   d_sb_ok, d_sb_code, d_wire, d_dirty_items, d_dirty_count, flags);
	else
		return -1;
	return (int)hipGetLastError();
}

extern "C" int tgk_rm_enable(const uint32_t *h_leader, const uint16_t *h_parity)
{
	static uint32_t *d_leader;	/* one table per process, never freed */
	if (!d_leader) {
		HIPCHK(hipMalloc((void **)&d_leader, 65536 * 4));
		HIPCHK(hipMemcpy(d_leader, h_leader, 65536 * 4, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_rm_leader), &d_leader, sizeof(d_leader)));
		HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_rm_parity), h_parity, 14 * 2));
	}
	return 0;
}

extern "C" int tgk_bbk_blocks(const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed, const uint32_t *d_masks,
			      const uint32_t *d_maskidx, uint8_t *d_rec, int flags, void *stream)
{
	if (!nitems)
		return 0;
	hipLaunchKernelGGL(k_bbk_blocks, dim3((nitems + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_items, nitems,
			   d_packed, d_masks, d_maskidx, d_rec, flags);
	return (int)hipGetLastError();
}

extern "C" int tgk_vit(int kind, const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed,
		       const uint32_t *d_masks, const uint32_t *d_maskidx, uint8_t *d_rec,
		       uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_wire, const uint32_t *d_soft, int flags,
		       const uint32_t *d_nitems /* NULL, or the device-side item count (nitems = its upper bound) */, void *stream)
{
	if (!nitems)
		return 0;
	const dim3 grid((nitems + 63) / 64), block(64);
	hipStream_t s = (hipStream_t)stream;
#define VIT_LAUNCH(K, H) hipLaunchKernelGGL((k_vit<K, H>), grid, block, 0, s, d_items, nitems, d_packed, d_masks, d_maskidx, d_rec, d_sb_ok, d_sb_code, d_wire, d_soft, flags, d_nitems)
	const int hm = d_soft ? 2 : 1;	/* survivor history in VGPRs (vit_core.h); 2 = the soft-input trellis */
	switch (kind) {
	case TG_KIND_SB1:
		if (hm == 2) VIT_LAUNCH(TG_KIND_SB1, 2); else VIT_LAUNCH(TG_KIND_SB1, 1);
		break;
	case TG_KIND_216:
		if (hm == 2) VIT_LAUNCH(TG_KIND_216, 2); else VIT_LAUNCH(TG_KIND_216, 1);
		break;
	case TG_KIND_432:
		if (hm == 2) VIT_LAUNCH(TG_KIND_432, 2); else VIT_LAUNCH(TG_KIND_432, 1);
		break;
	case TG_KIND_168:	/* hard input only */
		if (hm == 2) return -1; else VIT_LAUNCH(TG_KIND_168, 1);
		break;
	default:
		return -1;
	}
#undef VIT_LAUNCH
	return (int)hipGetLastError();
}

extern "C" int tgk_conv(int code, int g3, const uint8_t *d_type3, unsigned long long nblocks, uint32_t t3len, uint32_t L,
			const uint32_t *d_steps, uint8_t *d_type2, void *stream)
{
	if (!nblocks)
		return 0;
	const uint32_t nch = ((L + 7) / 8 + 7) / 8;
	const uint32_t span = (t3len + 3) / 4 > L ? (t3len + 3) / 4 : L;	/* class bytes in, decoded bits out */
	const uint32_t rawoff = (64 * span + 3) & ~3u;
	/* rows that are not whole dwords (or an odd base) are staged through a raw copy behind the working range */
	const bool odd = (t3len & 3) || ((uintptr_t)d_type3 & 3);
	const size_t lds = (size_t)rawoff + (odd ? ((((size_t)64 * t3len + 3) & ~(size_t)3) + 4) : 0);
	const unsigned long long nwg = (nblocks + 63) / 64;
	if (nch < 1 || nch > 8 || nwg > 0x7fffffffull || lds > 160 * 1024)
		return -1;
	hipStream_t s = (hipStream_t)stream;
	dim3 grid((unsigned)nwg), block(64);
#define CONV_LAUNCH(C, N, G) do {											\
		if (lds > 48 * 1024)										\
			HIPCHK(hipFuncSetAttribute((const void *)k_conv<C, N, G>,				\
						   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));	\
		hipLaunchKernelGGL((k_conv<C, N, G>), grid, block, lds, s, d_type3, nblocks, t3len, L, d_steps, d_type2, rawoff);	\
	} while (0)
#define CONV_CODE(C, G) switch (nch) {										\
	case 1: CONV_LAUNCH(C, 1, G); break; case 2: CONV_LAUNCH(C, 2, G); break; case 3: CONV_LAUNCH(C, 3, G); break;	\
	case 4: CONV_LAUNCH(C, 4, G); break; case 5: CONV_LAUNCH(C, 5, G); break; case 6: CONV_LAUNCH(C, 6, G); break;	\
	case 7: CONV_LAUNCH(C, 7, G); break; default: CONV_LAUNCH(C, 8, G); break; }
	if (code && g3)
		CONV_CODE(1, true)
	else if (code)
		CONV_CODE(1, false)
	else if (g3)
		CONV_CODE(0, true)
	else
		CONV_CODE(0, false)
#undef CONV_CODE
#undef CONV_LAUNCH
	return (int)hipGetLastError();
}

