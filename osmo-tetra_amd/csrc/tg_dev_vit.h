/*
 * tg_dev_vit.h -- device helpers the trellis-side HIP units share (tg_k_trellis.hip: k_vit and friends; tg_k_slot.hip: the
 * lane-per-slot kernels of round 6): bit-field extraction, a loop the language unrolls, decoded bits -> bytes through a 16-entry
 * LDS table, a survivor-history byte, the records' store forms.
 */
#ifndef TG_DEV_VIT_H
#define TG_DEV_VIT_H

#include "tg_dev.h"
#include <utility>

/* MSB-first value of 'len' consecutive decoded bits starting at bit n0 (bits are held
 * LSB-first in od[]): the reference's bits_to_uint(type2 + n0, len), tetra_common.c:31-39 */
__device__ __forceinline__ uint32_t field_msb(uint32_t lo, uint32_t hi, int sh, int len)
{
	const unsigned long long two = (unsigned long long)lo | ((unsigned long long)hi << 32);
	const uint32_t f = (uint32_t)(two >> sh) & ((1u << len) - 1);
	return __builtin_bitreverse32(f) >> (32 - len);
}
#define FIELD_MSB(od, n0, len) field_msb((od)[(n0) >> 5], (od)[((n0) >> 5) + 1], (n0) & 31, (len))

typedef uint32_t tg_v32 __attribute__((ext_vector_type(32)));
typedef uint32_t tg_v16 __attribute__((ext_vector_type(16)));

/* f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop over the history chunks that is unrolled by the
 * language, not by a pass with a size threshold (the packed operations of vit_core.h's difference form are inline assembly, which
 * the unroller prices like calls: "#pragma unroll" over the chunks gave up, and a chunk index that is not a constant puts the
 * whole survivor history into scratch memory) */
template <typename F, int... I>
__device__ __forceinline__ void tg_static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
	(f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void tg_static_for(F &&f)
{
	tg_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define TG_STAGE_PITCH 20	/* dwords per lane in the record staging area: 16 + 4 (dwordx4 rows of neighbouring lanes in different banks) */

/* sixteen decoded bits (half h of W) as sixteen bytes of 0 / 1, from a 16-entry table in LDS (nibble -> dword, behind the CRC tables:
 * TG_SP_LUT) instead of a multiply and two masks per nibble: the table index times four is bits 2..5 of a byte of W << 2 (even
 * nibbles) or of W >> 2 (odd nibbles) -- one masked byte select each */
#define TG_CRC_WORDS 512	/* uint16_t entries of the two CRC tables in front of the spread table */
#define TG_SP_LUT(s_crc) ((const uint32_t *)((s_crc) + TG_CRC_WORDS))
__device__ __forceinline__ uint32_t tg_sp_at(const uint32_t *lut, uint32_t byteoff)
{
	return *(const uint32_t *)((const uint8_t *)lut + byteoff);
}
__device__ __forceinline__ uint4 tg_bits16(const uint32_t *lut, uint32_t W, int h, bool with_last = true)
{
	const uint32_t W2 = W << 2, W6 = W >> 2;
	uint4 o;
	o.x = tg_sp_at(lut, (W2 >> (16 * h)) & 0x3cu);
	o.y = tg_sp_at(lut, (W6 >> (16 * h)) & 0x3cu);
	o.z = tg_sp_at(lut, (W2 >> (16 * h + 8)) & 0x3cu);
	o.w = with_last ? tg_sp_at(lut, (W6 >> (16 * h + 8)) & 0x3cu) : 0u;
	return o;
}
/* the table's fill (lanes 0..15 of a wave) */
__device__ __forceinline__ void tg_sp_fill(uint16_t *s_crc, uint32_t lane)
{
	if (lane < 16)
		((uint32_t *)(s_crc + TG_CRC_WORDS))[lane] = spread4(lane);
}

/* byte 's' (0..15) of the 16 history bytes held in four dwords: two v_perm_b32 + one select */
__device__ __forceinline__ uint32_t hist_byte(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t s)
{
	const uint32_t sel = s & 7;
	const uint32_t lo = __builtin_amdgcn_perm(w1, w0, sel);
	const uint32_t hi = __builtin_amdgcn_perm(w3, w2, sel);
	return ((s & 8) ? hi : lo) & 0xff;
}

/* the records' type-1 bits are output only: nobody on the device reads them again.  TG_REC_NT = 1 stores them non-temporally
 * (A/B builds; round 5: see DESIGN.md section 4) */
#ifndef TG_REC_NT
#define TG_REC_NT 1	/* bit 0: the SCH/F kernel's staged 64-byte segments (default), bit 1: the 16-byte stores of the other kernels */
#endif
#ifndef TG_REC_PAIR
#define TG_REC_PAIR 0	/* 1: the 216 kernel sends the records of NORM_2 slots whose two blocks sit in one wave out in 64-byte segments (measured: front end -7 us, this kernel +6 us per step -- off) */
#endif
typedef uint32_t tg_u4v __attribute__((ext_vector_type(4)));
#define TG_REC_STORE_NT(P, V) __builtin_nontemporal_store(tg_u4v{ (V).x, (V).y, (V).z, (V).w }, (tg_u4v *)(P))
#define TG_REC_STORE_PLAIN(P, V) (*(P) = (V))
#if TG_REC_NT & 1	/* the SCH/F kernel's staged stores: whole 64-byte segments */
#define TG_REC_STORE_SEG TG_REC_STORE_NT
#else
#define TG_REC_STORE_SEG TG_REC_STORE_PLAIN
#endif
#if TG_REC_NT & 2	/* the other kernels' stores: 16 bytes per lane, every lane in a record of its own */
#define TG_REC_STORE TG_REC_STORE_NT
#else
#define TG_REC_STORE TG_REC_STORE_PLAIN
#endif

#endif
