/*
 * tg_gsmtap.c -- GSMTAP wire format of a decoded block: what the reference's upper MAC
 * sends for every CRC-OK block (tetra_upper_mac.c:483-488 -> tetra_gsmtap.c:31-63), so that the GPU path can
 * feed Wireshark-based tooling directly.  Host code, no libosmocore: the 16-byte struct gsmtap_hdr and the
 * constants are libosmocore's published gsmtap.h (GSMTAP v2).
 */
#include <string.h>

#include "tetra_gpu.h"
#include "tg_internal.h"

#define GSMTAP_VERSION        0x02
#define GSMTAP_TYPE_TETRA_I1  0x05

int tgpu_gsmtap_makemsg(const struct tetra_tdma_time *tm, enum tetra_log_chan lchan, uint8_t ts, uint8_t ss,
			int8_t signal_dbm, uint8_t snr, const uint8_t *bitdata, unsigned int bitlen,
			uint8_t *out, size_t out_size)
{
	/* tetra_gsmtap.c:19-29: GSMTAP_TETRA_* channel sub-types; everything else is 0 */
	static const uint8_t sub_type[] = {
		[TETRA_LC_SCH_F] = 0x05, [TETRA_LC_SCH_HD] = 0x04, [TETRA_LC_SCH_HU] = 0x03, [TETRA_LC_STCH] = 0x07,
		[TETRA_LC_AACH] = 0x02, [TETRA_LC_TCH] = 0x08, [TETRA_LC_BSCH] = 0x01, [TETRA_LC_BNCH] = 0x06,
	};
	const unsigned int packed_len = (bitlen + 7) / 8;		/* osmo_pbit_bytesize() */

	if (!tm || !out || (bitlen && !bitdata) || out_size < 16u + packed_len)
		return TGPU_EINVAL;
	const uint32_t fn = (((tm->hn * 60) + tm->mn) * 18) + tm->fn;	/* tetra_tdma_time2fn(), tetra_tdma.c:96-99 */
	memset(out, 0, 16u + packed_len);				/* msgb_alloc() hands out zeroed memory */
	out[0] = GSMTAP_VERSION;
	out[1] = 16 / 4;						/* hdr_len in 32-bit words */
	out[2] = GSMTAP_TYPE_TETRA_I1;
	out[3] = ts;							/* timeslot; arfcn (bytes 4..5) stays 0 */
	out[6] = (uint8_t)signal_dbm;
	out[7] = snr;
	out[8] = (uint8_t)(fn >> 24);					/* htonl(fn) */
	out[9] = (uint8_t)(fn >> 16);
	out[10] = (uint8_t)(fn >> 8);
	out[11] = (uint8_t)fn;
	out[12] = ((unsigned)lchan < sizeof(sub_type)) ? sub_type[lchan] : 0;
	out[13] = 0;							/* antenna_nr */
	out[14] = ss;							/* sub_slot; res (byte 15) stays 0 */
	for (unsigned int i = 0; i < bitlen; i++)			/* osmo_ubit2pbit(): MSB first */
		if (bitdata[i])
			out[16 + (i >> 3)] |= (uint8_t)(0x80u >> (i & 7));
	return (int)(16u + packed_len);
}

/* the batch form: one launch for all CRC-OK blocks of a decoded batch (k_gsmtap, tg_k_aux.hip) */
int tgpu_gsmtap_batch(struct tgpu_engine *eng, const uint8_t *d_rec, const struct tetra_tdma_time *d_times, const uint8_t *d_traffic,
		      uint32_t nslots, uint8_t *d_msgs, uint8_t *d_lens, void *hip_stream)
{
	if (!eng || (nslots && (!d_rec || !d_times || !d_msgs || !d_lens)))
		return TGPU_EINVAL;
	_Static_assert(sizeof(struct tetra_tdma_time) == sizeof(tg_tdma_time_dev), "time layout");
	int rc = tgpi_engine_bind(eng);
	if (rc)
		return rc;
	return tgk_gsmtap(d_rec, d_times, d_traffic, nslots, d_msgs, d_lens, hip_stream);
}
