/*
 * tg_stages.c -- the lower MAC's intermediate bit strings (type 5 -> 4 -> 3 -> 3dp -> 2), what the reference prints
 * with DEBUGP (lower_mac/tetra_lower_mac.c:175, :188, :246, :251, :254) and the product kernels never materialise:
 * there the descrambler is a mask, the de-interleaver and the de-puncturer are a gather order, and the trellis reads
 * code words.  tgpu_stages_* runs the steps one after the other on the device for a batch of blocks of one type
 * (k_stages, the generic trellis of tgpu_conv_*, k_stages_crc): an instrument for looking inside a block, and a
 * second formulation of the same chain that the GPU suite holds the fused kernels -- and the oracle, step by step --
 * against.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdlib.h>

#include "tetra_gpu.h"
#include "tg_internal.h"

struct tgpu_stages {
	struct tgpu_engine *eng;
	enum tp_sap_data_type type;
	uint32_t K, a, type2_len, type1_len;
	struct tgpu_conv *cv;
};

/* block parameters: lower_mac/tetra_lower_mac.c:60-118 (tetra_blk_param) */
static int stage_params(enum tp_sap_data_type t, uint32_t *K, uint32_t *a, uint32_t *t2, uint32_t *t1)
{
	switch (t) {
	case TPSAP_T_SB1:    *K = 120; *a = 11;  *t2 = 80;  *t1 = 60;  return 0;
	case TPSAP_T_SB2:
	case TPSAP_T_NDB:    *K = 216; *a = 101; *t2 = 144; *t1 = 124; return 0;
	case TPSAP_T_SCH_HU: *K = 168; *a = 13;  *t2 = 112; *t1 = 92;  return 0;
	case TPSAP_T_SCH_F:  *K = 432; *a = 103; *t2 = 288; *t1 = 268; return 0;
	case TPSAP_T_BBK:    *K = 30;  *a = 0;   *t2 = 14;  *t1 = 14;  return 0;
	default:             return -1;
	}
}

int tgpu_stages_create(struct tgpu_engine *eng, enum tp_sap_data_type type, struct tgpu_stages **out)
{
	if (!eng || !out)
		return TGPU_EINVAL;
	*out = NULL;
	struct tgpu_stages *st = calloc(1, sizeof(*st));
	if (!st)
		return TGPU_ENOMEM;
	if (stage_params(type, &st->K, &st->a, &st->type2_len, &st->type1_len)) {
		free(st);
		return TGPU_EINVAL;
	}
	st->eng = eng;
	st->type = type;
	if (st->a) {
		int rc = tgpu_conv_create(eng, 0 /* the 2/3 puncturer, tetra_conv_enc.c:96-112 */, 4, st->K, st->type2_len, &st->cv);
		if (rc) {
			free(st);
			return rc;
		}
	}
	*out = st;
	return TGPU_OK;
}

void tgpu_stages_destroy(struct tgpu_stages *st)
{
	if (!st)
		return;
	if (st->cv)
		tgpu_conv_destroy(st->cv);
	free(st);
}

int tgpu_stages_lengths(const struct tgpu_stages *st, uint32_t *type345_len, uint32_t *mother_len, uint32_t *type2_len, uint32_t *type1_len)
{
	if (!st)
		return TGPU_EINVAL;
	if (type345_len)
		*type345_len = st->K;
	if (mother_len)
		*mother_len = st->a ? 4 * st->type2_len : 0;
	if (type2_len)
		*type2_len = st->type2_len;
	if (type1_len)
		*type1_len = st->type1_len;
	return TGPU_OK;
}

int tgpu_stages_execute(struct tgpu_stages *st, const uint8_t *d_type5, const uint32_t *d_codes, uint64_t nblocks, uint8_t *d_type4,
			uint8_t *d_type3, uint8_t *d_type3dp, uint8_t *d_type2, uint16_t *d_crc, void *hip_stream)
{
	if (!st || !d_type5 || !d_type4 || !d_type2 || (st->a && (!d_type3 || !d_type3dp)))
		return TGPU_EINVAL;
	if (st->type != TPSAP_T_SB1 && !d_codes)
		return TGPU_EINVAL;
	if (!nblocks)
		return TGPU_OK;
	int rc = tgpi_engine_bind(st->eng);
	if (rc)
		return rc;
	/* SB1 is always scrambled with the fixed code 3 (lower_mac/tetra_scramb.h:14, tetra_lower_mac.c:180-182) */
	rc = tgk_stages(d_type5, st->type == TPSAP_T_SB1 ? NULL : d_codes, 3, nblocks, st->K, st->a, 4 * st->type2_len, d_type4, d_type3,
			d_type3dp, hip_stream);
	if (rc)
		return rc;
	if (!st->a)	/* BBK: the first 14 descrambled bits are the block (tetra_lower_mac.c:268-274) */
		return (int)hipMemcpy2DAsync(d_type2, st->type2_len, d_type4, st->K, st->type2_len, (size_t)nblocks, hipMemcpyDeviceToDevice,
					     (hipStream_t)hip_stream);
	rc = tgpu_conv_execute(st->cv, d_type3, nblocks, d_type2, hip_stream);
	if (rc || !d_crc)
		return rc;
	return tgk_stages_crc(d_type2, nblocks, st->type2_len, st->type1_len + 16, d_crc, hip_stream);
}
