/*
 * tg_conv.c -- the remaining RCPC puncturers and the speech trellis.
 *
 * Host side of the generic trellis kernel (k_conv): a decoder object per block shape (the step program of
 * tg_conv.h lives in device memory), plus the reference's two puncturing entry points under their own names
 * for callers that keep using them on host buffers (lower_mac/tetra_conv_enc.c:201-248).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include "tetra_gpu.h"
#include "tg_conv.h"
#include "tg_internal.h"

struct tgpu_conv {
	struct tgpu_engine *eng;
	int code;			/* 0: rate-1/4 CCH code, 1: rate-1/3 speech code */
	int g3;				/* some step receives g3 */
	uint32_t type3_len, type2_len;
	uint32_t *d_steps;
};

int tgpu_conv_create(struct tgpu_engine *eng, int punct, int mother_rate, uint32_t type3_len, uint32_t type2_len,
		     struct tgpu_conv **out)
{
	uint32_t steps[(TG_CONV_MAX_T2 + 4) * TG_CONV_DESC_WORDS];

	if (!eng || !out)
		return TGPU_EINVAL;
	*out = NULL;
	if (tg_conv_build_steps(punct, mother_rate, type3_len, type2_len, steps))
		return TGPU_EINVAL;
	int brc = tgpi_engine_bind(eng);
	if (brc)
		return brc;
	struct tgpu_conv *cv = calloc(1, sizeof(*cv));
	if (!cv)
		return TGPU_ENOMEM;
	cv->eng = eng;
	cv->code = (mother_rate == 3);
	cv->g3 = tg_conv_uses_g3(steps, type2_len);
	cv->type3_len = type3_len;
	cv->type2_len = type2_len;
	const size_t bytes = (size_t)(type2_len + 4) * TG_CONV_DESC_WORDS * sizeof(uint32_t);
	if (hipMalloc((void **)&cv->d_steps, bytes) != hipSuccess) {
		free(cv);
		return TGPU_ENOMEM;
	}
	if (hipMemcpy(cv->d_steps, steps, bytes, hipMemcpyHostToDevice) != hipSuccess) {
		(void)hipFree(cv->d_steps);
		free(cv);
		return TGPU_ENODEV;
	}
	*out = cv;
	return TGPU_OK;
}

int tgpu_conv_execute(struct tgpu_conv *cv, const void *d_type3, uint64_t nblocks, void *d_type2, void *hip_stream)
{
	if (!cv || (nblocks && (!d_type3 || !d_type2)))
		return TGPU_EINVAL;
	int brc = tgpi_engine_bind(cv->eng);
	if (brc)
		return brc;
	return tgk_conv(cv->code, cv->g3, (const uint8_t *)d_type3, nblocks, cv->type3_len, cv->type2_len, cv->d_steps,
			(uint8_t *)d_type2, hip_stream);
}

void tgpu_conv_destroy(struct tgpu_conv *cv)
{
	if (!cv)
		return;
	(void)hipFree(cv->d_steps);
	free(cv);
}

/* lower_mac/tetra_conv_enc.c:201-223: out[j-1] = in[k(j)-1], -EINVAL for an unknown puncturer */
int get_punctured_rate(int pu, uint8_t *in, int len, uint8_t *out)
{
	if (pu < 0 || pu >= TG_CONV_NPUNCT)
		return -EINVAL;
	for (int j = 1; j <= len; j++)
		out[j - 1] = in[tg_conv_mother_pos(pu, (uint32_t)j) - 1];
	return 0;
}

/* lower_mac/tetra_conv_enc.c:226-248: out[k(j)-1] = in[j-1]; the caller pre-fills out (0xff = erased) */
int tetra_rcpc_depunct(int pu, const uint8_t *in, int len, uint8_t *out)
{
	if (pu < 0 || pu >= TG_CONV_NPUNCT)
		return -EINVAL;
	for (int j = 1; j <= len; j++)
		out[tg_conv_mother_pos(pu, (uint32_t)j) - 1] = in[j - 1];
	return 0;
}
