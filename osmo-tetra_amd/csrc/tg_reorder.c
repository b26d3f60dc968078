/*
 * tg_reorder.c -- ACELP bit re-ordering (second half) as an OPERATION.
 *
 * lower_mac/tch_reordering.c:94-140 moves the decoded type-2 bits of a full-rate speech block (classes one after the
 * other, every position twice: frame 0, frame 1) to two consecutive codec frames and back.  Which position a bit goes
 * to is EN 300 395-2 Table 4 -- data, not code: the tables are the caller's (three lists of 1-based positions inside
 * a codec frame), this file turns them into an index map, the device applies the map to batches of blocks
 * (k_reorder) and the reference's two entry points exist under their own names for host buffers.
 *
 * Index arithmetic is the reference's, including what follows from a table that is not a permutation (its own is
 * not: one position twice, one entry 0): a later entry overwrites an earlier one, a destination no entry names is
 * left as the caller's buffer held it, and an entry 0 -- out[-1] / in[-1] in the reference -- is skipped.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tetra_gpu.h"
#include "tg_internal.h"

struct tgpu_reorder {
	struct tgpu_engine *eng;
	uint32_t nbits;
	int32_t *d_src;		/* per destination position: source position or -1 */
};

int tgpu_acelp_build_map(const uint8_t *const cls[3], const unsigned int ncls[3], int to_codec, int32_t *src_of_dst)
{
	if (!cls || !ncls || !src_of_dst)
		return TGPU_EINVAL;
	const int nbits = (int)(ncls[0] + ncls[1] + ncls[2]);
	if (nbits <= 0 || nbits > 4096)
		return TGPU_EINVAL;
	for (int i = 0; i < 2 * nbits; i++)
		src_of_dst[i] = -1;
	int cur = 0;	/* position in the class-ordered (type-2) block */
	for (int c = 0; c < 3; c++) {
		if (ncls[c] && !cls[c])
			return TGPU_EINVAL;
		for (unsigned int bit = 0; bit < ncls[c]; bit++)
			for (int frame = 0; frame < 2; frame++) {
				if ((int)cls[c][bit] > nbits)
					return TGPU_EINVAL;
				const int codec = frame * nbits + (int)cls[c][bit] - 1;
				const int t2 = cur + 2 * (int)bit + frame;
				if (codec < 0)
					continue;
				if (to_codec)
					src_of_dst[codec] = t2;		/* tch_reordering.c:102 */
				else
					src_of_dst[t2] = codec;		/* tch_reordering.c:127 */
			}
		cur += 2 * (int)ncls[c];
	}
	return 2 * nbits;
}

int tgpu_reorder_create(struct tgpu_engine *eng, const int32_t *src_of_dst, uint32_t nbits, struct tgpu_reorder **out)
{
	if (!eng || !src_of_dst || !nbits || nbits > 8192 || !out)
		return TGPU_EINVAL;
	*out = NULL;
	for (uint32_t i = 0; i < nbits; i++)
		if (src_of_dst[i] < -1 || src_of_dst[i] >= (int32_t)nbits)
			return TGPU_EINVAL;
	struct tgpu_reorder *r = calloc(1, sizeof(*r));
	if (!r)
		return TGPU_ENOMEM;
	r->eng = eng;
	r->nbits = nbits;
	int brc = tgpi_engine_bind(eng);
	if (brc) {
		free(r);
		return brc;
	}
	hipError_t e = hipMalloc((void **)&r->d_src, (size_t)nbits * 4);
	if (e == hipSuccess)
		e = hipMemcpy(r->d_src, src_of_dst, (size_t)nbits * 4, hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		tgpu_reorder_destroy(r);
		return (int)e;
	}
	*out = r;
	return TGPU_OK;
}

int tgpu_reorder_execute(struct tgpu_reorder *r, const uint8_t *d_in, uint64_t nblocks, uint8_t *d_out, void *hip_stream)
{
	if (!r || !d_in || !d_out)
		return TGPU_EINVAL;
	int brc = tgpi_engine_bind(r->eng);
	if (brc)
		return brc;
	return tgk_reorder(d_in, nblocks, r->nbits, r->d_src, d_out, hip_stream);
}

void tgpu_reorder_destroy(struct tgpu_reorder *r)
{
	if (!r)
		return;
	if (r->d_src)
		(void)hipFree(r->d_src);
	free(r);
}

/* ---- the reference's entry points on host buffers ------------------------------------------------------------- */
static int32_t *g_map[2];	/* [0]: codec -> class order, [1]: class order -> codec */
static int g_map_len;

int tgpu_acelp_set_tables(const uint8_t *const cls[3], const unsigned int ncls[3])
{
	if (!cls || !ncls)
		return TGPU_EINVAL;
	const size_t n = 2 * ((size_t)ncls[0] + ncls[1] + ncls[2]);
	int32_t *m0 = malloc(n * 4 + 4), *m1 = malloc(n * 4 + 4);
	if (!m0 || !m1) {
		free(m0);
		free(m1);
		return TGPU_ENOMEM;
	}
	const int a = tgpu_acelp_build_map(cls, ncls, 0, m0), b = tgpu_acelp_build_map(cls, ncls, 1, m1);
	if (a < 0 || b < 0) {
		free(m0);
		free(m1);
		return TGPU_EINVAL;
	}
	free(g_map[0]);
	free(g_map[1]);
	g_map[0] = m0;
	g_map[1] = m1;
	g_map_len = a;
	return TGPU_OK;
}

static void apply_map(const int32_t *map, const uint8_t *in, uint8_t *out, const char *who)
{
	if (!map) {	/* no silent pass-through: the tables are the caller's to supply */
		fprintf(stderr, "%s: no class position tables (tgpu_acelp_set_tables)\n", who);
		abort();
	}
	for (int i = 0; i < g_map_len; i++)
		if (map[i] >= 0)
			out[i] = in[map[i]];
}

void tetra_acelp_type2_to_codec(const uint8_t *in, uint8_t *out)
{
	apply_map(g_map[1], in, out, "tetra_acelp_type2_to_codec");
}

void tetra_acelp_codec_to_acelp(const uint8_t *in, uint8_t *out)
{
	apply_map(g_map[0], in, out, "tetra_acelp_codec_to_acelp");
}
