/*
 * tg_pack.c -- packed ingest (optional): a capture in host memory, one bit per byte (the tetra-rx input format,
 * src/tetra-rx.c:83-94; what float_to_bits writes, src/float_to_bits.c:50-72), packed to one bit per BIT on the host so
 * that an eighth of the bytes cross PCIe.  The stream front end's first step on the device is this very packing
 * (k_front_stream: bytes -> bits -> LDS); tgpu_sync_multi_launch_packed() starts behind it.  The API's input format stays
 * one bit per byte: this is a host-side transport form, equivalent whenever every byte is 0 or 1 (the function says so).
 *
 * Bit i of packed byte k = bytes[8 k + i] & 1 (LSB first: the order the kernels' own bit string has).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "tetra_gpu.h"

struct pack_job {
	const uint8_t *in;
	uint8_t *out;
	uint64_t nbytes;	/* input bytes of this job: a multiple of 8 except for the last job */
	uint64_t nonbinary;
};

static uint64_t pack_scalar(const uint8_t *in, uint8_t *out, uint64_t n)
{
	uint64_t bad = 0, i = 0;
	for (; i + 8 <= n; i += 8) {
		uint64_t v;
		memcpy(&v, in + i, 8);
		bad += (v & 0xfefefefefefefefeull) != 0;
		/* the eight LSBs to the top byte: bit 8 k moves to bit 56 + k */
		out[i >> 3] = (uint8_t)(((v & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
	}
	if (i < n) {
		uint8_t b = 0;
		for (uint64_t k = i; k < n; k++) {
			b |= (uint8_t)((in[k] & 1u) << (k - i));
			bad += in[k] > 1;
		}
		out[i >> 3] = b;
	}
	return bad;
}

#if defined(__x86_64__)
__attribute__((target("avx2")))
static uint64_t pack_avx2(const uint8_t *in, uint8_t *out, uint64_t n)
{
	uint64_t i = 0;
	__m256i acc = _mm256_setzero_si256();
	const __m256i hi7 = _mm256_set1_epi8((char)0xfe);
	for (; i + 128 <= n; i += 128) {
		const __m256i a = _mm256_loadu_si256((const __m256i *)(in + i));
		const __m256i b = _mm256_loadu_si256((const __m256i *)(in + i + 32));
		const __m256i c = _mm256_loadu_si256((const __m256i *)(in + i + 64));
		const __m256i d = _mm256_loadu_si256((const __m256i *)(in + i + 96));
		acc = _mm256_or_si256(acc, _mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d)));
		/* bit 0 of every byte to its sign bit, the 32 sign bits to a word: byte i's bit lands at bit i */
		uint32_t w[4] = { (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(a, 7)), (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 7)),
				  (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 7)), (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(d, 7)) };
		memcpy(out + (i >> 3), w, 16);
	}
	uint64_t bad = !_mm256_testz_si256(acc, hi7);	/* (a count of blocks, not of bytes: != 0 is what matters) */
	return bad + pack_scalar(in + i, out + (i >> 3), n - i);
}
#endif

static void *pack_thread(void *arg)
{
	struct pack_job *j = arg;
#if defined(__x86_64__)
	if (__builtin_cpu_supports("avx2")) {
		j->nonbinary = pack_avx2(j->in, j->out, j->nbytes);
		return NULL;
	}
#endif
	j->nonbinary = pack_scalar(j->in, j->out, j->nbytes);
	return NULL;
}

int64_t tgpu_pack_bits(const uint8_t *bytes, uint64_t n, uint8_t *packed, unsigned int nthreads)
{
	if ((!bytes || !packed) && n)
		return TGPU_EINVAL;
	if (!nthreads)
		nthreads = 1;
	if (nthreads > 256)
		nthreads = 256;
	const uint64_t per = ((n / nthreads) + 1023) & ~(uint64_t)1023;	/* whole output bytes (and cache lines) per thread */
	struct pack_job job[256];
	pthread_t th[256];
	unsigned int nj = 0;
	for (uint64_t o = 0; o < n && nj < 256; o += per, nj++) {
		job[nj].in = bytes + o;
		job[nj].out = packed + (o >> 3);
		job[nj].nbytes = n - o < per || nj + 1 == nthreads ? n - o : per;
		job[nj].nonbinary = 0;
		if (job[nj].nbytes == n - o) {
			nj++;
			break;
		}
	}
	for (unsigned int k = 1; k < nj; k++)
		if (pthread_create(&th[k], NULL, pack_thread, &job[k])) {
			pack_thread(&job[k]);	/* no thread to be had: this one does the piece */
			th[k] = 0;
		}
	if (nj)
		pack_thread(&job[0]);
	int64_t bad = nj ? (int64_t)job[0].nonbinary : 0;
	for (unsigned int k = 1; k < nj; k++) {
		if (th[k])
			pthread_join(th[k], NULL);
		bad += (int64_t)job[k].nonbinary;
	}
	return bad;
}
