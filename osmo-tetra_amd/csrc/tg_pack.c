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

static uint64_t pack_piece(const uint8_t *in, uint8_t *out, uint64_t n)
{
#if defined(__x86_64__)
	if (__builtin_cpu_supports("avx2"))
		return pack_avx2(in, out, n);
#endif
	return pack_scalar(in, out, n);
}

/*
 * A pool of worker threads that lives as long as the process (created on first use, grown on demand): a 510 MB capture is
 * packed in a few milliseconds by some tens of threads, and creating that many threads per call would cost as much as the
 * packing.  One call at a time uses the pool (a mutex around the call); the work is cut into pieces of PIECE input bytes that
 * the workers -- and the calling thread -- take with an atomic counter.  Workers inherit the affinity of the thread that
 * created them: a caller pinned to its GPU's NUMA node keeps the packing there.
 */
#define PIECE (256u * 1024u)	/* input bytes per piece: a multiple of 1024, i.e. whole output cache lines */
#define MAX_WORKERS 255

static struct {
	pthread_mutex_t call;		/* one tgpu_pack_bits() at a time */
	pthread_mutex_t m;
	pthread_cond_t go, done;
	pthread_t th[MAX_WORKERS];
	unsigned int nworkers, wanted;	/* wanted: workers that may take part in the current call */
	uint64_t gen;			/* incremented per call */
	unsigned int running;		/* workers still inside the current call */
	const uint8_t *in;
	uint8_t *out;
	uint64_t n, npieces;
	uint64_t next;			/* next piece (atomic) */
	uint64_t bad;			/* pieces that held a byte other than 0 / 1 (atomic) */
} P = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, { 0 }, 0, 0, 0, 0,
	NULL, NULL, 0, 0, 0, 0 };

static void pack_drain(void)
{
	for (;;) {
		const uint64_t i = __atomic_fetch_add(&P.next, 1, __ATOMIC_RELAXED);
		if (i >= P.npieces)
			return;
		const uint64_t o = i * PIECE, len = P.n - o < PIECE ? P.n - o : PIECE;
		if (pack_piece(P.in + o, P.out + (o >> 3), len))
			__atomic_fetch_add(&P.bad, 1, __ATOMIC_RELAXED);
	}
}

static void *pack_worker(void *arg)
{
	const unsigned int me = (unsigned int)(uintptr_t)arg;
	uint64_t seen = 0;
	pthread_mutex_lock(&P.m);
	for (;;) {
		while (P.gen == seen || me >= P.wanted)
			pthread_cond_wait(&P.go, &P.m);
		seen = P.gen;
		pthread_mutex_unlock(&P.m);
		pack_drain();
		pthread_mutex_lock(&P.m);
		if (--P.running == 0)
			pthread_cond_signal(&P.done);
	}
	return NULL;
}

int64_t tgpu_pack_bits(const uint8_t *bytes, uint64_t n, uint8_t *packed, unsigned int nthreads)
{
	if ((!bytes || !packed) && n)
		return TGPU_EINVAL;
	if (!n)
		return 0;
	if (!nthreads)
		nthreads = 1;
	const uint64_t npieces = (n + PIECE - 1) / PIECE;
	if (nthreads > npieces)
		nthreads = (unsigned int)npieces;
	if (nthreads > MAX_WORKERS + 1)
		nthreads = MAX_WORKERS + 1;
	if (nthreads == 1)
		return (int64_t)(pack_piece(bytes, packed, n) != 0);
	pthread_mutex_lock(&P.call);
	pthread_mutex_lock(&P.m);
	while (P.nworkers < nthreads - 1) {		/* (the calling thread is one of the nthreads) */
		if (pthread_create(&P.th[P.nworkers], NULL, pack_worker, (void *)(uintptr_t)P.nworkers))
			break;				/* no more threads to be had: the ones there are do the work */
		pthread_detach(P.th[P.nworkers]);
		P.nworkers++;
	}
	P.in = bytes;
	P.out = packed;
	P.n = n;
	P.npieces = npieces;
	P.next = 0;
	P.bad = 0;
	P.wanted = P.nworkers < nthreads - 1 ? P.nworkers : nthreads - 1;
	P.running = P.wanted;
	P.gen++;
	pthread_cond_broadcast(&P.go);
	pthread_mutex_unlock(&P.m);
	pack_drain();
	pthread_mutex_lock(&P.m);
	while (P.running)
		pthread_cond_wait(&P.done, &P.m);
	P.wanted = 0;
	const int64_t bad = (int64_t)P.bad;
	pthread_mutex_unlock(&P.m);
	pthread_mutex_unlock(&P.call);
	return bad;
}
