/* Where a small flush spends its time: plan load / execute (launches) / wait, for n bursts per batch (plan API,
 * zero-copy mapped buffers like the channel API uses).
 * gcc -O2 -Iinclude -I/opt/rocm/include tools/flush_lat.c -Losmo-tetra_amd -ltetra_gpu -L/opt/rocm/lib -lamdhip64 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "tetra_gpu.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
	const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
	const int type = argc > 2 ? atoi(argv[2]) : 1;
	struct tgpu_engine *eng;
	struct tgpu_plan *plan;
	if (tgpu_engine_create(&eng, 0) || tgpu_plan_create(eng, n, 1, &plan))
		return 1;
	uint8_t *h_slots, *h_rec, *d_slots, *d_rec;
	if (hipHostMalloc((void **)&h_slots, (size_t)n * 512 + 64, hipHostMallocMapped) ||
	    hipHostMalloc((void **)&h_rec, (size_t)n * TGPU_REC_BYTES, hipHostMallocMapped) ||
	    hipHostGetDevicePointer((void **)&d_slots, h_slots, 0) || hipHostGetDevicePointer((void **)&d_rec, h_rec, 0))
		return 2;
	struct tgpu_synth_cfg cfg = { 1, 0x12345, 262, 42, 1, 0.0, 1 };
	uint8_t *types = malloc(n), *tmp = malloc((size_t)n * 510);
	uint64_t *off = malloc(8 * n);
	uint32_t *chan = calloc(n, 4), code = 0x12345;
	for (unsigned i = 0; i < n; i++) { types[i] = (uint8_t)type; off[i] = 512ull * i; }
	tgpu_synth_slots(&cfg, types, n, tmp, NULL);
	for (unsigned i = 0; i < n; i++) memcpy(h_slots + 512 * i, tmp + 510 * i, 510);
	hipStream_t s;
	if (hipStreamCreate(&s))
		return 3;
	double tl = 0, te = 0, tw = 0;
	const int N = 20000;
	for (int i = -500; i < N; i++) {
		if (i == 0) tl = te = tw = 0;
		double a = now();
		tgpu_plan_load(plan, n, off, types, chan, 1, &code);
		double b = now();
		tgpu_plan_execute(plan, d_slots, d_rec, s);
		double c = now();
		(void)hipStreamSynchronize(s);
		double d = now();
		tl += b - a; te += c - b; tw += d - c;
	}
	printf("n %u type %d: load %.2f us  execute (launches) %.2f us  wait %.2f us  total %.2f us; crc_ok %d\n", n, type,
	       tl / N * 1e6, te / N * 1e6, tw / N * 1e6, (tl + te + tw) / N * 1e6, h_rec[2]);
#ifdef TGB_TIMING
	{
		extern int tgk_burst_stamps(unsigned long long *out);
		unsigned long long st[16];
		static const char *const nm[] = { "descriptors -> LDS", "slot -> LDS, code", "de-interleave / descramble", "increments", "(wave 0 enters)",
						  "trellis", "traceback", "crc", "record + completion mark" };
		if (!tgk_burst_stamps(st))
			for (int k = 1; k <= 8; k++)
				printf("  %-28s %7.2f us\n", nm[k - 1], (double)(st[k] - st[k - 1]) / 100.0);	/* wall_clock64: 100 MHz */
	}
#endif
	return 0;
}
