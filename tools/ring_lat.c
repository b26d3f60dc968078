/* Where a flush of the channel API spends its time when it goes through the workgroups that stay (TGPU_OPT_RING) or through a
 * launch (k_burst): n bursts of one type handed over with tetra_burst_rx_cb()'s entry point, flushed, 20 000 times.
 * gcc -O2 -Iinclude tools/ring_lat.c -Losmo-tetra_amd -ltetra_gpu -o tools/ring_lat ; ring_lat <n> <type> <ring 0|1>
 * (a -DTGB_TIMING build of the library also prints the phases of the last decode on the device) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "tetra_gpu.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static unsigned long nb, nok;
static int on_block(const struct tgpu_unitdata *ud, unsigned int offset, void *priv)
{
	(void)priv; (void)offset;
	nb++;
	nok += ud->crc_ok != 0;
	return -1;
}

int main(int argc, char **argv)
{
	const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
	const int type = argc > 2 ? atoi(argv[2]) : 0;
	const int ring = argc > 3 ? atoi(argv[3]) : 1;
	struct tgpu_engine *eng;
	struct tgpu_channel *ch;
	if (tgpu_engine_create(&eng, 0))
		return 1;
	tgpu_engine_set_option(eng, TGPU_OPT_RING, ring);
	if (tgpu_channel_create(eng, n, on_block, NULL, NULL, &ch))
		return 2;
	struct tgpu_synth_cfg cfg = { 1, 3, 262, 42, 1, 0.0, 1 };
	uint8_t types[8], slots[8 * 510];
	for (unsigned i = 0; i < n && i < 8; i++) types[i] = (uint8_t)type;
	tgpu_synth_slots(&cfg, types, n, slots, NULL);
	const int N = 20000;
	double t0 = 0;
	for (int i = -500; i < N; i++) {
		if (i == 0) t0 = now();
		for (unsigned k = 0; k < n; k++)
			tgpu_channel_burst_rx(ch, slots + 510 * k, 510, (enum tetra_train_seq)types[k], 1);
		tgpu_channel_flush(ch);
	}
	const double us = (now() - t0) / N * 1e6;
	printf("%s, %u burst(s) of type %d per flush: %.2f us per flush (%lu blocks, %lu CRC ok)\n", ring ? "ring" : "launch", n, type, us, nb, nok);
#ifdef TGB_TIMING
	{
		extern int tgk_burst_stamps(unsigned long long *out);
		unsigned long long st[16];
		static const char *const nm[] = { "descriptors -> LDS", "slot -> LDS, code", "de-interleave / descramble", "increments", "(wave 0 enters)",
						  "trellis", "traceback", "crc", "record + completion mark" };
		if (!tgk_burst_stamps(st)) {
			if (ring)
				printf("  %-28s %7.2f us\n", "request seen -> decode starts", (double)(st[0] - st[9]) / 100.0);
			for (int k = 1; k <= 8; k++)
				printf("  %-28s %7.2f us\n", nm[k - 1], (double)(st[k] - st[k - 1]) / 100.0);
		}
	}
#endif
	tgpu_channel_destroy(ch);
	tgpu_engine_destroy(eng);
	return 0;
}
