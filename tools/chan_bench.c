/*
 * chan_bench.c -- end-to-end rate of the drop-in channel API in plain C, the way tetra-rx.c would use it
 * (INTEGRATION.md section 4): host bytes -> tetra_burst_sync_in() in 64-byte reads -> queued bursts -> GPU batches ->
 * callbacks in the reference's block order.  The callback only counts (an upper MAC would parse the PDU).
 *   gcc -O2 tools/chan_bench.c -Iinclude -Losmo-tetra_amd -ltetra_gpu -Wl,-rpath,$PWD/osmo-tetra_amd -o /tmp/chan_bench
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "tetra_gpu.h"

static unsigned long nblocks, ncrc;
static unsigned long long digest;	/* FNV-1a over what every block delivers: equal runs deliver equal blocks */
static int on_block(const struct tgpu_unitdata *ud, unsigned int offset, void *priv)
{
	(void)priv;
	if (offset == 0 || offset == 0xffffffffu) {
		nblocks++;
		ncrc += ud->crc_ok != 0;
		unsigned long long h = digest ? digest : 1469598103934665603ull;
#define MIX(v) do { h ^= (unsigned long long)(v); h *= 1099511628211ull; } while (0)
		MIX(ud->type); MIX(ud->blk_num); MIX(ud->crc_ok); MIX(ud->crc); MIX(ud->scrambling_code); MIX(ud->burst_seq);
		MIX(ud->tdma_time.tn); MIX(ud->tdma_time.fn); MIX(ud->tdma_time.mn);
		for (unsigned i = 0; ud->type1 && i < ud->type1_len; i++)
			MIX(ud->type1[i]);
#undef MIX
		digest = h;
	}
	return -1;
}

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv)
{
	const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 400000;
	static const uint8_t pat[8] = { 3, 0, 1, 0, 1, 0, 1, 0 };	/* SB, N1, N2, ... (enum tetra_train_seq) */
	uint8_t *types = malloc(n + 1), *stream = calloc(1, 100 + (n + 1) * 510 + 700);
	if (!types || !stream)
		return 1;
	types[0] = 3;
	for (size_t i = 0; i < n; i++)
		types[1 + i] = pat[i & 7];
	struct tgpu_synth_cfg cfg = { .seed = 11, .scramb_init = 0x41802A07, .mcc = 262, .mnc = 42, .cc = 1 };
	cfg.scramb_init = tetra_scramb_get_init(262, 42, 1);
	tgpu_synth_slots(&cfg, types, n + 1, stream + 100, NULL);
	const size_t len = 100 + (n + 1) * 510 + 700;

	struct tgpu_engine *eng;
	if (tgpu_engine_create(&eng, 0)) {
		fprintf(stderr, "no GPU\n");
		return 1;
	}
	unsigned batches[32] = { 1, 64, 1024, 16384 }, nb = 4;
	int a0 = 2;
	if (argc > 2 && (!strcmp(argv[2], "ring") || !strcmp(argv[2], "noring"))) {
		/* chan_bench N ring b1 ...: flushes of up to 4 bursts through workgroups that stay (the default since round 6);
		 * chan_bench N noring b1 ...: every flush by launch */
		tgpu_engine_set_option(eng, TGPU_OPT_RING, !strcmp(argv[2], "ring"));
		a0 = 3;
	}
	if (argc > a0) {		/* chan_bench N [ring] b1 b2 ...: the batch sizes to try */
		nb = 0;
		for (int a = a0; a < argc && nb < 32; a++)
			batches[nb++] = (unsigned)atoi(argv[a]);
	}
	for (unsigned b = 0; b < nb; b++) {
		struct tgpu_channel *ch;
		struct tetra_rx_state trs;
		memset(&trs, 0, sizeof(trs));
		if (tgpu_channel_create(eng, batches[b], on_block, NULL, NULL, &ch))
			return 1;
		trs.burst_cb_priv = ch;
		nblocks = ncrc = 0;
		digest = 0;
		const size_t use = batches[b] < 16 ? len / 8 : len;	/* near-synchronous modes are slow: shorter sample */
		const double t0 = now();
		for (size_t o = 0; o < use; o += 64)
			tetra_burst_sync_in(&trs, stream + o, (unsigned)(use - o < 64 ? use - o : 64));
		tgpu_channel_flush(ch);
		const double el = now() - t0;
		printf("batch %5u: %8.0f bursts/s end to end (%.2f s, %lu blocks delivered, %lu CRC ok, digest %016llx)\n", batches[b],
		       (double)(use / 510) / el, el, nblocks, ncrc, digest);
		tgpu_channel_destroy(ch);
	}
	tgpu_engine_destroy(eng);
	return 0;
}
