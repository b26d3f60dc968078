/*
 * tetra_rx_gpu.c -- the thin tetra-rx a maintainer would write on top of libtetra_gpu.so: plain C
 * against include/tetra_gpu.h, reading a 1-bit-per-byte capture and printing the reference's own observables --
 *     "CRC COMP: 0x%04x OK" + "<NAME> <mn/fn/tn/sn> type1: <bits>"  or  "CRC COMP: 0x%04x WRONG"
 * (lower_mac/tetra_lower_mac.c:258-266; src/tetra-rx-tests.sh:56 counts the OK lines), "found SYNC training sequence
 * in bit #%u" and the "####" loss-of-lock lines of phy/tetra_burst_sync.c:88,126,136,139 on stderr.
 *
 *   tetra_rx_gpu FILE [--seam sync_in|rx_cb] [--batch N]
 *     sync_in (default): tetra_burst_sync_in() fed 64 bytes at a time, as tetra-rx.c:82-95 does
 *     rx_cb            : the program plays the reference's PHY -- bursts and TDMA steps from the synchroniser walk,
 *                        t_phy_state.time stepped with tetra_tdma_time_add_tn(), every burst handed to
 *                        tetra_burst_rx_cb() under the reference's signature, which splits it into
 *                        tp_sap_udata_ind() calls.  Link the reference's own phy/tetra_burst.o in front of the
 *                        library and its tetra_burst_rx_cb() is the one that runs (it finds tp_sap_udata_ind here).
 *
 *   gcc -O2 tools/tetra_rx_gpu.c [oracle/_ref/tetra_burst.o] -Iinclude -Losmo-tetra_amd -ltetra_gpu \
 *       -Wl,-rpath,$PWD/osmo-tetra_amd -o /tmp/tetra_rx_gpu
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tetra_gpu.h"

static const char *const blk_name[6] = { "SB1", "SB2", "NDB", "BBK", "SCH/HU", "SCH/F" };	/* tetra_lower_mac.c:55-102 */
static unsigned long n_ok, n_wrong;

static int on_block(const struct tgpu_unitdata *ud, unsigned int offset, void *priv)
{
	(void)priv;
	if (offset != 0 || ud->type == TPSAP_T_BBK)	/* one line per block; the BBK has no CRC (and prints only under DEBUG) */
		return -1;
	printf("CRC COMP: 0x%04x ", ud->crc);
	if (ud->crc_ok) {
		char bits[300];
		for (unsigned i = 0; i < ud->type1_len; i++)
			bits[i] = ud->type1[i] ? '1' : '0';
		bits[ud->type1_len] = 0;
		printf("OK\n%s %02u/%02u/%u/%03u type1: %s\n", blk_name[ud->type], ud->time_str.mn, ud->time_str.fn,
		       ud->time_str.tn, ud->time_str.sn, bits);
		n_ok++;
	} else {
		printf("WRONG\n");
		n_wrong++;
	}
	return -1;
}

static void on_event(int event, uint32_t bitnum, uint32_t arg, void *priv)
{
	(void)priv;
	(void)bitnum;
	switch (event) {
	case TGPU_EV_FOUND_SYNC:
		printf("found SYNC training sequence in bit #%u\n", arg);
		break;
	case TGPU_EV_SYNC_MISPLACED:
	case TGPU_EV_NORM_MISPLACED:
		fprintf(stderr, "#### SYNC burst at offset %u?!?\n", arg);
		break;
	case TGPU_EV_NO_TRAIN:
		fprintf(stderr, "#### could not find successive burst training sequence\n");
		break;
	case TGPU_EV_ERROR:
		fprintf(stderr, "tetra_rx_gpu: a batch of %u could not be decoded: %s\n", arg, tgpu_strerror((int)bitnum));
		break;
	default:
		break;
	}
}

int main(int argc, char **argv)
{
	const char *path = NULL, *seam = "sync_in";
	unsigned batch = 64;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "--seam") && i + 1 < argc)
			seam = argv[++i];
		else if (!strcmp(argv[i], "--batch") && i + 1 < argc)
			batch = (unsigned)atoi(argv[++i]);
		else
			path = argv[i];
	}
	if (!path || !batch) {
		fprintf(stderr, "usage: %s FILE [--seam sync_in|rx_cb] [--batch N]\n", argv[0]);
		return 2;
	}
	FILE *f = fopen(path, "rb");
	if (!f) {
		perror(path);
		return 2;
	}
	fseek(f, 0, SEEK_END);
	const long len = ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *stream = malloc((size_t)len + 1);
	if (!stream || fread(stream, 1, (size_t)len, f) != (size_t)len)
		return 2;
	fclose(f);

	struct tgpu_engine *eng;
	struct tgpu_channel *ch;
	int rc = tgpu_engine_create(&eng, 0);
	if (rc) {
		fprintf(stderr, "tetra_rx_gpu: %s\n", tgpu_strerror(rc));
		return 1;
	}
	if ((rc = tgpu_channel_create(eng, batch, on_block, on_event, NULL, &ch))) {
		fprintf(stderr, "tetra_rx_gpu: %s\n", tgpu_strerror(rc));
		return 1;
	}
	if (!strcmp(seam, "sync_in")) {
		struct tetra_rx_state trs;
		memset(&trs, 0, sizeof(trs));
		trs.burst_cb_priv = ch;
		for (long o = 0; o < len; o += 64)
			tetra_burst_sync_in(&trs, stream + o, (unsigned)(len - o < 64 ? len - o : 64));
	} else {
		struct tgpu_sync_result res;
		if ((rc = tgpu_sync_walk(stream, (uint64_t)len, 64, 0, NULL, NULL, 0, TGPU_SYNC_NO_BURST_EVENTS, &res))) {
			fprintf(stderr, "tetra_rx_gpu: %s\n", tgpu_strerror(rc));
			return 1;
		}
		for (uint32_t e = 0; e < res.nevents; e++)
			on_event(res.events[e].ev, res.events[e].bitnum, res.events[e].arg, NULL);
		memset(&t_phy_state, 0, sizeof(t_phy_state));
		for (uint32_t i = 0; i < res.nslots; i++) {
			tetra_tdma_time_add_tn(&t_phy_state.time, res.slots[i].tn_adds);	/* phy/tetra_burst_sync.c:113 */
			tetra_burst_rx_cb(stream + res.slots[i].off, 510, (enum tetra_train_seq)res.slots[i].type, ch);
		}
		tgpu_sync_result_free(&res);
	}
	rc = tgpu_channel_flush(ch);
	if (rc)
		fprintf(stderr, "tetra_rx_gpu: %s\n", tgpu_strerror(rc));
	fprintf(stderr, "%lu CRC OK, %lu CRC WRONG\n", n_ok, n_wrong);
	tgpu_channel_destroy(ch);
	tgpu_engine_destroy(eng);
	free(stream);
	return rc ? 1 : 0;
}
