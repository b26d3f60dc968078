/*
 * pin_harness.c -- TEST INFRASTRUCTURE for tools/pin_with_libosmocore.sh, compiled ONLY on a machine that has libosmocore and
 * a checkout of the reference (it includes the reference's own headers and links the reference's own objects; nothing of either
 * is in this repository, and this build container can compile neither).
 *
 * What it is for: two rows that the block-level library (libtetra_ref_osmo.so) and tetra-rx cannot pin --
 *   - row P's uplink shape and row L on single blocks: tp_sap_udata_ind() (lower_mac/tetra_lower_mac.c:143-357) called with ANY
 *     block type, SCH/HU included (no downlink burst carries one, so tetra-rx never decodes it); what the lower MAC hands to
 *     upper_mac_prim_recv() is recorded: lchan, crc_ok, scrambling code, block number, TDMA time, the type-1 bits;
 *   - (f)3's GSMTAP constants: tetra_gsmtap_makemsg() (tetra_gsmtap.c:31-63) on given time / channel / bits: the message's bytes,
 *     header included (GSMTAP_VERSION, GSMTAP_TYPE_TETRA_I1, the GSMTAP_TETRA_* sub-types are libosmocore's).
 * This file's own upper_mac_prim_recv() stands in front of the reference's (link order + --allow-multiple-definition in the script).
 *
 * Protocol (stdin -> stdout, one line each):
 *   udata <type 0..5> <blk_num> <hex: one byte per bit>      -> zero or more "prim lchan=.. crc_ok=.. code=.. blk=.. tn=.. fn=.. mn=.. len=.. bits=<hex>"
 *                                                               lines, then "done"
 *   time <tn> <fn> <mn>                                       -> sets t_phy_state.time (what tp_sap_udata_ind copies), "done"
 *   gsmtap <tn> <fn> <mn> <lchan> <ts> <ss> <signal_dbm> <snr> <hex: one byte per bit>   -> "gsmtap <hex of the message>"
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include <osmocom/core/msgb.h>
#include <osmocom/core/prim.h>
#include <osmocom/core/talloc.h>

#include "tetra_common.h"
#include "tetra_tdma.h"
#include "tetra_prim.h"
#include "tetra_gsmtap.h"
#include "phy/tetra_burst.h"
#include "phy/tetra_burst_sync.h"

static struct tetra_mac_state g_tms;

/* what the lower MAC indicates (tetra_lower_mac.c:335): printed, and "nothing parsed" returned (-1: the multi-PDU loop ends) */
int upper_mac_prim_recv(struct osmo_prim_hdr *op, void *priv)
{
	struct tetra_tmvsap_prim *tmvp = (struct tetra_tmvsap_prim *)op;
	struct tmv_unitdata_param *tup = &tmvp->u.unitdata;
	struct msgb *msg = op->msg;
	const unsigned int len = msgb_l1len(msg);
	(void)priv;
	printf("prim lchan=%d crc_ok=%d code=%u blk=%d tn=%u fn=%u mn=%u len=%u bits=", (int)tup->lchan, tup->crc_ok,
	       (unsigned)tup->scrambling_code, tup->blk_num, (unsigned)tup->tdma_time.tn, (unsigned)tup->tdma_time.fn,
	       (unsigned)tup->tdma_time.mn, len);
	for (unsigned int i = 0; i < len; i++)
		printf("%02x", msg->l1h[i]);
	printf("\n");
	return -1;
}

static int unhex(const char *s, uint8_t *out, int max)
{
	int n = 0;
	while (s[0] && s[1] && n < max) {
		unsigned int v;
		if (sscanf(s, "%2x", &v) != 1)
			break;
		out[n++] = (uint8_t)v;
		s += 2;
	}
	return n;
}

int main(void)
{
	static char line[4096];
	static uint8_t bits[1024];
	tetra_mac_state_init(&g_tms);
	g_tms.dumpdir = NULL;
	while (fgets(line, sizeof(line), stdin)) {
		char hex[2200];
		int a, b, c, d, e, f, g, h;
		if (sscanf(line, "udata %d %d %2100s", &a, &b, hex) == 3) {
			const int n = unhex(hex, bits, (int)sizeof(bits));
			tp_sap_udata_ind((enum tp_sap_data_type)a, b, bits, (unsigned int)n, &g_tms);
			printf("done\n");
		} else if (sscanf(line, "time %d %d %d", &a, &b, &c) == 3) {
			t_phy_state.time.tn = (uint32_t)a;
			t_phy_state.time.fn = (uint32_t)b;
			t_phy_state.time.mn = (uint32_t)c;
			printf("done\n");
		} else if (sscanf(line, "gsmtap %d %d %d %d %d %d %d %d %2100s", &a, &b, &c, &d, &e, &f, &g, &h, hex) == 9) {
			struct tetra_tdma_time tm;
			memset(&tm, 0, sizeof(tm));
			tm.tn = (uint32_t)a;
			tm.fn = (uint32_t)b;
			tm.mn = (uint32_t)c;
			const int n = unhex(hex, bits, (int)sizeof(bits));
			struct msgb *msg = tetra_gsmtap_makemsg(&tm, (enum tetra_log_chan)d, (uint8_t)e, (uint8_t)f, (int8_t)g, (uint8_t)h, bits,
							       (unsigned int)n, &g_tms);
			printf("gsmtap ");
			if (msg) {
				for (unsigned int i = 0; i < msgb_length(msg); i++)
					printf("%02x", msgb_data(msg)[i]);
				msgb_free(msg);
			}
			printf("\n");
		} else
			printf("error\n");
		fflush(stdout);
	}
	return 0;
}
