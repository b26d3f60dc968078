/*
 * tgpu_adapter.c -- the one file a maintainer of osmo-tetra adds to put libtetra_gpu.so under the untouched upper MAC
 * (INTEGRATION.md section 3).  It is NOT part of the library and is not built into it: it needs libosmocore and the
 * reference's own headers (msgb, talloc, struct tetra_tmvsap_prim, struct tetra_mac_state), which this repository
 * neither has nor imitates.  __graft_entry__.build() compiles it (-fsyntax-only) when pkg-config finds libosmocore and a
 * reference tree is at hand, and skips it otherwise; tests/test_host_logic.py checks on every run that each
 * tgpu_* symbol and struct tgpu_unitdata field used below still exists in include/tetra_gpu.h.
 *
 * Reference seams: upper_mac_prim_recv() (tetra_upper_mac.h), struct tetra_tmvsap_prim / tmv_unitdata_param
 * (src/tetra_prim.h:25-47), what tp_sap_udata_ind() hands over (src/lower_mac/tetra_lower_mac.c:129-140 the
 * primitive's allocation, :198-241 the traffic dump, :279-280 and :326-352 the indication and its multi-PDU loop).
 *
 * Build (in the reference's src/, next to tetra-rx.c; the reference's headers go FIRST, tetra_gpu.h then takes the
 * mirrored types from them):
 *   cc -c tgpu_adapter.c -I. -Iphy -Ilower_mac -I$TGPU/include $(pkg-config --cflags libosmocore)
 */
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <osmocom/core/msgb.h>
#include <osmocom/core/talloc.h>

#include "tetra_common.h"
#include "tetra_prim.h"
#include "tetra_upper_mac.h"
#include "phy/tetra_burst.h"
#include "phy/tetra_burst_sync.h"
#include "lower_mac/tetra_scramb.h"

#include "tetra_gpu.h"

/* the primitive the upper MAC is handed (the allocation of tetra_lower_mac.c:129-140, which lives in the file the library
 * replaces) */
static struct tetra_tmvsap_prim *adapter_prim_alloc(void)
{
	struct tetra_tmvsap_prim *ttp = talloc_zero(NULL, struct tetra_tmvsap_prim);
	ttp->oph.msg = msgb_alloc(412, "tmvsap_prim");
	ttp->oph.sap = TETRA_SAP_TMV;
	ttp->oph.primitive = PRIM_TMV_UNITDATA;
	ttp->oph.operation = PRIM_OP_INDICATION;
	return ttp;
}

/* a traffic block: the 690-word block to <dumpdir>/traffic_<usage>_<tsn>.out, the SSI in use to the .txt beside it */
static void adapter_traffic(struct tetra_mac_state *tms, const struct tgpu_unitdata *ud)
{
	char fname[PATH_MAX];
	int16_t block[690];
	FILE *f;

	tgpu_traffic_block(ud->type4, ud->type4_len, block);
	snprintf(fname, sizeof(fname), "%s/traffic_%d_%d.out", tms->dumpdir, ud->traffic, tms->tsn);
	f = fopen(fname, "ab");
	if (!f) {
		fprintf(stderr, "Could not open dump file %s for writing: %s\n", fname, strerror(errno));
		exit(1);
	}
	fwrite(block, sizeof(int16_t), 690, f);
	fclose(f);
	snprintf(fname, sizeof(fname), "%s/traffic_%d_%d.txt", tms->dumpdir, ud->traffic, tms->tsn);
	f = fopen(fname, "a");
	if (f) {
		fprintf(f, "%d\n", tms->ssi);
		fclose(f);
	}
}

/* tgpu_unitdata_cb with upper_mac_prim_recv()'s return contract: priv = the struct tetra_mac_state the reference's
 * tetra-rx.c keeps (tgpu_channel_create(eng, batch, tgpu_adapter_unitdata, NULL, tms, &ch)) */
int tgpu_adapter_unitdata(const struct tgpu_unitdata *ud, unsigned int offset, void *priv)
{
	struct tetra_mac_state *tms = priv;
	struct tetra_tmvsap_prim *ttp;
	struct tmv_unitdata_param *tup;
	struct msgb *msg;
	int rc;

	if (ud->traffic) {
		adapter_traffic(tms, ud);
		return -1;
	}
	ttp = adapter_prim_alloc();
	tup = &ttp->u.unitdata;
	msg = ttp->oph.msg;
	tup->lchan = ud->lchan;
	tup->crc_ok = ud->crc_ok;
	tup->scrambling_code = ud->scrambling_code;
	tup->blk_num = ud->blk_num;
	memcpy(&tup->tdma_time, &ud->tdma_time, sizeof(tup->tdma_time));
	msg->l1h = msgb_put(msg, ud->type1_len);
	memcpy(msg->l1h, ud->type1, ud->type1_len);
	/* the multi-PDU loop of tetra_lower_mac.c:326-352 runs inside the library; the offset it has reached is ours to apply */
	msg->head += offset;
	msg->l1h = msg->head;
	msg->len = msg->tail - msg->head;
	rc = upper_mac_prim_recv(&ttp->oph, tms);
	talloc_free(msg);
	talloc_free(ttp);
	return rc;
}

/* what tetra-rx.c's main() does instead of handing tms to the PHY (src/tetra-rx.c:48-54): returns the channel to put
 * into trs->burst_cb_priv, or NULL */
struct tgpu_channel *tgpu_adapter_attach(struct tetra_mac_state *tms, unsigned int bursts_per_batch)
{
	struct tgpu_engine *eng;
	struct tgpu_channel *ch;

	if (tgpu_engine_create(&eng, 0))		/* no GPU: a hard error, there is no CPU fallback */
		return NULL;
	if (tgpu_channel_create(eng, bursts_per_batch, tgpu_adapter_unitdata, NULL, tms, &ch))
		return NULL;
	/* the upper MAC keeps writing tms->cur_burst; the library reads the same variables */
	tgpu_channel_bind_flags(ch, &tms->cur_burst.is_traffic, &tms->cur_burst.blk1_stolen, &tms->cur_burst.blk2_stolen);
	return ch;
}
