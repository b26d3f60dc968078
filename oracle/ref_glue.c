/*
 * ref_glue.c -- the one symbol the reference's PHY objects expect their
 * linker to supply (extern tp_sap_udata_ind, phy/tetra_burst.h:18; the
 * reference's own conv_enc_test.c:48-50 supplies an empty one).  Ours records
 * the calls so tests can check tetra_burst_rx_cb()'s demultiplexing.
 *
 * TEST INFRASTRUCTURE ONLY.  Linked into oracle/_ref/libtetra_ref.so together
 * with objects compiled straight from /root/reference/src (never copied).
 */
#include <stdint.h>
#include <string.h>

#define REF_MAX_CALLS 8

struct ref_tp_call {
	int type;
	int blk_num;
	unsigned len;
	uint8_t bits[432];
};

static struct ref_tp_call calls[REF_MAX_CALLS];
static int n_calls;

void tp_sap_udata_ind(int type, int blk_num, const uint8_t *bits, unsigned int len, void *priv)
{
	(void)priv;
	if (n_calls >= REF_MAX_CALLS)
		return;
	struct ref_tp_call *c = &calls[n_calls++];
	c->type = type;
	c->blk_num = blk_num;
	c->len = len > 432 ? 432 : len;
	memcpy(c->bits, bits, c->len);
}

void ref_glue_reset(void) { n_calls = 0; }
int ref_glue_count(void) { return n_calls; }
const struct ref_tp_call *ref_glue_get(int i) { return (i >= 0 && i < n_calls) ? &calls[i] : 0; }
