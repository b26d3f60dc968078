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

/*
 * conv_cch_decode(): the symbol lower_mac/viterbi.c:24 calls (lower_mac/viterbi_cch.h:5; the reference's own
 * definition, viterbi_cch.c:58-66, is a call into libosmocore, which this image lacks).  This one records what
 * the real viterbi_dec_sb1_wrapper() hands over -- n trellis steps of four int8 values plus the K - 1 = 4 flush
 * steps osmo_conv_decode() reads behind them -- and lets the decoder the test registered (the oracle's) fill
 * the output.
 */
#define REF_VIT_MAX ((864 + 4) * 4)
static int8_t vit_in[REF_VIT_MAX];
static int vit_n = -1;
static int (*vit_decoder)(const int8_t *in, uint8_t *out, int n);

int conv_cch_decode(int8_t *input, uint8_t *output, int n)
{
	vit_n = n;
	if (n >= 0 && (n + 4) * 4 <= REF_VIT_MAX)
		memcpy(vit_in, input, (size_t)(n + 4) * 4);
	return vit_decoder ? vit_decoder(input, output, n) : 0;
}

void ref_glue_set_decoder(int (*fn)(const int8_t *, uint8_t *, int)) { vit_decoder = fn; }
int ref_glue_vit_n(void) { return vit_n; }
const int8_t *ref_glue_vit_input(void) { return vit_in; }

void ref_glue_reset(void) { n_calls = 0; }
int ref_glue_count(void) { return n_calls; }
const struct ref_tp_call *ref_glue_get(int i) { return (i >= 0 && i < n_calls) ? &calls[i] : 0; }
