/*
 * tetra_oracle.h -- CPU restatement of the TETRA lower-MAC receive path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may load it.
 * The product library (osmo-tetra_amd/csrc) never includes, links or calls
 * anything from this directory.
 *
 * Every function restates, in plain C and from scratch, the behaviour of one
 * piece of osmocom/osmo-tetra (paths relative to /root/reference/src) and
 * cites the file:line it follows.
 *
 * Parity status:
 *   - rows X (scrambler), C (CRC-16), R/E (RM(30,14) encoder), F (training
 *     sequence search), D (burst demux), T (TDMA time), E (burst builders),
 *     B (float_to_bits), the ubit -> sbit map of viterbi.c:6-25 (0 -> +127,
 *     0xff -> 0, anything else -> -127, flush steps = 0) and the ACELP
 *     re-ordering of tch_reordering.c: pinned against the REAL reference
 *     objects compiled into oracle/_ref/ (see oracle/Makefile) and against
 *     tests/golden/.
 *   - rows I (interleaver), U (puncturer), E (conv. encoder), S (sync state
 *     machine), L (lower MAC orchestration): the reference files need
 *     libosmocore headers that are absent here, so they are pinned by the
 *     golden vectors recorded from the reference's own objects in SURVEY.md
 *     section 4 (tests/golden/survey_kat.json) and by the reference's own
 *     self-test properties (tetra_punct_test round trips, conv_enc_test
 *     loop-back).
 *   - row V (Viterbi): the arithmetic lives in libosmocore (osmo_conv_decode,
 *     un-vendored, version unpinned: src/Makefile:1-2, contrib/jenkins.sh:20,
 *     call site lower_mac/viterbi_cch.c:58-66).  Both published libosmocore
 *     algorithms (generic conv.c and accelerated conv_acc.c) are restated
 *     here and cross-checked against each other.  Noise-free parity is pinned
 *     by the reference's loop-back test (conv_enc_test.c:336-349); for NOISY
 *     input: PARITY UNPINNED (no libosmocore in this container, no reference
 *     test holds a noisy vector).
 *
 * In one line per row, what NO reference output pins (restatement + properties
 * only): I block (de)interleaver, U puncturers, the two convolutional
 * encoders, S tetra_burst_sync_in(), L tp_sap_udata_ind() orchestration, V the
 * add-compare-select / tie rule / traceback of osmo_conv_decode() on noisy
 * input, the GSMTAP header constants.  Everything else above is pinned by a
 * real reference object.
 */
#ifndef TETRA_ORACLE_H
#define TETRA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums mirrored from the reference (values are ABI) ---------------- */

/* phy/tetra_burst.h:30-36 */
enum orc_train_seq {
	ORC_TRAIN_NORM_1 = 0,
	ORC_TRAIN_NORM_2 = 1,
	ORC_TRAIN_NORM_3 = 2,
	ORC_TRAIN_SYNC   = 3,
	ORC_TRAIN_EXT    = 4,
};

/* phy/tetra_burst.h:9-16 */
enum orc_tpsap_type {
	ORC_T_SB1 = 0,
	ORC_T_SB2 = 1,
	ORC_T_NDB = 2,
	ORC_T_BBK = 3,
	ORC_T_SCH_HU = 4,
	ORC_T_SCH_F = 5,
};

/* tetra_common.h:22-39 */
enum orc_lchan {
	ORC_LC_UNKNOWN = 0,
	ORC_LC_SCH_F = 1,
	ORC_LC_AACH = 8,
	ORC_LC_BSCH = 10,
	ORC_LC_BNCH = 11,
};

/* lower_mac/tetra_conv_enc.h puncturer ids, same order as the reference enum */
enum orc_punct {
	ORC_PUNCT_2_3 = 0,
	ORC_PUNCT_1_3 = 1,
	ORC_PUNCT_292_432 = 2,
	ORC_PUNCT_148_432 = 3,
	ORC_PUNCT_112_168 = 4,
	ORC_PUNCT_72_162 = 5,
	ORC_PUNCT_38_80 = 6,
};

/* tetra_tdma.h:6-12 */
struct orc_tdma_time {
	uint16_t hn;
	uint32_t sn;
	uint32_t tn;
	uint32_t fn;
	uint32_t mn;
};

/* ---- row X: scrambler (lower_mac/tetra_scramb.c) ----------------------- */
uint32_t orc_scramb_get_init(uint16_t mcc, uint16_t mnc, uint8_t colour);
void orc_scramb_get_bits(uint32_t lfsr_init, uint8_t *out, int len);
void orc_scramb_bits(uint32_t lfsr_init, uint8_t *io, int len);

/* ---- row I: block interleaver (lower_mac/tetra_interleave.c) ----------- */
void orc_block_interleave(uint32_t K, uint32_t a, const uint8_t *in, uint8_t *out);
void orc_block_deinterleave(uint32_t K, uint32_t a, const uint8_t *in, uint8_t *out);

/* ---- rows E/U: RCPC mother code + (de)puncturing (tetra_conv_enc.c) ---- */
void orc_conv_encode(const uint8_t *in, int len, uint8_t *out4);
int orc_puncture(enum orc_punct pu, const uint8_t *mother, int len, uint8_t *out);
int orc_depuncture(enum orc_punct pu, const uint8_t *in, int len, uint8_t *mother);

/* ---- row V: Viterbi ---------------------------------------------------- */
/* libosmocore conv.c generic algorithm, as called through viterbi_cch.c:58-66 */
int orc_viterbi_generic(const int8_t *sbits, uint8_t *out, int len);
/* libosmocore conv_acc.c / conv_acc_generic.c formulation (N=4,K=5)       */
int orc_viterbi_acc(const int8_t *sbits, uint8_t *out, int len);
/* the same two algorithms on either mother code (lower_mac/viterbi_cch.c:28-47, viterbi_tch.c:29-47) */
#define ORC_CODE_CCH 0
#define ORC_CODE_TCH 1
unsigned orc_code_output(int code, unsigned state, unsigned bit);
int orc_viterbi_generic_code(int code, const int8_t *sbits, uint8_t *out, int len);
int orc_viterbi_acc_code(int code, const int8_t *sbits, uint8_t *out, int len);
void orc_conv_encode_tch(const uint8_t *in, int len, uint8_t *out3);
int orc_conv_decode_block(int pu, int mother_rate, const uint8_t *type3, unsigned type3_len, unsigned type2_len,
			  int use_acc, uint8_t *type2);
/* viterbi.c:6-25 : ubit/erasure -> sbit map, then the decoder above.
 * use_acc selects which restatement runs (0 = generic).                   */
void orc_viterbi_dec_wrapper(const uint8_t *in, uint8_t *out, unsigned sym_count, int use_acc);
/* soft-input extension (BASELINE config 5; the reference has no such path):
 * int8 soft values straight into the accelerated (correlation) decoder.   */
void orc_viterbi_soft(const int8_t *sbits_mother, uint8_t *out, unsigned sym_count);

/* ---- row C: CRC-16 (lower_mac/crc_simple.c) ---------------------------- */
uint16_t orc_crc16_itut_bits(uint16_t crc, const uint8_t *bits, int n);
uint16_t orc_crc16_ccitt_bits(const uint8_t *bits, unsigned n);
#define ORC_CRC_OK 0x1d0f	/* tetra_common.h:69 */

/* ---- row R: RM(30,14) (lower_mac/tetra_rm3014.c) ----------------------- */
uint32_t orc_rm3014_row(int i);
uint32_t orc_rm3014_compute(uint16_t in);

/* ---- row T: TDMA time (tetra_tdma.c) ----------------------------------- */
void orc_tdma_add_tn(struct orc_tdma_time *tm, uint32_t tn_count);
void orc_tdma_dump(const struct orc_tdma_time *tm, char *buf, size_t buflen);

/* ---- row F: training sequence search (phy/tetra_burst.c:269-339) ------- */
int orc_find_train_seq(const uint8_t *in, unsigned end_of_in, uint32_t mask, unsigned *offset);
const uint8_t *orc_train_bits(enum orc_train_seq t, unsigned *len);

/* ---- row E: burst builders (phy/tetra_burst.c:117-267) ----------------- */
int orc_build_sync_burst(uint8_t *buf510, const uint8_t *sb120, const uint8_t *bb30, const uint8_t *bkn216);
int orc_build_norm_burst(uint8_t *buf510, const uint8_t *bkn1_216, const uint8_t *bb30,
			 const uint8_t *bkn2_216, int two_log_chan);

/* ---- row E: whole-block channel encoders (conv_enc_test.c:88-134) ------ */
/* type-1 bits -> CRC16 -> tail -> RCPC 2/3 -> interleave -> scramble      */
void orc_encode_block(enum orc_tpsap_type type, const uint8_t *type1, uint32_t scramb_init, uint8_t *type5);
/* 14 AACH bits -> RM(30,14) -> scramble: 30 type-5 bits                   */
void orc_encode_bbk(const uint8_t *type1_14, uint32_t scramb_init, uint8_t *type5_30);

/* ---- rows P/L: one block through the type-5 -> type-1 chain ------------ */
struct orc_blk_param {
	const char *name;
	uint16_t type345_bits, type2_bits, type1_bits, interleave_a;
	uint8_t have_crc16;
};
const struct orc_blk_param *orc_blk_param(enum orc_tpsap_type t);

struct orc_block_result {
	uint8_t type1[432];	/* decoded bits, 1 per byte (type2 incl. crc+tail also kept) */
	uint8_t type2[288];
	uint8_t type4[432];
	uint16_t crc;
	int crc_ok;
};
void orc_decode_block(enum orc_tpsap_type type, const uint8_t *type5, uint32_t scramb_init,
		      int use_acc, struct orc_block_result *res);

/* ---- rows S/D/L/T: the streaming receiver ------------------------------ */

/* one record per tp_sap_udata_ind() call of the reference */
struct orc_record {
	uint32_t burst_seq;		/* ordinal of the LOCKED burst that produced it */
	uint8_t  burst_type;		/* enum orc_train_seq */
	uint8_t  type;			/* enum orc_tpsap_type */
	uint8_t  blk_num;
	uint8_t  lchan;			/* enum orc_lchan */
	uint8_t  crc_ok;
	uint8_t  traffic_dumped;	/* 1: block went to the traffic dump, no decode (tetra_lower_mac.c:198-241) */
	uint16_t crc;
	uint32_t scrambling_code;
	struct orc_tdma_time time;
	uint16_t type1_len;
	uint8_t  type1[268];
	uint8_t  type4[432];		/* descrambled bits (what the traffic dump is made of) */
	struct orc_tdma_time time_str;	/* tcd->time as copied from t_phy_state at entry (:167-168): what the
					 * reference's "<NAME> <time> type1:" line prints, also for an SB1 block
					 * whose SYNC PDU then moves the clock */
};

/* upper-MAC stand-in.  Same contract as upper_mac_prim_recv()
 * (tetra_upper_mac.c:549-566): returns parsed bits or -1.  'offset' is how far
 * msg->head has been advanced (tetra_lower_mac.c:326-352).                */
struct orc_rx;
typedef int (*orc_upper_cb)(struct orc_rx *rx, const struct orc_record *rec, unsigned offset, void *priv);

/* sync events (what the reference prints to stdout/stderr) */
enum orc_sync_event {
	ORC_EV_FOUND_SYNC = 1,	/* "found SYNC training sequence in bit #" */
	ORC_EV_BURST = 2,	/* "BURST" */
	ORC_EV_SYNC_MISPLACED = 3,	/* "#### SYNC burst at offset" from the SYNC branch -> unlock */
	ORC_EV_NORM_MISPLACED = 4,	/* same message from the NORM branch -> stay locked */
	ORC_EV_NO_TRAIN = 5,	/* "#### could not find successive burst training sequence" */
};
typedef void (*orc_event_cb)(int ev, uint32_t bitnum, uint32_t arg, void *priv);

struct orc_rx {
	/* struct tetra_rx_state, phy/tetra_burst_sync.h:12-20 */
	int state;
	unsigned bits_in_buf;
	uint8_t bitbuf[4096];
	unsigned bitbuf_start_bitnum;
	unsigned next_frame_start_bitnum;
	/* t_phy_state, phy/tetra_burst_sync.c:34 */
	struct orc_tdma_time phy_time;
	/* struct tetra_cell_data, lower_mac/tetra_lower_mac.c:104-113 */
	uint16_t mcc, mnc;
	uint8_t colour_code;
	struct orc_tdma_time cell_time;
	uint32_t scramb_init;
	/* tms->cur_burst, tetra_common.h:51-55 */
	int is_traffic;
	int blk1_stolen, blk2_stolen;
	/* plumbing */
	int use_acc;
	uint32_t burst_seq;
	uint8_t cur_burst_type;
	orc_upper_cb upper;
	orc_event_cb event;
	void *priv;
};

void orc_rx_init(struct orc_rx *rx, orc_upper_cb upper, orc_event_cb event, void *priv);
int orc_burst_sync_in(struct orc_rx *rx, const uint8_t *bits, unsigned len);
/* feed a whole buffer in 'chunk'-byte pieces like tetra-rx.c:82-95 (chunk=64) */
void orc_rx_feed(struct orc_rx *rx, const uint8_t *bits, size_t len, unsigned chunk);
/* direct entry points (for tests that bypass the synchroniser) */
void orc_burst_rx_cb(struct orc_rx *rx, const uint8_t *burst, unsigned len, int type);
void orc_tp_sap_udata_ind(struct orc_rx *rx, int type, int blk_num, const uint8_t *bits, unsigned len);

/* exhaustive minimum-distance decoder of the (30,14) code (no counterpart in the reference; checks the product's
 * optional decoder).  rx30 bit 29 = first received bit.  Returns the 14 data bits (bit 13 = first). */
uint16_t orc_rm3014_decode_ml(uint32_t rx30, unsigned *nerr);

/* GSMTAP message of a decoded block (tetra_gsmtap.c:31-63 restated).  PARITY UNPINNED for the header: libosmocore's
 * gsmtap.h is absent here, so GSMTAP_VERSION (2), GSMTAP_TYPE_TETRA_I1 (5), the GSMTAP_TETRA_* sub-types (BSCH 1, AACH 2,
 * SCH_HU 3, SCH_HD 4, SCH_F 5, BNCH 6, STCH 7, TCH_F 8) and the 16-byte struct gsmtap_hdr layout are RECALLED from its
 * published header, not read from a file in this image; the bit packing (osmo_ubit2pbit: MSB first) and the frame-number
 * arithmetic (tetra_tdma.c:96-99) follow files that are here. */
int orc_gsmtap_makemsg(const struct orc_tdma_time *tm, int lchan, uint8_t ts, uint8_t ss, int8_t signal_dbm,
		       uint8_t snr, const uint8_t *bits, unsigned bitlen, uint8_t *out);
/* traffic dump block (tetra_lower_mac.c:213-231): 690 int16 from the descrambled type-4 bits of a traffic block */
void orc_traffic_block(const uint8_t *type4, unsigned len, int16_t *block690);
/* ACELP bit re-ordering (lower_mac/tch_reordering.c:94-140) with caller-supplied class position tables */
void orc_acelp_type2_to_codec(const uint8_t *in, uint8_t *out, const uint8_t *const cls[3], const unsigned ncls[3]);
void orc_acelp_codec_to_acelp(const uint8_t *in, uint8_t *out, const uint8_t *const cls[3], const unsigned ncls[3]);

/* ---- row B: float_to_bits (float_to_bits.c) ---------------------------- */
void orc_float_to_bits(const float *in, size_t n, uint8_t *out2n, int afc,
		       float filter_val, float filter_goal, float *filter_state);

/* ---- soft-input extension (config 5; our definition, the reference has none) ---- */
void orc_float_to_soft(const float *in, size_t n, int8_t *out2n);
void orc_decode_block_soft(enum orc_tpsap_type type, const int8_t *soft5, uint32_t scramb_init,
			   struct orc_block_result *res);

/* ---- CPU baseline: decode n aligned slots of known type, no callbacks -- */
/* returns number of CRC-OK blocks; types[i] is enum orc_train_seq         */
uint64_t orc_bench_decode_slots(const uint8_t *slots, const uint8_t *types, size_t n,
				uint32_t scramb_init, int use_acc, uint8_t *type1_out /* n*288 or NULL */,
				uint16_t *crc_out /* n*2 or NULL */);
uint64_t orc_bench_decode_slots_soft(const int8_t *slots, const uint8_t *types, size_t n, uint32_t scramb_init,
				     uint8_t *type1_out, uint16_t *crc_out);

#ifdef __cplusplus
}
#endif
#endif
