/*
 * tg_sync.c -- host side of the channel API: the burst synchroniser with the exact
 * semantics of the reference, burst queueing, and the in-order delivery ("replay") of
 * GPU-decoded blocks to the upper-MAC callback.
 *
 * Mirrors (reference paths under src/):
 *   tetra_find_train_seq()   phy/tetra_burst.c:269-339   incl. its look-ahead filter quirk
 *   tetra_burst_sync_in()    phy/tetra_burst_sync.c:38-154
 *   tetra_burst_rx_cb()      phy/tetra_burst.c:341-379   (block order)
 *   tp_sap_udata_ind()       lower_mac/tetra_lower_mac.c:143-357  (everything except the
 *                            descramble/deinterleave/Viterbi/CRC arithmetic, which the GPU did)
 *   tetra_tdma_time_add_tn() tetra_tdma.c:27-81
 *
 * What is sequential stays here, per channel and cheap: TDMA time, cell data, the
 * is_traffic / blk2_stolen feedback (SURVEY.md 3.3 loops 2 and 3).  What is arithmetic
 * runs on the GPU for a whole batch of bursts; the GPU decodes every block speculatively
 * and this file decides what is delivered.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tetra_gpu.h"
#include "tg_layout.h"
#include "tg_internal.h"

/* ------------------------------------------------------------------------- */
/* small reference-compatible helpers                                         */
/* ------------------------------------------------------------------------- */
void tetra_tdma_time_add_tn(struct tetra_tdma_time *tm, uint32_t tn_count)
{
	/* the three normalisers always run in this order, tetra_tdma.c:27-58 */
	tm->tn += tn_count;
	if (tm->tn > 4) {
		tm->fn += tm->tn / 4;
		tm->tn %= 4;
	}
	if (tm->fn > 18) {
		tm->mn += tm->fn / 18;
		tm->fn %= 18;
	}
	if (tm->mn > 60)
		tm->mn %= 60;
}

uint32_t tetra_scramb_get_init(uint16_t mcc, uint16_t mnc, uint8_t colour)
{
	uint32_t v = ((uint32_t)(mcc & 0x3ff) << 20) | ((uint32_t)(mnc & 0x3fff) << 6) | (colour & 0x3f);
	return (v << 2) | SCRAMB_INIT;
}

static uint32_t scramb_next(uint32_t *st)
{
	/* taps 32 26 23 22 16 12 11 10 8 7 5 4 2 1 -> state bits 0 6 9 10 16 20 21 22 24 25 27 28 30 31 */
	const uint32_t s = *st;
	const uint32_t fb = __builtin_parity(s & 0xdb710641u);
	*st = (s >> 1) | (fb << 31);
	return fb;
}

/* training sequences, EN 300 392-2 clause 9.4.4.3 */
static const uint8_t ts_n[22] = { 1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0 };
static const uint8_t ts_p[22] = { 0,1,1,1,1,0,1,0,0,1,0,0,0,0,1,1,0,1,1,1,1,0 };
static const uint8_t ts_q[22] = { 1,0,1,1,0,1,1,1,0,0,0,0,0,1,1,0,1,0,1,1,0,1 };
static const uint8_t ts_x[30] = { 1,0,0,1,1,1,0,1,0,0,0,0,1,1,1,0,1,0,0,1,1,1,0,1,0,0,0,0,1,1 };
static const uint8_t ts_y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };

static uint32_t prefix22(const uint8_t *s)
{
	uint32_t v = 0;
	for (int i = 0; i < 22; i++)
		v = (v << 1) | s[i];
	return v;
}

int tetra_find_train_seq(const uint8_t *in, unsigned int end_of_in, uint32_t mask, unsigned int *offset)
{
	/*
	 * The reference gates every position with a 22-bit window compared against the first
	 * 22 bits of y, n, p, q, x.  The window is primed with in[0..19] and then fed in[cur+21],
	 * so it only equals in[cur..cur+21] from cur = 21 on; before that it is the stream with
	 * in[20] missing.  A hit needs gate AND exact memcmp, tested in the order SYNC, NORM_1,
	 * NORM_2, NORM_3, EXT, each only if enough bits remain.  All of that is kept.
	 */
	const uint32_t g0 = prefix22(ts_y), g1 = prefix22(ts_n), g2 = prefix22(ts_p), g3 = prefix22(ts_q),
		       g4 = prefix22(ts_x);
	uint32_t win = 0;
	for (int i = 0; i < 20; i++)
		win = (win << 1) | in[i];
	for (unsigned int cur = 0; cur < end_of_in; cur++) {
		win = ((win << 1) | in[cur + 21]) & 0x3fffffu;
		if (win != g0 && win != g1 && win != g2 && win != g3 && win != g4)
			continue;
		const unsigned int left = end_of_in - cur;
		const uint8_t *p = in + cur;
		int hit = -1;
		if ((mask & (1u << TETRA_TRAIN_SYNC)) && left >= 38 && !memcmp(p, ts_y, 38))
			hit = TETRA_TRAIN_SYNC;
		else if ((mask & (1u << TETRA_TRAIN_NORM_1)) && left >= 22 && !memcmp(p, ts_n, 22))
			hit = TETRA_TRAIN_NORM_1;
		else if ((mask & (1u << TETRA_TRAIN_NORM_2)) && left >= 22 && !memcmp(p, ts_p, 22))
			hit = TETRA_TRAIN_NORM_2;
		else if ((mask & (1u << TETRA_TRAIN_NORM_3)) && left >= 22 && !memcmp(p, ts_q, 22))
			hit = TETRA_TRAIN_NORM_3;
		else if ((mask & (1u << TETRA_TRAIN_EXT)) && left >= 30 && !memcmp(p, ts_x, 30))
			hit = TETRA_TRAIN_EXT;
		if (hit >= 0) {
			*offset = cur;
			return hit;
		}
	}
	return -1;
}

/* ------------------------------------------------------------------------- */
/* channel                                                                    */
/* ------------------------------------------------------------------------- */
#define SLOT_STRIDE 512		/* staging stride: 510 rounded up, keeps slots 16-byte aligned */

struct pending {
	uint32_t burst_seq;
	uint32_t tn_adds;	/* tetra_tdma_time_add_tn(,1) calls since the previously queued burst */
	uint8_t type;
};

struct tgpu_channel {
	struct tgpu_engine *eng;
	struct tgpu_plan *plan;
	uint32_t batch_slots;
	tgpu_unitdata_cb cb;
	tgpu_event_cb ev;
	void *priv;

	/* queue */
	uint32_t n_pending;
	struct pending *pend;
	uint8_t *h_slots;	/* pinned, batch_slots * SLOT_STRIDE */
	uint8_t *h_rec;		/* pinned, batch_slots * TGPU_REC_BYTES */
	uint8_t *d_slots, *d_rec;
	uint64_t *h_off;
	uint8_t *h_type;
	uint32_t *h_chan;
	hipStream_t stream;

	/* what the reference keeps in globals: t_phy_state (phy/tetra_burst_sync.c:34),
	 * tcd (lower_mac/tetra_lower_mac.c:113) */
	struct tetra_tdma_time phy_time, cell_time;
	uint16_t mcc, mnc;
	uint8_t colour_code;
	uint32_t scramb_init;
	uint32_t burst_seq, tn_adds;

	/* tms->cur_burst */
	int loc_is_traffic;
	bool loc_blk1, loc_blk2;
	int *is_traffic;
	bool *blk1_stolen, *blk2_stolen;
	int last_error;
	int zero_copy;		/* h_slots / h_rec are mapped: d_slots / d_rec alias them */

	/* TGPU_OPT_RING: flushes go to workgroups that stay (k_burst_ring) */
	int ring;			/* 1: in use; 0: not asked for, or given up after a failure */
	struct tg_ring_msg *h_ring, *d_ring;	/* mapped host memory: the request line and the kernel's two words back */
	struct tg_ring_box *d_box;
	hipStream_t rstream;		/* the kernel's own stream */
	uint32_t ring_seq, ring_launches;
	int ring_fails;		/* flushes in a row the ring did not answer (ring_failed()) */
	int ring_fail_score;	/* + RING_FAIL_COST per unanswered flush, - 1 per answered one: failures over a sliding count */
	int ring_counted;		/* this channel is one of RING_MAX_CHANNELS */

	/* block queue of the tp_sap_udata_ind() seam (allocated on first use) */
	struct tgpu_plan *bplan;
	uint32_t bq_cap, bq_n;
	struct bq_item *bq;
	uint8_t *bq_bits;	/* pinned, bq_cap * BQ_STRIDE */
	uint8_t *bq_rec;	/* pinned, bq_cap * TGPU_REC_BYTES */
	uint8_t *d_bq_bits, *d_bq_rec;
	uint64_t *bq_off;
	uint8_t *bq_type;
	uint32_t *bq_code;
	uint8_t bq_burst, bq_burst_known;	/* burst type of the blocks tetra_burst_rx_cb() is handing over right now */
};

#define BQ_STRIDE 432u

struct bq_item {
	uint8_t type;			/* enum tp_sap_data_type */
	uint8_t blk_num;
	uint16_t len;
	uint8_t burst_type;		/* enum tetra_train_seq of the burst the block came in (see bq_burst_type) */
	uint32_t burst_seq;
	struct tetra_tdma_time time;	/* t_phy_state.time when the block was handed over (tetra_lower_mac.c:167) */
};

/* ---- TGPU_OPT_RING: the decoder workgroups that stay (k_burst_ring; protocol in tg_k_trellis.hip) ---- */
#define RING_IDLE_TICKS 2000000ull	/* 20 ms of the device's 100 MHz clock without a request: the workgroups leave */
#define RING_MAX_CHANNELS 32		/* channels of a process that may hold workgroups at a time (each up to four, 43 KB of LDS
					 * apiece, for as long as its flushes keep coming): the others flush by launch */
static int ring_channels;

static void ring_post(struct tg_ring_msg *m, uint32_t seq, uint32_t n, uint32_t have_sync, uint32_t code, const uint64_t *desc)
{
	/* either half of the line carries the request number behind its own fields (k_burst_ring takes the line in one load and
	 * believes a half whose number is new) */
	volatile struct tg_ring_msg *v = m;
	v->desc[2] = n > 2 ? desc[2] : 0;
	v->desc[3] = n > 3 ? desc[3] : 0;
	__atomic_store_n(&m->req2, seq, __ATOMIC_RELEASE);
	v->n = n;
	v->have_sync = have_sync;
	v->code = code;
	v->desc[0] = desc[0];
	v->desc[1] = n > 1 ? desc[1] : 0;
	__atomic_store_n(&m->req, seq, __ATOMIC_RELEASE);
}

static int ring_start(struct tgpu_channel *ch)
{
	uint32_t *sb_ok, *sb_code, *maskidx, *masks;
	int rc = tgpi_engine_bind(ch->eng);
	if (!rc)
		rc = tgpi_plan_ring(ch->plan, &sb_ok, &sb_code, &maskidx, &masks);
	if (rc)
		return rc;
	__atomic_store_n(&ch->h_ring->alive, 1u, __ATOMIC_RELEASE);
	rc = tgk_burst_ring(ch->d_ring, ch->d_box, ch->d_slots, ch->batch_slots, sb_ok, sb_code, ch->d_rec, maskidx, masks,
			    __atomic_load_n(&ch->h_ring->served, __ATOMIC_ACQUIRE), ++ch->ring_launches, RING_IDLE_TICKS, ch->rstream);
	if (rc)
		__atomic_store_n(&ch->h_ring->alive, 0u, __ATOMIC_RELEASE);
	return rc;
}

/* make the workgroups leave and wait until they have (at most one poll of theirs away) */
static void ring_stop(struct tgpu_channel *ch)
{
	if (!ch->h_ring || !ch->rstream)
		return;
	if (__atomic_load_n(&ch->h_ring->alive, __ATOMIC_ACQUIRE)) {
		static const uint64_t none[TG_RING_MAX] = { 0 };
		ring_post(ch->h_ring, TG_RING_STOP, 0, 0, 0, none);
	}
	(void)hipStreamSynchronize(ch->rstream);
	ch->ring = 0;
}

/* a flush the ring did not answer in time.  The protocol assumes that all of a channel's workgroups (up to four, 43 KB of LDS
 * each) are RESIDENT together -- they meet at a spin barrier -- which a GPU saturated by other work does not promise: the
 * workgroups are told to leave, this batch goes by launch, and the next flush starts them again.  Only a ring that fails
 * RING_MAX_FAILS flushes in a row is given up for good. */
#define RING_MAX_FAILS 4
/* ... and a ring that keeps alternating between answering and not (a busy GPU: every failure costs the 20 ms wait, a stream
 * synchronise and a relaunch) is bounded over a sliding count as well: ring_fail_score gains RING_FAIL_COST per failure, loses 1 per
 * answered flush, and the ring is given up when it passes RING_FAIL_LIMIT -- i.e. a steady failure rate above 1 in
 * RING_FAIL_COST + 1 flushes ends it after a few dozen flushes, a rare failure never does. */
#define RING_FAIL_COST  16
#define RING_FAIL_LIMIT 64
static void ring_failed(struct tgpu_channel *ch)
{
	ch->ring_fail_score += RING_FAIL_COST;
	const int give_up = ++ch->ring_fails >= RING_MAX_FAILS || ch->ring_fail_score > RING_FAIL_LIMIT;
	ring_stop(ch);
	if (!give_up)
		ch->ring = 1;
}

/* one flush through the ring: 1 = every record complete; 0 = given up (the caller decodes the batch the ordinary way) */
static int ring_flush(struct tgpu_channel *ch, uint32_t n)
{
	uint64_t desc[TG_RING_MAX] = { 0 };
	uint32_t hs = 0;
	for (uint32_t i = 0; i < n; i++) {
		desc[i] = ch->h_off[i] | ((uint64_t)ch->h_type[i] << 56);
		hs |= ch->h_type[i] == TETRA_TRAIN_SYNC;
	}
	if (++ch->ring_seq == TG_RING_STOP || !ch->ring_seq)
		ch->ring_seq = 1;
	ring_post(ch->h_ring, ch->ring_seq, n, hs, ch->scramb_init, desc);
	int starts = 0;
	struct timespec t0, t;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (uint32_t i = 0; i < n; i++) {
		const volatile uint8_t *mark = ch->h_rec + (size_t)i * TGPU_REC_BYTES + TG_REC_TYPE;
		for (unsigned spins = 0; *mark == TG_REC_PENDING; spins++) {
			if ((spins & 255) == 0 && !__atomic_load_n(&ch->h_ring->alive, __ATOMIC_ACQUIRE) &&
			    __atomic_load_n(&ch->h_ring->served, __ATOMIC_ACQUIRE) != ch->ring_seq) {
				/* nobody there (the first flush, or the workgroups left while idle): start them; they begin behind
				 * `served`, i.e. with this request */
				if (starts++ == 3 || ring_start(ch))
					return 0;
			}
			if ((spins & 1023) == 1023) {
				clock_gettime(CLOCK_MONOTONIC, &t);
				if ((t.tv_sec - t0.tv_sec) * 1000000000L + (t.tv_nsec - t0.tv_nsec) > 20000000L)
					return 0;
			}
#if defined(__x86_64__) || defined(__i386__)
			__builtin_ia32_pause();
#endif
		}
	}
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	return 1;
}

int tgpu_channel_create(struct tgpu_engine *eng, uint32_t batch_slots, tgpu_unitdata_cb cb, tgpu_event_cb ev,
			void *priv, struct tgpu_channel **out)
{
	if (!eng || !out || !batch_slots)
		return TGPU_EINVAL;
	int brc = tgpi_engine_bind(eng);
	if (brc)
		return brc;
	struct tgpu_channel *ch = calloc(1, sizeof(*ch));
	if (!ch)
		return TGPU_ENOMEM;
	ch->eng = eng;
	ch->batch_slots = batch_slots;
	ch->cb = cb;
	ch->ev = ev;
	ch->priv = priv;
	ch->is_traffic = &ch->loc_is_traffic;
	ch->blk1_stolen = &ch->loc_blk1;
	ch->blk2_stolen = &ch->loc_blk2;
	int rc = tgpu_plan_create(eng, batch_slots, 1, &ch->plan);
	if (rc) {
		free(ch);
		return rc;
	}
	const size_t n = batch_slots;
	hipError_t e = hipSuccess;
	ch->pend = calloc(n, sizeof(*ch->pend));
	ch->h_off = calloc(n, sizeof(uint64_t));
	ch->h_type = calloc(n, 1);
	ch->h_chan = calloc(n, 4);
	/* small batches are round trips: the bursts stay in pinned host memory, the kernels read them and write the
	 * records in place over PCIe (no copy operations in a flush); larger ones go through device buffers */
	ch->zero_copy = batch_slots <= 64;
	if (e == hipSuccess) e = hipHostMalloc((void **)&ch->h_slots, n * SLOT_STRIDE + 64, ch->zero_copy ? (hipHostMallocMapped | hipHostMallocCoherent) : 0);
	if (e == hipSuccess) e = hipHostMalloc((void **)&ch->h_rec, n * TGPU_REC_BYTES, ch->zero_copy ? (hipHostMallocMapped | hipHostMallocCoherent) : 0);
	if (ch->zero_copy) {
		tgpi_plan_set_marks(ch->plan, 1);	/* records in mapped memory: wait_records() polls their completion marks */
		if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ch->d_slots, ch->h_slots, 0);
		if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ch->d_rec, ch->h_rec, 0);
	} else {
		if (e == hipSuccess) e = hipMalloc((void **)&ch->d_slots, n * SLOT_STRIDE + 64);
		if (e == hipSuccess) e = hipMalloc((void **)&ch->d_rec, n * TGPU_REC_BYTES);
	}
	if (e == hipSuccess) e = hipStreamCreate(&ch->stream);
	if (e == hipSuccess && ch->zero_copy && batch_slots <= TG_RING_MAX && tgi_option(TGPU_OPT_RING) &&
	    __atomic_add_fetch(&ring_channels, 1, __ATOMIC_RELAXED) > RING_MAX_CHANNELS)
		__atomic_sub_fetch(&ring_channels, 1, __ATOMIC_RELAXED);
	else if (e == hipSuccess && ch->zero_copy && batch_slots <= TG_RING_MAX && tgi_option(TGPU_OPT_RING)) {
		ch->ring_counted = 1;
		e = hipHostMalloc((void **)&ch->h_ring, sizeof(*ch->h_ring), hipHostMallocMapped | hipHostMallocCoherent);
		if (e == hipSuccess) {
			memset(ch->h_ring, 0, sizeof(*ch->h_ring));
			e = hipHostGetDevicePointer((void **)&ch->d_ring, ch->h_ring, 0);
		}
		if (e == hipSuccess) e = hipMalloc((void **)&ch->d_box, sizeof(*ch->d_box));
		if (e == hipSuccess) e = hipMemset(ch->d_box, 0, sizeof(*ch->d_box));
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&ch->rstream, hipStreamNonBlocking);
		ch->ring = e == hipSuccess;
	}
	if (e != hipSuccess || !ch->pend || !ch->h_off || !ch->h_type || !ch->h_chan) {
		tgpu_channel_destroy(ch);
		return e != hipSuccess ? (int)e : TGPU_ENOMEM;
	}
	for (size_t i = 0; i < n; i++)
		ch->h_off[i] = i * SLOT_STRIDE;
	*out = ch;
	return TGPU_OK;
}

void tgpu_channel_destroy(struct tgpu_channel *ch)
{
	if (!ch)
		return;
	ring_stop(ch);
	if (ch->ring_counted)
		__atomic_sub_fetch(&ring_channels, 1, __ATOMIC_RELAXED);
	if (ch->rstream) (void)hipStreamDestroy(ch->rstream);
	if (ch->d_box) (void)hipFree(ch->d_box);
	if (ch->h_ring) (void)hipHostFree(ch->h_ring);
	if (ch->stream) (void)hipStreamDestroy(ch->stream);
	if (ch->h_slots) (void)hipHostFree(ch->h_slots);
	if (ch->h_rec) (void)hipHostFree(ch->h_rec);
	if (ch->d_slots && !ch->zero_copy) (void)hipFree(ch->d_slots);
	if (ch->d_rec && !ch->zero_copy) (void)hipFree(ch->d_rec);
	tgpu_plan_destroy(ch->plan);
	if (ch->bq_bits) (void)hipHostFree(ch->bq_bits);
	if (ch->bq_rec) (void)hipHostFree(ch->bq_rec);
	if (ch->d_bq_bits) (void)hipFree(ch->d_bq_bits);
	if (ch->d_bq_rec) (void)hipFree(ch->d_bq_rec);
	if (ch->bplan) tgpu_plan_destroy(ch->bplan);
	free(ch->bq);
	free(ch->bq_off);
	free(ch->bq_type);
	free(ch->bq_code);
	free(ch->pend);
	free(ch->h_off);
	free(ch->h_type);
	free(ch->h_chan);
	free(ch);
}

void tgpu_channel_bind_flags(struct tgpu_channel *ch, int *is_traffic, bool *blk1_stolen, bool *blk2_stolen)
{
	ch->is_traffic = is_traffic ? is_traffic : &ch->loc_is_traffic;
	ch->blk1_stolen = blk1_stolen ? blk1_stolen : &ch->loc_blk1;
	ch->blk2_stolen = blk2_stolen ? blk2_stolen : &ch->loc_blk2;
}

int tgpu_channel_set_rm_decode(struct tgpu_channel *ch, int on)
{
	return ch ? tgpu_plan_set_rm_decode(ch->plan, on) : TGPU_EINVAL;
}

void tgpu_channel_set_traffic(struct tgpu_channel *ch, int is_traffic)
{
	*ch->is_traffic = is_traffic;
}

void tgpu_channel_set_blk2_stolen(struct tgpu_channel *ch, bool stolen)
{
	*ch->blk2_stolen = stolen;
}

/* ------------------------------------------------------------------------- */
/* delivery: the non-arithmetic part of tp_sap_udata_ind()                    */
/* ------------------------------------------------------------------------- */
static int is_bnch(const struct tetra_tdma_time *tm)
{
	return tm->fn == 18 && tm->tn == 4 - ((tm->mn + 3) % 4);	/* lower_mac/tetra_lower_mac.c:122-127 */
}

/* raw type-5 bits of a block inside a queued slot (phy/tetra_burst.c:341-379) */
static unsigned gather_type5(const uint8_t *slot, int btype, enum tp_sap_data_type t, int blk_num, uint8_t *out)
{
	switch (t) {
	case TPSAP_T_SB1:
		memcpy(out, slot + TG_SB_BLK1_OFF, 120);
		return 120;
	case TPSAP_T_SB2:
		memcpy(out, slot + TG_SB_BLK2_OFF, 216);
		return 216;
	case TPSAP_T_NDB:
		memcpy(out, slot + (blk_num == BLK_1 ? TG_NDB_BLK1_OFF : TG_NDB_BLK2_OFF), 216);
		return 216;
	case TPSAP_T_SCH_F:
		memcpy(out, slot + TG_NDB_BLK1_OFF, 216);
		memcpy(out + 216, slot + TG_NDB_BLK2_OFF, 216);
		return 432;
	case TPSAP_T_BBK:
		if (btype == TETRA_TRAIN_SYNC) {
			memcpy(out, slot + TG_SB_BBK_OFF, 30);
		} else {
			memcpy(out, slot + TG_NDB_BBK1_OFF, 14);
			memcpy(out + 14, slot + TG_NDB_BBK2_OFF, 16);
		}
		return 30;
	default:
		return 0;
	}
}

/* SYNC-PDU fields of a record (slot mode: a SYNC burst's record; block mode: an SB1 block's record) */
static void sync_info_of(const uint8_t *rec, struct tgpu_sync_info *out)
{
	uint32_t f0, f1, code;
	memcpy(&f0, rec + TG_REC_SBF0, 4);
	memcpy(&f1, rec + TG_REC_SBF1, 4);
	memcpy(&code, rec + TG_REC_SBCODE, 4);
	out->cc = (uint8_t)f0;
	out->tn = (uint8_t)(f0 >> 8);
	out->fn = (uint8_t)(f0 >> 16);
	out->mn = (uint8_t)(f0 >> 24);
	out->mcc = (uint16_t)f1;
	out->mnc = (uint16_t)(f1 >> 16);
	out->scramb_init = code;
}

/* slot != NULL: the block sits in a queued burst; else raw5 / nraw5 are the block's own type-5 bits (block queue).
 * phy_time: the channel's clock, or the reference's global t_phy_state.time for the tp_sap_udata_ind() seam. */
static void deliver_block(struct tgpu_channel *ch, const struct pending *pd, const uint8_t *slot,
			  const uint8_t *rec, const struct tgpu_block *b, const uint8_t *raw5, unsigned nraw5,
			  struct tetra_tdma_time *phy_time)
{
	struct tgpu_unitdata ud;
	uint8_t type4[432];

	memset(&ud, 0, sizeof(ud));
	ud.type = b->type;
	ud.blk_num = b->blk_num;
	ud.burst_seq = pd->burst_seq;
	ud.burst_type = (enum tetra_train_seq)pd->type;
	ud.lchan = TETRA_LC_UNKNOWN;

	ch->cell_time = *phy_time;							/* :167 */
	ud.time_str = ch->cell_time;							/* :168 */
	if (b->type == TPSAP_T_SB2 && is_bnch(&ch->cell_time))				/* :170-173 */
		ud.lchan = TETRA_LC_BNCH;

	ud.scrambling_code = (b->type == TPSAP_T_SB1) ? SCRAMB_INIT : ch->scramb_init;	/* :179-186 */

	if (*ch->is_traffic && b->type == TPSAP_T_NDB && b->blk_num == BLK_1)		/* :194-195 */
		*ch->blk1_stolen = true;

	if (*ch->is_traffic && (b->type == TPSAP_T_SCH_F || (b->blk_num == BLK_2 && !*ch->blk2_stolen))) {
		/* :198-241 -- traffic block: not decoded, handed over as descrambled type-4 bits */
		unsigned n;
		if (slot)
			n = gather_type5(slot, pd->type, b->type, b->blk_num, type4);
		else {
			n = nraw5 < 432 ? nraw5 : 432;
			memcpy(type4, raw5, n);
		}
		uint32_t st = ud.scrambling_code;
		for (unsigned i = 0; i < n; i++)
			type4[i] ^= (uint8_t)scramb_next(&st);
		ud.traffic = *ch->is_traffic;
		ud.type4 = type4;
		ud.type4_len = (uint16_t)n;
		ud.tdma_time = ch->cell_time;
		if (ch->cb)
			ch->cb(&ud, 0xffffffffu, ch->priv);
		return;
	}

	ud.crc_ok = b->crc_ok;
	ud.crc = b->crc;
	ud.type1_len = b->type1_len;
	ud.type1 = b->type1;

	switch (b->type) {
	case TPSAP_T_SB1: {								/* :283-310 */
		if (ud.crc_ok) {
			struct tgpu_sync_info si;
			sync_info_of(rec, &si);
			ch->colour_code = si.cc;
			ch->cell_time.tn = si.tn;
			ch->cell_time.fn = si.fn;
			ch->cell_time.mn = si.mn;
			ch->mcc = si.mcc;
			ch->mnc = si.mnc;
			ch->scramb_init = si.scramb_init;
		}
		*phy_time = ch->cell_time;						/* :302 */
		ud.lchan = TETRA_LC_BSCH;
		break;
	}
	case TPSAP_T_BBK:
		ud.lchan = TETRA_LC_AACH;
		break;
	case TPSAP_T_SCH_F:
		ud.lchan = TETRA_LC_SCH_F;
		break;
	default:
		break;
	}

	/* :326-352; limit computed like the reference: (int)type1_bits - 16 seen as unsigned */
	uint32_t offset = 0;
	const uint32_t limit = (uint32_t)((int)b->type1_len - 16);
	while (offset < limit) {
		ud.tdma_time = ch->cell_time;
		int n = ch->cb ? ch->cb(&ud, offset, ch->priv) : -1;
		if (n <= 0)	/* the reference never terminates on 0; we stop */
			break;
		offset += (uint32_t)n;
	}
}

static int flush_blocks(struct tgpu_channel *ch);

static int channel_fail(struct tgpu_channel *ch, int rc, uint32_t arg)
{
	ch->last_error = rc;
	if (ch->ev)
		ch->ev(TGPU_EV_ERROR, (uint32_t)rc, arg, ch->priv);
	return rc;
}

/* decode the queued bursts on the GPU; on failure the batch is given up, but not silently: its time steps still
 * advance the channel's clock (later bursts keep the reference's TDMA time), last_error stays set until
 * tgpu_channel_clear_error(), and the event callback gets TGPU_EV_ERROR(error, bursts lost) */
/*
 * Small batches in mapped memory: k_burst writes a record's burst type last, after a system-wide fence, so the host can
 * watch the bytes it preset to TG_REC_PENDING instead of calling hipStreamSynchronize() (~5 us less per flush,
 * tools/ubench/sync_lat.hip).  Returns 1 when every record is complete, 0 after ~2 ms without (the caller then
 * synchronises the stream the ordinary way, which also surfaces a launch error).
 */
static int wait_records(struct tgpu_channel *ch, uint32_t n)
{
	struct timespec t0, t;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (uint32_t i = 0; i < n; i++) {
		const volatile uint8_t *mark = ch->h_rec + (size_t)i * TGPU_REC_BYTES + TG_REC_TYPE;
		for (unsigned spins = 0; *mark == TG_REC_PENDING; spins++) {
			if ((spins & 1023) == 1023) {
				clock_gettime(CLOCK_MONOTONIC, &t);
				if ((t.tv_sec - t0.tv_sec) * 1000000000L + (t.tv_nsec - t0.tv_nsec) > 2000000L)
					return 0;
			}
#if defined(__x86_64__) || defined(__i386__)
			__builtin_ia32_pause();
#endif
		}
	}
	__atomic_thread_fence(__ATOMIC_ACQUIRE);
	return 1;
}

static int flush_slots(struct tgpu_channel *ch)
{
	const uint32_t n = ch->n_pending;
	if (!n)
		return TGPU_OK;
	hipError_t e = hipSuccess;
	int rc = TGPU_OK;
	for (uint32_t i = 0; i < n; i++)
		ch->h_type[i] = ch->pend[i].type;
	if (ch->zero_copy)		/* completion marks: see wait_records() */
		for (uint32_t i = 0; i < n; i++)
			ch->h_rec[(size_t)i * TGPU_REC_BYTES + TG_REC_TYPE] = TG_REC_PENDING;
	int ringed = 0;
	if (ch->ring && n <= TG_RING_MAX) {	/* (the workgroups that stay need nothing of the plan but its scratch arrays: no load) */
		ringed = ring_flush(ch, n);
		if (ringed) {
			ch->ring_fails = 0;
			if (ch->ring_fail_score > 0)
				ch->ring_fail_score--;
		}
		if (!ringed) {		/* the ring did not answer: this batch by launch (ring_failed() says what happens next) */
			ring_failed(ch);
			for (uint32_t i = 0; i < n; i++)
				ch->h_rec[(size_t)i * TGPU_REC_BYTES + TG_REC_TYPE] = TG_REC_PENDING;
		}
	}
	if (!ringed) {
		if ((rc = tgpi_engine_bind(ch->eng)))
			return channel_fail(ch, rc, n);
		rc = tgpu_plan_load(ch->plan, n, ch->h_off, ch->h_type, ch->h_chan, 1, &ch->scramb_init);
	}
	if (!rc && !ch->zero_copy &&
	    (e = hipMemcpyAsync(ch->d_slots, ch->h_slots, (size_t)n * SLOT_STRIDE, hipMemcpyHostToDevice, ch->stream)))
		rc = (int)e;
	if (!rc && !ringed)
		rc = tgpu_plan_execute(ch->plan, ch->d_slots, ch->d_rec, ch->stream);
	if (!rc && !ch->zero_copy &&
	    (e = hipMemcpyAsync(ch->h_rec, ch->d_rec, (size_t)n * TGPU_REC_BYTES, hipMemcpyDeviceToHost, ch->stream)))
		rc = (int)e;
	if (!rc && !ringed && !(ch->zero_copy && tgpi_plan_last_burst(ch->plan) && wait_records(ch, n)) &&
	    (e = hipStreamSynchronize(ch->stream)))
		rc = (int)e;
	ch->n_pending = 0;
	if (rc) {
		for (uint32_t i = 0; i < n; i++)
			tetra_tdma_time_add_tn(&ch->phy_time, ch->pend[i].tn_adds);
		return channel_fail(ch, rc, n);
	}
	for (uint32_t i = 0; i < n; i++) {
		const struct pending *pd = &ch->pend[i];
		const uint8_t *rec = ch->h_rec + (size_t)i * TGPU_REC_BYTES;
		const uint8_t *slot = ch->h_slots + (size_t)i * SLOT_STRIDE;
		struct tgpu_block blk[3];
		for (uint32_t k = 0; k < pd->tn_adds; k++)
			tetra_tdma_time_add_tn(&ch->phy_time, 1);			/* phy/tetra_burst_sync.c:113 */
		int nb = tgpu_record_blocks(rec, blk);
		for (int k = 0; k < nb; k++)
			deliver_block(ch, pd, slot, rec, &blk[k], NULL, 0, &ch->phy_time);
	}
	return TGPU_OK;
}

int tgpu_channel_flush(struct tgpu_channel *ch)
{
	if (!ch)
		return TGPU_EINVAL;
	int rc = flush_slots(ch);
	int rc2 = flush_blocks(ch);
	if (rc || rc2)
		return rc ? rc : rc2;
	return ch->last_error;		/* sticky: a batch an earlier automatic flush lost is reported here too */
}

int tgpu_channel_last_error(const struct tgpu_channel *ch)
{
	return ch ? ch->last_error : TGPU_EINVAL;
}

void tgpu_channel_clear_error(struct tgpu_channel *ch)
{
	if (ch)
		ch->last_error = TGPU_OK;
}

int tgpu_channel_scramb_init(const struct tgpu_channel *ch, uint32_t *code)
{
	if (!ch || !code)
		return TGPU_EINVAL;
	*code = ch->scramb_init;
	return TGPU_OK;
}

/* deliver records that were decoded outside the channel's own queue (stream / plan API): the same
 * in-order replay as tgpu_channel_flush(), slot bytes taken from the host copy of the stream */
int tgpu_channel_deliver(struct tgpu_channel *ch, uint32_t n, const struct tgpu_sync_slot *slots,
			 const uint8_t *h_stream, const uint8_t *h_rec)
{
	if (!ch || (n && (!slots || !h_stream || !h_rec)))
		return TGPU_EINVAL;
	for (uint32_t i = 0; i < n; i++) {
		struct pending pd = { slots[i].burst_seq, slots[i].tn_adds, slots[i].type };
		const uint8_t *rec = h_rec + (size_t)i * TGPU_REC_BYTES;
		struct tgpu_block blk[3];
		for (uint32_t k = 0; k < pd.tn_adds; k++)
			tetra_tdma_time_add_tn(&ch->phy_time, 1);
		int nb = tgpu_record_blocks(rec, blk);
		for (int k = 0; k < nb; k++)
			deliver_block(ch, &pd, h_stream + slots[i].off, rec, &blk[k], NULL, 0, &ch->phy_time);
	}
	return TGPU_OK;
}

static void queue_burst(struct tgpu_channel *ch, const uint8_t *burst, int type)
{
	struct pending *pd = &ch->pend[ch->n_pending];
	pd->burst_seq = ch->burst_seq;
	pd->tn_adds = ch->tn_adds;
	pd->type = (uint8_t)type;
	ch->tn_adds = 0;
	memcpy(ch->h_slots + (size_t)ch->n_pending * SLOT_STRIDE, burst, TG_SLOT_BITS);
	ch->n_pending++;
	if (ch->n_pending >= ch->batch_slots)
		(void)flush_slots(ch);	/* a failure is kept in last_error and reported through TGPU_EV_ERROR */
}

/* ------------------------------------------------------------------------- */
/* traffic dump block: lower_mac/tetra_lower_mac.c:213-231                     */
/* ------------------------------------------------------------------------- */
void tgpu_traffic_block(const uint8_t *type4, unsigned int len, int16_t out[690])
{
	/* six 115-word frames: marker 0x6b21 + i, then 114 soft bits (the last frame carries 90);
	 * bit 1 -> -127, bit 0 -> +127; everything else 0 */
	static const uint16_t first[4] = { 1, 116, 231, 346 }, from[4] = { 0, 114, 228, 342 }, count[4] = { 114, 114, 114, 90 };
	memset(out, 0, 690 * sizeof(int16_t));
	for (int i = 0; i < 6; i++)
		out[115 * i] = (int16_t)(0x6b21 + i);
	for (int f = 0; f < 4; f++)
		for (unsigned i = 0; i < count[f]; i++) {
			const unsigned k = from[f] + i;
			/* a 216-bit half-slot block leaves bits 216..431 unwritten in the reference (its type4[] is an
			 * uninitialised local there); they are emitted as bit 0 */
			out[first[f] + i] = (k < len && type4[k]) ? -127 : 127;
		}
}

/* ------------------------------------------------------------------------- */
/* the tetra_burst_rx_cb() seam: phy/tetra_burst.c:341-379                     */
/* ------------------------------------------------------------------------- */
int tgpu_channel_burst_rx(struct tgpu_channel *ch, const uint8_t *burst, unsigned int len, int type,
			  uint32_t tn_steps)
{
	if (!ch || !burst || len < TG_SLOT_BITS)
		return TGPU_EINVAL;
	ch->tn_adds += tn_steps;
	ch->burst_seq += tn_steps;	/* ordinal of the LOCKED step, as in tetra_burst_sync_in() below */
	/* NORM_3 / EXT bursts are ignored like the reference's switch (:374-377); their time steps count */
	if (type == TETRA_TRAIN_SYNC || type == TETRA_TRAIN_NORM_1 || type == TETRA_TRAIN_NORM_2)
		queue_burst(ch, burst, type);
	return ch->last_error ? ch->last_error : TGPU_OK;
}

/* ------------------------------------------------------------------------- */
/* tp_sap_udata_ind() / tetra_burst_rx_cb() under the reference's own signatures */
/* ------------------------------------------------------------------------- */
/*
 * libosmo-tetra-phy.a (phy/tetra_burst_sync.o + phy/tetra_burst.o) references two symbols it does not define:
 * tp_sap_udata_ind (phy/tetra_burst.h:18) and tetra_tdma_time_add_tn.  Both are exported here, so a host can keep the
 * reference's PHY objects unchanged and link this library in place of libosmo-tetra-mac.a: priv = the
 * struct tgpu_channel * (where the reference passes tms).  The reference's lower MAC reads and, after a good SYNC
 * PDU, writes the global t_phy_state.time (tetra_lower_mac.c:167,302), which phy/tetra_burst_sync.c:34 defines; the
 * definition below is weak, i.e. the PHY object's own one is used when it is linked, and this one when the host has
 * no such object.
 *
 * Blocks are queued and decoded in batches (block mode of the plan: tgpu_plan_load_blocks).  An SB1 block ends its
 * batch and is decoded before tp_sap_udata_ind() returns: the scrambling code and the time it brings are in force
 * for the very next block (the BBK and SB2 of the same burst), exactly as in the reference.  Everything else is
 * delivered when the queue is full, with the next SB1, or by tgpu_channel_flush().
 */
struct tetra_phy_state t_phy_state __attribute__((weak));

static void bq_free(struct tgpu_channel *ch)
{
	free(ch->bq);
	free(ch->bq_off);
	free(ch->bq_type);
	free(ch->bq_code);
	if (ch->bq_bits) (void)hipHostFree(ch->bq_bits);
	if (ch->bq_rec) (void)hipHostFree(ch->bq_rec);
	if (ch->d_bq_bits) (void)hipFree(ch->d_bq_bits);
	if (ch->d_bq_rec) (void)hipFree(ch->d_bq_rec);
	if (ch->bplan) tgpu_plan_destroy(ch->bplan);
	ch->bq = NULL;
	ch->bq_off = NULL;
	ch->bq_type = NULL;
	ch->bq_code = NULL;
	ch->bq_bits = ch->bq_rec = ch->d_bq_bits = ch->d_bq_rec = NULL;
	ch->bplan = NULL;
	ch->bq_cap = 0;
}

/* all or nothing: bq_cap is set last and is what the callers test, a partial allocation is taken back so that the
 * next call starts from scratch */
static int bq_alloc(struct tgpu_channel *ch)
{
	if (ch->bq_cap)
		return TGPU_OK;
	int rc = tgpi_engine_bind(ch->eng);
	if (rc)
		return rc;
	const size_t n = (size_t)ch->batch_slots * 3 + 3;
	rc = tgpu_plan_create(ch->eng, (uint32_t)n, 4, &ch->bplan);
	if (rc) {
		ch->bplan = NULL;
		return rc;
	}
	hipError_t e = hipSuccess;
	ch->bq = calloc(n, sizeof(*ch->bq));
	ch->bq_off = calloc(n, sizeof(uint64_t));
	ch->bq_type = calloc(n, 1);
	ch->bq_code = calloc(n, 4);
	if (e == hipSuccess) e = hipHostMalloc((void **)&ch->bq_bits, n * BQ_STRIDE, 0);
	if (e == hipSuccess) e = hipHostMalloc((void **)&ch->bq_rec, n * TGPU_REC_BYTES, 0);
	if (e == hipSuccess) e = hipMalloc((void **)&ch->d_bq_bits, n * BQ_STRIDE);
	if (e == hipSuccess) e = hipMalloc((void **)&ch->d_bq_rec, n * TGPU_REC_BYTES);
	if (e != hipSuccess || !ch->bq || !ch->bq_off || !ch->bq_type || !ch->bq_code) {
		bq_free(ch);
		return e != hipSuccess ? (int)e : TGPU_ENOMEM;
	}
	for (size_t i = 0; i < n; i++)
		ch->bq_off[i] = i * BQ_STRIDE;
	ch->bq_cap = (uint32_t)n;
	return TGPU_OK;
}

static int flush_blocks(struct tgpu_channel *ch)
{
	const uint32_t n = ch->bq_n;
	if (!n)
		return TGPU_OK;
	hipError_t e = hipSuccess;
	{
		const int brc = tgpi_engine_bind(ch->eng);	/* the calling thread may have another device current */
		if (brc) {
			ch->bq_n = 0;
			return channel_fail(ch, brc, n);
		}
	}
	for (uint32_t i = 0; i < n; i++) {
		ch->bq_type[i] = ch->bq[i].type;
		ch->bq_code[i] = ch->scramb_init;	/* the code in force: an SB1 only ever ends a batch */
	}
	int rc = tgpu_plan_load_blocks(ch->bplan, n, ch->bq_off, ch->bq_type, ch->bq_code);
	if (!rc && (e = hipMemcpyAsync(ch->d_bq_bits, ch->bq_bits, (size_t)n * BQ_STRIDE, hipMemcpyHostToDevice, ch->stream)))
		rc = (int)e;
	if (!rc)
		rc = tgpu_plan_execute(ch->bplan, ch->d_bq_bits, ch->d_bq_rec, ch->stream);
	if (!rc && (e = hipMemcpyAsync(ch->bq_rec, ch->d_bq_rec, (size_t)n * TGPU_REC_BYTES, hipMemcpyDeviceToHost, ch->stream)))
		rc = (int)e;
	if (!rc && (e = hipStreamSynchronize(ch->stream)))
		rc = (int)e;
	ch->bq_n = 0;
	if (rc)
		return channel_fail(ch, rc, n);
	static const uint16_t t1len[6] = { 60, 124, 124, 14, 92, 268 };
	for (uint32_t i = 0; i < n; i++) {
		const struct bq_item *it = &ch->bq[i];
		const uint8_t *rec = ch->bq_rec + (size_t)i * TGPU_REC_BYTES;
		struct tgpu_block b;
		struct pending pd = { it->burst_seq, 0, it->burst_type };
		memset(&b, 0, sizeof(b));
		b.type = (enum tp_sap_data_type)it->type;
		b.blk_num = it->blk_num;
		b.crc_ok = rec[TG_REC_CRC_OK];
		memcpy(&b.crc, rec + TG_REC_CRC, 2);
		memcpy(&b.scrambling_code, rec + TG_REC_CODE, 4);
		b.type1_len = t1len[it->type];
		b.type1 = rec + (it->type == TPSAP_T_BBK ? TG_REC_BBK : TG_REC_BITS1);
		struct tetra_tdma_time tm = it->time;
		deliver_block(ch, &pd, NULL, rec, &b, ch->bq_bits + (size_t)i * BQ_STRIDE, it->len, &tm);
		if (it->type == TPSAP_T_SB1)
			t_phy_state.time = tm;		/* tetra_lower_mac.c:302 (deliver_block wrote it on a good CRC) */
	}
	return TGPU_OK;
}

/* burst type of a block handed over on its own: tetra_burst_rx_cb() below knows it and leaves it in ch->bq_burst for the
 * blocks it hands over; for a bare tp_sap_udata_ind() call it follows from the block type where that is unambiguous
 * (phy/tetra_burst.c:350-372: SB1 / SB2 only come in SYNC bursts, SCH/F in NORM_1, NDB halves in NORM_2), and is
 * TGPU_BURST_UNKNOWN (0xff) for a BBK or SCH/HU block, which any burst type (or none of the downlink ones) carries */
static uint8_t bq_burst_type(const struct tgpu_channel *ch, enum tp_sap_data_type type)
{
	if (ch->bq_burst_known)
		return ch->bq_burst;
	switch (type) {
	case TPSAP_T_SB1: case TPSAP_T_SB2: return TETRA_TRAIN_SYNC;
	case TPSAP_T_SCH_F: return TETRA_TRAIN_NORM_1;
	case TPSAP_T_NDB: return TETRA_TRAIN_NORM_2;
	default: return TGPU_BURST_UNKNOWN;
	}
}

void tp_sap_udata_ind(enum tp_sap_data_type type, int blk_num, const uint8_t *bits, unsigned int len, void *priv)
{
	static const uint16_t t5len[6] = { 120, 216, 216, 30, 168, 432 };	/* lower_mac/tetra_lower_mac.c:55-102 */
	struct tgpu_channel *ch = priv;
	if (!ch || !bits)
		return;
	if ((unsigned)type > TPSAP_T_SCH_F || len != t5len[type]) {
		(void)channel_fail(ch, TGPU_EINVAL, 0);
		return;
	}
	int rc = bq_alloc(ch);
	if (rc) {
		(void)channel_fail(ch, rc, 0);
		return;
	}
	struct bq_item *it = &ch->bq[ch->bq_n];
	it->type = (uint8_t)type;
	it->blk_num = (uint8_t)blk_num;
	it->len = (uint16_t)len;
	it->time = t_phy_state.time;
	it->burst_type = bq_burst_type(ch, type);
	it->burst_seq = ch->burst_seq;
	memcpy(ch->bq_bits + (size_t)ch->bq_n * BQ_STRIDE, bits, len);
	ch->bq_n++;
	if (type == TPSAP_T_SB1 || ch->bq_n >= ch->bq_cap)
		(void)flush_blocks(ch);
}

/* phy/tetra_burst.c:341-379: a burst is handed to the lower MAC block by block */
void tetra_burst_rx_cb(const uint8_t *burst, unsigned int len, enum tetra_train_seq type, void *priv)
{
	uint8_t bbk[30], both[432];
	struct tgpu_channel *ch = priv;
	if (!burst || len < TG_SLOT_BITS || !ch)
		return;
	if (type != TETRA_TRAIN_SYNC && type != TETRA_TRAIN_NORM_1 && type != TETRA_TRAIN_NORM_2)
		return;		/* uplink training sequences: ignored */
	ch->burst_seq++;	/* ordinal of the burst, reported with its blocks (tgpu_unitdata.burst_seq) */
	ch->bq_burst = (uint8_t)type;
	ch->bq_burst_known = 1;
	if (type == TETRA_TRAIN_SYNC) {
		tp_sap_udata_ind(TPSAP_T_SB1, BLK_1, burst + TG_SB_BLK1_OFF, 120, priv);
		tp_sap_udata_ind(TPSAP_T_BBK, 0, burst + TG_SB_BBK_OFF, 30, priv);
		tp_sap_udata_ind(TPSAP_T_SB2, BLK_2, burst + TG_SB_BLK2_OFF, 216, priv);
		ch->bq_burst_known = 0;
		return;
	}
	memcpy(bbk, burst + TG_NDB_BBK1_OFF, 14);
	memcpy(bbk + 14, burst + TG_NDB_BBK2_OFF, 16);
	tp_sap_udata_ind(TPSAP_T_BBK, 0, bbk, 30, priv);
	if (type == TETRA_TRAIN_NORM_2) {
		tp_sap_udata_ind(TPSAP_T_NDB, BLK_1, burst + TG_NDB_BLK1_OFF, 216, priv);
		tp_sap_udata_ind(TPSAP_T_NDB, BLK_2, burst + TG_NDB_BLK2_OFF, 216, priv);
	} else {
		memcpy(both, burst + TG_NDB_BLK1_OFF, 216);
		memcpy(both + 216, burst + TG_NDB_BLK2_OFF, 216);
		tp_sap_udata_ind(TPSAP_T_SCH_F, 0, both, 432, priv);
	}
	ch->bq_burst_known = 0;
}

/* ------------------------------------------------------------------------- */
/* tetra_burst_sync_in(): phy/tetra_burst_sync.c:54-154                        */
/* ------------------------------------------------------------------------- */
int tetra_burst_sync_in(struct tetra_rx_state *trs, uint8_t *bits, unsigned int len)
{
	struct tgpu_channel *ch = trs->burst_cb_priv;
	unsigned int offs = 0;
	int rc;

	/* make_bitbuf_space(), :38-52 */
	unsigned int space = (unsigned int)sizeof(trs->bitbuf) - trs->bits_in_buf;
	if (space < len) {
		unsigned int delta = len - space;
		memmove(trs->bitbuf, trs->bitbuf + delta, trs->bits_in_buf - delta);
		trs->bits_in_buf -= delta;
		trs->bitbuf_start_bitnum += delta;
	}
	memcpy(trs->bitbuf + trs->bits_in_buf, bits, len);
	trs->bits_in_buf += len;

	if (trs->state == RX_S_UNLOCKED) {
		if (trs->bits_in_buf < TG_SLOT_BITS * 2)
			return (int)len;
		rc = tetra_find_train_seq(trs->bitbuf, trs->bits_in_buf, 1u << TETRA_TRAIN_SYNC, &offs);
		if (rc < 0)
			return rc;
		if (ch->ev)
			ch->ev(TGPU_EV_FOUND_SYNC, trs->bitbuf_start_bitnum, offs, ch->priv);
		trs->state = RX_S_KNOW_FSTART;
		trs->next_frame_start_bitnum = trs->bitbuf_start_bitnum + offs + 296;
		return (int)len;
	}

	if (trs->state == RX_S_KNOW_FSTART) {
		if (trs->bitbuf_start_bitnum + trs->bits_in_buf < trs->next_frame_start_bitnum)
			return 0;
		int skip = (int)(trs->next_frame_start_bitnum - trs->bitbuf_start_bitnum);
		int keep = (int)trs->bits_in_buf - skip;
		memmove(trs->bitbuf, trs->bitbuf + skip, (size_t)keep);
		trs->bits_in_buf = (unsigned int)keep;
		trs->bitbuf_start_bitnum += (unsigned int)skip;
		trs->next_frame_start_bitnum += TG_SLOT_BITS;
		trs->state = RX_S_LOCKED;
		/* the reference has no break here (:105): the same call goes on as LOCKED */
	}

	/* RX_S_LOCKED, :106-149 */
	if (trs->bits_in_buf < TG_SLOT_BITS)
		return (int)len;

	ch->tn_adds++;				/* tetra_tdma_time_add_tn(&t_phy_state.time, 1), applied at delivery */
	ch->burst_seq++;
	if (ch->ev)
		ch->ev(TGPU_EV_BURST, trs->bitbuf_start_bitnum, trs->bits_in_buf, ch->priv);

	rc = tetra_find_train_seq(trs->bitbuf, trs->bits_in_buf,
				  (1u << TETRA_TRAIN_NORM_1) | (1u << TETRA_TRAIN_NORM_2) | (1u << TETRA_TRAIN_SYNC), &offs);
	if (rc == TETRA_TRAIN_SYNC) {
		if (offs == TG_SYNC_TRAIN_OFF)
			queue_burst(ch, trs->bitbuf, rc);
		else {
			if (ch->ev)
				ch->ev(TGPU_EV_SYNC_MISPLACED, trs->bitbuf_start_bitnum, offs, ch->priv);
			trs->state = RX_S_UNLOCKED;
		}
	} else if (rc == TETRA_TRAIN_NORM_1 || rc == TETRA_TRAIN_NORM_2) {
		if (offs == TG_NORM_TRAIN_OFF)
			queue_burst(ch, trs->bitbuf, rc);
		else if (ch->ev)
			ch->ev(TGPU_EV_NORM_MISPLACED, trs->bitbuf_start_bitnum, offs, ch->priv);
	} else {
		if (ch->ev)
			ch->ev(TGPU_EV_NO_TRAIN, trs->bitbuf_start_bitnum, 0, ch->priv);
		trs->state = RX_S_UNLOCKED;
	}

	trs->bits_in_buf -= TG_SLOT_BITS;
	memmove(trs->bitbuf, trs->bitbuf + TG_SLOT_BITS, trs->bits_in_buf);
	trs->bitbuf_start_bitnum += TG_SLOT_BITS;
	trs->next_frame_start_bitnum += TG_SLOT_BITS;
	return (int)len;
}
