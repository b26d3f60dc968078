/*
 * tg_stream.c -- burst synchronisation of a whole recorded stream, GPU-assisted.
 *
 * tetra_burst_sync_in() (phy/tetra_burst_sync.c:54-154) is a per-call state machine.  For a stream of
 * 'len' bytes fed 'chunk' bytes per call (tetra-rx.c:82-95 uses 64) its behaviour is a closed-form
 * function of the byte positions:
 *   call k has been fed F(k) = min(k*chunk, len) bytes;
 *   UNLOCKED  : from the first call with >= 1020 buffered bytes, every call looks for the first SYNC
 *               training sequence in the buffer (which starts at bs and, once 4096 bytes are buffered,
 *               slides); a hit at p gives next_frame_start = p + 296 (:81);
 *   KNOW_FSTART: the first LATER call with F(k) >= next_frame_start moves the buffer start there (:93-105)
 *               and, in the same call, goes on as LOCKED;
 *   LOCKED    : a call with F(k) - bs >= 510 handles exactly one burst: the search window is everything
 *               buffered, w = F(k) - bs (:117-120); then bs += 510 (:145-148).
 * tgpu_sync_walk() evaluates exactly that, slot by slot instead of call by call.  Where a slot lies on
 * the grid the GPU front end has classified (k_front_stream: first training sequence in the window),
 * the classification word replaces the byte scan; everything the kernel cannot settle (hits below
 * offset 21 where the reference's look-ahead filter is skewed, windows larger than its view, slots off
 * the grid) is settled here with tetra_find_train_seq() on the bytes, i.e. by the reference's own rule.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <sys/prctl.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <pthread.h>

#include "tetra_gpu.h"
#include "tg_layout.h"
#include "tg_internal.h"
#include "tg_walk_core.h"

#define MASK_LOCKED ((1u << TETRA_TRAIN_NORM_1) | (1u << TETRA_TRAIN_NORM_2) | (1u << TETRA_TRAIN_SYNC))

struct walk {
	const uint8_t *s;
	uint64_t len;
	uint64_t chunk;
	uint64_t ncalls;
	struct tgpu_sync_result *out;
	uint32_t cap_slots, cap_events;
	/* optional GPU summary: ysum[g] describes the y sequences starting in [anchor + 510 g, anchor + 510 (g + 1)) */
	const uint16_t *ysum;
	uint64_t anchor, nys;
	/* grid mode (TGPU_SYNC_GRID): delivered bursts are marked in a bitmap over the classified grid instead of
	 * being listed; one that is not a grid slot (or whose classification word disagrees) is only counted */
	uint32_t *grid_bits;
	const uint32_t *cls;
	uint32_t ncls;
	int cshift;	/* log2(chunk) if chunk is a power of two (tetra-rx.c feeds 64), else -1 */
};

static inline uint64_t div_chunk(const struct walk *w, uint64_t x)
{
	return w->cshift >= 0 ? x >> w->cshift : x / w->chunk;
}


static inline uint64_t fed(const struct walk *w, uint64_t k)
{
	uint64_t f = k * w->chunk;
	return f > w->len ? w->len : f;
}

/* first call index whose fed count reaches 'pos' (>= 1); ncalls + 1 if never */
static inline uint64_t call_reaching(const struct walk *w, uint64_t pos)
{
	if (pos > w->len)
		return w->ncalls + 1;
	uint64_t k = div_chunk(w, pos + w->chunk - 1);
	return k ? k : 1;
}

static int push_event(struct walk *w, int ev, uint64_t bitnum, uint32_t arg)
{
	struct tgpu_sync_result *o = w->out;
	if (o->nevents == w->cap_events) {
		uint32_t nc = w->cap_events ? w->cap_events * 2 : 1024;
		void *p = realloc(o->events, (size_t)nc * sizeof(*o->events));
		if (!p)
			return TGPU_ENOMEM;
		o->events = p;
		w->cap_events = nc;
	}
	o->events[o->nevents].ev = ev;
	o->events[o->nevents].bitnum = (uint32_t)bitnum;
	o->events[o->nevents].arg = arg;
	o->nevents++;
	return 0;
}

static int push_slot(struct walk *w, uint64_t off, int type, uint32_t seq, uint32_t tn_adds)
{
	struct tgpu_sync_result *o = w->out;
	if (o->nslots == w->cap_slots) {
		uint32_t nc = w->cap_slots ? w->cap_slots * 2 : 4096;
		void *p = realloc(o->slots, (size_t)nc * sizeof(*o->slots));
		if (!p)
			return TGPU_ENOMEM;
		o->slots = p;
		w->cap_slots = nc;
	}
	o->slots[o->nslots].off = off;
	o->slots[o->nslots].burst_seq = seq;
	o->slots[o->nslots].tn_adds = tn_adds;
	o->slots[o->nslots].type = (uint8_t)type;
	o->nslots++;
	return 0;
}

static int deliver_slot(struct walk *w, uint64_t off, int type, uint32_t seq, uint32_t tn_adds, int ongrid, uint64_t gi)
{
	if (!w->grid_bits)
		return push_slot(w, off, type, seq, tn_adds);
	const unsigned want_offs = (type == TETRA_TRAIN_SYNC) ? TG_SYNC_TRAIN_OFF : TG_NORM_TRAIN_OFF;
	if (ongrid && gi < w->ncls && (w->cls[gi] & 0xff) == (uint32_t)type && ((w->cls[gi] >> 8) & 0xffff) == want_offs) {
		w->grid_bits[gi >> 5] |= 1u << (gi & 31);
		w->out->nslots++;
	} else
		w->out->noffgrid++;
	return 0;
}

/* bit j set: classification word c[j] (flags EARLY21 / NONBINARY included) is none of the three "delivered" words */
static inline uint32_t not_plain32(const uint32_t *c, uint32_t A, uint32_t B, uint32_t C)
{
#if defined(__SSE2__)
	const __m128i m = _mm_set1_epi32(0x03ffffff), a = _mm_set1_epi32((int)A), b = _mm_set1_epi32((int)B), cc = _mm_set1_epi32((int)C);
	uint32_t r = 0;
	for (int q = 0; q < 8; q++) {
		const __m128i v = _mm_and_si128(_mm_loadu_si128((const __m128i *)(c + 4 * q)), m);
		const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi32(v, a), _mm_cmpeq_epi32(v, b)), _mm_cmpeq_epi32(v, cc));
		r |= (uint32_t)(_mm_movemask_ps(_mm_castsi128_ps(ok)) ^ 0xf) << (4 * q);
	}
	return r;
#else
	uint32_t r = 0;
	for (int j = 0; j < 32; j++) {
		const uint32_t v = c[j] & 0x03ffffffu;
		r |= ((uint32_t)(v != A) & (uint32_t)(v != B) & (uint32_t)(v != C)) << j;
	}
	return r;
#endif
}

static const uint8_t tsq_y38[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };

/* first start position of the SYNC training sequence in [from, last] by looking at the bytes */
static uint64_t scan_sync_seq(const struct walk *w, uint64_t from, uint64_t last)
{
	const uint8_t *y = tsq_y38;
	uint64_t y8;
	memcpy(&y8, y, 8);
	for (uint64_t p = from; p <= last; p++) {
		uint64_t v;
		memcpy(&v, w->s + p, 8);
		if (v == y8 && !memcmp(w->s + p + 8, y + 8, 30))
			return p;
	}
	return UINT64_MAX;
}

/* first start position of the SYNC training sequence in [from, last], UINT64_MAX if none.  Grid slots
 * covered by the GPU summary cost one table look-up; the bytes are only read before the grid, after
 * it, and inside a slot that holds several sequences when the first one lies before 'from'. */
static uint64_t next_sync_seq(const struct walk *w, uint64_t from, uint64_t last)
{
	if (w->len < 38)
		return UINT64_MAX;
	if (last > w->len - 38)
		last = w->len - 38;
	uint64_t p = from;
	while (p <= last) {
		if (!w->ysum || p < w->anchor) {
			uint64_t e = last;
			if (w->ysum && w->nys && e >= w->anchor)
				e = w->anchor - 1;
			const uint64_t r = scan_sync_seq(w, p, e);
			if (r != UINT64_MAX)
				return r;
			p = e + 1;
			continue;
		}
		const uint64_t g = (p - w->anchor) / TG_SLOT_BITS;
		if (g >= w->nys)
			return scan_sync_seq(w, p, last);
		const uint64_t s0 = w->anchor + g * TG_SLOT_BITS;
		const uint32_t v = w->ysum[g];
		if (v != TG_YS_NONE) {
			const uint64_t fp = s0 + TG_YS_FIRST(v);
			/* the kernel reads any non-zero byte as 1: where this slot's or the next one's window holds a byte other
			 * than 0 / 1 a summary entry is only a candidate and the bytes decide (such a sequence is none for
			 * the reference's memcmp()) */
			const int doubt = !w->cls || g + 1 >= w->ncls || (((w->cls[g] | w->cls[g + 1]) >> 24) & TG_CLS_NONBINARY);
			const int real = !doubt || !memcmp(w->s + fp, tsq_y38, 38);
			if (fp >= p && real)
				return fp <= last ? fp : UINT64_MAX;
			if ((v & TG_YS_MULTI) || !real) {
				const uint64_t e = s0 + TG_SLOT_BITS - 1 < last ? s0 + TG_SLOT_BITS - 1 : last;
				const uint64_t r = scan_sync_seq(w, p, e);
				if (r != UINT64_MAX)
					return r;
			}
		}
		p = s0 + TG_SLOT_BITS;
	}
	return UINT64_MAX;
}

/* the reference's routine on the bytes; the buffer it sees is [pos, pos + n) followed by whatever the
 * stream holds next (it peeks up to 21 bytes past the window, never using them for a hit) */
static int exact_find(const struct walk *w, uint64_t pos, uint32_t n, uint32_t mask, unsigned int *offs)
{
	if (pos + n + 22 <= w->len)
		return tetra_find_train_seq(w->s + pos, n, mask, offs);
	uint8_t *tmp = calloc(1, (size_t)n + 32);
	if (!tmp)
		return -1;
	memcpy(tmp, w->s + pos, (size_t)(w->len - pos < n ? w->len - pos : n));
	if (pos + n < w->len)
		memcpy(tmp + n, w->s + pos + n, (size_t)(w->len - pos - n));
	int rc = tetra_find_train_seq(tmp, n, mask, offs);
	free(tmp);
	return rc;
}

/*
 * The reference's state machine call by call (phy/tetra_burst_sync.c:54-154 on a 4096-byte window that slides over
 * the stream), every search on the bytes.  This is what the closed form below is checked against; the walk itself
 * uses it for feed sizes outside the range the closed form was derived for (below 21 a sequence rejected in the
 * skewed look-ahead zone can become acceptable when the window slides; above 296 a call can push the window past
 * the expected frame start).  Where the reference itself would compute a negative memmove() offset the input is
 * refused (TGPU_EINVAL) rather than guessed at.
 */
static int walk_per_call(struct walk *w, uint32_t flags, uint64_t anchor)
{
	uint64_t bs = 0, nfs = 0, in_buf = 0;
	uint32_t seq = 0, tn_adds = 0;
	int state = RX_S_UNLOCKED, rc;
	unsigned int offs = 0;
	for (uint64_t k = 1; k <= w->ncalls; k++) {
		const uint64_t n = fed(w, k) - fed(w, k - 1);
		if (4096 - in_buf < n) {	/* make_bitbuf_space(), :38-52 */
			const uint64_t delta = n - (4096 - in_buf);
			in_buf -= delta;
			bs += delta;
		}
		in_buf += n;
		if (state == RX_S_UNLOCKED) {
			if (in_buf < 2 * TG_SLOT_BITS)
				continue;
			if (exact_find(w, bs, (uint32_t)in_buf, 1u << TETRA_TRAIN_SYNC, &offs) < 0)
				continue;
			if ((rc = push_event(w, TGPU_EV_FOUND_SYNC, bs, offs)))
				return rc;
			state = RX_S_KNOW_FSTART;
			nfs = bs + offs + 296;
			continue;
		}
		if (state == RX_S_KNOW_FSTART) {
			if (bs + in_buf < nfs)
				continue;
			if (nfs < bs)
				return TGPU_EINVAL;	/* the reference's offset would be negative here */
			in_buf -= nfs - bs;
			bs = nfs;
			nfs += TG_SLOT_BITS;
			state = RX_S_LOCKED;
		}
		if (in_buf < TG_SLOT_BITS)
			continue;
		seq++;
		tn_adds++;
		if (!(flags & TGPU_SYNC_NO_BURST_EVENTS) && (rc = push_event(w, TGPU_EV_BURST, bs, (uint32_t)in_buf)))
			return rc;
		const int type = exact_find(w, bs, (uint32_t)in_buf, MASK_LOCKED, &offs);
		const int ongrid = w->cls && bs >= anchor && (bs - anchor) % TG_SLOT_BITS == 0;
		const uint64_t gi = ongrid ? (bs - anchor) / TG_SLOT_BITS : 0;
		if (type == TETRA_TRAIN_SYNC || type == TETRA_TRAIN_NORM_1 || type == TETRA_TRAIN_NORM_2) {
			if (offs == (type == TETRA_TRAIN_SYNC ? TG_SYNC_TRAIN_OFF : TG_NORM_TRAIN_OFF)) {
				if ((rc = deliver_slot(w, bs, type, seq, tn_adds, ongrid, gi)))
					return rc;
				tn_adds = 0;
			} else {
				if ((rc = push_event(w, type == TETRA_TRAIN_SYNC ? TGPU_EV_SYNC_MISPLACED : TGPU_EV_NORM_MISPLACED, bs, offs)))
					return rc;
				if (type == TETRA_TRAIN_SYNC)
					state = RX_S_UNLOCKED;
			}
		} else {
			if ((rc = push_event(w, TGPU_EV_NO_TRAIN, bs, 0)))
				return rc;
			state = RX_S_UNLOCKED;
		}
		in_buf -= TG_SLOT_BITS;
		bs += TG_SLOT_BITS;
		nfs += TG_SLOT_BITS;
	}
	w->out->final_state = state;
	w->out->tail_tn_adds = tn_adds;
	w->out->burst_seq = seq;
	return TGPU_OK;
}

static int sync_walk(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor, const uint32_t *cls,
		     const uint16_t *ysum, const uint32_t *plain, uint32_t ncls, uint32_t flags, struct tgpu_sync_result *out);

int tgpu_sync_walk(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor,
		   const uint32_t *cls, const uint16_t *ysum, uint32_t ncls, uint32_t flags, struct tgpu_sync_result *out)
{
	return sync_walk(h_stream, len, chunk, anchor, cls, ysum, NULL, ncls, flags, out);
}

int tgpu_sync_walk_plain(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor, const uint32_t *cls,
			 const uint16_t *ysum, const uint32_t *plain, uint32_t ncls, uint32_t flags, struct tgpu_sync_result *out)
{
	return sync_walk(h_stream, len, chunk, anchor, cls, ysum, plain, ncls, flags, out);
}

/* plain (optional): k_cls_plain's bitmap of the same grid -- bit i = word i is one of the three "delivered" words */
static int sync_walk(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor, const uint32_t *cls,
		     const uint16_t *ysum, const uint32_t *plain, uint32_t ncls, uint32_t flags, struct tgpu_sync_result *out)
{
	/* chunk <= 510: a call never feeds more than a burst consumes, so the 4096-byte buffer can only
	 * overflow (and drop data) while UNLOCKED -- which is modelled.  tetra-rx.c uses 64. */
	if (!h_stream || !out || !chunk || chunk > TG_SLOT_BITS)
		return TGPU_EINVAL;
	memset(out, 0, sizeof(*out));
	struct walk w = { h_stream, len, chunk, (len + chunk - 1) / chunk, out, 0, 0, ysum, anchor, ysum ? ncls : 0,
			  NULL, cls, ncls, (chunk & (chunk - 1)) ? -1 : __builtin_ctz(chunk) };
	int rc;
	if (flags & TGPU_SYNC_GRID) {
		if (!cls || !ncls)
			return TGPU_EINVAL;
		out->grid_bits = calloc(((size_t)ncls + 31) / 32, 4);
		if (!out->grid_bits)
			return TGPU_ENOMEM;
		out->ngrid = ncls;
		w.grid_bits = out->grid_bits;
	} else if (ncls) {	/* nearly every grid slot is delivered: reserve once instead of growing by doubling */
		out->slots = malloc(((size_t)ncls + 16) * sizeof(*out->slots));
		if (!out->slots)
			return TGPU_ENOMEM;
		w.cap_slots = ncls + 16;
	}

	if (chunk < TGPU_SYNC_CHUNK_MIN || chunk > TGPU_SYNC_CHUNK_MAX || (flags & TGPU_SYNC_PER_CALL))
		return walk_per_call(&w, flags, anchor);

	uint64_t bs = 0;	/* bitbuf_start_bitnum */
	uint64_t k = 0;		/* index of the last call that has run */
	uint64_t kceil = 0, fceil = 0, fceil_for = UINT64_MAX;
	uint64_t gi = 0, grid_for = UINT64_MAX, gfit = UINT64_MAX;
	int ongrid = 0;
	uint32_t seq = 0, tn_adds = 0;
	int state = RX_S_UNLOCKED;
	uint64_t nfs = 0;

#ifdef TG_WALK_TIMING
	static unsigned long long c_unl, c_blk, c_gen, n_walk;
	unsigned long long tq = __builtin_ia32_rdtsc();
#define TQ(acc) do { const unsigned long long t_ = __builtin_ia32_rdtsc(); acc += t_ - tq; tq = t_; } while (0)
#else
#define TQ(acc) do { } while (0)
#endif
	while (k < w.ncalls) {
		TQ(c_gen);
		if (state == RX_S_UNLOCKED) {
			/* calls k+1, k+2, ...: buffer = [b, F(k)), b = max(bs, F(k) - 4096).  The reference
			 * re-scans the whole buffer on every call; the outcome only depends on where SYNC
			 * sequences start, so positions are looked at once ('clean_to' = everything before
			 * it is known to hold no valid hit for the current buffer start). */
			uint64_t found_p = 0, found_k = 0, found_bs = 0;
			int found = 0;
			uint64_t kk = k + 1;
			uint64_t k1020 = call_reaching(&w, bs + 2 * TG_SLOT_BITS);	/* nothing happens below 1020 buffered bytes */
			if (kk < k1020)
				kk = k1020;
			uint64_t clean_to = bs;
			for (; kk <= w.ncalls && !found; kk++) {
				const uint64_t f = fed(&w, kk);
				const uint64_t b = (f > 4096 && f - 4096 > bs) ? f - 4096 : bs;
				if (clean_to < b)
					clean_to = b;
				if (f - b < 2 * TG_SLOT_BITS || f < 38)
					continue;	/* only at the stream tail */
				const uint64_t last = f - 38;
				while (clean_to <= last) {
					const uint64_t p = next_sync_seq(&w, clean_to, last);
					if (p == UINT64_MAX) {
						/* nothing in this buffer: jump to the call that first sees the next sequence
						 * of the stream (the calls in between would find nothing either) */
						const uint64_t pn = next_sync_seq(&w, last + 1, UINT64_MAX - 64);
						if (pn == UINT64_MAX) {
							kk = w.ncalls;	/* no SYNC sequence left: UNLOCKED to the end */
							clean_to = w.len;
							break;
						}
						clean_to = pn;
						const uint64_t kn = call_reaching(&w, pn + 38);
						if (kn > kk + 1)
							kk = kn - 1;	/* the for-loop's increment makes it kn */
						break;
					}
					if (p - b >= 21) {
						found = 1;
						found_p = p;
					} else {
						/* inside the skewed zone of the look-ahead filter: ask the exact routine
						 * about this very buffer; its answer is final for this call */
						unsigned int offs;
						if (exact_find(&w, b, (uint32_t)(f - b), 1u << TETRA_TRAIN_SYNC, &offs) >= 0) {
							found = 1;
							found_p = b + offs;
						} else
							clean_to = last + 1;
					}
					break;
				}
				if (found) {
					found_k = kk;
					found_bs = b;
				}
			}
			if (!found) {
				k = w.ncalls;
				break;
			}
			if ((rc = push_event(&w, TGPU_EV_FOUND_SYNC, found_bs, (uint32_t)(found_p - found_bs))))
				return rc;
			bs = found_bs;
			nfs = found_p + 296;
			k = found_k;
			state = RX_S_KNOW_FSTART;
			TQ(c_unl);
			continue;
		}
		if (state == RX_S_KNOW_FSTART) {
			uint64_t kl = call_reaching(&w, nfs);
			if (kl <= k)
				kl = k + 1;
			if (kl > w.ncalls) {
				k = w.ncalls;
				break;
			}
			bs = nfs;
			nfs += TG_SLOT_BITS;
			state = RX_S_LOCKED;
			k = kl - 1;	/* the LOCKED branch below may use call kl itself */
		}
		/* LOCKED, steady state: a run of grid slots whose classification word alone says "delivered" (a
		 * training sequence of the right type at its nominal offset, found at an offset >= 21: the first hit
		 * of any window that holds it, whatever the backlog).  Per slot: the call that consumes it,
		 * k = max(k + 1, ceil((bs + 510) / chunk)), the ordinal, the delivery.  Only without per-burst
		 * events and with a power-of-two chunk; everything else takes the general path below. */
		if (cls && (flags & TGPU_SYNC_NO_BURST_EVENTS) && w.cshift >= 0) {
			if (grid_for != bs) {
				ongrid = bs >= anchor && (bs - anchor) % TG_SLOT_BITS == 0;
				gi = ongrid ? (bs - anchor) / TG_SLOT_BITS : 0;
				grid_for = bs;
			}
			if (ongrid) {
				const uint64_t bs0 = bs;
				/* 32 slots at a time while nothing is backlogged (k < the call that completes the first of
				 * them: every slot is then consumed by "its" call, k ends at the last one's) and all 32
				 * classification words say "delivered": one word of the bitmap, closed-form ordinals */
				const uint32_t A = TETRA_TRAIN_SYNC | TG_SYNC_TRAIN_OFF << 8, B = TETRA_TRAIN_NORM_1 | TG_NORM_TRAIN_OFF << 8,
					       C = TETRA_TRAIN_NORM_2 | TG_NORM_TRAIN_OFF << 8;
				if (gfit == UINT64_MAX) {	/* once per walk: the grid slots that fit into the stream */
					gfit = len >= anchor ? (len - anchor) / TG_SLOT_BITS : 0;
					if (gfit > ncls)
						gfit = ncls;
				}
				for (;;) {
				const uint64_t gi_before = gi;
				/* bitmap mode: up to the end of the current bitmap word at a time (a whole word in the steady state,
				 * the rest of one after a re-lock) while nothing is backlogged (k < the call that completes the first
				 * slot: every slot is then consumed by "its" call, k ends at the last one's).  The leading run of
				 * slots whose classification words say "delivered" goes in closed form: its bits, the ordinals,
				 * the call that consumes the last of them; the first other slot is left to the general path. */
				while (w.grid_bits && gi < ncls && ((bs + TG_SLOT_BITS + chunk - 1) >> w.cshift) > k) {
					uint32_t m = 32 - (uint32_t)(gi & 31);
					if (gi >= gfit)
						break;
					if (m > gfit - gi)
						m = (uint32_t)(gfit - gi);	/* grid slots that lie inside the stream (and have a word) */
					uint32_t bad;
					if (ysum)	/* the re-lock search reads the SYNC summaries every hundred slots or so: keep them streaming */
						__builtin_prefetch(ysum + gi + 512);
					if (plain)	/* (and with the bitmap the words themselves are only read at the exceptions) */
						__builtin_prefetch(cls + gi + 256);
					if (plain) {
						bad = ~plain[gi >> 5] >> (gi & 31);
						if (m < 32)
							bad &= (1u << m) - 1u;
					} else if (gi + 32 <= ncls) {
						bad = not_plain32(cls + gi, A, B, C);
						if (m < 32)
							bad &= (1u << m) - 1u;
					} else {
						bad = 0;
						for (uint32_t j = 0; j < m; j++) {
							const uint32_t v = cls[gi + j] & 0x03ffffffu;
							bad |= ((uint32_t)(v != A) & (uint32_t)(v != B) & (uint32_t)(v != C)) << j;
						}
					}
					uint32_t g = bad ? (uint32_t)__builtin_ctz(bad) : m;
					while (g && ((bs + (uint64_t)g * TG_SLOT_BITS + chunk - 1) >> w.cshift) > w.ncalls)
						g--;		/* (the stream's last calls) */
					if (!g)
						break;
					w.grid_bits[gi >> 5] |= (g == 32 ? 0xffffffffu : ((1u << g) - 1u)) << (gi & 31);
					out->nslots += g;
					k = (bs + (uint64_t)g * TG_SLOT_BITS + chunk - 1) >> w.cshift;
					seq += g;
					tn_adds = 0;
					bs += (uint64_t)g * TG_SLOT_BITS;
					nfs += (uint64_t)g * TG_SLOT_BITS;
					gi += g;
					if (g < m)
						break;		/* the next slot is not a plain delivery */
				}
				/* one slot at a time: a backlog (the bursts already buffered when lock was found are consumed one per
				 * call), and the slot-table form */
				while (gi < ncls && bs + TG_SLOT_BITS <= len) {
					const uint32_t v = cls[gi] & 0x03ffffffu;	/* type, offset, TG_CLS_EARLY21, TG_CLS_NONBINARY */
					if (v != A && v != B && v != C)
						break;
					const uint64_t kc = (bs + TG_SLOT_BITS + chunk - 1) >> w.cshift;
					const uint64_t kj = kc > k ? kc : k + 1;
					if (kj > w.ncalls)
						break;
					k = kj;
					seq++;
					tn_adds++;
					if (w.grid_bits) {
						w.grid_bits[gi >> 5] |= 1u << (gi & 31);
						out->nslots++;
					} else if ((rc = push_slot(&w, bs, (int)(v & 0xff), seq, tn_adds)))
						return rc;
					tn_adds = 0;
					bs += TG_SLOT_BITS;
					nfs += TG_SLOT_BITS;
					gi++;
					if (w.grid_bits && ((bs + TG_SLOT_BITS + chunk - 1) >> w.cshift) > k)
						break;	/* the backlog is gone: back to the closed form */
				}
				if (gi == gi_before || !w.grid_bits)
					break;
				}
				grid_for = bs;
				TQ(c_blk);
				if (bs != bs0)
					continue;
			}
		}
		/* LOCKED: the next burst is handled by the first call >= k+1 that has 510 bytes of it.
		 * kceil / fceil = that call and its fed count when nothing is backlogged, tracked incrementally
		 * (bs only moves forward) so that the steady state costs no division. */
		const uint64_t need = bs + TG_SLOT_BITS;
		if (need > len) {
			k = w.ncalls;
			break;
		}
		if (fceil_for != bs) {
			if (fceil_for == UINT64_MAX || bs < fceil_for || need > fceil + 64 * (uint64_t)chunk) {
				kceil = div_chunk(&w, need + chunk - 1);
				fceil = kceil * chunk;
			}
			while (fceil < need) {
				fceil += chunk;
				kceil++;
			}
			fceil_for = bs;
		}
		uint64_t kj = kceil > k ? kceil : k + 1;
		if (kj > w.ncalls) {
			k = w.ncalls;
			break;
		}
		const uint64_t fj = (kj == kceil) ? (fceil > len ? len : fceil) : fed(&w, kj);
		const uint32_t win = (uint32_t)(fj - bs);
		k = kj;
		seq++;
		tn_adds++;
		if (!(flags & TGPU_SYNC_NO_BURST_EVENTS) && (rc = push_event(&w, TGPU_EV_BURST, bs, win)))
			return rc;

		int type = -1;
		unsigned int offs = 0;
		int settled = 0;
		if (cls) {
			if (grid_for != bs) {
				ongrid = bs >= anchor && (bs - anchor) % TG_SLOT_BITS == 0;
				gi = ongrid ? (bs - anchor) / TG_SLOT_BITS : 0;
				grid_for = bs;
			}
			if (ongrid && gi < ncls) {
				const uint32_t cw = cls[gi];
				const uint32_t cflags = cw >> 24;
				/* (a byte other than 0 / 1 in the window: the kernel read it as 1, the reference's memcmp() matches
				 * nothing with it -- settled on the bytes like a hit below offset 21) */
				if (!(cflags & (TG_CLS_EARLY21 | TG_CLS_NONBINARY))) {
					if ((cw & 0xff) != TG_BURST_NONE) {
						/* first hit at an offset >= 21 inside the window the kernel assumed (the steady-state
						 * one): it is the first hit of any window that contains it */
						type = (int)(cw & 0xff);
						offs = (cw >> 8) & 0xffff;
						settled = 1;
					} else if (kj == kceil && !(cflags & TG_CLS_CLIPPED)) {
						type = -1;	/* same window as the kernel's, and nothing in it */
						settled = 1;
					}
				}
			}
		}
		if (!settled)
			type = exact_find(&w, bs, win, MASK_LOCKED, &offs);

		if (type == TETRA_TRAIN_SYNC) {
			if (offs == TG_SYNC_TRAIN_OFF) {
				if ((rc = deliver_slot(&w, bs, type, seq, tn_adds, cls && ongrid && grid_for == bs, gi)))
					return rc;
				tn_adds = 0;
			} else {
				if ((rc = push_event(&w, TGPU_EV_SYNC_MISPLACED, bs, offs)))
					return rc;
				state = RX_S_UNLOCKED;
			}
		} else if (type == TETRA_TRAIN_NORM_1 || type == TETRA_TRAIN_NORM_2) {
			if (offs == TG_NORM_TRAIN_OFF) {
				if ((rc = deliver_slot(&w, bs, type, seq, tn_adds, cls && ongrid && grid_for == bs, gi)))
					return rc;
				tn_adds = 0;
			} else if ((rc = push_event(&w, TGPU_EV_NORM_MISPLACED, bs, offs)))
				return rc;
		} else {
			if ((rc = push_event(&w, TGPU_EV_NO_TRAIN, bs, 0)))
				return rc;
			state = RX_S_UNLOCKED;
		}
		bs += TG_SLOT_BITS;
		nfs += TG_SLOT_BITS;
		if (ongrid && grid_for + TG_SLOT_BITS == bs) {	/* stay on the grid without dividing */
			grid_for = bs;
			gi++;
		}
	}
#ifdef TG_WALK_TIMING
	TQ(c_gen);
	if (++n_walk % 512 == 0)
		fprintf(stderr, "walk cycles per call: unlocked %.0f  block path %.0f  general %.0f (events %zu)\n", (double)c_unl / n_walk,
			(double)c_blk / n_walk, (double)c_gen / n_walk, (size_t)out->nevents);
#endif
	out->final_state = state;
	out->tail_tn_adds = tn_adds;
	out->burst_seq = seq;
	return TGPU_OK;
}

void tgpu_sync_result_free(struct tgpu_sync_result *r)
{
	if (!r)
		return;
	free(r->slots);
	free(r->events);
	free(r->grid_bits);
	memset(r, 0, sizeof(*r));
}

/* first lock of a stream, host only: where the slot grid starts (0 if the stream never locks) */
static int find_anchor_root(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t *anchor, int *locks,
			    struct tg_walk_root *root);

static int find_anchor(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t *anchor, int *locks)
{
	return find_anchor_root(h_stream, len, chunk, anchor, locks, NULL);
}

/* root (optional): buffer start and index of the call that finds the first SYNC sequence -- where the device walk
 * (k_walk) takes over */
static int find_anchor_root(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t *anchor, int *locks,
			    struct tg_walk_root *root)
{
	/* run the walk without classification until the first LOCKED burst: cheap, it stops early */
	struct walk w = { h_stream, len, chunk, (len + chunk - 1) / chunk, NULL, 0, 0, NULL, 0, 0, NULL, NULL, 0,
			  (chunk & (chunk - 1)) ? -1 : __builtin_ctz(chunk) };
	*locks = 0;
	uint64_t kk = call_reaching(&w, 2 * TG_SLOT_BITS);
	for (; kk <= w.ncalls; kk++) {
		uint64_t f = fed(&w, kk);
		uint64_t b = f > 4096 ? f - 4096 : 0;
		if (f - b < 2 * TG_SLOT_BITS)
			continue;
		unsigned int offs;
		if (exact_find(&w, b, (uint32_t)(f - b), 1u << TETRA_TRAIN_SYNC, &offs) >= 0) {
			*anchor = b + offs + 296;
			*locks = 1;
			if (root) {
				root->found_bs = b;
				root->found_k = kk;
			}
			return TGPU_OK;
		}
	}
	return TGPU_OK;
}

int tgpu_sync_classify(struct tgpu_engine *eng, const uint8_t *d_stream, uint64_t len, uint32_t chunk,
		       uint64_t anchor, uint32_t nslots, uint32_t *h_cls, uint16_t *h_ysum, void *stream)
{
	if (!eng || !d_stream || !h_cls || !chunk)
		return TGPU_EINVAL;
	if (!nslots)
		return TGPU_OK;
	int brc = tgpi_engine_bind(eng);
	if (brc)
		return brc;
	uint32_t *d_cls = NULL, *d_packed = NULL, *d_defer = NULL;
	/* classification words, then (same allocation) the SYNC-sequence summaries */
	hipError_t e = hipMalloc((void **)&d_cls, (size_t)nslots * 6);
	if (e == hipSuccess)
		e = hipMalloc((void **)&d_packed, (size_t)nslots * TG_PACKED_WORDS * 4);
	if (e == hipSuccess)
		e = hipMalloc((void **)&d_defer, TG_DEFER_WORDS(nslots) * 4);
	int rc = (int)e;
	if (!rc)
		rc = tgk_front_stream(d_stream, anchor, len, nslots, chunk, d_packed, d_cls,
				      h_ysum ? (uint16_t *)(d_cls + nslots) : NULL, d_defer, stream, NULL);
	if (!rc)
		rc = (int)hipMemcpyAsync(h_cls, d_cls, (size_t)nslots * 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
	if (!rc && h_ysum)
		rc = (int)hipMemcpyAsync(h_ysum, d_cls + nslots, (size_t)nslots * 2, hipMemcpyDeviceToHost, (hipStream_t)stream);
	if (!rc)
		rc = (int)hipStreamSynchronize((hipStream_t)stream);
	if (d_cls)
		(void)hipFree(d_cls);
	if (d_packed)
		(void)hipFree(d_packed);
	if (d_defer)
		(void)hipFree(d_defer);
	return rc;
}


int tgpu_sync_stream(struct tgpu_engine *eng, const uint8_t *h_stream, const uint8_t *d_stream, uint64_t len,
		     uint32_t chunk, uint32_t flags, struct tgpu_sync_result *out, void *stream)
{
	if (!eng || !h_stream || !d_stream || !out || !chunk)
		return TGPU_EINVAL;
	uint64_t anchor = 0;
	int locks = 0;
	int rc = find_anchor(h_stream, len, chunk, &anchor, &locks);
	if (rc)
		return rc;
	uint32_t *cls = NULL;
	uint16_t *ysum = NULL;
	uint32_t ncls = 0;
	if (locks && anchor + TG_SLOT_BITS <= len) {
		uint64_t n = (len - anchor) / TG_SLOT_BITS;
		if (n > 0xfffffff0u)
			return TGPU_ECAPACITY;
		ncls = (uint32_t)n;
		cls = malloc((size_t)ncls * 6);
		if (!cls)
			return TGPU_ENOMEM;
		ysum = (uint16_t *)(cls + ncls);
		rc = tgpu_sync_classify(eng, d_stream, len, chunk, anchor, ncls, cls, ysum, stream);
		if (rc) {
			free(cls);
			return rc;
		}
	}
	rc = tgpu_sync_walk(h_stream, len, chunk, anchor, cls, ysum, ncls, flags, out);
	out->anchor = anchor;
	free(cls);
	return rc;
}

/*
 * Stream mode end to end: classification (which also packs every grid slot into the plan's buffer) ->
 * host walk in grid mode -> the plan's lists are built on the device from the classification words and the
 * walk's bitmap.  Slot i of the plan is grid slot i: record i of tgpu_plan_execute() belongs to stream offset
 * out->anchor + 510 i and is valid iff bit i of out->grid_bits is set.  No slot table, no second front-end pass.
 */
int tgpu_sync_stream_grid_begin(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *h_stream,
				const uint8_t *d_stream, uint64_t len, uint32_t chunk, struct tgpu_sync_result *out, void *stream)
{
	if (!eng || !plan || !h_stream || !d_stream || !out || !chunk)
		return TGPU_EINVAL;
	memset(out, 0, sizeof(*out));
	uint64_t anchor = 0;
	int locks = 0;
	int rc = find_anchor(h_stream, len, chunk, &anchor, &locks);
	if (rc)
		return rc;
	out->anchor = anchor;
	if (!locks || anchor + TG_SLOT_BITS > len)
		return TGPU_OK;		/* never locks (or nothing after the lock): ngrid stays 0, nothing is launched */
	const uint64_t n = (len - anchor) / TG_SLOT_BITS;
	if (n > 0xfffffff0u)
		return TGPU_ECAPACITY;
	const uint32_t ncls = (uint32_t)n;
	uint32_t *d_packed, *d_cls, *cls;
	uint16_t *d_ysum, *ysum;
	if ((rc = tgpi_plan_grid_begin(plan, ncls, &d_packed, &d_cls, &d_ysum, &cls, &ysum)))
		return rc;
	rc = tgk_front_stream(d_stream, anchor, len, ncls, chunk, d_packed, d_cls, d_ysum, tgpi_plan_defer_scratch(plan), stream, NULL);
	if (!rc) {
		uint32_t *d_plain, *h_plain;
		tgpi_plan_grid_plain(plan, ncls, &d_plain, &h_plain);
		rc = tgk_cls_plain(d_cls, ncls, d_plain, stream);
	}
	if (!rc)	/* words, summaries and the bitmap are adjacent on both sides: one copy into the plan's pinned mirror */
		rc = (int)hipMemcpyAsync(cls, d_cls, TG_GRID_COPY_BYTES(ncls), hipMemcpyDeviceToHost, (hipStream_t)stream);
	out->ngrid = ncls;
	return rc;
}

int tgpu_sync_stream_grid_finish(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *h_stream, uint64_t len,
				 uint32_t chunk, uint32_t flags, uint32_t scramb_init, struct tgpu_sync_result *out,
				 void *stream)
{
	if (!eng || !plan || !h_stream || !out || !chunk)
		return TGPU_EINVAL;
	const uint64_t anchor = out->anchor;
	const uint32_t ncls = out->ngrid;
	int rc;
	if (!ncls) {
		/* never locks: the plain walk settles it, nothing to decode */
		rc = tgpu_sync_walk(h_stream, len, chunk, anchor, NULL, NULL, 0, flags & ~TGPU_SYNC_GRID, out);
		out->anchor = anchor;
		out->noffgrid = out->nslots;	/* anything it found would not be in a plan */
		return rc;
	}
	uint32_t *d_packed, *d_cls, *cls;
	uint16_t *d_ysum, *ysum;
	if ((rc = tgpi_plan_grid_begin(plan, ncls, &d_packed, &d_cls, &d_ysum, &cls, &ysum)))
		return rc;
	rc = (int)hipStreamSynchronize((hipStream_t)stream);
	if (!rc) {
		uint32_t *d_plain, *h_plain;
		tgpi_plan_grid_plain(plan, ncls, &d_plain, &h_plain);
		rc = sync_walk(h_stream, len, chunk, anchor, cls, ysum, h_plain, ncls, flags | TGPU_SYNC_GRID, out);
	}
	out->anchor = anchor;
	if (!rc && !out->noffgrid)
		rc = tgpi_plan_grid_load(plan, ncls, out->grid_bits, 1, &scramb_init, NULL, stream);
	return rc;
}

int tgpu_sync_front_prof(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *d_stream, uint64_t len,
			 uint32_t chunk, uint64_t anchor, uint32_t nrep, float us[2], void *stream)
{
	if (!eng || !plan || !d_stream || !chunk || !nrep || !us || anchor + TG_SLOT_BITS > len)
		return TGPU_EINVAL;
	const uint64_t n = (len - anchor) / TG_SLOT_BITS;
	if (n > 0xfffffff0u)
		return TGPU_ECAPACITY;
	uint32_t *d_packed, *d_cls, *cls;
	uint16_t *d_ysum, *ysum;
	int rc = tgpi_plan_grid_begin(plan, (uint32_t)n, &d_packed, &d_cls, &d_ysum, &cls, &ysum);
	if (rc)
		return rc;
	hipEvent_t ev[3] = { NULL, NULL, NULL };
	for (int i = 0; i < 3 && !rc; i++)
		rc = (int)hipEventCreate(&ev[i]);
	double acc[2] = { 0, 0 };
	for (uint32_t r = 0; r < nrep && !rc; r++) {
		rc = (int)hipEventRecord(ev[0], (hipStream_t)stream);
		if (!rc)
			rc = tgk_front_stream(d_stream, anchor, len, (uint32_t)n, chunk, d_packed, d_cls, d_ysum,
					      tgpi_plan_defer_scratch(plan), stream, ev[1]);
		if (!rc)
			rc = (int)hipEventRecord(ev[2], (hipStream_t)stream);
		if (!rc)
			rc = (int)hipEventSynchronize(ev[2]);
		float a = 0, b = 0;
		if (!rc)
			rc = (int)hipEventElapsedTime(&a, ev[0], ev[1]);
		if (!rc)
			rc = (int)hipEventElapsedTime(&b, ev[1], ev[2]);
		acc[0] += a;
		acc[1] += b;
	}
	for (int i = 0; i < 3; i++)
		if (ev[i])
			(void)hipEventDestroy(ev[i]);
	us[0] = (float)(acc[0] * 1e3 / nrep);
	us[1] = (float)(acc[1] * 1e3 / nrep);
	return rc;
}

/*
 * Several recorded channels in ONE grid and one plan batch (BASELINE config 4: a GPU's share of the channels).
 * begin: first lock of every channel on the host, one classification launch over all of them, one copy back;
 * finish: the synchroniser walks (one per channel, on up to nthreads host threads -- channels are independent,
 * the reference runs a process per channel), one bitmap, device-built lists with the channel of every slot.
 * Record index of grid slot i of channel c = out[c].grid_base + i.
 */
struct tgpu_sync_multi {
	struct tgpu_engine *eng;
	struct tgpu_plan *plan;
	uint32_t nchan, chunk, ngrid;
	struct tgpu_multi_chan *ch;
	struct tg_chan_ent *ent;
	int *locks;
};

void tgpu_sync_multi_free(struct tgpu_sync_multi *st)
{
	if (!st)
		return;
	free(st->ch);
	free(st->ent);
	free(st->locks);
	free(st);
}

int tgpu_sync_multi_begin(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			  const uint8_t *d_base, uint32_t chunk, struct tgpu_sync_multi **out, void *stream)
{
	if (!eng || !plan || !nchan || nchan > 64 || !ch || !d_base || !chunk || !out)
		return TGPU_EINVAL;
	*out = NULL;
	struct tgpu_sync_multi *st = calloc(1, sizeof(*st));
	if (!st)
		return TGPU_ENOMEM;
	st->eng = eng;
	st->plan = plan;
	st->nchan = nchan;
	st->chunk = chunk;
	st->ch = malloc((size_t)nchan * sizeof(*ch));
	st->ent = calloc(nchan, sizeof(*st->ent));
	st->locks = calloc(nchan, sizeof(int));
	int rc = (st->ch && st->ent && st->locks) ? TGPU_OK : TGPU_ENOMEM;
	if (!rc)
		memcpy(st->ch, ch, (size_t)nchan * sizeof(*ch));
	uint64_t total = 0;
	for (uint32_t c = 0; c < nchan && !rc; c++) {
		uint64_t anchor = 0;
		if (!ch[c].h_stream) {
			rc = TGPU_EINVAL;
			break;
		}
		rc = find_anchor(ch[c].h_stream, ch[c].len, chunk, &anchor, &st->locks[c]);
		uint64_t n = 0;
		if (!rc && st->locks[c] && anchor + TG_SLOT_BITS <= ch[c].len)
			n = (ch[c].len - anchor) / TG_SLOT_BITS;
		else
			st->locks[c] = 0;
		st->ent[c].d_off = ch[c].d_off;
		st->ent[c].anchor = anchor;
		st->ent[c].len = ch[c].len;
		st->ent[c].gbase = (uint32_t)total;
		st->ent[c].ncls = (uint32_t)n;
		total += (n + 31) & ~(uint64_t)31;
		if (total > 0xfffffff0u)
			rc = TGPU_ECAPACITY;
	}
	st->ngrid = (uint32_t)total;
	if (!rc && st->ngrid) {
		uint32_t *d_packed, *d_cls, *cls;
		uint16_t *d_ysum, *ysum;
		struct tg_chan_ent *d_tab;
		rc = tgpi_plan_grid_begin(plan, st->ngrid, &d_packed, &d_cls, &d_ysum, &cls, &ysum);
		if (!rc)
			rc = tgpi_plan_chan_table(plan, st->ent, nchan, &d_tab, stream);
		if (!rc)
			rc = tgk_front_stream_multi(d_base, d_tab, nchan, st->ngrid, chunk, d_packed, d_cls, d_ysum,
						    tgpi_plan_defer_scratch(plan), stream, NULL, 0);
		if (!rc) {
			uint32_t *d_plain, *h_plain;
			tgpi_plan_grid_plain(plan, st->ngrid, &d_plain, &h_plain);
			rc = tgk_cls_plain(d_cls, st->ngrid, d_plain, stream);
		}
		if (!rc)
			rc = (int)hipMemcpyAsync(cls, d_cls, TG_GRID_COPY_BYTES(st->ngrid), hipMemcpyDeviceToHost, (hipStream_t)stream);
	}
	if (rc) {
		tgpu_sync_multi_free(st);
		return rc;
	}
	*out = st;
	return TGPU_OK;
}

struct multi_job {
	struct tgpu_sync_multi *st;
	const uint32_t *cls;
	const uint16_t *ysum;
	const uint32_t *plain;
	uint32_t flags;
	struct tgpu_sync_result *out;
	volatile uint32_t next;
	volatile int rc;
};

static void *multi_worker(void *arg)
{
	struct multi_job *j = arg;
	for (;;) {
		const uint32_t c = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
		if (c >= j->st->nchan)
			return NULL;
		const struct tgpu_multi_chan *ch = &j->st->ch[c];
		const struct tg_chan_ent *e = &j->st->ent[c];
		int rc;
		if (e->ncls)
			rc = sync_walk(ch->h_stream, ch->len, j->st->chunk, e->anchor, j->cls + e->gbase, j->ysum + e->gbase,
				       j->plain + e->gbase / 32, e->ncls, j->flags | TGPU_SYNC_GRID, &j->out[c]);	/* (grids start at multiples of 32) */
		else {	/* never locks: nothing of it is in the grid */
			rc = tgpu_sync_walk(ch->h_stream, ch->len, j->st->chunk, e->anchor, NULL, NULL, 0, j->flags & ~TGPU_SYNC_GRID,
					    &j->out[c]);
			j->out[c].noffgrid = j->out[c].nslots;
		}
		j->out[c].anchor = e->anchor;
		j->out[c].grid_base = e->gbase;
		if (rc)
			j->rc = rc;
	}
}

int tgpu_sync_multi_finish(struct tgpu_sync_multi *st, uint32_t flags, unsigned int nthreads, struct tgpu_sync_result *out,
			   void *stream)
{
	if (!st || !out)
		return TGPU_EINVAL;
	memset(out, 0, (size_t)st->nchan * sizeof(*out));
	int rc = (int)hipStreamSynchronize((hipStream_t)stream);
	if (rc)
		return rc;
	uint32_t *d_packed, *d_cls, *cls = NULL;
	uint16_t *d_ysum, *ysum = NULL;
	if (st->ngrid && (rc = tgpi_plan_grid_begin(st->plan, st->ngrid, &d_packed, &d_cls, &d_ysum, &cls, &ysum)))
		return rc;
	uint32_t *d_plain_, *h_plain = NULL;
	if (st->ngrid)
		tgpi_plan_grid_plain(st->plan, st->ngrid, &d_plain_, &h_plain);
	struct multi_job job = { st, cls, ysum, h_plain, flags, out, 0, 0 };
	if (nthreads > st->nchan)
		nthreads = st->nchan;
#ifdef TG_WALK_TIMING
	struct timespec ta, tb;
	clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ta);
#endif
	if (nthreads <= 1)
		multi_worker(&job);
	else {
		pthread_t th[64];
		unsigned int started = 0;
		if (nthreads > 64)
			nthreads = 64;
		for (unsigned int i = 0; i + 1 < nthreads; i++)
			if (!pthread_create(&th[started], NULL, multi_worker, &job))
				started++;
		multi_worker(&job);
		for (unsigned int i = 0; i < started; i++)
			pthread_join(th[i], NULL);
	}
#ifdef TG_WALK_TIMING
	{
		static double acc; static unsigned cnt;
		clock_gettime(CLOCK_THREAD_CPUTIME_ID, &tb);
		acc += (tb.tv_sec - ta.tv_sec) * 1e3 + (tb.tv_nsec - ta.tv_nsec) * 1e-6;
		if (++cnt % 64 == 0)
			fprintf(stderr, "walks: %.3f ms cpu per finish (mean over %u)\n", acc / cnt, cnt);
	}
#endif
	if (job.rc)
		return job.rc;
	if (!st->ngrid)
		return TGPU_OK;
	/* one bitmap over the whole grid (channel grids start at multiples of 32 slots) */
	uint32_t *bits = calloc(((size_t)st->ngrid + 31) / 32, 4);
	uint32_t *codes = malloc((size_t)st->nchan * 4);
	if (!bits || !codes) {
		free(bits);
		free(codes);
		return TGPU_ENOMEM;
	}
	uint32_t off = 0;
	for (uint32_t c = 0; c < st->nchan; c++) {
		codes[c] = st->ch[c].scramb_init;
		if (out[c].grid_bits && !out[c].noffgrid)
			memcpy(bits + st->ent[c].gbase / 32, out[c].grid_bits, (((size_t)st->ent[c].ncls + 31) / 32) * 4);
		off += out[c].noffgrid;
	}
	rc = tgpi_plan_grid_load(st->plan, st->ngrid, bits, st->nchan, codes, st->ent, stream);
	free(bits);
	free(codes);
	(void)off;
	return rc;
}

int tgpu_sync_front_prof_multi(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			       const uint8_t *d_base, uint32_t chunk, uint32_t nrep, float us[2], void *stream)
{
	struct tgpu_sync_multi *st = NULL;
	if (!nrep || !us)
		return TGPU_EINVAL;
	int rc = tgpu_sync_multi_begin(eng, plan, nchan, ch, d_base, chunk, &st, stream);	/* (also the warm-up run) */
	if (rc)
		return rc;
	us[0] = us[1] = 0;
	uint32_t *d_packed, *d_cls, *cls;
	uint16_t *d_ysum, *ysum;
	struct tg_chan_ent *d_tab = NULL;
	hipEvent_t ev[3] = { NULL, NULL, NULL };
	if (st->ngrid) {
		rc = tgpi_plan_grid_begin(plan, st->ngrid, &d_packed, &d_cls, &d_ysum, &cls, &ysum);
		if (!rc)
			rc = tgpi_plan_chan_table(plan, st->ent, nchan, &d_tab, stream);
		for (int i = 0; i < 3 && !rc; i++)
			rc = (int)hipEventCreate(&ev[i]);
		double acc[2] = { 0, 0 };
		for (uint32_t r = 0; r < nrep && !rc; r++) {
			float a = 0, b = 0;
			rc = (int)hipEventRecord(ev[0], (hipStream_t)stream);
			if (!rc)
				rc = tgk_front_stream_multi(d_base, d_tab, nchan, st->ngrid, chunk, d_packed, d_cls, d_ysum,
							    tgpi_plan_defer_scratch(plan), stream, ev[1], 0);
			if (!rc)
				rc = (int)hipEventRecord(ev[2], (hipStream_t)stream);
			if (!rc)
				rc = (int)hipEventSynchronize(ev[2]);
			if (!rc)
				rc = (int)hipEventElapsedTime(&a, ev[0], ev[1]);
			if (!rc)
				rc = (int)hipEventElapsedTime(&b, ev[1], ev[2]);
			acc[0] += a;
			acc[1] += b;
		}
		for (int i = 0; i < 3; i++)
			if (ev[i])
				(void)hipEventDestroy(ev[i]);
		us[0] = (float)(acc[0] * 1e3 / nrep);
		us[1] = (float)(acc[1] * 1e3 / nrep);
	}
	tgpu_sync_multi_free(st);
	return rc;
}

uint32_t tgpu_sync_multi_ngrid(const struct tgpu_sync_multi *st)
{
	return st ? st->ngrid : 0;
}

int tgpu_sync_stream_grid(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *h_stream,
			  const uint8_t *d_stream, uint64_t len, uint32_t chunk, uint32_t flags, uint32_t scramb_init,
			  struct tgpu_sync_result *out, void *stream)
{
	int rc = tgpu_sync_stream_grid_begin(eng, plan, h_stream, d_stream, len, chunk, out, stream);
	if (rc)
		return rc;
	return tgpu_sync_stream_grid_finish(eng, plan, h_stream, len, chunk, flags, scramb_init, out, stream);
}

/* ------------------------------------------------------------------------- */
/* the walk in the form the device runs it (k_walk), on the host              */
/* ------------------------------------------------------------------------- */
/*
 * tgpu_sync_walk_emul(): k_walk's phases one after the other on the CPU, over the same tg_walk_core.h -- nodes from
 * the plain bitmap, every node through tgw_run(), reachability by pointer doubling, bitmap / events / counts from
 * the visited nodes.  Same outputs as tgpu_sync_walk_plain(..., TGPU_SYNC_GRID | TGPU_SYNC_NO_BURST_EVENTS) whenever
 * *status comes back 0; *status = 1 (TGW_FALLBACK, *why = TGW_WHY_*) where the device form hands the channel to the
 * host walk.  This is what the CPU tests fuzz against sync_walk() and the oracle; it is not a product path.
 */
int tgpu_sync_walk_emul(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor, const uint32_t *cls,
			const uint16_t *ysum, const uint32_t *plain, uint32_t ncls, struct tgpu_sync_result *out, int *status,
			int *why)
{
	if (!h_stream || !out || !cls || !ysum || !plain || !ncls || !status || !why || !chunk || (chunk & (chunk - 1)) ||
	    chunk < TGPU_SYNC_CHUNK_MIN || chunk > TGPU_SYNC_CHUNK_MAX)
		return TGPU_EINVAL;
	memset(out, 0, sizeof(*out));
	*status = TGW_OK;
	*why = 0;
	uint64_t a2 = 0;
	int locks = 0;
	struct tg_walk_root root = { 0, 0 };
	int rc = find_anchor_root(h_stream, len, chunk, &a2, &locks, &root);
	if (rc)
		return rc;
	if (!locks || a2 != anchor)
		return TGPU_EINVAL;
	const uint32_t W = (ncls + 31) / 32;
	/* (a channel beyond TGW_WCAP words runs the same steps with the caps of k_walk_big's scratch area -- there made for
	 * the plan's capacity, here for the channel's own length) */
	const int big = W > TGW_WCAP;
	const uint32_t ncap = big ? tg_walk_big_ncap(ncls) : TGW_NCAP, evcap = big ? 4 * ncap : TGW_EVCAP;
	struct tgw_chan wc = { cls, ysum, h_stream, 0, 0, len, anchor, (len + chunk - 1) / chunk, ncls, chunk, (uint32_t)__builtin_ctz(chunk) };
	uint32_t *bm = malloc((size_t)W * 4), *wpre = malloc((size_t)W * 4);
	uint32_t *nslot = malloc((size_t)ncap * 4);
	uint32_t *Ja = malloc(((size_t)ncap + 8) * 4), *Jb = malloc(((size_t)ncap + 8) * 4);
	uint8_t *mark = calloc((size_t)ncap + 8, 1);
	struct tgw_rec *recs = malloc(((size_t)ncap + 1) * sizeof(*recs));
	rc = TGPU_ENOMEM;
	if (!bm || !wpre || !nslot || !Ja || !Jb || !mark || !recs)
		goto done;
	rc = TGPU_OK;
	/* A, B */
	uint32_t N = 0;
	for (uint32_t w = 0; w < W; w++) {
		uint32_t v = plain[w];
		if (w == W - 1 && (ncls & 31))
			v |= ~0u << (ncls & 31);
		bm[w] = v;
		wpre[w] = N;
		for (uint32_t z = ~v; z; z &= z - 1) {
			if (N < ncap)
				nslot[N] = 32 * w + (uint32_t)__builtin_ctz(z);
			N++;
		}
	}
	if (N > ncap) {
		*status = TGW_FALLBACK;
		*why = TGW_WHY_NODES;
		goto done;
	}
#define RANK(t) ((t) >= ncls ? N : wpre[(t) >> 5] + (uint32_t)__builtin_popcount(~bm[(t) >> 5] & ((1u << ((t) & 31)) - 1u)))
	/* C */
	for (uint32_t i = 0; i < N; i++) {
		const uint64_t bs = anchor + (uint64_t)nslot[i] * TG_SLOT_BITS;
		const uint64_t kc = (bs + TG_SLOT_BITS + chunk - 1) >> wc.cshift;
		tgw_run(&wc, TGW_S_LOCKED, bs, bs + TG_SLOT_BITS, kc - 1, &recs[i]);
		Ja[i] = recs[i].status == TGW_OK ? RANK(recs[i].next) : N;
	}
	struct tgw_rec *rr = &recs[ncap];
	tgw_run(&wc, TGW_S_KNOW_FSTART, root.found_bs, anchor, root.found_k, rr);
	if (rr->status != TGW_OK) {
		*status = TGW_FALLBACK;
		*why = rr->why;
		goto done;
	}
	const uint32_t head = RANK(rr->next);
#undef RANK
	Ja[N] = Jb[N] = N;
	/* D */
	if (head < N)
		mark[head] = 1;
	{
		uint32_t *J = Ja, *Jn = Jb;
		for (uint32_t span = 1; span <= N; span <<= 1) {
			for (uint32_t v = 0; v < N; v++)
				if (mark[v] && J[v] < N)
					mark[J[v]] = 1;
			for (uint32_t v = 0; v < N; v++)
				Jn[v] = J[v] < N ? J[J[v]] : N;
			uint32_t *t = J;
			J = Jn;
			Jn = t;
		}
	}
	/* E */
	uint32_t lastnode = 0;
	for (uint32_t i = 0; i <= N; i++) {
		const int isroot = (i == N);
		if (!isroot && !mark[i])
			continue;
		const struct tgw_rec *r = isroot ? rr : &recs[i];
		if (r->status != TGW_OK) {
			*status = TGW_FALLBACK;
			*why = r->why;
			goto done;
		}
		uint32_t from = isroot ? 0 : nslot[i], to = r->next > ncls ? ncls : r->next;
		for (; from < to; from++)
			bm[from >> 5] &= ~(1u << (from & 31));
		if (!isroot)
			lastnode = i + 1;
	}
	for (uint32_t i = 0; i <= N; i++) {
		const int isroot = (i == N);
		if (!isroot && !mark[i])
			continue;
		const struct tgw_rec *r = isroot ? rr : &recs[i];
		for (uint32_t d = 0; d < r->ndel; d++)
			bm[r->del[d] >> 5] |= 1u << (r->del[d] & 31);
	}
	/* F */
	uint32_t ns = 0, lastdel = 0;
	if (ncls & 31)
		bm[W - 1] &= (1u << (ncls & 31)) - 1u;
	for (uint32_t w = 0; w < W; w++) {
		ns += (uint32_t)__builtin_popcount(bm[w]);
		if (bm[w])
			lastdel = 32 * w + 32 - (uint32_t)__builtin_clz(bm[w]);
	}
	/* G */
	uint32_t etot = rr->nev;
	for (uint32_t i = 0; i < N; i++)
		if (mark[i])
			etot += recs[i].nev;
	if (etot > evcap) {
		*status = TGW_FALLBACK;
		*why = TGW_WHY_EVENTS;
		goto done;
	}
	out->events = malloc((size_t)(etot ? etot : 1) * sizeof(*out->events));
	if (!out->events) {
		rc = TGPU_ENOMEM;
		goto done;
	}
	uint32_t nd = 0, tail = 0, e = 0;
	for (uint32_t i = 0; i <= N; i++) {
		const uint32_t j = i ? i - 1 : N;	/* the head run first, then the nodes in slot order */
		if (j != N && !mark[j])
			continue;
		const struct tgw_rec *r = j == N ? rr : &recs[j];
		for (uint32_t q = 0; q < r->nev; q++, e++) {
			out->events[e].ev = (int32_t)r->ev[q][0];
			out->events[e].bitnum = r->ev[q][1];
			out->events[e].arg = r->ev[q][2];
			if (r->evslot[q] != TGW_NOSLOT) {
				nd++;
				if (r->evslot[q] + 1 > lastdel)
					tail++;
			}
		}
	}
	out->nevents = etot;
	out->nslots = ns;
	out->ngrid = ncls;
	out->anchor = anchor;
	out->burst_seq = ns + nd;
	out->tail_tn_adds = tail;
	{
		const struct tgw_rec *lr = lastnode ? &recs[lastnode - 1] : rr;
		out->final_state = lr->next == TGW_END ? lr->end_state : RX_S_LOCKED;
	}
	out->grid_bits = bm;
	bm = NULL;
done:
	free(bm);
	free(wpre);
	free(nslot);
	free(Ja);
	free(Jb);
	free(mark);
	free(recs);
	return rc;
}

/* ------------------------------------------------------------------------- */
/* multi-channel batches with the walk on the device                          */
/* ------------------------------------------------------------------------- */
/*
 * tgpu_sync_multi_launch(): everything a batch needs is enqueued on the caller's stream and the call returns -- first
 * lock of every channel (host, a few kB of each stream), channel table + roots up, k_front_stream (+ fix), k_cls_plain,
 * k_walk (the synchroniser per channel: delivered bitmap, events, counts), the list builder on that bitmap, SB1 -> code
 * fill -> masks -> trellis kernels with the item counts read on the device, summaries + events + bitmap back into pinned
 * memory.  No host wait in between: a step costs the host its launches.
 * tgpu_sync_multi_collect(): waits for the batch and hands out the per-channel outcomes.  Where the device walk could
 * not settle a channel (tg_walk_core.h: only the bytes can decide, more than TGW_NCAP exceptions, ...) the batch is
 * redone through the host walks (tgpu_sync_multi_finish) and decoded again before the call returns -- same results,
 * one batch later in time.
 */
struct tgpu_sync_dev {
	struct tgpu_sync_multi *st;
	const uint8_t *d_base;
	uint8_t *d_rec;
	hipStream_t stream;
	hipEvent_t done;
	struct tg_walk_io io;	/* the batch's blocks: one copy up, one copy down */
	int fellback;
	int fused;		/* the front end and the trellises of this batch ran as one launch (k_slot) */
	int cwire;		/* the compact transport form was enqueued behind the decode (tgpu_plan_set_cwire) */
	uint64_t cwire_bytes;	/* ... its size, known after collect */
	uint64_t cwire_needed;	/* ... or, when the caller's buffer was too small, the bytes it would have taken */
};

void tgpu_sync_dev_free(struct tgpu_sync_dev *sd)
{
	if (!sd)
		return;
	if (sd->done)
		(void)hipEventDestroy(sd->done);
	tgpu_sync_multi_free(sd->st);
	free(sd);
}

/* evs (optional): TGPU_NDEVSTAGES + 1 events recorded around the stages in front of the decode; prof / step: per-stage events
 * of the decode itself (tgpu_plan_execute_prof) */
static int multi_launch(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			const uint8_t *d_base, uint32_t chunk, uint8_t *d_rec, struct tgpu_sync_dev **out, void *stream,
			hipEvent_t *evs, struct tgpu_prof *prof, uint32_t step, int packed_input);

int tgpu_sync_multi_launch(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			   const uint8_t *d_base, uint32_t chunk, uint8_t *d_rec, struct tgpu_sync_dev **out, void *stream)
{
	return multi_launch(eng, plan, nchan, ch, d_base, chunk, d_rec, out, stream, NULL, NULL, 0, 0);
}

int tgpu_sync_multi_launch_packed(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
				  const uint8_t *d_packed_base, uint32_t chunk, uint8_t *d_rec, struct tgpu_sync_dev **out, void *stream)
{
	/* the front end fetches a group from the 16-byte aligned address below its first byte, counted from the buffer's start: a
	 * base that is not 16-byte aligned itself (a sub-buffer) would be read up to 15 bytes in front of it */
	if ((uintptr_t)d_packed_base & 15)
		return TGPU_EINVAL;
	if (ch)
		for (uint32_t c = 0; c < nchan; c++)
			if (ch[c].d_off & 7)
				return TGPU_EINVAL;	/* a channel starts on a byte of the packed buffer */
	return multi_launch(eng, plan, nchan, ch, d_packed_base, chunk, d_rec, out, stream, NULL, NULL, 0, 1);
}

const char *tgpu_sync_dev_stage_name(int stage)
{
	static const char *const names[TGPU_NDEVSTAGES] = { "k_front_stream", "k_front_stream_fix", "k_cls_plain2", "k_vit<SB1>", "k_masks2",
							    "k_walk", "k_lb_scan", "k_lists2" };
	return (stage >= 0 && stage < TGPU_NDEVSTAGES) ? names[stage] : "?";
}

/* measurement aid: one batch, synchronously, with HIP events between all of its stages on hip_stream: dev_ms[] = the
 * TGPU_NDEVSTAGES stages in front of the decode (tgpu_sync_dev_stage_name), the decode's own stages in prof / step as
 * tgpu_plan_execute_prof() leaves them (stage 0, k_front, is empty in stream mode) */
int tgpu_sync_multi_launch_prof(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
				const uint8_t *d_base, uint32_t chunk, uint8_t *d_rec, void *stream, struct tgpu_prof *prof,
				uint32_t step, float dev_ms[TGPU_NDEVSTAGES])
{
	if (!prof || !dev_ms)
		return TGPU_EINVAL;
	int rc = tgpi_engine_bind(eng);
	if (rc)
		return rc;
	hipEvent_t evs[TGPU_NDEVSTAGES + 1];
	memset(evs, 0, sizeof(evs));
	for (int i = 0; i <= TGPU_NDEVSTAGES && !rc; i++)
		rc = (int)hipEventCreate(&evs[i]);
	struct tgpu_sync_dev *sd = NULL;
	if (!rc)
		rc = multi_launch(eng, plan, nchan, ch, d_base, chunk, d_rec, &sd, stream, evs, prof, step, 0);
	if (!rc)
		rc = (int)hipStreamSynchronize((hipStream_t)stream);
	for (int i = 0; i < TGPU_NDEVSTAGES && !rc; i++)
		rc = (int)hipEventElapsedTime(&dev_ms[i], evs[i], evs[i + 1]);
	for (int i = 0; i <= TGPU_NDEVSTAGES; i++)
		if (evs[i])
			(void)hipEventDestroy(evs[i]);
	tgpu_sync_dev_free(sd);
	return rc;
}

static int multi_launch(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			const uint8_t *d_base, uint32_t chunk, uint8_t *d_rec, struct tgpu_sync_dev **out, void *stream,
			hipEvent_t *evs, struct tgpu_prof *prof, uint32_t step, int packed_input)
{
	if (!eng || !plan || !nchan || nchan > 64 || !ch || !d_base || !d_rec || !out)
		return TGPU_EINVAL;
	if (!chunk || (chunk & (chunk - 1)) || chunk < TGPU_SYNC_CHUNK_MIN || chunk > TGPU_SYNC_CHUNK_MAX)
		return TGPU_EINVAL;	/* (other feed sizes: tgpu_sync_multi_begin / _finish) */
	*out = NULL;
	int rc = tgpi_engine_bind(eng);
	if (rc)
		return rc;
	struct tgpu_sync_dev *sd = calloc(1, sizeof(*sd));
	struct tgpu_sync_multi *st = calloc(1, sizeof(*st));
	if (!sd || !st) {
		free(sd);
		free(st);
		return TGPU_ENOMEM;
	}
	sd->st = st;
	sd->d_base = d_base;
	sd->d_rec = d_rec;
	sd->stream = (hipStream_t)stream;
	st->eng = eng;
	st->plan = plan;
	st->nchan = nchan;
	st->chunk = chunk;
	st->ch = malloc((size_t)nchan * sizeof(*ch));
	st->ent = calloc(nchan, sizeof(*st->ent));
	st->locks = calloc(nchan, sizeof(int));
	rc = (st->ch && st->ent && st->locks) ? TGPU_OK : TGPU_ENOMEM;
#ifdef TG_LAUNCH_TIMING	/* measurement builds: which call of a launch takes the time (stderr, launches above 1 ms only) */
	struct timespec lt_[16];
	int ltn_ = 0;
#define LT_MARK() do { if (ltn_ < 16) clock_gettime(CLOCK_MONOTONIC, &lt_[ltn_++]); } while (0)
#else
#define LT_MARK() do { } while (0)
#endif
	LT_MARK();
	if (!rc)
		rc = (int)hipEventCreateWithFlags(&sd->done, hipEventDisableTiming);
	LT_MARK();
	if (!rc)
		memcpy(st->ch, ch, (size_t)nchan * sizeof(*ch));
	uint64_t total = 0;
	struct tg_walk_root roots[64];
	uint32_t codes[64];
	for (uint32_t c = 0; c < nchan && !rc; c++) {
		uint64_t anchor = 0;
		if (!ch[c].h_stream) {
			rc = TGPU_EINVAL;
			break;
		}
		roots[c].found_bs = roots[c].found_k = 0;
		rc = find_anchor_root(ch[c].h_stream, ch[c].len, chunk, &anchor, &st->locks[c], &roots[c]);
		uint64_t n = 0;
		if (!rc && st->locks[c] && anchor + TG_SLOT_BITS <= ch[c].len)
			n = (ch[c].len - anchor) / TG_SLOT_BITS;
		else
			st->locks[c] = 0;
		st->ent[c].d_off = ch[c].d_off | (packed_input ? TG_CHAN_PACKED : 0);	/* (packed ingest: a bit offset) */
		st->ent[c].anchor = anchor;
		st->ent[c].len = ch[c].len;
		st->ent[c].gbase = (uint32_t)total;
		st->ent[c].ncls = (uint32_t)n;
		codes[c] = ch[c].scramb_init;
		total += (n + 31) & ~(uint64_t)31;
		if (total > 0xfffffff0u)
			rc = TGPU_ECAPACITY;
	}
	st->ngrid = (uint32_t)total;
	if (!rc && st->ngrid) {
		uint32_t *d_packed, *d_cls, *cls, *d_plain, *h_plain, *d_bits = NULL;
		uint16_t *d_ysum, *ysum;
		struct tg_walk_io *io = &sd->io;
		rc = tgpi_plan_grid_begin(plan, st->ngrid, &d_packed, &d_cls, &d_ysum, &cls, &ysum);
		if (!rc)
			rc = tgpi_plan_walk_io(plan, nchan, st->ngrid, io);
		if (!rc) {	/* channel table, roots and carry-in codes: one block, one copy */
			memcpy(io->h_tab, st->ent, (size_t)nchan * sizeof(*st->ent));
			memcpy(io->h_roots, roots, (size_t)nchan * sizeof(*roots));
			memcpy(io->h_codes, codes, (size_t)nchan * 4);
			LT_MARK();
			/* (a few KB: fetched by a kernel from the mapped block -- no copy engine, no dependency between engines in front of
			 * the front end, a cheaper call; the block is this plan's until the batch is collected) */
			rc = io->hd_up0 ? tgk_copy16(io->hd_up0, io->d_up0, io->up_bytes, sd->stream)
					: (int)hipMemcpyAsync(io->d_up0, io->h_up0, io->up_bytes, hipMemcpyHostToDevice, sd->stream);
			LT_MARK();
		}
#define EVMARK(i) do { if (evs && !rc) rc = (int)hipEventRecord(evs[i], sd->stream); } while (0)
		/* per-kernel timing: the front end's start event must not be passed while the copy up is still on its way (the copy engine's
		 * work is a dependency of the kernel, not of the event: the kernel's "duration" then began with the wait for it, ~20 us):
		 * a 16-byte kernel in between waits in the event's place */
		if (evs && !rc && !io->hd_up0)
			rc = tgk_copy16(io->d_down0, io->d_down0, 16, sd->stream);
		/* the batch's arena (counters, code table, okbits: cleared) in front of the front end: a k_slot launch runs the code
		 * look-back's atomics itself */
		if (!rc) {
			tgpi_plan_grid_plain(plan, st->ngrid, &d_plain, &h_plain);
			tgpi_plan_set_rec(plan, d_rec);
			rc = tgpi_plan_dev_prepare(plan, st->ngrid, nchan, io->d_codes, &d_bits, stream);
		}
		if (evs && !rc)		/* (armed only when the launch it is for follows: the caller destroys the event) */
			tgk_front_stream_ev_start(evs[0]);
		/* round 6: front end and trellises in one launch (k_slot) where the channels have a code to decode on -- the caller's carry-in,
		 * else what the plan's last batch ended with; otherwise (a plan's first batch, TGPU_OPT_SLOT < 2) the front end on its own */
		int fused = 0;
		if (!rc)
			rc = tgpi_plan_dev_front_fused(plan, d_base, io->d_tab, nchan, st->ngrid, chunk, codes, stream, evs ? evs[1] : NULL, packed_input, &fused);
		sd->fused = fused;
		if (!rc && !fused)
			rc = tgk_front_stream_multi(d_base, io->d_tab, nchan, st->ngrid, chunk, d_packed, d_cls, d_ysum,
						    tgpi_plan_defer_scratch(plan), stream, evs ? evs[1] : NULL, packed_input);
		/* TGPU_OPT_SLOT 3: the trellises of every plain slot on the hinted codes start NOW, beside what follows (k_slot_e) */
		if (!rc && !fused && !evs)
			rc = tgpi_plan_dev_early(plan, io->d_tab, nchan, st->ngrid, codes, stream);
		if (!rc && tgpi_plan_is_early(plan))
			sd->fused = 2;
		LT_MARK();
		EVMARK(2);
		if (!rc)	/* plain bitmap + SYNC list; SB1 and the masks beside the walk (side stream), or in line when stages are timed */
			rc = tgpi_plan_dev_stage1(plan, st->ngrid, nchan, io->d_tab, io->d_codes, d_plain, stream, evs != NULL,
						  evs ? (void **)(evs + 3) : NULL);
		/* channels of more than 262 144 slots -- or of fewer, once the plan has seen the LDS form overflow with exceptions
		 * (tgpi_plan_walk_threshold) --: the same walk with its arrays in global memory, behind the first */
		struct tg_walk_big big = { 0 };
		unsigned long long skip = 0;
		const uint32_t thr = tgpi_plan_walk_threshold(plan);
		uint32_t wmax = 0;	/* bitmap words of the longest channel that stays in the LDS form */
		for (uint32_t c = 0; c < nchan; c++)
			if (st->ent[c].ncls > thr && big.n < TGW_BIG_MAX) {
				big.chan[big.n++] = c;
				skip |= 1ull << c;
			} else if ((st->ent[c].ncls + 31) / 32 > wmax && st->ent[c].ncls <= TGW_WCAP * 32u)
				wmax = (st->ent[c].ncls + 31) / 32;
		if (!rc)
			rc = tgk_walk(d_base, io->d_tab, io->d_roots, nchan, chunk, d_cls, d_ysum, d_plain, d_bits, io->d_bits2, io->d_sums,
				      io->d_eager, io->d_evbig, io->d_recs, tgi_option(TGPU_OPT_WALK_MONO) ? NULL : io->d_tmp, skip,
				      wmax ? wmax : 1, tgpi_plan_walk_ncap(plan), io->rec_stride, (int)tgi_option(TGPU_OPT_WALK_WIDE), stream);
		if (!rc) {
			if (big.n) {
				rc = tgpi_plan_walk_big(plan, big.n, io);
				io->big.n = big.n;
				memcpy(io->big.chan, big.chan, sizeof(big.chan));
				if (!rc)
					rc = tgk_walk_big(&io->big, io->d_big, d_base, io->d_tab, io->d_roots, chunk, d_cls, d_ysum, d_plain,
							  d_bits, io->d_bits2, io->d_sums, io->d_eager, tgi_option(TGPU_OPT_WALK_MONO) ? NULL : io->d_tmp, stream);
			}
		}
		EVMARK(6);
		if (!rc)
			rc = tgpi_plan_dev_stage2(plan, io->d_tab, io->d_final, stream, evs != NULL, evs ? (void **)(evs + 7) : NULL);
#undef EVMARK
		if (!rc)
			rc = prof ? tgpu_plan_execute_prof(plan, d_base, d_rec, stream, prof, step) : tgpu_plan_execute(plan, d_base, d_rec, stream);
		/* the compact transport form of the batch's wire records (tgpu_plan_set_cwire), its size into the block below */
		if (!rc && tgpi_plan_has_cwire(plan)) {
			struct tg_cw_chans cw;
			memset(&cw, 0, sizeof(cw));
			cw.n = nchan;
			for (uint32_t c = 0; c < nchan; c++) {
				cw.gbase[c] = st->ent[c].gbase;
				cw.ncls[c] = st->ent[c].ncls;
			}
			sd->cwire = 1;
			rc = tgpi_plan_cwire(plan, &cw, io->d_final + 66, stream);
		}
		/* what the host wants to know -- summaries, the first events of every channel, the bitmap -- in one copy */
		LT_MARK();
		if (!rc)
			rc = io->hd_down0 ? tgk_copy16(io->d_down0, io->hd_down0, io->down_bytes, sd->stream)
					  : (int)hipMemcpyAsync(io->h_down0, io->d_down0, io->down_bytes, hipMemcpyDeviceToHost, sd->stream);
		LT_MARK();
	}
	if (!rc)
		rc = (int)hipEventRecord(sd->done, sd->stream);
	LT_MARK();
#ifdef TG_LAUNCH_TIMING
	if (ltn_ > 1 && (lt_[ltn_ - 1].tv_sec - lt_[0].tv_sec) * 1e3 + (lt_[ltn_ - 1].tv_nsec - lt_[0].tv_nsec) * 1e-6 > 1.0) {
		fprintf(stderr, "launch marks (ms):");
		for (int i = 1; i < ltn_; i++)
			fprintf(stderr, " %.3f", (lt_[i].tv_sec - lt_[i - 1].tv_sec) * 1e3 + (lt_[i].tv_nsec - lt_[i - 1].tv_nsec) * 1e-6);
		fprintf(stderr, "\n");
	}
#endif
	if (rc) {
		/* stage 1 may have forked work onto the plan's side stream and part of the batch may be running: nothing of it may
		 * outlive the objects freed here, and this thread's armed start event must not reach a later launch */
		tgk_front_stream_ev_start(NULL);
		(void)hipDeviceSynchronize();
		tgpu_sync_dev_free(sd);
		return rc;
	}
	*out = sd;
	return TGPU_OK;
}

uint32_t tgpu_sync_dev_ngrid(const struct tgpu_sync_dev *sd)
{
	return sd ? sd->st->ngrid : 0;
}

int tgpu_sync_dev_fellback(const struct tgpu_sync_dev *sd)
{
	return sd ? sd->fellback : 0;
}

int tgpu_sync_dev_fused(const struct tgpu_sync_dev *sd)
{
	return sd ? sd->fused : 0;
}

/* why the device walk handed channel c over (after collect): 0 = it did not; TGPU_WHY_* otherwise */
int tgpu_sync_dev_why(const struct tgpu_sync_dev *sd, uint32_t c)
{
	if (!sd || c >= sd->st->nchan || !sd->st->ngrid || !sd->io.h_sums || !sd->st->ent[c].ncls)
		return 0;
	if (sd->io.h_sums[c].status == TGW_OK)
		return 0;
	return sd->io.h_sums[c].why ? (int)sd->io.h_sums[c].why : -1;
}

uint64_t tgpu_sync_dev_cwire_bytes(const struct tgpu_sync_dev *sd)
{
	return sd ? sd->cwire_bytes : 0;
}

uint64_t tgpu_sync_dev_cwire_needed(const struct tgpu_sync_dev *sd)
{
	return sd ? sd->cwire_needed : 0;
}

int tgpu_sync_multi_collect(struct tgpu_sync_dev *sd, struct tgpu_sync_result *out)
{
	if (!sd || !out)
		return TGPU_EINVAL;
	struct tgpu_sync_multi *st = sd->st;
	memset(out, 0, (size_t)st->nchan * sizeof(*out));
	int rc = tgpi_engine_bind(st->eng);
	if (rc)
		return rc;
	/* wait for the batch: the stream's completion event is polled with short sleeps in between -- a spinning
	 * hipEventSynchronize() costs a whole core per GPU for as long as the GPU works */
	{
		/* (a thread's timer slack, 50 us by default, is added to every sleep: for the duration of the wait it is set to
		 * 1 us, so that a nap is the 20 us asked for and the batch's successor is launched that much sooner) */
		hipError_t q;
		const struct timespec nap = { 0, 20000 };
		const int slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
		if (slack > 1000)
			prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
		while ((q = hipEventQuery(sd->done)) == hipErrorNotReady)
			nanosleep(&nap, NULL);
		if (slack > 1000)
			prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
		rc = (int)q;
	}
	if (rc)
		return rc;
	int fb = 0;
	for (uint32_t c = 0; c < st->nchan; c++)
		if (st->ent[c].ncls && sd->io.h_sums[c].status != TGW_OK) {
			fb = 1;
			if (sd->io.h_sums[c].why == TGW_WHY_NODES && sd->io.h_sums[c].nnodes > TGW_NCAP)	/* more exceptions than the LDS form holds: the long form next time */
				tgpi_plan_walk_overflow(st->plan, st->ent[c].ncls, sd->io.h_sums[c].nnodes);
		}
	for (uint32_t c = 0; c < st->nchan; c++)	/* (what sizes the next launch's node arrays) */
		if (st->ent[c].ncls && sd->io.h_sums[c].nnodes <= TGW_NCAP)
			tgpi_plan_walk_seen(st->plan, sd->io.h_sums[c].nnodes);
	if (st->ngrid && sd->io.h_final[64])
		fb = 1;		/* more scrambling codes in the batch than the device path's table holds */
	if (fb || tgi_option(TGPU_OPT_WALK_HOST)) {
		/* the host walks decide: classification words, summaries and plain bitmap over, walks, bitmap up, lists, decode */
		sd->fellback = 1;
		if (st->ngrid) {
			uint32_t *d_packed, *d_cls, *cls;
			uint16_t *d_ysum, *ysum;
			struct tg_chan_ent *d_tab;
			rc = tgpi_plan_grid_begin(st->plan, st->ngrid, &d_packed, &d_cls, &d_ysum, &cls, &ysum);
			if (!rc)
				rc = tgpi_plan_chan_table(st->plan, st->ent, st->nchan, &d_tab, sd->stream);	/* (the host path's own copy of the table) */
			if (!rc)
				rc = (int)hipMemcpyAsync(cls, d_cls, TG_GRID_COPY_BYTES(st->ngrid), hipMemcpyDeviceToHost, sd->stream);
			if (rc)
				return rc;
		}
		rc = tgpu_sync_multi_finish(st, TGPU_SYNC_NO_BURST_EVENTS, 4, out, sd->stream);
		if (!rc && st->ngrid)
			rc = tgpu_plan_execute(st->plan, sd->d_base, sd->d_rec, sd->stream);
		if (!rc)
			rc = (int)hipStreamSynchronize(sd->stream);
		return rc;
	}
	for (uint32_t c = 0; c < st->nchan && !rc; c++) {
		const struct tg_chan_ent *e = &st->ent[c];
		struct tgpu_sync_result *o = &out[c];
		if (!e->ncls) {	/* never locks (or nothing behind the lock): nothing of it is in the grid; the bytes settle it */
			rc = tgpu_sync_walk(st->ch[c].h_stream, st->ch[c].len, st->chunk, e->anchor, NULL, NULL, 0, TGPU_SYNC_NO_BURST_EVENTS, o);
			o->noffgrid = o->nslots;
			o->anchor = e->anchor;
			o->grid_base = e->gbase;
			continue;
		}
		const struct tg_walk_sum *s = &sd->io.h_sums[c];
		o->nslots = s->nslots;
		o->nevents = s->nevents;
		o->final_state = (int)s->final_state;
		o->tail_tn_adds = s->tail_tn_adds;
		o->burst_seq = s->burst_seq;
		o->anchor = e->anchor;
		o->ngrid = e->ncls;
		o->grid_base = e->gbase;
		const size_t nw = ((size_t)e->ncls + 31) / 32;
		o->grid_bits = malloc(nw * 4);
		o->events = malloc((size_t)(s->nevents ? s->nevents : 1) * sizeof(*o->events));
		if (!o->grid_bits || !o->events) {
			for (uint32_t k = 0; k <= c; k++)	/* nothing half-filled is handed back with an error */
				tgpu_sync_result_free(&out[k]);
			rc = TGPU_ENOMEM;
			break;
		}
		memcpy(o->grid_bits, sd->io.h_bits2 + e->gbase / 32, nw * 4);
		const uint32_t eager = s->nevents < TGW_EVEAGER ? s->nevents : TGW_EVEAGER;
		memcpy(o->events, sd->io.h_eager + (size_t)c * TGW_EVEAGER, (size_t)eager * sizeof(*o->events));
		if (s->nevents > eager) {	/* a channel with many exceptions: the rest of its events in a second copy */
			const tgpu_sync_event_rec_dev *src = sd->io.d_evbig + (size_t)c * TGW_EVCAP;
			for (uint32_t k = 0; k < sd->io.big.n; k++)
				if (sd->io.big.chan[k] == c) {		/* (a long channel's are in its scratch slot) */
					struct tg_walk_big_layout L;
					tg_walk_big_offsets(sd->io.big.wcap, sd->io.big.ncap, sd->io.big.evcap, &L);
					src = (const tgpu_sync_event_rec_dev *)(sd->io.d_big + (size_t)k * L.slot_bytes + L.o_ev);
				}
			/* (on the batch's own stream, which is idle by now: a plain hipMemcpy() goes through the null stream and
			 * waits for every other batch in flight) */
			rc = (int)hipMemcpyAsync(o->events + eager, src + eager, (size_t)(s->nevents - eager) * sizeof(*o->events),
						 hipMemcpyDeviceToHost, sd->stream);
			if (!rc)
				rc = (int)hipStreamSynchronize(sd->stream);
		}
		uint32_t last = 0xffffffffu;
		for (size_t wd = nw; wd-- > 0;)
			if (o->grid_bits[wd]) {
				last = e->gbase + (uint32_t)(wd * 32 + 31 - (uint32_t)__builtin_clz(o->grid_bits[wd]));
				break;
			}
		tgpi_plan_set_last_slot(st->plan, c, last);
	}
	if (!rc && st->ngrid)
		tgpi_plan_set_final_codes(st->plan, sd->io.h_final, st->nchan);
	if (!rc && st->ngrid && sd->cwire) {
		/* the buffer given to tgpu_plan_set_cwire() may be smaller than tgpu_cwire_bound(); a batch that needed more is a
		 * valid batch all the same (records, bitmap, events are the caller's): it has no compact form (cwire_bytes 0) and
		 * says what it would have taken */
		if (sd->io.h_final[67] == 0xffffffffu)
			sd->cwire_needed = sd->io.h_final[66] ? sd->io.h_final[66] : tgpu_cwire_bound(st->ngrid, st->nchan);
		else
			sd->cwire_bytes = sd->io.h_final[66];
	}
	return rc;
}


/* Hand every delivered burst's wire record of a gathered batch to a callback (host): wire = the 40-byte records of ngrid
 * grid slots, grid_bits = the delivered bitmap of the same slots (NULL: every slot whose record carries a burst type).
 * cb == NULL counts only.  Returns the number of records handed over.  What the end-to-end measurement of bench.py ends
 * in: the consumer's side of "records delivered to a callback" without unpacking (tgpu_wire_unpack does that per record). */
uint64_t tgpu_wire_foreach(const uint8_t *wire, const uint32_t *grid_bits, uint32_t ngrid, tgpu_wire_cb cb, void *priv)
{
	uint64_t n = 0;
	if (!wire)
		return 0;
	if (grid_bits) {
		for (uint32_t w = 0; w < (ngrid + 31) / 32; w++)
			for (uint32_t z = grid_bits[w]; z; z &= z - 1) {
				const uint32_t g = 32 * w + (uint32_t)__builtin_ctz(z);
				if (g >= ngrid)
					break;
				if (cb)
					cb(wire + (size_t)g * TG_WIRE_BYTES, g, priv);
				n++;
			}
		return n;
	}
	for (uint32_t g = 0; g < ngrid; g++)
		if (wire[(size_t)g * TG_WIRE_BYTES] != 0xff) {
			if (cb)
				cb(wire + (size_t)g * TG_WIRE_BYTES, g, priv);
			n++;
		}
	return n;
}

static void wire_noop(const uint8_t *rec, uint32_t slot, void *priv)
{
	(void)slot;
	*(volatile uint64_t *)priv += rec[0];	/* touches the record: the consumer at least reads the header */
}

tgpu_wire_cb tgpu_wire_noop_cb(void)
{
	return wire_noop;
}
