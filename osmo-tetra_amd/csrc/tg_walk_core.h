/*
 * tg_walk_core.h -- the burst synchroniser's walk over a classified grid in the form the GPU runs it.
 *
 * tetra_burst_sync_in() (phy/tetra_burst_sync.c:54-154) is sequential per channel, but on a classified grid almost
 * all of it is the same step: a slot whose classification word alone says "delivered" (k_cls_plain's bitmap) is
 * delivered whatever happened before, as long as the synchroniser is LOCKED on it.  Everything else hangs off the
 * other slots, the NODES:
 *
 *   canonical arrival at grid slot g := LOCKED, buffer starts at the slot, and the call that completes the slot has not
 *   run yet (k < ceil((bs + 510) / chunk)): the slot is then handled by exactly that call with the steady-state
 *   window the classification kernel assumed.  After a run of plain deliveries the synchroniser is in that state at
 *   the next slot (tg_stream.c, closed form of the steady state).
 *
 *   tgw_run() takes the synchroniser from a canonical arrival at a node through whatever the node causes -- a
 *   misplaced-sequence event, loss of lock, the UNLOCKED search over the SYNC summaries, KNOW_FSTART, the bursts
 *   that are consumed one per call while a backlog lasts -- up to the next canonical arrival, and records what
 *   happened on the way: events, the slots delivered meanwhile, where it arrives.  Its result depends on nothing
 *   but the node, so all nodes of a channel are processed at once (one lane each); which of them the walk really
 *   visits is then a reachability question on the arrival pointers (pointer doubling, k_walk), and the delivered
 *   bitmap is the plain bitmap minus the spans of the visited nodes plus their own deliveries.
 *
 * The arithmetic is tg_stream.c's (sync_walk: same call / window bookkeeping, same search), restricted to what can be
 * settled without tetra_find_train_seq() on the bytes; where the host walk would call that (a hit below offset 21 of a
 * buffer, a byte other than 0 / 1 nearby, a window the kernel did not see, a re-lock off the grid) the result is
 * TGW_FALLBACK and the caller runs the host walk for the batch.  Only power-of-two feed sizes inside the closed form's
 * range, no per-burst events (TGPU_SYNC_NO_BURST_EVENTS), grid mode.
 *
 * The same source is compiled for the device (tg_k_walk.hip) and for the host (tgpu_sync_walk_emul(), which runs the
 * kernel's phases one after the other on the CPU and is what the CPU test-suite fuzzes against sync_walk() and the
 * oracle).
 */
#ifndef TG_WALK_CORE_H
#define TG_WALK_CORE_H

#include <stdint.h>
#include "tg_layout.h"

#if defined(__HIPCC__)
#define TGW_FN __device__ static
#else
#define TGW_FN static inline
#endif

#define TGW_MAX_EV   6		/* events one node may cause before the next canonical arrival */
#define TGW_MAX_DEL  10		/* bursts delivered meanwhile (consumed one per call while a backlog lasts: < 4096 / 510) */
#define TGW_MAX_ITER 64		/* slots one node may walk through */
#define TGW_MAX_SCAN 65536	/* grid slots one UNLOCKED search may look ahead */
#define TGW_END      0xffffffffu
#define TGW_NOSLOT   0xffffffffu

enum { TGW_OK = 0, TGW_FALLBACK = 1 };
/* why a node gave up (status = TGW_FALLBACK): diagnostics only */
enum { TGW_WHY_FLAGS = 1, TGW_WHY_WINDOW, TGW_WHY_EARLY, TGW_WHY_OFFGRID, TGW_WHY_ITER, TGW_WHY_EVENTS, TGW_WHY_MULTI,
       TGW_WHY_SCAN, TGW_WHY_NEGATIVE, TGW_WHY_NODES, TGW_WHY_SIZE };

/* states = enum rx_state of phy/tetra_burst_sync.h:6-10 */
#define TGW_S_UNLOCKED    0
#define TGW_S_KNOW_FSTART 1
#define TGW_S_LOCKED      2

/* events = enum tgpu_sync_event (include/tetra_gpu.h) */
#define TGW_EV_FOUND_SYNC     1
#define TGW_EV_SYNC_MISPLACED 3
#define TGW_EV_NORM_MISPLACED 4
#define TGW_EV_NO_TRAIN       5

struct tgw_chan {
	const uint32_t *cls;	/* the channel's ncls classification words */
	const uint16_t *ysum;	/* its SYNC-sequence summaries */
	const uint8_t *s;	/* its stream bytes (read only at the stream's tail and where a summary is in doubt) */
	uint64_t sbit;		/* packed ingest (TG_CHAN_PACKED): s = the packed buffer, the channel's bit 0 is its bit sbit */
	int packed;
	uint64_t len, anchor, ncalls;
	uint32_t ncls, chunk, cshift;
};

struct tgw_rec {
	uint32_t next;		/* grid slot of the canonical arrival that follows; TGW_END: the stream ends first */
	uint8_t nev, ndel, status, end_state;	/* end_state: enum rx_state when next == TGW_END */
	uint8_t why, pad[3];
	uint32_t del[TGW_MAX_DEL];	/* grid slots delivered on the way */
	uint32_t ev[TGW_MAX_EV][3];	/* ev, bitnum, arg as struct tgpu_sync_event_rec */
	uint32_t evslot[TGW_MAX_EV];	/* grid slot of a burst that was handled and not delivered, else TGW_NOSLOT */
};

TGW_FN uint64_t tgw_fed(const struct tgw_chan *c, uint64_t k)
{
	const uint64_t f = k << c->cshift;
	return f > c->len ? c->len : f;
}

/* first call index whose fed count reaches 'pos' (>= 1); ncalls + 1 if never */
TGW_FN uint64_t tgw_call_reaching(const struct tgw_chan *c, uint64_t pos)
{
	if (pos > c->len)
		return c->ncalls + 1;
	const uint64_t k = (pos + c->chunk - 1) >> c->cshift;
	return k ? k : 1;
}

/* stream position p of the channel as the byte the reference would read there */
TGW_FN uint32_t tgw_byte(const struct tgw_chan *c, uint64_t p)
{
	if (c->packed) {
		const uint64_t b = c->sbit + p;
		return (c->s[b >> 3] >> (b & 7)) & 1u;
	}
	return c->s[p];
}

TGW_FN int tgw_is_y(const struct tgw_chan *c, uint64_t p)
{
	/* EN 300 392-2 9.4.4.3.4, the 38-bit synchronisation training sequence */
	const uint64_t Y = 0x3983973983ull;	/* bit i = y[i] */
	for (int i = 0; i < 38; i++)
		if (tgw_byte(c, p + i) != ((Y >> i) & 1))
			return 0;
	return 1;
}

/* first start of the SYNC sequence in [from, last] by looking at the bytes (the stream's tail behind the last grid
 * slot, a slot with several sequences): rare, short ranges */
TGW_FN uint64_t tgw_scan_bytes(const struct tgw_chan *c, uint64_t from, uint64_t last)
{
	for (uint64_t p = from; p <= last; p++)
		if (tgw_byte(c, p) == 1 && tgw_byte(c, p + 1) == 1 && tgw_byte(c, p + 2) == 0 && tgw_is_y(c, p))
			return p;
	return UINT64_MAX;
}

/* tg_stream.c:next_sync_seq() on the summaries; *fb is set where the host version would have to decide more than this
 * one can */
TGW_FN uint64_t tgw_next_sync(const struct tgw_chan *c, uint64_t from, uint64_t last, int *fb, uint8_t *why)
{
	if (c->len < 38)
		return UINT64_MAX;
	if (last > c->len - 38)
		last = c->len - 38;
	uint64_t p = from;
	uint32_t looked = 0;
	while (p <= last) {
		if (p < c->anchor) {	/* (not after the first lock) */
			*fb = 1;
			*why = TGW_WHY_NEGATIVE;
			return UINT64_MAX;
		}
		const uint64_t g = (p - c->anchor) / TG_SLOT_BITS;
		if (g >= c->ncls)
			return tgw_scan_bytes(c, p, last);
		if (++looked > TGW_MAX_SCAN) {
			*fb = 1;
			*why = TGW_WHY_SCAN;
			return UINT64_MAX;
		}
		const uint64_t s0 = c->anchor + g * TG_SLOT_BITS;
		const uint32_t v = c->ysum[g];
		if (v != TG_YS_NONE) {
			const uint64_t fp = s0 + TG_YS_FIRST(v);
			/* a byte other than 0 / 1 in this slot's or the next one's window (the kernel reads it as 1), or no
			 * next window: the bytes decide whether the summary's sequence is one */
			const int doubt = g + 1 >= c->ncls || (((c->cls[g] | c->cls[g + 1]) >> 24) & TG_CLS_NONBINARY);
			const int real = !doubt || tgw_is_y(c, fp);
			if (fp >= p && real)
				return fp <= last ? fp : UINT64_MAX;
			if ((v & TG_YS_MULTI) || !real) {
				const uint64_t e = s0 + TG_SLOT_BITS - 1 < last ? s0 + TG_SLOT_BITS - 1 : last;
				const uint64_t r = tgw_scan_bytes(c, p, e);
				if (r != UINT64_MAX)
					return r;
			}
		}
		p = s0 + TG_SLOT_BITS;
	}
	return UINT64_MAX;
}

TGW_FN int tgw_event(struct tgw_rec *r, uint32_t ev, uint64_t bitnum, uint32_t arg, uint32_t gslot)
{
	if (r->nev == TGW_MAX_EV) {
		r->status = TGW_FALLBACK;
		r->why = TGW_WHY_EVENTS;
		return 1;
	}
	r->ev[r->nev][0] = ev;
	r->ev[r->nev][1] = (uint32_t)bitnum;
	r->ev[r->nev][2] = arg;
	r->evslot[r->nev] = gslot;
	r->nev++;
	return 0;
}

/*
 * From (state, bs, nfs, k) to the next canonical arrival.
 *   a node            : state = LOCKED, bs = the node's slot, k = any call before the one that completes the slot
 *   the stream's head : state = KNOW_FSTART right after the first lock (bs = buffer start of the call that found the
 *                       SYNC sequence, nfs = the grid's anchor, k = that call) -- tg_stream.c:find_anchor()
 * 'first' slots are handled whatever their word says; the run ends in front of the first slot that is reached
 * canonically after them.
 */
TGW_FN void tgw_run(const struct tgw_chan *c, int state, uint64_t bs, uint64_t nfs, uint64_t k, struct tgw_rec *r)
{
	const uint32_t A = TG_BURST_SYNC | TG_SYNC_TRAIN_OFF << 8, B = TG_BURST_NORM_1 | TG_NORM_TRAIN_OFF << 8,
		       C = TG_BURST_NORM_2 | TG_NORM_TRAIN_OFF << 8;
	r->next = TGW_END;
	r->nev = r->ndel = 0;
	r->status = TGW_OK;
	r->why = 0;
	r->end_state = (uint8_t)state;
	int first = (state == TGW_S_LOCKED);	/* a node's own slot is handled even though it is reached canonically */
	if (state == TGW_S_KNOW_FSTART)		/* the stream's head: the first lock is this run's first event */
		(void)tgw_event(r, TGW_EV_FOUND_SYNC, bs, (uint32_t)(nfs - 296 - bs), TGW_NOSLOT);
	for (uint32_t iter = 0;; iter++) {
		if (iter == TGW_MAX_ITER) {
			r->status = TGW_FALLBACK;
			r->why = TGW_WHY_ITER;
			return;
		}
		if (state == TGW_S_UNLOCKED) {
			/* tg_stream.c, UNLOCKED: calls k + 1, ...: buffer [b, F(kk)), b = max(bs, F(kk) - 4096); nothing below 1020
			 * buffered bytes; positions are looked at once ('clean_to') */
			uint64_t found_p = 0, found_k = 0, found_bs = 0;
			int found = 0, fb = 0;
			uint64_t kk = k + 1;
			const uint64_t k1020 = tgw_call_reaching(c, bs + 2 * TG_SLOT_BITS);
			if (kk < k1020)
				kk = k1020;
			uint64_t clean_to = bs;
			for (uint32_t tries = 0; kk <= c->ncalls && !found; kk++, tries++) {
				if (tries == TGW_MAX_ITER) {
					r->status = TGW_FALLBACK;
					r->why = TGW_WHY_ITER;
					return;
				}
				const uint64_t f = tgw_fed(c, kk);
				const uint64_t b = (f > 4096 && f - 4096 > bs) ? f - 4096 : bs;
				if (clean_to < b)
					clean_to = b;
				if (f - b < 2 * TG_SLOT_BITS || f < 38)
					continue;
				const uint64_t last = f - 38;
				while (clean_to <= last) {
					const uint64_t p = tgw_next_sync(c, clean_to, last, &fb, &r->why);
					if (fb) {
						r->status = TGW_FALLBACK;
						return;
					}
					if (p == UINT64_MAX) {
						const uint64_t pn = tgw_next_sync(c, last + 1, UINT64_MAX - 64, &fb, &r->why);
						if (fb) {
							r->status = TGW_FALLBACK;
							return;
						}
						if (pn == UINT64_MAX) {
							kk = c->ncalls;
							clean_to = c->len;
							break;
						}
						clean_to = pn;
						const uint64_t kn = tgw_call_reaching(c, pn + 38);
						if (kn > kk + 1)
							kk = kn - 1;
						break;
					}
					if (p - b >= 21) {
						found = 1;
						found_p = p;
					} else {
						/* inside the skewed zone of the reference's look-ahead filter: its own routine on the bytes
						 * of this very buffer decides -- the host's business */
						r->status = TGW_FALLBACK;
						r->why = TGW_WHY_EARLY;
						return;
					}
					break;
				}
				if (found) {
					found_k = kk;
					found_bs = b;
				}
			}
			if (!found) {
				r->end_state = TGW_S_UNLOCKED;
				return;
			}
			if (tgw_event(r, TGW_EV_FOUND_SYNC, found_bs, (uint32_t)(found_p - found_bs), TGW_NOSLOT))
				return;
			bs = found_bs;
			nfs = found_p + 296;
			k = found_k;
			state = TGW_S_KNOW_FSTART;
		}
		if (state == TGW_S_KNOW_FSTART) {
			uint64_t kl = tgw_call_reaching(c, nfs);
			if (kl <= k)
				kl = k + 1;
			if (kl > c->ncalls) {
				r->end_state = TGW_S_KNOW_FSTART;
				return;
			}
			if (nfs < c->anchor || (nfs - c->anchor) % TG_SLOT_BITS) {
				r->status = TGW_FALLBACK;	/* a lock off the classified grid: the slot-table path */
				r->why = TGW_WHY_OFFGRID;
				return;
			}
			bs = nfs;
			nfs += TG_SLOT_BITS;
			state = TGW_S_LOCKED;
			k = kl - 1;
		}
		/* LOCKED at a grid slot */
		const uint64_t gi = (bs - c->anchor) / TG_SLOT_BITS;
		const uint64_t need = bs + TG_SLOT_BITS;
		r->end_state = TGW_S_LOCKED;
		if (gi >= c->ncls || need > c->len)
			return;
		const uint64_t kc = (need + c->chunk - 1) >> c->cshift;
		const uint64_t kj = kc > k ? kc : k + 1;
		if (kj > c->ncalls)
			return;
		const int canonical = (kj == kc);
		if (canonical && !first) {
			r->next = (uint32_t)gi;
			return;
		}
		first = 0;
		k = kj;
		const uint32_t cw = c->cls[gi];
		const uint32_t v = cw & 0x03ffffffu;
		if (v == A || v == B || v == C) {
			/* (only while a backlog lasts: a plain slot reached canonically ends the run above) */
			if (r->ndel == TGW_MAX_DEL) {
				r->status = TGW_FALLBACK;
				r->why = TGW_WHY_EVENTS;
				return;
			}
			r->del[r->ndel++] = (uint32_t)gi;
		} else {
			const uint32_t cflags = cw >> 24;
			if (cflags & (TG_CLS_EARLY21 | TG_CLS_NONBINARY)) {
				r->status = TGW_FALLBACK;
				r->why = TGW_WHY_FLAGS;
				return;
			}
			const uint32_t type = cw & 0xff, offs = (cw >> 8) & 0xffff;
			if (type == TG_BURST_NONE) {
				/* "nothing" is the kernel's answer for the window it looked at (the steady-state one, unless clipped) and,
				 * with TG_CLS_NOVIEW, for every longer window its view (TG_VIEW_OF) covers: a slot that is handled one or two
				 * calls late (the first burst after a re-lock onto the very next SYNC burst) */
				const uint64_t fj = tgw_fed(c, kj);
				int vt = -1;	/* the longer window's find, if it has one */
				int settled = (canonical && !(cflags & TG_CLS_CLIPPED)) ||
					      ((cflags & TG_CLS_NOVIEW) && fj - bs <= TG_VIEW_OF(c->chunk));
				if (!settled && !(cflags & TG_CLS_NOVIEW) && TG_CLS_VIEWHIT(cflags) && fj - bs <= TG_VIEW_OF(c->chunk)) {
					/* the first sequence that starts in view and ends past the kernel's window: a window that holds all of
					 * it finds it (nothing starts before it); one that ends before a 22-bit sequence would fit there finds
					 * nothing; in between (a 38-bit one cut off, room for a 22-bit one behind its start) only the bytes tell */
					const uint32_t t1 = TG_CLS_VIEWHIT(cflags) - 1u, l1 = t1 == TG_BURST_SYNC ? 38u : 22u;
					if (offs + l1 <= fj - bs) {
						vt = (int)t1;
						settled = 1;
					} else if (offs + 22u > fj - bs)
						settled = 1;
				}
				if (!settled) {
					r->status = TGW_FALLBACK;
					r->why = TGW_WHY_WINDOW;
					return;
				}
				if (vt == (int)TG_BURST_SYNC || vt < 0) {
					if (tgw_event(r, vt < 0 ? TGW_EV_NO_TRAIN : TGW_EV_SYNC_MISPLACED, bs, vt < 0 ? 0 : offs, (uint32_t)gi))
						return;
					state = TGW_S_UNLOCKED;
				} else if (tgw_event(r, TGW_EV_NORM_MISPLACED, bs, offs, (uint32_t)gi))
					return;
			} else if (type == TG_BURST_SYNC) {
				if (tgw_event(r, TGW_EV_SYNC_MISPLACED, bs, offs, (uint32_t)gi))
					return;
				state = TGW_S_UNLOCKED;
			} else if (tgw_event(r, TGW_EV_NORM_MISPLACED, bs, offs, (uint32_t)gi))
				return;
		}
		bs += TG_SLOT_BITS;
		nfs += TG_SLOT_BITS;
	}
}

#endif
