/*
 * tg_k_walk.hip -- k_walk / k_walk_big / k_walk_nodes: the burst synchroniser's walk of every channel on the device
 * (one of the four HIP units of the library: tg_dev.h has the map)
 */
#include "tg_dev.h"

/* ------------------------------------------------------------------------- */
/* k_walk: the burst synchroniser's walk over a classified grid, on the device */
/* ------------------------------------------------------------------------- */
/*
 * tetra_burst_sync_in() (phy/tetra_burst_sync.c:54-154) per channel, from the classification words, the SYNC
 * summaries and k_cls_plain's bitmap, without the host: what tg_stream.c:sync_walk() computes in grid mode
 * (delivered bitmap, events, counts, final state), in the node form of tg_walk_core.h.
 *
 * One workgroup of 1024 threads per channel; bitmap, node list and arrival pointers live in LDS:
 *   A  the channel's plain bitmap -> LDS; nodes = its zero bits; per-word prefix counts (block scan)
 *   B  node list (slot of the i-th node)
 *   C  every node through tgw_run(), one lane each (a dozen dependent reads of cls / ysum per node: latency bound,
 *      hidden by the thousand lanes); the stream's head (the first lock, found on the host) likewise; arrival slot
 *      -> index of the first node at or after it
 *   D  which nodes does the walk visit?  reachability from the head's arrival along the arrival pointers: pointer
 *      doubling, marks {succ^n(head) : n < 2^r} after r rounds
 *   E  delivered bitmap = plain bitmap - spans of the visited nodes [slot, arrival) + their own deliveries
 *   F  bitmap -> global, number of delivered bursts, last delivered slot
 *   G  events of the visited nodes in slot order (block scan of the counts), bursts handled but not delivered,
 *      those after the last delivery (tail_tn_adds), final state
 * A channel with more than TGW_NCAP nodes or TGW_WCAP bitmap words, or whose walk meets something only the bytes can
 * settle (tg_walk_core.h), reports TGW_FALLBACK: the host walk takes over.
 */
#include "tg_walk_core.h"
static_assert(sizeof(tgw_rec) <= TGW_REC_BYTES, "TGW_REC_BYTES");

#define TGW_THREADS 1024
#define TGW_LDS_BYTES (TGW_WCAP * 4 + TGW_NCAP * 4 + TGW_WCAP * 2 + 2 * (TGW_NCAP + 8) * 2 + 2 * (TGW_NCAP + 8))	/* MODE 0, full caps */
#define TGW_LDS2_BYTES(wcap, ncap) ((wcap) * 4u + 2u * ((ncap) + 8u) * 2u + 2u * ((ncap) + 8u))	/* MODE 2 */
#ifndef TGW_THREADS_LIGHT
#define TGW_THREADS_LIGHT 256
#endif
#ifndef TGW_NODES_THREADS
#define TGW_NODES_THREADS 256
#endif
/* TGW_THREADS_LIGHT: the three-launch form: a workgroup that takes one wave slot per SIMD and 20-70 KB of LDS finds a
				 * place beside the heavy kernels of the other batches; 1024 threads and 128 KB wait for a nearly empty
				 * compute unit */

__device__ __forceinline__ uint32_t tgw_block_excl_scan(uint32_t v, uint32_t *sm /* 17 words */, uint32_t &total)
{
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(inc, d);
		if (lane >= (uint32_t)d)
			inc += o;
	}
	__syncthreads();	/* sm may still be read from the previous scan */
	if (lane == 63)
		sm[w] = inc;
	__syncthreads();
	uint32_t pre = 0, tot = 0;
	for (uint32_t q = 0; q < blockDim.x / 64; q++) {
		const uint32_t x = sm[q];
		if (q < w)
			pre += x;
		tot += x;
	}
	total = tot;
	return pre + inc - v;
}

#ifdef TGW_TIMING
__device__ unsigned long long g_tgw_stamp[64][12];
#define TGW_STAMP(i) do { if (threadIdx.x == 0) g_tgw_stamp[blockIdx.x][i] = __builtin_readcyclecounter(); } while (0)
extern "C" int tgk_walk_stamps(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tgw_stamp), sizeof(g_tgw_stamp));
}
#else
#define TGW_STAMP(i) do { } while (0)
#endif

/*
 * The body, for both homes of its working arrays: BIG = false, LDS (a channel of up to TGW_WCAP bitmap words and TGW_NCAP
 * nodes: 16-bit node indices, loop counts known at compile time); BIG = true, global memory (k_walk_big: a channel
 * beyond that -- a recording of more than 262 144 slots -- with the caps the plan's scratch area was made for; the same
 * steps at L2 latency, a millisecond or two for a million slots, on one compute unit beside the other batches' kernels).
 */
/* MODE 0: all of it in one launch.  MODE 1 + k_walk_nodes + MODE 2: the node pass (C) -- a dozen dependent reads per node, the
 * longest stretch of the walk -- as a launch of its own over the whole chip: MODE 1 runs A and B and leaves node list, word
 * prefixes and the node count in global memory (tg_walk_tmp), k_walk_nodes takes every node of every channel through tgw_run()
 * (256 nodes per workgroup), MODE 2 runs A and B again (the LDS form; the other keeps its arrays), picks the arrival pointers
 * up and goes on with D .. G. */
struct tg_walk_tmp {
	uint32_t *nslot, *wpre, *J;	/* this channel's node list, per-word prefix counts, arrival pointers */
	uint32_t *meta;			/* this channel's {node count or ~0: nothing to do, head, fallback flag, reason} */
};
template <bool BIG, typename idx_t, int MODE>
__device__ __forceinline__ void walk_body(uint32_t c, uint32_t *bm, uint32_t *nslot, idx_t *wpre, idx_t *Ja, idx_t *Jb, uint8_t *mark,
					  const tg_walk_tmp tmp, const bool skip,
					  tgw_rec *recs, tgpu_sync_event_rec_dev *ev_big, const uint32_t wcap, const uint32_t ncap,
					  const uint32_t rootidx /* the head's record: recs[rootidx] */, const uint32_t evcap, const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan,
					  const tg_walk_root *__restrict__ roots, uint32_t chunk, uint32_t cshift,
					  const uint32_t *__restrict__ g_cls, const uint16_t *__restrict__ g_ysum,
					  const uint32_t *__restrict__ g_plain, uint32_t *__restrict__ g_bits, uint32_t *__restrict__ g_bits2,
					  tg_walk_sum *__restrict__ sums, tgpu_sync_event_rec_dev *__restrict__ g_eager)
{
	__shared__ uint32_t sm[20];
	__shared__ uint32_t s_head, s_fb, s_why, s_nd, s_tail, s_last, s_lastdel, s_ns;

	const uint32_t tid = threadIdx.x, NT = blockDim.x;
	/* the light form (the LDS form split in three launches, MODE 1 / 2): MODE 1 keeps nothing in LDS -- node list and word
	 * prefixes go straight to the hand-over area --, MODE 2 keeps the bitmap, the arrival pointers and the marks */
	constexpr bool LIGHT1 = !BIG && MODE == 1, LIGHT2 = !BIG && MODE == 2;
	const tg_chan_ent ce = chan[c];
	const tg_walk_root rt = roots[c];
	tg_walk_sum *sum = sums + c;
	/* event e of the channel: the first TGW_EVEAGER in the block that is copied to the host with the batch, the rest behind */
	tgpu_sync_event_rec_dev *ev_eager = g_eager + (size_t)c * TGW_EVEAGER;
	const uint32_t ncls = ce.ncls, W = (ncls + 31) >> 5, w0 = ce.gbase >> 5;

	if (tid == 0) {
		s_head = 0xffffffffu;
		s_fb = 0;
		s_why = 0;
		s_nd = s_tail = 0;
		s_last = 0;		/* 1 + index of the last visited node (0: only the head run) */
		s_lastdel = 0;
		s_ns = 0;
	}
	__syncthreads();
	if (MODE == 1 && tid == 0)
		tmp.meta[0] = 0xffffffffu;	/* until A and B are through: nothing for k_walk_nodes / MODE 2 to do */
	if (!ncls || W > wcap || skip) {	/* nothing classified (the host settles such a channel), too long for this form's arrays, or left to the long form */
		if (tid == 0) {
			sum->nslots = sum->nevents = sum->tail_tn_adds = sum->burst_seq = 0;
			sum->final_state = TGW_S_UNLOCKED;
			sum->status = ncls ? TGW_FALLBACK : TGW_OK;
			sum->why = ncls ? TGW_WHY_SIZE : 0;	/* (k_walk_big overwrites this where it runs) */
			sum->nnodes = 0;
		}
		return;
	}
	tgw_chan wc;
	wc.cls = g_cls + ce.gbase;
	wc.ysum = g_ysum + ce.gbase;
	wc.packed = (ce.d_off & TG_CHAN_PACKED) != 0;
	wc.sbit = ce.d_off & ~TG_CHAN_PACKED;
	wc.s = wc.packed ? d_base : d_base + ce.d_off;
	wc.len = ce.len;
	wc.anchor = ce.anchor;
	wc.ncalls = (ce.len + chunk - 1) >> cshift;
	wc.ncls = ncls;
	wc.chunk = chunk;
	wc.cshift = cshift;

	TGW_STAMP(0);
	/* A: bitmap -> LDS (bits at and past ncls read "plain" so that they are no nodes; they are cleared again in F) */
	const uint32_t WPT = (W + NT - 1) / NT;	/* consecutive words per thread */
	uint32_t cnt = 0;
	uint32_t N, base = 0;
	auto plain_word = [&](uint32_t w) -> uint32_t {
		uint32_t v = g_plain[w0 + w];
		if (w == W - 1 && (ncls & 31))
			v |= ~0u << (ncls & 31);
		return v;
	};
	if (BIG && MODE == 2) {		/* (bitmap, node list and prefixes are where MODE 1 left them) */
		N = tmp.meta[0];
		if (N == 0xffffffffu)
			return;
	} else if (LIGHT2) {		/* (node count from MODE 1; ~0: that launch has settled the channel's summary already) */
		N = tmp.meta[0];
		if (N == 0xffffffffu)
			return;
		for (uint32_t w = tid; w < W; w += NT)
			bm[w] = plain_word(w);
	} else {
	for (uint32_t q = 0; q < WPT; q++) {
		const uint32_t w = WPT * tid + q;
		if (w < W) {
			const uint32_t v = plain_word(w);
			if (!LIGHT1)
				bm[w] = v;
			cnt += __popc(~v);
		}
	}
	base = tgw_block_excl_scan(cnt, sm, N);
	}
	if (N > ncap) {
		if (tid == 0) {
			sum->nslots = sum->nevents = sum->tail_tn_adds = sum->burst_seq = 0;
			sum->final_state = TGW_S_UNLOCKED;
			sum->status = TGW_FALLBACK;
			sum->why = TGW_WHY_NODES;
			sum->nnodes = N;
		}
		return;
	}
	/* B: prefix counts per word, node list */
	if (MODE != 2) {
	for (uint32_t q = 0; q < WPT; q++) {
		const uint32_t w = WPT * tid + q;
		if (w < W) {
			uint32_t z;
			if (LIGHT1) {
				tmp.wpre[w] = base;
				z = ~plain_word(w);
			} else {
				wpre[w] = (idx_t)base;
				z = ~bm[w];
			}
			while (z) {
				const uint32_t b = __builtin_ctz(z);
				z &= z - 1;
				if (LIGHT1)
					tmp.nslot[base++] = 32 * w + b;
				else
					nslot[base++] = 32 * w + b;
			}
		}
	}
	}
	__syncthreads();
	if (MODE == 1) {	/* the lists are k_walk_nodes' now (BIG: the body's arrays are the hand-over area) */
		if (tid == 0)
			tmp.meta[0] = N;
		return;
	}
	auto rank = [&](uint32_t t) -> uint32_t {	/* index of the first node at or after grid slot t */
		if (t >= ncls)
			return N;
		const uint32_t w = t >> 5;
		return (uint32_t)wpre[w] + __popc(~bm[w] & ((1u << (t & 31)) - 1u));
	};
	TGW_STAMP(1);
	/* C: every node, and the stream's head */
	if (MODE == 2) {	/* (done by k_walk_nodes) */
		if (!BIG)
			for (uint32_t i = tid; i < N; i += NT)
				Ja[i] = (idx_t)tmp.J[i];
		if (tid == 0) {
			s_head = tmp.meta[1];
			s_fb = tmp.meta[2];
			s_why = tmp.meta[3];
		}
	} else {
	for (uint32_t i = tid; i < N; i += NT) {
		const uint64_t bs = wc.anchor + (uint64_t)nslot[i] * TG_SLOT_BITS;
		const uint64_t kc = (bs + TG_SLOT_BITS + chunk - 1) >> cshift;
		tgw_rec r;
		tgw_run(&wc, TGW_S_LOCKED, bs, bs + TG_SLOT_BITS, kc - 1, &r);
		recs[i] = r;
		Ja[i] = (idx_t)(r.status == TGW_OK ? rank(r.next) : N);
	}
	if (tid == NT - 1) {
		tgw_rec r;
		tgw_run(&wc, TGW_S_KNOW_FSTART, rt.found_bs, wc.anchor, rt.found_k, &r);
		recs[rootidx] = r;
		if (r.status != TGW_OK) {
			s_fb = 1;
			s_why = r.why;
		} else
			s_head = rank(r.next);
	}
	}
	if (tid == 0)
		Ja[N] = Jb[N] = (idx_t)N;
	for (uint32_t i = tid; i <= N; i += NT)
		mark[i] = 0;
	__syncthreads();
	TGW_STAMP(2);
	/* D: reachability from the head along the arrival pointers */
	{
		const uint32_t head = s_head;
		if (tid == 0 && head < N)
			mark[head] = 1;
		__syncthreads();
		idx_t *J = Ja, *Jn = Jb;
		for (uint32_t span = 1; span <= N; span <<= 1) {
			for (uint32_t v = tid; v < N; v += NT)
				if (mark[v] && J[v] < N)
					mark[J[v]] = 1;
			for (uint32_t v = tid; v < N; v += NT) {
				const uint32_t j = J[v];
				Jn[v] = j < N ? J[j] : (idx_t)N;
			}
			__syncthreads();
			idx_t *t = J;
			J = Jn;
			Jn = t;
		}
	}
	TGW_STAMP(3);
	/* E: spans of the visited nodes (and of the head run) leave the bitmap, their own deliveries enter it */
	auto clear_span = [&](uint32_t from, uint32_t to) {	/* grid slots [from, to) */
		if (to > ncls)
			to = ncls;
		while (from < to) {
			const uint32_t w = from >> 5, b = from & 31;
			const uint32_t n = (32 - b < to - from) ? 32 - b : to - from;
			const uint32_t m = (n == 32 ? 0xffffffffu : ((1u << n) - 1u)) << b;
			atomicAnd(&bm[w], ~m);
			from += n;
		}
	};
	for (uint32_t i = tid; i < N; i += NT)
		if (mark[i]) {
			const tgw_rec *r = recs + i;
			if (r->status != TGW_OK) {
				s_fb = 1;
				s_why = r->why;
			}
			clear_span(MODE == 2 ? tmp.nslot[i] : nslot[i], r->next);
			atomicMax(&s_last, i + 1);
		}
	if (tid == NT - 1 && !s_fb)
		clear_span(0, recs[rootidx].next);
	__syncthreads();
	if (s_fb) {
		if (tid == 0) {
			sum->nslots = sum->nevents = sum->tail_tn_adds = sum->burst_seq = 0;
			sum->final_state = TGW_S_UNLOCKED;
			sum->status = TGW_FALLBACK;
			sum->why = s_why;
			sum->nnodes = N;
		}
		return;
	}
	for (uint32_t i = tid; i < N + 1; i += NT) {
		const bool root = (i == N);
		if (root || mark[i]) {
			const tgw_rec *r = recs + (root ? rootidx : i);
			for (uint32_t d = 0; d < r->ndel; d++)
				atomicOr(&bm[r->del[d] >> 5], 1u << (r->del[d] & 31));
		}
	}
	__syncthreads();
	TGW_STAMP(4);
	/* F: bitmap out, delivered bursts, last delivered slot */
	{
		uint32_t ns = 0, lastd = 0xffffffffu;
		for (uint32_t q = 0; q < WPT; q++) {
			const uint32_t w = WPT * tid + q;
			if (w < W) {
				uint32_t v = bm[w];
				if (w == W - 1 && (ncls & 31))
					v &= (1u << (ncls & 31)) - 1u;
				g_bits[w0 + w] = v;
				g_bits2[w0 + w] = v;
				ns += __popc(v);
				if (v)
					lastd = 32 * w + 31 - __builtin_clz(v);
			}
		}
		if (ns)
			atomicAdd(&s_ns, ns);
		if (lastd != 0xffffffffu)
			atomicMax(&s_lastdel, lastd + 1);	/* 1 + last delivered slot, 0 = none */
	}
	__syncthreads();
	TGW_STAMP(5);
	/* G: events in slot order */
	const uint32_t NPT = (N + NT - 1) / NT;
	const tgw_rec *root = recs + rootidx;
	uint32_t ecnt = 0;
	for (uint32_t q = 0; q < NPT; q++) {
		const uint32_t i = NPT * tid + q;
		if (i < N && mark[i])
			ecnt += recs[i].nev;
	}
	uint32_t etot;
	uint32_t eoff = tgw_block_excl_scan(ecnt, sm, etot) + root->nev;
	etot += root->nev;
	const uint32_t lastdel = s_lastdel;
	uint32_t nd = 0, tail = 0;
	auto emit = [&](const tgw_rec *r, uint32_t at) {
		for (uint32_t e = 0; e < r->nev; e++) {
			if (at + e < evcap) {
				tgpu_sync_event_rec_dev *o = at + e < TGW_EVEAGER ? ev_eager + at + e : ev_big + at + e;
				o->ev = (int32_t)r->ev[e][0];
				o->bitnum = r->ev[e][1];
				o->arg = r->ev[e][2];
			}
			if (r->evslot[e] != TGW_NOSLOT) {
				nd++;
				if (r->evslot[e] + 1 > lastdel)
					tail++;
			}
		}
	};
	if (tid == NT - 1)
		emit(root, 0);
	for (uint32_t q = 0; q < NPT; q++) {
		const uint32_t i = NPT * tid + q;
		if (i < N && mark[i]) {
			emit(recs + i, eoff);
			eoff += recs[i].nev;
		}
	}
	if (nd)
		atomicAdd(&s_nd, nd);
	if (tail)
		atomicAdd(&s_tail, tail);
	__syncthreads();
	if (tid == 0) {
		const tgw_rec *lastrec = s_last ? recs + (s_last - 1) : root;
		sum->nslots = s_ns;
		sum->nevents = etot;
		sum->final_state = lastrec->next == TGW_END ? lastrec->end_state : TGW_S_LOCKED;
		sum->tail_tn_adds = s_tail;
		sum->burst_seq = s_ns + s_nd;
		sum->status = etot > evcap ? TGW_FALLBACK : TGW_OK;
		sum->why = etot > evcap ? TGW_WHY_EVENTS : 0;
		sum->nnodes = N;
	}
	TGW_STAMP(6);
}

/* layout of the split form's hand-over area (the LDS form): per channel four words of meta data, then node list, word prefixes
 * and arrival pointers as 32-bit words */
#define TGW_TMP_META_BYTES 1024u	/* 64 channels x {N, head, fallback, reason} */
#define TGW_TMP_CHAN_WORDS (TGW_NCAP + TGW_WCAP + TGW_NCAP + 8u)
__device__ __forceinline__ tg_walk_tmp walk_tmp_small(uint8_t *d_tmp, uint32_t c)
{
	tg_walk_tmp t;
	uint32_t *w = (uint32_t *)(d_tmp + TGW_TMP_META_BYTES) + (size_t)c * TGW_TMP_CHAN_WORDS;
	t.nslot = w;
	t.wpre = w + TGW_NCAP;
	t.J = w + TGW_NCAP + TGW_WCAP;
	t.meta = (uint32_t *)d_tmp + 4 * c;
	return t;
}
__device__ __forceinline__ tg_walk_tmp walk_tmp_big(uint8_t *slot, const tg_walk_big_layout &L, uint8_t *d_tmp, uint32_t c)
{
	tg_walk_tmp t;
	t.nslot = (uint32_t *)(slot + L.o_nslot);
	t.wpre = (uint32_t *)(slot + L.o_wpre);
	t.J = (uint32_t *)(slot + L.o_ja);
	t.meta = (uint32_t *)d_tmp + 4 * c;
	return t;
}

template <int MODE>
__global__ __launch_bounds__(TGW_THREADS)
void k_walk(const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan, const tg_walk_root *__restrict__ roots,
	    uint32_t chunk, uint32_t cshift, const uint32_t *__restrict__ g_cls, const uint16_t *__restrict__ g_ysum,
	    const uint32_t *__restrict__ g_plain, uint32_t *__restrict__ g_bits, uint32_t *__restrict__ g_bits2,
	    tg_walk_sum *__restrict__ sums, tgpu_sync_event_rec_dev *__restrict__ g_eager,
	    tgpu_sync_event_rec_dev *__restrict__ g_evbig, tgw_rec *__restrict__ g_recs, uint8_t *__restrict__ d_tmp,
	    unsigned long long skip_mask, uint32_t wcap, uint32_t ncap, uint32_t rec_stride)
{
	/* working arrays in LDS, laid out for the caps of this launch (tgk_walk): MODE 0 all of them, MODE 2 the bitmap, the two
	 * arrival-pointer arrays and the marks, MODE 1 none */
	extern __shared__ uint32_t s_dyn[];
	uint32_t *bm = s_dyn;
	uint32_t *nslot = bm + wcap;
	uint16_t *wpre = (uint16_t *)(nslot + ncap);
	uint16_t *Ja = (MODE == 2) ? (uint16_t *)(bm + wcap) : wpre + wcap, *Jb = Ja + ncap + 8;
	uint8_t *mark = (uint8_t *)(Jb + ncap + 8);
	const uint32_t c = blockIdx.x;
	tg_walk_tmp tmp = { nullptr, nullptr, nullptr, nullptr };
	if (MODE)
		tmp = walk_tmp_small(d_tmp, c);
	walk_body<false, uint16_t, MODE>(c, bm, nslot, wpre, Ja, Jb, mark, tmp, ((skip_mask >> c) & 1) != 0, g_recs + (size_t)c * rec_stride,
					 g_evbig + (size_t)c * TGW_EVCAP, wcap, ncap < rec_stride - 1 ? ncap : rec_stride - 1, rec_stride - 1, TGW_EVCAP, d_base, chan, roots, chunk, cshift, g_cls,
					 g_ysum, g_plain, g_bits, g_bits2, sums, g_eager);
}

/* the channels of the batch that are too long for the form above (their indices in `big`), one workgroup each, working
 * arrays in the plan's scratch area (tg_walk_big_layout): runs behind k_walk, which has reported them as TGW_WHY_SIZE */
template <int MODE>
__global__ __launch_bounds__(TGW_THREADS)
void k_walk_big(tg_walk_big big, uint8_t *__restrict__ scratch, const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan,
		const tg_walk_root *__restrict__ roots, uint32_t chunk, uint32_t cshift, const uint32_t *__restrict__ g_cls,
		const uint16_t *__restrict__ g_ysum, const uint32_t *__restrict__ g_plain, uint32_t *__restrict__ g_bits,
		uint32_t *__restrict__ g_bits2, tg_walk_sum *__restrict__ sums, tgpu_sync_event_rec_dev *__restrict__ g_eager,
		uint8_t *__restrict__ d_tmp)
{
	tg_walk_big_layout L;
	tg_walk_big_offsets(big.wcap, big.ncap, big.evcap, &L);
	uint8_t *base = scratch + (size_t)blockIdx.x * L.slot_bytes;
	const uint32_t c = big.chan[blockIdx.x];
	tg_walk_tmp tmp = { nullptr, nullptr, nullptr, nullptr };
	if (MODE)
		tmp = walk_tmp_big(base, L, d_tmp, c);
	walk_body<true, uint32_t, MODE>(c, (uint32_t *)(base + L.o_bm), (uint32_t *)(base + L.o_nslot), (uint32_t *)(base + L.o_wpre),
					(uint32_t *)(base + L.o_ja), (uint32_t *)(base + L.o_jb), base + L.o_mark, tmp, false, (tgw_rec *)(base + L.o_recs),
					(tgpu_sync_event_rec_dev *)(base + L.o_ev), big.wcap, big.ncap, big.ncap, big.evcap, d_base, chan, roots, chunk,
					cshift, g_cls, g_ysum, g_plain, g_bits, g_bits2, sums, g_eager);
}

/* phase C of the walk over the whole chip: workgroup (x, y) takes nodes 256 x .. of channel y (BIG: of the y-th long channel)
 * through tgw_run(); its first thread also runs the stream's head */
template <bool BIG>
__global__ __launch_bounds__(256)
void k_walk_nodes(tg_walk_big big, uint8_t *__restrict__ scratch, uint8_t *__restrict__ d_tmp, tgw_rec *__restrict__ g_recs,
		  const uint8_t *__restrict__ d_base, const tg_chan_ent *__restrict__ chan, const tg_walk_root *__restrict__ roots,
		  uint32_t chunk, uint32_t cshift, const uint32_t *__restrict__ g_cls, const uint16_t *__restrict__ g_ysum,
		  const uint32_t *__restrict__ g_plain, uint32_t rec_stride)
{
	const uint32_t c = BIG ? big.chan[blockIdx.y] : blockIdx.y;
	tg_walk_tmp tmp;
	tgw_rec *recs;
	uint32_t ncap;
	if (BIG) {
		tg_walk_big_layout L;
		tg_walk_big_offsets(big.wcap, big.ncap, big.evcap, &L);
		uint8_t *slot = scratch + (size_t)blockIdx.y * L.slot_bytes;
		tmp = walk_tmp_big(slot, L, d_tmp, c);
		recs = (tgw_rec *)(slot + L.o_recs);
		ncap = big.ncap;
	} else {
		tmp = walk_tmp_small(d_tmp, c);
		recs = g_recs + (size_t)c * rec_stride;
		ncap = rec_stride - 1;
	}
	const uint32_t N = tmp.meta[0];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (N == 0xffffffffu || (i >= N && i != 0))
		return;
	const tg_chan_ent ce = chan[c];
	const uint32_t ncls = ce.ncls, W = (ncls + 31) >> 5, w0 = ce.gbase >> 5;
	tgw_chan wc;
	wc.cls = g_cls + ce.gbase;
	wc.ysum = g_ysum + ce.gbase;
	wc.packed = (ce.d_off & TG_CHAN_PACKED) != 0;
	wc.sbit = ce.d_off & ~TG_CHAN_PACKED;
	wc.s = wc.packed ? d_base : d_base + ce.d_off;
	wc.len = ce.len;
	wc.anchor = ce.anchor;
	wc.ncalls = (ce.len + chunk - 1) >> cshift;
	wc.ncls = ncls;
	wc.chunk = chunk;
	wc.cshift = cshift;
	auto rank = [&](uint32_t t) -> uint32_t {	/* index of the first node at or after grid slot t */
		if (t >= ncls)
			return N;
		const uint32_t w = t >> 5;
		uint32_t v = g_plain[w0 + w];
		if (w == W - 1 && (ncls & 31))
			v |= ~0u << (ncls & 31);
		return tmp.wpre[w] + __popc(~v & ((1u << (t & 31)) - 1u));
	};
	if (i < N) {
		const uint64_t bs = wc.anchor + (uint64_t)tmp.nslot[i] * TG_SLOT_BITS;
		const uint64_t kc = (bs + TG_SLOT_BITS + chunk - 1) >> cshift;
		tgw_rec r;
		tgw_run(&wc, TGW_S_LOCKED, bs, bs + TG_SLOT_BITS, kc - 1, &r);
		recs[i] = r;
		tmp.J[i] = r.status == TGW_OK ? rank(r.next) : N;
	}
	if (i == 0) {
		const tg_walk_root rt = roots[c];
		tgw_rec r;
		tgw_run(&wc, TGW_S_KNOW_FSTART, rt.found_bs, wc.anchor, rt.found_k, &r);
		recs[ncap] = r;
		tmp.meta[1] = r.status == TGW_OK ? rank(r.next) : 0xffffffffu;
		tmp.meta[2] = r.status != TGW_OK;
		tmp.meta[3] = r.status != TGW_OK ? r.why : 0u;
	}
}

extern "C" int tgk_walk(const uint8_t *d_base, const struct tg_chan_ent *d_chan, const struct tg_walk_root *d_roots, uint32_t nchan,
			uint32_t chunk, const uint32_t *d_cls, const uint16_t *d_ysum, const uint32_t *d_plain, uint32_t *d_bits,
			uint32_t *d_bits2, struct tg_walk_sum *d_sums, void *d_eager, void *d_evbig, void *d_recs, void *d_tmp,
			unsigned long long skip_mask, uint32_t wcap, uint32_t ncap, uint32_t rec_stride, int wide, void *stream)
{
	if (!nchan)
		return 0;
	if (!chunk || (chunk & (chunk - 1)) || nchan > 64 || rec_stride < 2 || rec_stride > TGW_NCAP + 1)
		return -1;
	if (!d_tmp || wide || !wcap || !ncap || wcap > TGW_WCAP || ncap > TGW_NCAP) {
		wcap = TGW_WCAP;
		ncap = TGW_NCAP;
	}
	wcap = (wcap + 1u) & ~1u;
	ncap = (ncap + 255u) & ~255u;
	const uint32_t cshift = (uint32_t)__builtin_ctz(chunk);
	hipStream_t s = (hipStream_t)stream;
#define WALK_ARGS d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, d_bits, d_bits2, d_sums, \
		  (tgpu_sync_event_rec_dev *)d_eager, (tgpu_sync_event_rec_dev *)d_evbig, (tgw_rec *)d_recs, (uint8_t *)d_tmp, skip_mask, \
		  wcap, ncap, rec_stride
	if (!d_tmp) {
		HIPCHK(hipFuncSetAttribute((const void *)k_walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TGW_LDS_BYTES));
		hipLaunchKernelGGL(k_walk<0>, dim3(nchan), dim3(TGW_THREADS), TGW_LDS_BYTES, s, WALK_ARGS);
		return (int)hipGetLastError();
	}
	static __thread int attr_set_dev = -1;	/* (the attribute is per device and process: once per thread and device is enough) */
	int dev = 0;
	HIPCHK(hipGetDevice(&dev));
	if (attr_set_dev != dev) {
		HIPCHK(hipFuncSetAttribute((const void *)k_walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TGW_LDS_BYTES));
		HIPCHK(hipFuncSetAttribute((const void *)k_walk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TGW_LDS_BYTES));
		attr_set_dev = dev;
	}
	const uint32_t nt = wide ? TGW_THREADS : TGW_THREADS_LIGHT;
	hipLaunchKernelGGL(k_walk<1>, dim3(nchan), dim3(nt), wide ? TGW_LDS_BYTES : 0, s, WALK_ARGS);
	tg_walk_big none = {};
	hipLaunchKernelGGL(k_walk_nodes<false>, dim3(ncap / TGW_NODES_THREADS, nchan), dim3(TGW_NODES_THREADS), 0, s, none, (uint8_t *)nullptr, (uint8_t *)d_tmp,
			   (tgw_rec *)d_recs, d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, rec_stride);
	hipLaunchKernelGGL(k_walk<2>, dim3(nchan), dim3(nt), wide ? TGW_LDS_BYTES : TGW_LDS2_BYTES(wcap, ncap), s, WALK_ARGS);
#undef WALK_ARGS
	return (int)hipGetLastError();
}

extern "C" int tgk_walk_big(const struct tg_walk_big *big, void *d_scratch, const uint8_t *d_base, const struct tg_chan_ent *d_chan,
			    const struct tg_walk_root *d_roots, uint32_t chunk, const uint32_t *d_cls, const uint16_t *d_ysum,
			    const uint32_t *d_plain, uint32_t *d_bits, uint32_t *d_bits2, struct tg_walk_sum *d_sums, void *d_eager, void *d_tmp,
			    void *stream)
{
	if (!big || !big->n)
		return 0;
	if (big->n > TGW_BIG_MAX || !d_scratch || !chunk || (chunk & (chunk - 1)))
		return -1;
	const uint32_t cshift = (uint32_t)__builtin_ctz(chunk);
	hipStream_t s = (hipStream_t)stream;
#define WALK_ARGS *big, (uint8_t *)d_scratch, d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, d_bits, d_bits2, d_sums, \
		  (tgpu_sync_event_rec_dev *)d_eager, (uint8_t *)d_tmp
	if (!d_tmp) {
		hipLaunchKernelGGL(k_walk_big<0>, dim3(big->n), dim3(TGW_THREADS), 0, s, WALK_ARGS);
		return (int)hipGetLastError();
	}
	hipLaunchKernelGGL(k_walk_big<1>, dim3(big->n), dim3(TGW_THREADS), 0, s, WALK_ARGS);
	hipLaunchKernelGGL(k_walk_nodes<true>, dim3((big->ncap + 255) / 256, big->n), dim3(256), 0, s, *big, (uint8_t *)d_scratch,
			   (uint8_t *)d_tmp, (tgw_rec *)nullptr, d_base, d_chan, d_roots, chunk, cshift, d_cls, d_ysum, d_plain, 0u);
	hipLaunchKernelGGL(k_walk_big<2>, dim3(big->n), dim3(TGW_THREADS), 0, s, WALK_ARGS);
#undef WALK_ARGS
	return (int)hipGetLastError();
}

