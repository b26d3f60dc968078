/* tg_internal.h -- launch layer between the C host code and the HIP units (tg_dev.h has the map: tg_k_front / _trellis / _walk / _aux .hip) */
#ifndef TG_INTERNAL_H
#define TG_INTERNAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int tgk_init(void);
long tgi_option(int opt);	/* enum tgpu_option (tgpu_engine_set_option): process-wide switches, never the environment */
/* d_slot_desc[i] = byte offset | (uint64_t)burst type << 56 */
int tgk_front(const uint8_t *d_stream, const uint64_t *d_slot_desc,
	      uint32_t nslots, uint32_t *d_packed, uint8_t *d_rec, void *stream);
/* stream mode: grid slot n at anchor + 510 n; writes packed slots and one classification word per slot.
 * d_defer: scratch of TG_DEFER_WORDS(nslots) dwords (the slots the packed-bit pass hands to the exact pass);
 * ev_mid: hipEvent_t recorded between the two passes, or NULL */
/* (a count per wave of the first pass + 16 + a list per wave of 4 x the groups it can meet: <= nslots + 5 waves + 16 with at most a
 * wave per group) */
#define TG_DEFER_WORDS(nslots) ((size_t)(nslots) + 5 * (((size_t)(nslots) + 3) / 4 + 4) + 80)
int tgk_front_stream(const uint8_t *d_stream, uint64_t anchor, uint64_t len, uint32_t nslots,
		     uint32_t chunk, uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum, uint32_t *d_defer,
		     void *stream, void *ev_mid);
/* several channels in one grid (BASELINE config 4): channel c owns grid slots gbase .. gbase + ncls - 1, gbase a
 * multiple of 32 (padding slots behind ncls are classified "nothing"); its stream lies at byte d_off of d_base */
struct tg_chan_ent {
	uint64_t d_off, anchor, len;	/* d_off | TG_CHAN_PACKED: the device buffer holds the stream one BIT per position (LSB first), d_off counts bits */
	uint32_t gbase, ncls;
};
#define TG_CHAN_PACKED (1ull << 63)
void tgk_front_stream_ev_start(void *ev);	/* timing: record 'ev' right in front of this thread's next k_front_stream launch */
int tgk_front_stream_multi(const uint8_t *d_base, const struct tg_chan_ent *d_chan, uint32_t nchan, uint32_t nslots,
			   uint32_t chunk, uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum, uint32_t *d_defer,
			   void *stream, void *ev_mid, int packed_input /* every channel's d_off carries TG_CHAN_PACKED */);
int tgk_vit(int kind, const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed,
	    const uint32_t *d_masks, const uint32_t *d_maskidx, uint8_t *d_rec,
	    uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_wire /* or NULL */,
	    const uint32_t *d_softarea /* NULL: hard input */, int flags /* TGK_F_* */,
	    const uint32_t *d_nitems /* NULL, or device-side item count with nitems as its upper bound */, void *stream);
/* one lane per SLOT (tg_k_slot.hip, slot_core.h): d_items = the batch's slots of type NORM_1 / NORM_2 / SYNC, any order; replaces the
 * 216 and 432 launches of tgk_vit for hard input (a SYNC burst's SB1 is decoded again on the way: its lane would idle otherwise) */
int tgk_slot_t(const uint32_t *d_items, uint32_t nitems, const uint32_t *d_nitems, const uint32_t *d_items2 /* the SYNC slots */,
	       uint32_t nitems2, const uint32_t *d_nitems2, uint32_t ntotal /* 0, or an upper bound of both lists together */,
	       const uint32_t *d_packed, const uint32_t *d_masks, const uint32_t *d_maskidx,
	       uint8_t *d_rec, uint8_t *d_wire /* or NULL */, int flags, void *stream);
/* clean-block pre-pass (kinds 216 / 432): finishes the blocks that are code words, lists the others for tgk_vit */
int tgk_clean(int kind, const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed, const uint32_t *d_masks,
	      const uint32_t *d_maskidx, uint8_t *d_rec, uint32_t *d_sb_ok, uint32_t *d_sb_code, uint8_t *d_wire,
	      uint32_t *d_dirty_items, uint32_t *d_dirty_count, int flags, void *stream);
/* small batches: one workgroup per burst, the trellis states across lanes; two launches per batch (SB1 pass first
 * when the batch holds a SYNC slot).  d_sb_ok / d_sb_code are indexed by SLOT here. */
int tgk_burst(const uint8_t *d_stream, const uint64_t *d_slot_desc, const uint32_t *d_slot_chan,
	      const uint32_t *d_chan_code, uint32_t nslots, uint32_t nchan, int have_sync, uint32_t *d_sb_ok,
	      uint32_t *d_sb_code, uint8_t *d_rec, uint32_t *d_maskidx, uint32_t *d_masks,
	      int marks /* records in mapped host memory: completion marks behind a system-wide fence */, void *stream);
/* block mode: descriptor = byte offset | table index (TG_KIND_* or 4 = BBK) << 56 | tp_sap type << 48 */
int tgk_front_blocks(const uint8_t *d_bits, const uint64_t *d_desc, uint32_t nblocks, uint32_t *d_packed, void *stream);
int tgk_bbk_blocks(const uint32_t *d_items, uint32_t nitems, const uint32_t *d_packed, const uint32_t *d_masks,
		   const uint32_t *d_maskidx, uint8_t *d_rec, int flags, void *stream);
int tgk_front_soft(const int8_t *d_soft, const uint64_t *d_slot_desc, uint32_t nslots,
		   uint32_t *d_area, uint32_t *d_packed, uint8_t *d_rec, void *stream);
int tgk_front_soft_f32(const float *d_phi, unsigned long long nfloats, const uint64_t *d_slot_desc, uint32_t nslots,
		       uint32_t *d_area, uint32_t *d_packed, uint8_t *d_rec, void *stream);
int tgk_float_to_bits(const float *d_in, unsigned long long n, uint8_t *d_bits, int8_t *d_soft, void *stream);
int tgk_float_to_bits_afc(const float *d_in, unsigned long long n, uint8_t *d_bits, float filter_val,
			  float filter_goal, float *d_state, void *stream);
/* generic trellis (tg_conv.h step program in d_steps); code 0: rate-1/4 CCH code, 1: rate-1/3 speech code;
 * g3: the program receives g3 somewhere (tg_conv_uses_g3) */
int tgk_conv(int code, int g3, const uint8_t *d_type3, unsigned long long nblocks, uint32_t t3len, uint32_t L,
	     const uint32_t *d_steps, uint8_t *d_type2, void *stream);
/* d_list_sb: slot of the k-th SYNC slot; a SYNC slot whose code repeats its predecessor's (same channel) shares that entry */
int tgk_fill(const uint32_t *d_slot_chan, const int32_t *d_slot_sbord, const uint32_t *d_sb_ok, const uint32_t *d_sb_code,
	     const uint32_t *d_list_sb, uint32_t nchan, uint32_t nslots, unsigned long long *d_block_tmp, uint32_t *d_maskidx,
	     void *stream);
int tgk_masks(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_sb_ok,
	      const uint32_t *d_sb_code, uint32_t nsb, const uint32_t *d_list_sb, const uint32_t *d_slot_chan,
	      uint32_t *d_masks, void *stream);
/* d_nsb != NULL: the number of SYNC slots is read from the device, nsb is its upper bound */
int tgk_masks_dev(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_sb_ok,
		  const uint32_t *d_sb_code, uint32_t nsb, const uint32_t *d_nsb, const uint32_t *d_list_sb,
		  const uint32_t *d_slot_chan, uint32_t *d_masks, void *stream);

/* out[b][j] = in[b][src[j]] for src[j] >= 0 (tg_reorder.c) */
int tgk_reorder(const uint8_t *d_in, unsigned long long nblocks, uint32_t nbits, const int32_t *d_src, uint8_t *d_out,
		void *stream);

/* stream mode: per-slot arrays and item lists from classification words + "delivered" bitmap;
 * d_blk: 3 * (ceil(n / 1024) + 1) words, the totals (sb, 216 items, 432 items) end up at d_blk[3 * nblocks] */
int tgk_grid_lists(const uint32_t *d_cls, const uint32_t *d_bits, uint32_t n, uint32_t *d_blk,
		   uint32_t *d_slot_chan, int32_t *d_slot_sbord, uint32_t *d_list_sb, uint32_t *d_list_216,
		   uint32_t *d_list_432, const struct tg_chan_ent *d_chan /* NULL: one channel */, uint32_t nchan, void *stream);

/* the synchroniser's walk on the device (k_walk, tg_walk_core.h): one workgroup per channel of the table */
#ifndef TGW_NCAP
#define TGW_NCAP 8192u	/* nodes (grid slots that are no plain delivery) per channel */
#endif
#ifndef TGW_WCAP
#define TGW_WCAP 8192u	/* bitmap words per channel: 262 144 grid slots */
#endif
struct tg_walk_root {	/* the stream's first lock (tg_stream.c:find_anchor): buffer start and index of the call that found it */
	uint64_t found_bs, found_k;
};
struct tg_walk_sum {
	uint32_t nslots, nevents, final_state, tail_tn_adds, burst_seq, status /* TGW_OK / TGW_FALLBACK */, why, nnodes;
};
typedef struct { int32_t ev; uint32_t bitnum, arg; } tgpu_sync_event_rec_dev;	/* = struct tgpu_sync_event_rec */
#define TGW_EVCAP 16384u	/* events per channel the device walk can report */
#define TGW_EVEAGER 4096u	/* of those, copied to the host with the batch; the rest on demand */
#define TGW_REC_BYTES 152u	/* sizeof(struct tgw_rec), checked where it is allocated */
/* d_bits: the delivered bitmap for the list builder; d_bits2: a second copy in the block that goes to the host;
 * events 0 .. TGW_EVEAGER - 1 of channel c at d_eager + c * TGW_EVEAGER (same block), the others at d_evbig + c * TGW_EVCAP */
int tgk_walk(const uint8_t *d_base, const struct tg_chan_ent *d_chan, const struct tg_walk_root *d_roots, uint32_t nchan,
	     uint32_t chunk, const uint32_t *d_cls, const uint16_t *d_ysum, const uint32_t *d_plain, uint32_t *d_bits,
	     uint32_t *d_bits2, struct tg_walk_sum *d_sums, void *d_eager, void *d_evbig, void *d_recs, void *d_tmp,
	     unsigned long long skip_mask /* channels left to tgk_walk_big */,
	     uint32_t wcap, uint32_t ncap /* the split form's LDS arrays hold channels of this many bitmap words / nodes (0: the full caps) */,
	     uint32_t rec_stride /* node records per channel in d_recs, the head's record last: min(TGW_NCAP, the plan's slots) + 1 */,
	     int wide /* 1: the split form's per-channel launches as 1024 threads with 128 KB of LDS (rounds 3 and 4) */, void *stream);
/* k_burst_ring (TGPU_OPT_RING): the request line in mapped host memory and the workgroups' box in device memory.  The two
 * begin alike on purpose: dwords 1 .. 11 of the line are copied to the box as they are. */
#define TG_RING_MAX  4u			/* bursts per request = workgroups that stay */
#define TG_RING_STOP 0xffffffffu
struct tg_ring_msg {
	uint32_t req;			/* request number (written last); TG_RING_STOP: leave */
	uint32_t n, have_sync, code;	/* slots, "a SYNC slot among them", the channel's carry-in scrambling code */
	uint64_t desc[TG_RING_MAX];	/* slot descriptors (offset | type << 56) */
	uint32_t pad[3];
	uint32_t req2;			/* = req, in the line's other half (written before that half's fields ... see k_burst_ring) */
	/* device -> host */
	uint32_t served;		/* last request the workgroups have finished */
	uint32_t alive;			/* the host sets it in front of a launch, workgroup 0 clears it when it leaves */
	uint32_t pad2[14];
};
struct tg_ring_box {
	uint32_t seq;			/* the request the workgroups are to work on (workgroup 0 writes it last) */
	uint32_t n, have_sync, code;
	uint64_t desc[TG_RING_MAX];
	uint32_t chan[TG_RING_MAX];	/* zeros: one channel */
	uint32_t bar;			/* barrier counter, only grows */
	uint32_t stop;			/* = the launch's number when workgroup 0 has decided to leave */
	uint32_t pad[2];
};
int tgk_burst_ring(struct tg_ring_msg *d_ring, struct tg_ring_box *d_box, const uint8_t *d_stream, uint32_t nwg, uint32_t *d_sb_ok,
		   uint32_t *d_sb_code, uint8_t *d_rec, uint32_t *d_maskidx, uint32_t *d_masks, uint32_t start_seq,
		   uint32_t launch_no, unsigned long long idle_ticks, void *stream);
/* the node cap a plan asks for (tg_host.c: twice what its batches have shown so far) */
uint32_t tgpi_plan_walk_ncap(const struct tgpu_plan *p);
void tgpi_plan_walk_seen(struct tgpu_plan *p, uint32_t nnodes);
/* d_tmp != NULL: the split form (node pass as a grid-wide launch between two per-channel ones); TGW_TMP_BYTES of device memory */
#define TGW_TMP_BYTES (1024u + 64u * (TGW_NCAP + TGW_WCAP + TGW_NCAP + 8u) * 4u)
/* channels beyond TGW_WCAP words (recordings of more than 262 144 slots): k_walk_big, one workgroup each, with its working
 * arrays in global memory -- a scratch slot per such channel, laid out by tg_walk_big_offsets() for the caps the plan was
 * made with (wcap: bitmap words, ncap: nodes, evcap: events of the channel behind the TGW_EVEAGER first ones) */
#define TGW_BIG_MAX 8u
struct tg_walk_big {
	uint32_t n, wcap, ncap, evcap;
	uint32_t chan[TGW_BIG_MAX];
};
struct tg_walk_big_layout {
	size_t o_bm, o_nslot, o_wpre, o_ja, o_jb, o_mark, o_recs, o_ev, slot_bytes;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline void tg_walk_big_offsets(uint32_t wcap, uint32_t ncap, uint32_t evcap, struct tg_walk_big_layout *L)
{
	size_t o = 0;
#define TGW_AT(field, bytes) do { L->field = o; o = (o + (size_t)(bytes) + 255) & ~(size_t)255; } while (0)
	TGW_AT(o_bm, (size_t)wcap * 4);
	TGW_AT(o_nslot, (size_t)ncap * 4);
	TGW_AT(o_wpre, (size_t)wcap * 4);
	TGW_AT(o_ja, ((size_t)ncap + 8) * 4);
	TGW_AT(o_jb, ((size_t)ncap + 8) * 4);
	TGW_AT(o_mark, (size_t)ncap + 8);
	TGW_AT(o_recs, ((size_t)ncap + 1) * TGW_REC_BYTES);
	TGW_AT(o_ev, (size_t)evcap * 12);
#undef TGW_AT
	L->slot_bytes = o;
}
static inline uint32_t tg_walk_big_ncap(uint32_t slots) { return slots / 8 < 16384u ? 16384u : slots / 8; }
int tgk_walk_big(const struct tg_walk_big *big, void *d_scratch, const uint8_t *d_base, const struct tg_chan_ent *d_chan,
		 const struct tg_walk_root *d_roots, uint32_t chunk, const uint32_t *d_cls, const uint16_t *d_ysum,
		 const uint32_t *d_plain, uint32_t *d_bits, uint32_t *d_bits2, struct tg_walk_sum *d_sums, void *d_eager, void *d_tmp,
		 void *stream);
/* buffers of a device-walk batch: ONE block up (channel table, roots, carry-in codes), ONE block down (summaries, the
 * first TGW_EVEAGER events of every channel, the delivered bitmap), device-only scratch (further events, node records) */
struct tg_walk_io {
	uint8_t *d_up0, *h_up0;
	size_t up_bytes;
	struct tg_chan_ent *d_tab, *h_tab;
	struct tg_walk_root *d_roots, *h_roots;
	uint32_t *d_codes, *h_codes;
	uint8_t *d_down0, *h_down0;
	uint8_t *hd_down0;	/* h_down0 as the device sees it (mapped, coherent host memory), or NULL */
	uint8_t *hd_up0;	/* ... and h_up0 */
	size_t down_bytes;
	struct tg_walk_sum *d_sums, *h_sums;
	tgpu_sync_event_rec_dev *d_eager, *h_eager;
	uint32_t *d_bits2, *h_bits2;
	uint32_t *d_final, *h_final;	/* 64 codes after the batch + the code table's overflow flag */
	tgpu_sync_event_rec_dev *d_evbig;
	void *d_recs;
	uint32_t rec_stride;	/* node records per channel in d_recs */
	void *d_tmp;			/* hand-over area of the split walk (TGW_TMP_BYTES) */
	struct tg_walk_big big;		/* the batch's channels beyond TGW_WCAP words (n = 0: none) and the scratch caps */
	uint8_t *d_big;			/* big.n scratch slots (tg_walk_big_offsets) */
};
struct tgpu_plan;
int tgpi_plan_walk_io(struct tgpu_plan *p, uint32_t nchan, uint32_t ngrid, struct tg_walk_io *io);
/* scratch for k_walk_big: nbig slots at the plan's caps (allocated on the first batch that needs it); fills io->big's caps and io->d_big */
int tgpi_plan_walk_big(struct tgpu_plan *p, uint32_t nbig, struct tg_walk_io *io);
void tgpi_plan_walk_overflow(struct tgpu_plan *p, uint32_t nslots, uint32_t nnodes);
uint32_t tgpi_plan_walk_threshold(const struct tgpu_plan *p);

/* GSMTAP messages of a decoded batch (k_gsmtap); tg_tdma_time_dev = struct tetra_tdma_time */
typedef struct { uint16_t hn; uint32_t sn, tn, fn, mn; } tg_tdma_time_dev;
int tgk_gsmtap(const uint8_t *d_rec, const void *d_times, const uint8_t *d_traffic, uint32_t nslots, uint8_t *d_msgs,
	       uint8_t *d_lens, void *stream);
int tgk_stages(const uint8_t *d_type5, const uint32_t *d_codes, uint32_t fixed_code, unsigned long long nblocks, uint32_t K,
	       uint32_t a, uint32_t mother_len, uint8_t *d_type4, uint8_t *d_type3, uint8_t *d_type3dp, void *stream);
int tgk_stages_crc(const uint8_t *d_type2, unsigned long long nblocks, uint32_t type2_len, uint32_t n, uint16_t *d_crc,
		   void *stream);

/* traffic blocks of a decoded batch (tg_traffic.hip): driven by the batch's item lists (d_counts: their device-side lengths or
 * NULL), d_rec / d_wire: the records to mark (either may be NULL), d_type4 / d_blocks may be NULL, d_lens is cleared first */
int tgk_traffic(const uint32_t *d_items432, uint32_t n432, const uint32_t *d_items216, uint32_t n216, const uint32_t *d_counts,
		const uint8_t *d_traffic, const uint32_t *d_packed, const uint32_t *d_masks, const uint32_t *d_maskidx,
		uint8_t *d_rec, uint8_t *d_wire, uint32_t nslots, uint8_t *d_type4, int16_t *d_blocks, uint16_t *d_lens,
		void *stream);

/* nbytes (rounded up to 16; both ends 16-byte aligned and that long) from one device-visible address to another, by a kernel */
int tgk_copy16(const void *d_src, void *d_dst, size_t nbytes, void *stream);

/* optional RM(30,14) decoder (tg_rm.c): coset-leader table (65536 words, built on first use) and the generator's
 * parity rows; tgk_rm_enable() uploads both for the kernels (flag TGK_F_RM of tgk_vit / tgk_bbk_blocks) */
const uint32_t *tgi_rm_leader_table(void);
const uint16_t *tgi_rm_parity(void);
int tgk_rm_enable(const uint32_t *h_leader, const uint16_t *h_parity);
#define TGK_F_BLOCK 1	/* tgk_vit flags: items are blocks on their own */
#define TGK_F_RM    2	/* correct the BBK with the RM(30,14) decoder before its first 14 bits are kept */
#define TGK_F_WIREONLY 8	/* only the 40-byte wire records are written (tgpu_plan_set_wire_only): no 320-byte records */
#define TGK_F_LOOKBACK 16	/* SB1 launch of a device-walk batch: d_sb_ok = okbits (bit per grid slot), d_sb_code = mask entry per slot, d_masks =
				 * the code table (TGK_LB_TBL + 1 words), flags >> 8 = number of channels (tg_k_aux.hip, k_lists2) */
#define TGK_LB_TBL 4096u

/* make the engine's device the calling thread's current HIP device (every allocating / launching entry point does) */
struct tgpu_engine;
int tgpi_engine_bind(const struct tgpu_engine *eng);
/* stream-mode grid buffer of a plan: ngrid classification words, ngrid 16-bit SYNC summaries, then (dword aligned) one
 * bit per slot "plain delivery" */
#define TG_GRID_PLAIN_WORD(n) (((size_t)(n) * 6 + 3) / 4)
#define TG_GRID_COPY_BYTES(n) ((TG_GRID_PLAIN_WORD(n) + ((size_t)(n) + 31) / 32) * 4)
#define TG_GRID_BYTES(n)      (TG_GRID_COPY_BYTES(n) + 16)
void tgpi_plan_grid_plain(struct tgpu_plan *p, uint32_t ngrid, uint32_t **d_plain, uint32_t **h_plain);
int tgk_cls_plain(const uint32_t *d_cls, uint32_t n, uint32_t *d_plain, void *stream);
int tgpi_plan_last_burst(const struct tgpu_plan *p);
int tgpi_plan_ring(struct tgpu_plan *p, uint32_t **sb_ok, uint32_t **sb_code, uint32_t **maskidx, uint32_t **masks);
void tgpi_plan_set_marks(struct tgpu_plan *p, int on);	/* the caller polls completion marks in mapped records (tg_sync.c) */	/* the last execute wrote completion marks (k_burst) */

/* plan internals used by the stream synchroniser (tg_stream.c) */
struct tgpu_plan;
int tgpi_plan_grid_begin(struct tgpu_plan *p, uint32_t ngrid, uint32_t **d_packed, uint32_t **d_cls, uint16_t **d_ysum,
			 uint32_t **h_cls, uint16_t **h_ysum);	/* h_*: pinned mirrors owned by the plan */
uint32_t *tgpi_plan_defer_scratch(struct tgpu_plan *p);	/* TG_DEFER_WORDS(max_slots) dwords, after tgpi_plan_grid_begin() */
int tgpi_plan_chan_table(struct tgpu_plan *p, const struct tg_chan_ent *ents, uint32_t nchan, struct tg_chan_ent **d_out,
			 void *stream);
/* ents == NULL, nchan == 1: one channel owns the grid; else the table given to tgpi_plan_chan_table() */
int tgpi_plan_grid_load(struct tgpu_plan *p, uint32_t ngrid, const uint32_t *h_bits, uint32_t nchan, const uint32_t *codes,
			const struct tg_chan_ent *ents, void *stream);

/* device-walk batches: everything between the front end and the trellis kernels (tg_k_aux.hip, k_lists2) */
int tgk_cls_plain2(const uint32_t *d_cls, uint32_t n, uint32_t *d_plain, uint32_t *d_list_sb, uint32_t *d_cnt_sb,
		   uint8_t *d_word_chan, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_specbits /* or NULL */,
		   uint32_t *d_specbits_out /* or NULL */, const uint32_t *hints /* host, with d_specbits_out */, void *stream);
/* k_slot_e: the trellises of every plain grid slot on its channel's hinted code, needs nothing of the walk (TGPU_OPT_SLOT 3) */
int tgk_slot_early(const uint32_t *d_cls, uint32_t nslots, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_packed,
		   const uint32_t *d_masks, uint32_t hint_base, const uint32_t *hints, uint8_t *d_rec, uint8_t *d_wire, int flags, void *stream);
int tgpi_plan_is_early(const struct tgpu_plan *p);
int tgpi_plan_dev_early(struct tgpu_plan *p, const struct tg_chan_ent *d_tab, uint32_t nchan, uint32_t ngrid, const uint32_t *carry, void *stream);
/* k_slot batches (tg_k_slot.hip): front end + trellis in one launch on hinted codes, then the exact pass; tgk_masks_list builds the
 * hints' mask entries (codes by value, n <= 64) */
int tgk_slot_fused(const uint8_t *d_base, const struct tg_chan_ent *d_chan, uint32_t nchan, uint32_t nslots, uint32_t chunk,
		   uint32_t *d_packed, uint32_t *d_cls, uint16_t *d_ysum, uint32_t *d_defer, const uint32_t *d_masks,
		   uint32_t hint_base, const uint32_t *hints, uint32_t *d_specbits, uint8_t *d_rec, uint8_t *d_wire,
		   uint32_t *d_tbl, uint32_t *d_sb_ok, uint32_t *d_sb_entry, int flags, void *stream, void *ev_mid, int packed_input);
int tgk_masks_list(const uint32_t *codes, uint32_t n, uint32_t *d_masks_out, void *stream);
int tgk_masks2(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_tbl, uint32_t *d_masks, void *stream);
int tgk_lb_scan(const uint32_t *d_okbits, const uint32_t *d_dbits, const uint8_t *d_word_chan, uint32_t nwords,
		uint32_t *d_prevw, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_chan_code,
		const uint32_t *d_slot_entry, const uint32_t *d_masks, const uint32_t *d_tbl, uint32_t *d_final_code, void *stream);
int tgk_lists2(const uint32_t *d_cls, const uint32_t *d_dbits, uint32_t n, const uint32_t *d_okbits, const uint32_t *d_prevw,
	       const uint8_t *d_word_chan, const uint32_t *d_slot_entry, uint32_t *d_maskidx, uint32_t *d_list_216,
	       uint32_t *d_list_432, uint32_t *d_list_all /* or NULL */, uint32_t *d_list_sync /* with d_list_all */, uint32_t *d_cnt /* 5 words */,
	       const uint32_t *d_specbits /* or NULL */, const uint32_t *hints /* host, nchan words */, const uint32_t *d_chan_code,
	       const uint32_t *d_tbl, uint32_t nchan, void *stream);
/* compact transport form of a batch (tg_cwire.h / tg_cwire.hip): the delivered slots' 40-byte wire records -> one buffer of
 * header, channel table, bitmap, block table and 25 / 33 / 36 (41) byte records; d_total (optional): two words, the bytes the
 * batch needs and its delivered bursts (0xffffffff: cap was too small and no record was written) */
struct tg_cw_chans {
	uint32_t n;
	uint32_t gbase[64], ncls[64];
};
int tgk_cwire(const uint8_t *d_wire, const uint32_t *d_bits, uint32_t ngrid, const struct tg_cw_chans *ch, uint8_t *d_out,
	      uint32_t cap, uint32_t *d_total, void *stream);
int tgpi_plan_cwire(struct tgpu_plan *p, const struct tg_cw_chans *ch, uint32_t *d_total, void *stream);
int tgpi_plan_has_cwire(const struct tgpu_plan *p);
/* stream mode with the walk on the device (tg_stream.c: tgpu_sync_multi_launch) */
/* evs (optional, serial mode): HIP events recorded behind stage 1's three launches / stage 2's two */
int tgpi_plan_dev_prepare(struct tgpu_plan *p, uint32_t ngrid, uint32_t nchan, uint32_t *d_codes, uint32_t **d_bits_out, void *stream);
int tgpi_plan_dev_front_fused(struct tgpu_plan *p, const uint8_t *d_base, const struct tg_chan_ent *d_tab, uint32_t nchan, uint32_t ngrid,
			      uint32_t chunk, const uint32_t *carry, void *stream, void *ev_mid, int packed_input, int *fused);
int tgpi_plan_dev_stage1(struct tgpu_plan *p, uint32_t ngrid, uint32_t nchan, const struct tg_chan_ent *d_tab, uint32_t *d_codes,
			 uint32_t *d_plain, void *stream, int serial, void **evs);
int tgpi_plan_dev_stage2(struct tgpu_plan *p, const struct tg_chan_ent *d_tab, uint32_t *d_final, void *stream, int serial, void **evs);
void tgpi_plan_set_rec(struct tgpu_plan *p, uint8_t *d_rec);
void tgpi_plan_set_final_codes(struct tgpu_plan *p, const uint32_t *codes, uint32_t nchan);
void tgpi_plan_set_last_slot(struct tgpu_plan *p, uint32_t chan, uint32_t slot);
uint32_t *tgpi_plan_bits_dev(struct tgpu_plan *p);

#ifdef __cplusplus
}
#endif
#endif
