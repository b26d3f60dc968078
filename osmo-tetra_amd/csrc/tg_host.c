/*
 * tg_host.c -- engine and plan objects: device memory, batch descriptors and the launch
 * sequence of the device pipeline.  Plain C on top of the HIP runtime API.
 *
 * Launch sequence of tgpu_plan_execute() (all on the caller's stream, no host sync):
 *   k_front                      every slot  -> packed code words
 *   k_vit<SB1>                   SYNC slots  -> SB1 type-1 bits, CRC, SYNC-PDU fields, new code
 *   k_fill_{reduce,scan,apply}   forward-fill of the scrambling code (mask entry per slot)
 *   k_masks                      scrambling masks for every code in play
 *   k_vit<216>, k_vit<432>       all remaining blocks + BBK + record headers
 * This order is feedback loop 1 of SURVEY.md 3.3: SB1 is descrambled with the fixed
 * code 3, and a CRC-OK SB1 sets the code for everything after it, starting with the
 * BBK and SB2 of the same burst (lower_mac/tetra_lower_mac.c:179-186, 291-300).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tetra_gpu.h"
#include "tg_layout.h"
#include "tg_internal.h"
#include "tg_cwire.h"

struct tgpu_engine {
	int device;
};

int tgpu_plan_set_fastpath(struct tgpu_plan *p, int on);

/* process-wide switches (tgpu_engine_set_option): test aids and documented alternatives -- set by explicit calls only,
 * nothing in this library reads the environment */
static long tg_options[TGPU_OPT__COUNT] = {
	[TGPU_OPT_BURST_MAX] = 1024,	/* measured crossover of k_burst and the lane-per-trellis kernels */
	[TGPU_OPT_SLOT] = 1,		/* round 6: the trellises of a batch by one lane per slot (k_slot_t, tg_k_slot.hip); 0 = k_vit<216> + k_vit<432>; 2 = k_slot as
					 * well: front end and trellises of a device-walk batch in one launch -- built, bit-exact, and measured SLOWER than
					 * 1 on the metric's workload (0.45 against 0.39 ms per step: DESIGN.md section 4), hence not the default */
	[TGPU_OPT_RING] = 1,		/* round 6: on by default -- channels of up to four bursts per flush (the reference's own usage pattern,
					 * tetra-rx.c:82-95) answer a flush 30-40 % sooner through workgroups that stay; a flush the ring does not
					 * answer goes by launch, a ring that keeps failing is given up (tg_sync.c: ring_failed) */
};

long tgi_option(int opt)
{
	return opt > 0 && opt < TGPU_OPT__COUNT ? tg_options[opt] : 0;
}

int tgpu_engine_set_option(struct tgpu_engine *eng, int opt, long value)
{
	(void)eng;	/* (process-wide: the launch layer has no engine; the argument keeps the call next to the engine it is meant for) */
	if (opt <= 0 || opt >= TGPU_OPT__COUNT || value < 0)
		return TGPU_EINVAL;
	tg_options[opt] = value;
	return TGPU_OK;
}

long tgpu_engine_get_option(const struct tgpu_engine *eng, int opt)
{
	(void)eng;
	return tgi_option(opt);
}

/* HIP's current device is per thread: every entry point that allocates, copies or launches makes the engine's
 * device current first (a host with one thread per channel, or engines on several GPUs in one process) */
int tgpi_engine_bind(const struct tgpu_engine *eng)
{
	int cur = -1;
	if (!eng)
		return TGPU_EINVAL;
	if (hipGetDevice(&cur) == hipSuccess && cur == eng->device)
		return TGPU_OK;
	hipError_t e = hipSetDevice(eng->device);
	return e == hipSuccess ? TGPU_OK : (int)e;
}
#define BIND(eng) do { int b_ = tgpi_engine_bind(eng); if (b_) return b_; } while (0)

#define TGPU_SMALL_PLAN 256u	/* plans up to this many slots keep their descriptors in mapped host memory */
#define TGPU_NKINDS 4	/* trellis kinds TG_KIND_SB1 / _216 / _432 / _168; index 4 = BBK in block-mode lists */

struct tgpu_plan {
	struct tgpu_engine *eng;
	uint32_t max_slots, max_chan;
	uint32_t nslots, nchan, nsb, n216, n432;
	int loaded;
	int static_masks;	/* batch has no SYNC slot: mask entries are known at load time */
	int static_pending;	/* ... and the copy of the indices + the mask kernel still have to run (first execute) */
	int up_mapped;		/* small plans: the upload arena is pinned host memory the kernels read in place */
	int last_burst;		/* the last execute took the workgroup-per-burst path (records carry a completion mark) */
	int marks;		/* the owner keeps d_rec in mapped host memory and polls the marks (tgpi_plan_set_marks) */
	int wire_only;		/* tgpu_plan_set_wire_only: the trellis kernels write the wire records only */
	int no_side;		/* tgpu_plan_set_side_stream(plan, 0): everything of a batch on the caller's stream */
	/* device */
	uint8_t *d_up, *h_up;	/* upload arena (device / pinned host mirror): one copy per load */
	uint8_t *d_up_dev;	/* small plans (up_mapped): the arena of device-walk batches, whose kernels run atomics on it -- device memory */
	size_t up_bytes;
	uint64_t *d_slot_off;	/* the next seven point into d_up, laid out per load */
	uint32_t *d_slot_chan;
	int32_t *d_slot_sbord;
	uint32_t *d_list_sb, *d_list_216, *d_list_432;
	uint32_t *d_list_all;	/* the batch's slots of type NORM_1 / NORM_2, once each: the lane-per-slot kernel's items (NULL: not built) */
	uint32_t *d_list_sync;	/* ... and its SYNC slots: a list of their own (k_slot_t runs a shorter schedule where every lane holds one) */
	uint32_t nall, nsync;
	uint32_t *d_packed;
	uint32_t *d_maskidx;
	uint32_t *d_idx_stage;	/* static batches: the mask indices as uploaded (copied to d_maskidx by the first execute) */
	uint32_t *d_masks;
	uint32_t *d_chan_code;	/* in d_up */
	uint32_t *d_sb_ok, *d_sb_code;
	unsigned long long *d_block_tmp;
	uint8_t *d_wire;	/* caller-owned, optional */
	uint8_t *d_cwire;	/* caller-owned, optional: compact form of a device-walk batch's wire records (tgpu_plan_set_cwire) */
	size_t cwire_cap;
	const uint8_t *d_traffic;	/* caller-owned, optional (tgpu_plan_set_traffic): byte per slot, bit 0 traffic burst, bit 1 second block stolen */
	uint8_t *d_type4;		/* ... descrambled type-4 bits of the dumped blocks, 432 bytes per slot */
	int16_t *d_tblocks;		/* ... their 690-word dump blocks */
	uint16_t *d_tlens;		/* ... bits per slot (0: nothing dumped) */
	uint32_t *d_softarea;	/* max_slots * 512 B, allocated on the first soft execute */
	uint32_t *d_grid;	/* stream mode: classification words + SYNC summaries, max_slots * 6 B, allocated on first use */
	uint32_t *h_grid;	/* pinned host mirror of d_grid */
	int packed_ready;	/* stream mode: d_packed was filled by k_front_stream (slot = grid slot), k_front is skipped */
	int block_mode;		/* tgpu_plan_load_blocks(): items are type-5 blocks, not slots */
	int rm_decode;		/* tgpu_plan_set_rm_decode(): correct the BBK with the RM(30,14) decoder */
	int fastpath;		/* tgpu_plan_set_fastpath(): k_clean pre-pass, the trellis kernels only see the other blocks */
	uint32_t *d_dirty;	/* fastpath: [0..1] counters (216, 432), then the two item lists; 4 * (2 + 3 * max_slots) bytes */
	uint32_t *d_list_168, *d_list_bbk;	/* block mode only (in d_up) */
	uint32_t n168, nbbk;
	hipStream_t side;	/* k_vit<216> and k_vit<432> are independent: they run side by side */
	hipEvent_t ev_fork, ev_join;
	uint32_t *h_last_slot_of_chan;
	struct tg_chan_ent *d_chan_tab, *h_chan_tab;	/* multi-channel stream mode: channel table (64 entries) */
	uint32_t *d_defer;	/* stream mode: slots the packed-bit front end hands to its exact pass (count + list) */
	uint64_t max_off;	/* slot mode: largest slot offset of the load (bounds check of tgpu_plan_execute_float) */
	/* stream mode with the walk on the device (tgpu_sync_multi_launch): item counts stay on the device */
	const uint32_t *d_counts;	/* [nsb, n216, n432] behind the list builder's block sums; NULL: the host knows them */
	uint32_t *d_lb_tbl, *d_lb_ok, *d_lb_prevw;	/* device-walk batches: code table, okbits, look-back words (in d_up) */
	uint8_t *d_lb_wchan;
	uint8_t *d_rec_dev;		/* ... the batch's records (the SB1 launch runs before tgpu_plan_execute gets them) */
	int dev_mid;			/* ... SB1 / code fill / masks were done by the device-walk stages */
	int have_final;			/* ... and the codes after the batch are in h_final_code */
	uint32_t h_final_code[64];
	uint8_t *d_walk, *h_walk;	/* k_walk's blocks (tg_walk_io): up, down, device-only events (device / pinned mirror) */
	void *d_walk_recs;		/* max_chan * (min(TGW_NCAP, max_slots) + 1) node records */
	void *d_walk_tmp;		/* hand-over area of the split walk */
	uint8_t *d_walk_big;		/* scratch slots of k_walk_big (channels beyond TGW_WCAP bitmap words), on first need */
	uint32_t walk_big_slots;
	uint32_t walk_big_nodes;	/* nodes a channel of the LDS form had when it overflowed (0: never): sizes the long form's threshold */
	uint32_t walk_nodes_seen;	/* most nodes any channel of this plan's batches had (0: no batch yet): sizes the LDS form's arrays */
	uint32_t *d_bits_dev;		/* the delivered bitmap k_walk left in the upload arena */
	/* k_slot batches (TGPU_OPT_SLOT 2; tg_k_slot.hip): front end and trellis in one launch, decoding on hinted scrambling codes */
	uint32_t *d_specbits;		/* bit per grid slot "decoded by k_slot under its channel's hint" (upload arena) */
	uint32_t hint_base;		/* mask-table entry of channel 0's hint (behind the batch kernels' entries) */
	uint32_t hint_last[64];		/* the codes this plan's last device-walk batch ended with, per channel index (hints of the next one) */
	uint32_t hint_last_n;
	uint32_t hint_built[64];	/* the codes whose masks sit in entries hint_base + c */
	uint32_t hint_now[64];		/* this batch's hints (0: none) */
	int dev_prepared;		/* tgpi_plan_dev_prepare() laid this batch's arena out */
	int fused;			/* this batch's front end ran as k_slot */
	int early;			/* this batch's plain slots are decoded by k_slot_e beside the walk (TGPU_OPT_SLOT 3); early_pending: the join is still to come */
	int early_pending;
	hipStream_t side2;		/* ... on this stream (created on first use) */
	hipEvent_t ev_early0, ev_early1;
};

const char *tgpu_strerror(int err)
{
	switch (err) {
	case TGPU_OK: return "ok";
	case TGPU_EINVAL: return "invalid argument";
	case TGPU_ENOMEM: return "out of memory";
	case TGPU_ENODEV: return "no usable GPU (this library has no CPU fallback)";
	case TGPU_ECAPACITY: return "batch exceeds plan capacity";
	case TGPU_ESTATE: return "call order violated";
	case TGPU_ENOSYS: return "component not available in this process (no RCCL library to load)";
	case TGPU_ECOMM: return "communication library error";
	default: break;
	}
	if (err > 0)
		return hipGetErrorString((hipError_t)err);
	return "unknown error";
}

int tgpu_engine_create(struct tgpu_engine **out, int device)
{
	int n = 0;
	if (!out)
		return TGPU_EINVAL;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
		return TGPU_ENODEV;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess)
		return (int)e;
	int rc = tgk_init();
	if (rc)
		return rc;
	struct tgpu_engine *eng = calloc(1, sizeof(*eng));
	if (!eng)
		return TGPU_ENOMEM;
	eng->device = device;
	*out = eng;
	return TGPU_OK;
}

void tgpu_engine_destroy(struct tgpu_engine *eng)
{
	free(eng);
}

int tgpu_device_host_locality(int device, char bdf[16], int *numa_node, char *cpulist, size_t n)
{
	char id[32] = "", path[96];
	int cnt = 0;
	if (!bdf)
		return TGPU_EINVAL;
	if (hipGetDeviceCount(&cnt) != hipSuccess || device < 0 || device >= cnt)
		return TGPU_ENODEV;
	hipError_t e = hipDeviceGetPCIBusId(id, (int)sizeof(id), device);
	if (e != hipSuccess)
		return (int)e;
	for (char *q = id; *q; q++)
		if (*q >= 'A' && *q <= 'F')
			*q = (char)(*q - 'A' + 'a');	/* sysfs spells the address in lower case */
	snprintf(bdf, 16, "%s", id);
	if (numa_node) {
		*numa_node = -1;
		snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id);
		FILE *f = fopen(path, "r");
		if (f) {
			if (fscanf(f, "%d", numa_node) != 1)
				*numa_node = -1;
			fclose(f);
		}
	}
	if (cpulist && n) {
		cpulist[0] = 0;
		snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", id);
		FILE *f = fopen(path, "r");
		if (f) {
			if (fgets(cpulist, (int)n, f)) {
				size_t l = strlen(cpulist);
				while (l && (cpulist[l - 1] == '\n' || cpulist[l - 1] == ' '))
					cpulist[--l] = 0;
			} else
				cpulist[0] = 0;
			fclose(f);
		}
	}
	return TGPU_OK;
}

#define UP_ALIGN 256u
#define DALLOC(ptr, bytes) do { size_t b_ = (bytes); hipError_t e_ = hipMalloc((void **)&(ptr), b_ ? b_ : 16); \
	if (e_ != hipSuccess) { tgpu_plan_destroy(p); return (int)e_; } } while (0)

int tgpu_plan_create(struct tgpu_engine *eng, uint32_t max_slots, uint32_t max_chan, struct tgpu_plan **out)
{
	if (!eng || !out || !max_slots || !max_chan)
		return TGPU_EINVAL;
	BIND(eng);
	struct tgpu_plan *p = calloc(1, sizeof(*p));
	if (!p)
		return TGPU_ENOMEM;
	p->eng = eng;
	p->max_slots = max_slots;
	p->max_chan = max_chan;
	const size_t n = max_slots;
	/* descriptors 8n, chan 4n, sbord 4n, lists <= 8n in total, the slot list 4n, static mask indices 4n, codes, padding */
	p->up_bytes = 36 * n + 4 * (size_t)max_chan + 24 * UP_ALIGN + 4 * (TGK_LB_TBL + 1);	/* (+ the code table of device-walk batches) */
	/* small plans (the drop-in channel API at small batch sizes: a flush is a round trip, and every copy in it costs
	 * more than the bytes): descriptors and lists stay in pinned host memory and the kernels read them in place.
	 * Consequence for callers: a small plan must be idle (its last execute complete) before the next tgpu_plan_load*()
	 * rewrites that memory -- larger plans copy at load time and may be reloaded while an execute is in flight only
	 * in so far as the copy is ordered behind it by the caller (include/tetra_gpu.h, tgpu_plan_load) */
	p->up_mapped = max_slots <= TGPU_SMALL_PLAN;
	if (hipHostMalloc((void **)&p->h_up, p->up_bytes, p->up_mapped ? (hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocDefault) != hipSuccess) {
		p->h_up = NULL;
		tgpu_plan_destroy(p);
		return TGPU_ENOMEM;
	}
	if (p->up_mapped) {
		hipError_t e_ = hipHostGetDevicePointer((void **)&p->d_up, p->h_up, 0);
		if (e_ != hipSuccess) {
			p->d_up = NULL;
			tgpu_plan_destroy(p);
			return (int)e_;
		}
	} else
		DALLOC(p->d_up, p->up_bytes);
	DALLOC(p->d_packed, n * TG_PACKED_WORDS * 4);
	DALLOC(p->d_maskidx, n * 4);
	/* mask entries: 0 = the fixed code 3, 1 + c = channel c's carry-in code, then one per SYNC slot of a batch (<= n) or --
	 * device-walk batches -- one per slot of the batch's code hash table (1 + nchan + h, h < TGK_LB_TBL), whatever n is */
	/* ... and behind those, k_slot's hints: one entry per channel */
	p->hint_base = (uint32_t)(1 + (size_t)max_chan + (n > TGK_LB_TBL ? n : TGK_LB_TBL));
	DALLOC(p->d_masks, ((size_t)p->hint_base + 64) * TG_MASK_WORDS * 4);
	DALLOC(p->d_sb_ok, n * 4);
	DALLOC(p->d_sb_code, n * 4);
	DALLOC(p->d_block_tmp, ((n + 1023) / 1024 + 1) * sizeof(unsigned long long));
	p->h_last_slot_of_chan = malloc((size_t)max_chan * 4);
	if (!p->h_last_slot_of_chan) {
		tgpu_plan_destroy(p);
		return TGPU_ENOMEM;
	}
	if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
		tgpu_plan_destroy(p);
		return TGPU_ENOMEM;
	}
	*out = p;
	return TGPU_OK;
}

void tgpu_plan_destroy(struct tgpu_plan *p)
{
	if (!p)
		return;
	if (p->side) (void)hipStreamDestroy(p->side);
	if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
	if (p->ev_join) (void)hipEventDestroy(p->ev_join);
	if (p->side2) (void)hipStreamDestroy(p->side2);
	if (p->ev_early0) (void)hipEventDestroy(p->ev_early0);
	if (p->ev_early1) (void)hipEventDestroy(p->ev_early1);
	void *d[] = { p->up_mapped ? NULL : p->d_up, p->d_up_dev, p->d_packed, p->d_maskidx, p->d_masks, p->d_sb_ok, p->d_sb_code,
		      p->d_block_tmp, p->d_softarea, p->d_grid, p->d_dirty, p->d_chan_tab, p->d_defer, p->d_walk, p->d_walk_recs,
		      p->d_walk_big, p->d_walk_tmp };
	for (size_t i = 0; i < sizeof(d) / sizeof(d[0]); i++)
		if (d[i])
			(void)hipFree(d[i]);
	if (p->h_up)
		(void)hipHostFree(p->h_up);
	if (p->h_grid)
		(void)hipHostFree(p->h_grid);
	if (p->h_chan_tab)
		(void)hipHostFree(p->h_chan_tab);
	if (p->h_walk)
		(void)hipHostFree(p->h_walk);
	free(p->h_last_slot_of_chan);
	free(p);
}

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

/* the three per-slot inputs are read through byte strides so that separate arrays (tgpu_plan_load) and
 * the slot table of the stream synchroniser (tgpu_plan_load_slots) share one implementation */
#define SLOT_OFF(i)  (*(const uint64_t *)(off_b + (size_t)(i) * off_st))
#define SLOT_TYPE(i) (*(type_b + (size_t)(i) * type_st))
#define SLOT_CHAN(i) (*(const uint32_t *)(chan_b + (size_t)(i) * chan_st))
static int plan_load_strided(struct tgpu_plan *p, uint32_t nslots, const uint8_t *off_b, size_t off_st,
			     const uint8_t *type_b, size_t type_st, const uint8_t *chan_b, size_t chan_st,
			     uint32_t nchan, const uint32_t *chan_code)
{
	if (p)
		BIND(p->eng);
	if (nslots > p->max_slots || nchan > p->max_chan)
		return TGPU_ECAPACITY;
	/* pass 1: validate and count, so that the upload arena can be laid out exactly */
	uint32_t nsb = 0, n216 = 0, n432 = 0, prev = 0;
	uint64_t max_off = 0;
	for (uint32_t i = 0; i < nslots; i++) {
		if (SLOT_CHAN(i) >= nchan || SLOT_CHAN(i) < prev || (SLOT_OFF(i) >> 56))
			return TGPU_EINVAL;
		prev = SLOT_CHAN(i);
		if (SLOT_OFF(i) > max_off)
			max_off = SLOT_OFF(i);
		const uint8_t t = SLOT_TYPE(i);
		nsb += t == TETRA_TRAIN_SYNC;
		n216 += (t == TETRA_TRAIN_SYNC) + 2 * (t == TETRA_TRAIN_NORM_2);
		n432 += t == TETRA_TRAIN_NORM_1;
	}
	const int is_static = nsb == 0 && nslots;
	size_t o = 0;
#define UP_PLACE(dptr, hptr, type, count) do { dptr = (type *)(p->d_up + o); hptr = (type *)(p->h_up + o); \
		o = (o + (size_t)(count) * sizeof(type) + UP_ALIGN - 1) & ~(size_t)(UP_ALIGN - 1); } while (0)
	uint64_t *h_desc;
	uint32_t *h_chan, *h_list_sb, *h_list_216, *h_list_432, *h_list_all, *h_list_sync, *h_code, *h_idx, *d_idx_stage;
	int32_t *h_sbord;
	UP_PLACE(p->d_slot_off, h_desc, uint64_t, nslots);
	UP_PLACE(p->d_slot_chan, h_chan, uint32_t, nslots);
	UP_PLACE(p->d_slot_sbord, h_sbord, int32_t, nslots);
	UP_PLACE(p->d_list_sb, h_list_sb, uint32_t, nsb);
	UP_PLACE(p->d_list_216, h_list_216, uint32_t, n216);
	UP_PLACE(p->d_list_432, h_list_432, uint32_t, n432);
	UP_PLACE(p->d_list_all, h_list_all, uint32_t, n432 + (n216 - nsb) / 2);
	UP_PLACE(p->d_list_sync, h_list_sync, uint32_t, nsb);
	UP_PLACE(p->d_chan_code, h_code, uint32_t, nchan);
	UP_PLACE(d_idx_stage, h_idx, uint32_t, is_static ? nslots : 0);
#undef UP_PLACE
	if (o > p->up_bytes)
		return TGPU_ECAPACITY;	/* cannot happen: sized for the worst case in tgpu_plan_create */

	/* pass 2: fill the pinned mirror */
	for (uint32_t c = 0; c < nchan; c++)
		p->h_last_slot_of_chan[c] = 0xffffffffu;
	uint32_t isb = 0, i216 = 0, i432 = 0, iall = 0, isync = 0;
	for (uint32_t i = 0; i < nslots; i++) {
		const uint8_t t = SLOT_TYPE(i);
		const uint32_t ch = SLOT_CHAN(i);
		if (t == TETRA_TRAIN_NORM_2 || t == TETRA_TRAIN_NORM_1)
			h_list_all[iall++] = i;
		else if (t == TETRA_TRAIN_SYNC)
			h_list_sync[isync++] = i;
		p->h_last_slot_of_chan[ch] = i;
		/* descriptor = offset | type << 56 (one scalar load per slot in the front kernel) */
		h_desc[i] = SLOT_OFF(i) | ((uint64_t)t << 56);
		h_chan[i] = ch;
		h_sbord[i] = -1;
		switch (t) {
		case TETRA_TRAIN_SYNC:
			h_sbord[i] = (int32_t)isb;
			h_list_sb[isb++] = i;
			h_list_216[i216++] = (i << 1) | 1;	/* SB2 */
			break;
		case TETRA_TRAIN_NORM_2:
			h_list_216[i216++] = (i << 1);
			h_list_216[i216++] = (i << 1) | 1;
			break;
		case TETRA_TRAIN_NORM_1:
			h_list_432[i432++] = i;
			break;
		default:
			break;
		}
		if (is_static)
			h_idx[i] = 1 + ch;
	}
	memcpy(h_code, chan_code, (size_t)nchan * 4);
	if (!p->up_mapped)
		HCHK(hipMemcpy(p->d_up, p->h_up, o, hipMemcpyHostToDevice));
	/* no SYNC burst in the batch: every slot keeps its channel's carry-in code, so the forward fill degenerates to
	 * entry 1 + chan and the masks depend on the carry-in codes alone: one index copy + k_masks over 1 + nchan entries,
	 * issued on the caller's stream in front of the first execute (static_pending) */
	p->static_masks = is_static;
	p->static_pending = is_static;
	p->d_counts = NULL;
	p->dev_mid = 0;
	p->have_final = 0;
	p->max_off = max_off;
	p->d_idx_stage = d_idx_stage;
	p->packed_ready = 0;
	p->block_mode = 0;
	p->nslots = nslots;
	p->nchan = nchan;
	p->nsb = nsb;
	p->n216 = n216;
	p->n432 = n432;
	p->nall = iall;
	p->nsync = isync;
	p->loaded = 1;
	return TGPU_OK;
}
#undef SLOT_OFF
#undef SLOT_TYPE
#undef SLOT_CHAN

/* ---- stream mode (tg_stream.c): slot i of the plan = grid slot i of the classified stream ---- */
int tgpi_plan_grid_begin(struct tgpu_plan *p, uint32_t ngrid, uint32_t **d_packed, uint32_t **d_cls, uint16_t **d_ysum,
			 uint32_t **h_cls, uint16_t **h_ysum)
{
	if (!p || !ngrid)
		return TGPU_EINVAL;
	BIND(p->eng);
	if (ngrid > p->max_slots)
		return TGPU_ECAPACITY;
	if (!p->d_grid) {
		hipError_t e = hipMalloc((void **)&p->d_grid, TG_GRID_BYTES(p->max_slots));
		if (e != hipSuccess)
			return (int)e;
	}
	if (!p->d_defer) {
		hipError_t e = hipMalloc((void **)&p->d_defer, TG_DEFER_WORDS(p->max_slots) * 4);
		if (e != hipSuccess)
			return (int)e;
	}
	if (!p->h_grid && hipHostMalloc((void **)&p->h_grid, TG_GRID_BYTES(p->max_slots), hipHostMallocDefault) != hipSuccess) {
		p->h_grid = NULL;
		return TGPU_ENOMEM;
	}
	p->loaded = 0;
	*d_packed = p->d_packed;
	*d_cls = p->d_grid;
	*d_ysum = (uint16_t *)(p->d_grid + ngrid);
	*h_cls = p->h_grid;
	*h_ysum = (uint16_t *)(p->h_grid + ngrid);
	return TGPU_OK;
}

/* the "plain delivery" bitmap of the grid (k_cls_plain): behind the words and summaries on both sides, so that one
 * copy of TG_GRID_COPY_BYTES(ngrid) brings all three to the host */
void tgpi_plan_grid_plain(struct tgpu_plan *p, uint32_t ngrid, uint32_t **d_plain, uint32_t **h_plain)
{
	*d_plain = p->d_grid + TG_GRID_PLAIN_WORD(ngrid);
	*h_plain = p->h_grid + TG_GRID_PLAIN_WORD(ngrid);
}

uint32_t *tgpi_plan_defer_scratch(struct tgpu_plan *p)
{
	return p ? p->d_defer : NULL;
}

/* device copy of a channel table for the multi-channel stream mode (owned by the plan, <= 64 entries) */
int tgpi_plan_chan_table(struct tgpu_plan *p, const struct tg_chan_ent *ents, uint32_t nchan, struct tg_chan_ent **d_out,
			 void *stream)
{
	if (!p || !ents || !nchan || nchan > 64 || nchan > p->max_chan || !d_out)
		return TGPU_EINVAL;
	BIND(p->eng);
	if (!p->d_chan_tab)
		HCHK(hipMalloc((void **)&p->d_chan_tab, 64 * sizeof(struct tg_chan_ent)));
	if (!p->h_chan_tab && hipHostMalloc((void **)&p->h_chan_tab, 64 * sizeof(struct tg_chan_ent), hipHostMallocDefault) != hipSuccess) {
		p->h_chan_tab = NULL;
		return TGPU_ENOMEM;
	}
	memcpy(p->h_chan_tab, ents, (size_t)nchan * sizeof(*ents));
	HCHK(hipMemcpyAsync(p->d_chan_tab, p->h_chan_tab, (size_t)nchan * sizeof(*ents), hipMemcpyHostToDevice, (hipStream_t)stream));
	*d_out = p->d_chan_tab;
	return TGPU_OK;
}

/* nchan == 1, ents == NULL: one channel owning the whole grid.  Otherwise ents = the table given to
 * tgpi_plan_chan_table() (channel c: grid slots gbase .. gbase + ncls - 1), codes[c] = its carry-in scrambling code */
int tgpi_plan_grid_load(struct tgpu_plan *p, uint32_t ngrid, const uint32_t *h_bits, uint32_t nchan, const uint32_t *codes,
			const struct tg_chan_ent *ents, void *stream)
{
	if (!p || !ngrid || !h_bits || !p->d_grid || !nchan || !codes || (ents && !p->d_chan_tab))
		return TGPU_EINVAL;
	BIND(p->eng);
	if (ngrid > p->max_slots || nchan > p->max_chan)
		return TGPU_ECAPACITY;
	const size_t nwords = ((size_t)ngrid + 31) / 32, nblk = ((size_t)ngrid + 1023) / 1024;
	size_t o = 0;
#define UP_AT(ptr, type, count) do { ptr = (type *)(p->d_up + o); \
		o = (o + (size_t)(count) * sizeof(type) + UP_ALIGN - 1) & ~(size_t)(UP_ALIGN - 1); } while (0)
	uint32_t *d_bits, *d_blk;
	UP_AT(p->d_chan_code, uint32_t, nchan);
	UP_AT(d_bits, uint32_t, nwords);
	const size_t upload = o;
	UP_AT(p->d_slot_chan, uint32_t, ngrid);
	UP_AT(p->d_slot_sbord, int32_t, ngrid);
	UP_AT(p->d_list_sb, uint32_t, ngrid);
	UP_AT(p->d_list_216, uint32_t, 2 * (size_t)ngrid);
	UP_AT(p->d_list_432, uint32_t, ngrid);
	UP_AT(d_blk, uint32_t, 3 * (nblk + 1));
#undef UP_AT
	p->d_slot_off = NULL;
	p->d_list_all = p->d_list_sync = NULL;	/* (host-walk grid batches keep the lane-per-block kernels: their lists come from k_grid_lists) */
	p->nall = p->nsync = 0;
	if (o > p->up_bytes)
		return TGPU_ECAPACITY;
	memcpy(p->h_up, codes, (size_t)nchan * 4);
	memcpy(p->h_up + ((uint8_t *)d_bits - p->d_up), h_bits, nwords * 4);
	if (!p->up_mapped)
		HCHK(hipMemcpyAsync(p->d_up, p->h_up, upload, hipMemcpyHostToDevice, (hipStream_t)stream));
	int rc = tgk_grid_lists(p->d_grid, d_bits, ngrid, d_blk, p->d_slot_chan, p->d_slot_sbord, p->d_list_sb,
				p->d_list_216, p->d_list_432, ents ? p->d_chan_tab : NULL, ents ? nchan : 1, stream);
	if (rc)
		return rc;
	uint32_t tot[3];
	HCHK(hipMemcpyAsync(tot, d_blk + 3 * nblk, sizeof(tot), hipMemcpyDeviceToHost, (hipStream_t)stream));
	HCHK(hipStreamSynchronize((hipStream_t)stream));
	/* last delivered slot of every channel (tgpu_plan_final_codes) */
	for (uint32_t c = 0; c < nchan; c++) {
		const size_t w0 = ents ? ents[c].gbase / 32 : 0;
		const size_t w1 = ents ? (c + 1 < nchan ? ents[c + 1].gbase / 32 : nwords) : nwords;
		p->h_last_slot_of_chan[c] = 0xffffffffu;
		for (size_t wd = w1; wd-- > w0;)
			if (h_bits[wd]) {
				p->h_last_slot_of_chan[c] = (uint32_t)(wd * 32 + 31 - (uint32_t)__builtin_clz(h_bits[wd]));
				break;
			}
	}
	p->static_masks = 0;
	p->static_pending = 0;
	p->packed_ready = 1;
	p->block_mode = 0;
	p->d_counts = NULL;
	p->dev_mid = 0;
	p->have_final = 0;
	p->nslots = ngrid;
	p->nchan = nchan;
	p->nsb = tot[0];
	p->n216 = tot[1];
	p->n432 = tot[2];
	p->loaded = 1;
	return TGPU_OK;
}

/*
 * Device-walk batches (tg_stream.c: tgpu_sync_multi_launch): nothing comes back to the host before the decode, and what
 * lies between the front end and the trellis kernels is five small launches (tg_k_aux.hip, k_lists2):
 *   stage 1 (before k_walk)  one memset (counters, code table, okbits), k_cls_plain2; then beside the walk, on the plan's
 *                            side stream: k_vit<SB1> over the SYNC-classified slots, k_masks2
 *   stage 2 (after k_walk)   k_lb_scan, k_lists2 (mask entry per delivered slot, item lists), the two trellis kernels with
 *                            their item counts read on the device
 * d_codes / d_tab: the channels' carry-in codes and the channel table, already on the device (tg_walk_io's upload block);
 * d_plain: where k_cls_plain2 leaves the plain bitmap; d_final: 65 words for the codes after the batch + the overflow flag
 * of the code table (in the block that goes to the host); *d_bits_out: where k_walk is to leave the delivered bitmap.
 * serial != 0: everything on the caller's stream in program order (per-stage profiling).
 */
/* round 6: the arena's layout and the one memset moved in front of the front end (tgpi_plan_dev_prepare) -- k_slot's SYNC lanes run the
 * code look-back's atomics from inside the front-end launch */
int tgpi_plan_dev_prepare(struct tgpu_plan *p, uint32_t ngrid, uint32_t nchan, uint32_t *d_codes, uint32_t **d_bits_out, void *stream)
{
	if (!p || !ngrid || !p->d_grid || !nchan || !d_codes || !d_bits_out)
		return TGPU_EINVAL;
	BIND(p->eng);
	if (ngrid > p->max_slots || nchan > p->max_chan)
		return TGPU_ECAPACITY;
	const size_t nwords = ((size_t)ngrid + 31) / 32;
	size_t o = 0;
	/* counters, code table and okbits are the targets of device atomics (k_vit<SB1> LOOKBACK, k_cls_plain2, k_lists2): never
	 * in mapped host memory (PCIe atomics are slow where they exist at all) -- a small plan gets a device arena for these batches */
	uint8_t *base = p->d_up;
	if (p->up_mapped) {
		if (!p->d_up_dev)
			HCHK(hipMalloc((void **)&p->d_up_dev, p->up_bytes));
		base = p->d_up_dev;
	}
#define UP_AT(ptr, type, count) do { ptr = (type *)(base + o); \
		o = (o + (size_t)(count) * sizeof(type) + UP_ALIGN - 1) & ~(size_t)(UP_ALIGN - 1); } while (0)
	uint32_t *d_cnt, *d_tbl, *d_ok, *d_bits, *d_prevw;
	uint8_t *d_wchan;
	UP_AT(d_cnt, uint32_t, 8);
	UP_AT(d_tbl, uint32_t, TGK_LB_TBL + 1);
	UP_AT(d_ok, uint32_t, nwords);
	const size_t zero_bytes = o;		/* counters, code table (+ overflow flag), okbits: cleared per batch in one go */
	UP_AT(d_bits, uint32_t, nwords);
	UP_AT(d_prevw, uint32_t, nwords);
	UP_AT(d_wchan, uint8_t, nwords);
	UP_AT(p->d_slot_sbord, int32_t, ngrid);		/* here: the mask entry a SYNC slot's good SB1 took (k_vit<SB1>) */
	UP_AT(p->d_list_sb, uint32_t, ngrid);
	UP_AT(p->d_list_216, uint32_t, 2 * (size_t)ngrid);
	UP_AT(p->d_list_432, uint32_t, ngrid);
	UP_AT(p->d_list_all, uint32_t, ngrid);
	UP_AT(p->d_list_sync, uint32_t, ngrid);
	UP_AT(p->d_specbits, uint32_t, nwords + 2);
#undef UP_AT
	p->d_slot_off = NULL;
	p->d_slot_chan = NULL;
	if (o > p->up_bytes)
		return TGPU_ECAPACITY;
	p->d_chan_code = d_codes;
	p->d_bits_dev = d_bits;
	p->d_counts = d_cnt;
	p->d_lb_tbl = d_tbl;
	p->d_lb_ok = d_ok;
	p->d_lb_prevw = d_prevw;
	p->d_lb_wchan = d_wchan;
	*d_bits_out = d_bits;
	p->loaded = 0;
	p->nslots = ngrid;
	p->nchan = nchan;
	hipStream_t s = (hipStream_t)stream;
	HCHK(hipMemsetAsync(base, 0, zero_bytes, s));
	p->dev_prepared = 1;
	p->fused = 0;
	p->early = 0;
	return TGPU_OK;
}

/* the hints of a batch: the caller's carry-in code, else what this plan's last batch of the channel ended with; their masks into the
 * entries hint_base + c when they changed.  Returns 1 when some channel has one, 0 when none has, < 0 (-error) on failure */
static int plan_hints(struct tgpu_plan *p, uint32_t nchan, const uint32_t *carry, void *stream)
{
	int any = 0, rebuild = 0;
	for (uint32_t c = 0; c < nchan; c++) {
		p->hint_now[c] = carry[c] ? carry[c] : (c < p->hint_last_n ? p->hint_last[c] : 0u);
		any |= p->hint_now[c] != 0;
		rebuild |= p->hint_now[c] != p->hint_built[c];
	}
	if (any && rebuild) {
		int rc = tgk_masks_list(p->hint_now, nchan, p->d_masks + (size_t)p->hint_base * TG_MASK_WORDS, stream);
		if (rc)
			return rc > 0 ? -rc : rc;
		memcpy(p->hint_built, p->hint_now, (size_t)nchan * 4);
	}
	return any;
}

/*
 * TGPU_OPT_SLOT 3: behind the front end (both passes), every plain grid slot's trellises on the channels' hinted codes (k_slot_e) on a
 * stream of the plan's own, BESIDE the small kernels, the walk and the code look-back; the batch's own stream meets it again in front of
 * the launch that re-decodes what the look-back says was decoded under another code (plan_run).  Meant for a caller who waits for every
 * batch: on paper the trellises leave the critical path (front end 130 + max(trellises 245, walk chain 195) instead of 130 + 195 +
 * 216 us); measured (tools/experiments/one_batch.py, one plan and one stream in the process): 0.603 against 0.620 ms -- the walk's
 * small kernels make little headway beside a kernel that holds every SIMD, even with that kernel held to two waves per SIMD (at three
 * the two forms are level, at one the kernel itself takes 0.73 ms).  With the plan's side streams off (tgpu_plan_set_side_stream(plan,
 * 0): several batches in flight) it runs in line.
 */
int tgpi_plan_dev_early(struct tgpu_plan *p, const struct tg_chan_ent *d_tab, uint32_t nchan, uint32_t ngrid, const uint32_t *carry, void *stream)
{
	if (!p || !p->dev_prepared || !p->d_rec_dev || nchan > 64)
		return TGPU_EINVAL;
	if (tgi_option(TGPU_OPT_SLOT) != 3 || p->rm_decode || p->fastpath || p->d_traffic || ngrid < 64)
		return TGPU_OK;
	BIND(p->eng);
	int any = plan_hints(p, nchan, carry, stream);
	if (any < 0)
		return -any;
	if (!any)
		return TGPU_OK;
	hipStream_t s = (hipStream_t)stream, se = s;
	if (!p->no_side) {
		if (!p->side2) {
			HCHK(hipStreamCreateWithFlags(&p->side2, hipStreamNonBlocking));
			HCHK(hipEventCreateWithFlags(&p->ev_early0, hipEventDisableTiming));
			HCHK(hipEventCreateWithFlags(&p->ev_early1, hipEventDisableTiming));
		}
		HCHK(hipEventRecord(p->ev_early0, s));
		HCHK(hipStreamWaitEvent(p->side2, p->ev_early0, 0));
		se = p->side2;
	}
	const int kf = (p->wire_only && p->d_wire ? TGK_F_WIREONLY : 0);
	int rc = tgk_slot_early(p->d_grid, ngrid, d_tab, nchan, p->d_packed, p->d_masks, p->hint_base, p->hint_now, p->d_rec_dev, p->d_wire, kf, (void *)se);
	if (rc)
		return rc;
	if (se != s) {
		HCHK(hipEventRecord(p->ev_early1, se));
		p->early_pending = 1;
	}
	p->early = 1;
	return TGPU_OK;
}

/*
 * k_slot batches (TGPU_OPT_SLOT 2): the front end and the trellises in one launch, decoding on hinted codes -- hints[c] (host, nchan
 * words; 0 = none): the caller's carry-in code, else what this plan's last batch of the channel ended with.  *fused = 1 when the batch
 * was launched that way, 0 when it is not for this form (no hint at all, an option only the other kernels implement: the caller runs
 * tgk_front_stream_multi).  Wants tgpi_plan_dev_prepare() and tgpi_plan_set_rec() in front.
 */
int tgpi_plan_dev_front_fused(struct tgpu_plan *p, const uint8_t *d_base, const struct tg_chan_ent *d_tab, uint32_t nchan, uint32_t ngrid,
			      uint32_t chunk, const uint32_t *carry, void *stream, void *ev_mid, int packed_input, int *fused)
{
	if (!p || !p->dev_prepared || !p->d_rec_dev || nchan > 64 || !fused)
		return TGPU_EINVAL;
	*fused = 0;
	if (tgi_option(TGPU_OPT_SLOT) != 2 || tgi_option(TGPU_OPT_STREAM_EXACT) || p->rm_decode || p->fastpath || p->d_traffic || ngrid < 64)
		return TGPU_OK;
	BIND(p->eng);
	const int any = plan_hints(p, nchan, carry, stream);
	if (any < 0)
		return -any;
	if (!any)
		return TGPU_OK;
	int rc;
	const int kf = TGK_F_LOOKBACK | (int)(nchan << 8) | (p->wire_only && p->d_wire ? TGK_F_WIREONLY : 0);
	rc = tgk_slot_fused(d_base, d_tab, nchan, ngrid, chunk, p->d_packed, p->d_grid, (uint16_t *)(p->d_grid + ngrid), p->d_defer,
			    p->d_masks, p->hint_base, p->hint_now, p->d_specbits, p->d_rec_dev, p->d_wire, p->d_lb_tbl, p->d_lb_ok,
			    (uint32_t *)p->d_slot_sbord, kf, stream, ev_mid, packed_input);
	if (rc)
		return rc;
	p->fused = 1;
	*fused = 1;
	return TGPU_OK;
}

int tgpi_plan_dev_stage1(struct tgpu_plan *p, uint32_t ngrid, uint32_t nchan, const struct tg_chan_ent *d_tab, uint32_t *d_codes,
			 uint32_t *d_plain, void *stream, int serial, void **evs)
{
	if (!p || !p->dev_prepared || !ngrid || !nchan || !d_codes || !d_tab)
		return TGPU_EINVAL;
	BIND(p->eng);
	hipStream_t s = (hipStream_t)stream;
	uint32_t *d_cnt = (uint32_t *)p->d_counts, *d_tbl = p->d_lb_tbl, *d_ok = p->d_lb_ok;
	uint8_t *d_wchan = p->d_lb_wchan;
	p->dev_prepared = 0;
	int rc = tgk_cls_plain2(p->d_grid, ngrid, d_plain, p->d_list_sb, d_cnt, d_wchan, d_tab, nchan, p->fused ? p->d_specbits : NULL,
				p->early ? p->d_specbits : NULL, p->hint_now, stream);
	if (rc)
		return rc;
	if (evs)
		HCHK(hipEventRecord((hipEvent_t)evs[0], s));
	serial = serial || p->no_side;
	hipStream_t s2 = serial ? s : p->side;
	if (!serial) {
		HCHK(hipEventRecord(p->ev_fork, s));
		HCHK(hipStreamWaitEvent(p->side, p->ev_fork, 0));
	}
	const int kf = TGK_F_LOOKBACK | (int)(nchan << 8) | (p->wire_only && p->d_wire ? TGK_F_WIREONLY : 0);
	rc = tgk_vit(TG_KIND_SB1, p->d_list_sb, ngrid, p->d_packed, d_tbl, NULL, p->d_rec_dev, d_ok, (uint32_t *)p->d_slot_sbord, p->d_wire,
		     NULL, kf, d_cnt, (void *)s2);
	if (rc)
		return rc;
	if (evs)
		HCHK(hipEventRecord((hipEvent_t)evs[1], s2));
	rc = tgk_masks2(d_codes, nchan, d_tbl, p->d_masks, (void *)s2);
	if (rc)
		return rc;
	if (evs)
		HCHK(hipEventRecord((hipEvent_t)evs[2], s2));
	if (!serial)
		HCHK(hipEventRecord(p->ev_join, p->side));
	return TGPU_OK;
}

int tgpi_plan_dev_stage2(struct tgpu_plan *p, const struct tg_chan_ent *d_tab, uint32_t *d_final, void *stream, int serial, void **evs)
{
	if (!p || !p->d_counts || !p->d_bits_dev || !d_tab || !d_final)
		return TGPU_EINVAL;
	BIND(p->eng);
	const uint32_t ngrid = p->nslots;
	hipStream_t s = (hipStream_t)stream;
	serial = serial || p->no_side;
	if (!serial)
		HCHK(hipStreamWaitEvent(s, p->ev_join, 0));
	int rc = tgk_lb_scan(p->d_lb_ok, p->d_bits_dev, p->d_lb_wchan, (ngrid + 31) / 32, p->d_lb_prevw, d_tab, p->nchan, p->d_chan_code,
			     (const uint32_t *)p->d_slot_sbord, p->d_masks, p->d_lb_tbl, d_final, stream);
	if (rc)
		return rc;
	if (evs)
		HCHK(hipEventRecord((hipEvent_t)evs[0], s));
	rc = tgk_lists2(p->d_grid, p->d_bits_dev, ngrid, p->d_lb_ok, p->d_lb_prevw, p->d_lb_wchan, (const uint32_t *)p->d_slot_sbord,
			p->d_maskidx, p->d_list_216, p->d_list_432, p->d_list_all, p->d_list_sync, (uint32_t *)p->d_counts,
			(p->fused || p->early) ? p->d_specbits : NULL, p->hint_now, p->d_chan_code, p->d_lb_tbl, p->nchan, stream);
	if (rc)
		return rc;
	if (evs)
		HCHK(hipEventRecord((hipEvent_t)evs[1], s));
	for (uint32_t c = 0; c < p->nchan; c++)
		p->h_last_slot_of_chan[c] = 0xffffffffu;
	p->static_masks = 0;
	p->static_pending = 0;
	p->packed_ready = 1;
	p->block_mode = 0;
	p->dev_mid = 1;		/* plan_run: SB1 / fill / masks are done, the trellis kernels read their counts on the device */
	p->have_final = 0;
	p->nsb = ngrid;		/* upper bounds: the launches are sized for them */
	p->n216 = 2 * ngrid;
	p->n432 = ngrid;
	p->nall = ngrid;
	p->nsync = ngrid;
	p->loaded = 1;
	return TGPU_OK;
}

/* the SB1 launch of stage 1 writes into the batch's records: tell the plan where they are before stage 1 */
void tgpi_plan_set_rec(struct tgpu_plan *p, uint8_t *d_rec)
{
	if (p)
		p->d_rec_dev = d_rec;
}

/* codes in force after a device-walk batch (k_lb_scan left them in the block that came to the host) */
void tgpi_plan_set_final_codes(struct tgpu_plan *p, const uint32_t *codes, uint32_t nchan)
{
	if (!p || nchan > p->max_chan)
		return;
	memcpy(p->h_final_code, codes, (size_t)nchan * 4);
	p->have_final = 1;
	/* what the next batch of these channels on this plan may decode on before its own SYNC bursts are looked at (k_slot) */
	if (nchan <= 64) {
		memcpy(p->hint_last, codes, (size_t)nchan * 4);
		p->hint_last_n = nchan;
	}
}

void tgpi_plan_set_last_slot(struct tgpu_plan *p, uint32_t chan, uint32_t slot)
{
	if (p && chan < p->max_chan)
		p->h_last_slot_of_chan[chan] = slot;
}

/* blocks of a device-walk batch (tg_walk_io): allocated once per plan for max_chan channels and max_slots grid slots, the
 * down block laid out for this batch's nchan / ngrid so that ONE copy brings summaries, eager events and bitmap */
int tgpi_plan_walk_io(struct tgpu_plan *p, uint32_t nchan, uint32_t ngrid, struct tg_walk_io *io)
{
	if (!p || !io || !nchan || nchan > 64 || nchan > p->max_chan || ngrid > p->max_slots)
		return TGPU_EINVAL;
	BIND(p->eng);
	const size_t nc = p->max_chan < 64 ? p->max_chan : 64;
	const size_t up = 64 * (sizeof(struct tg_chan_ent) + sizeof(struct tg_walk_root) + 4);
	const size_t down_max = 64 * sizeof(struct tg_walk_sum) + 68 * 4 + nc * (size_t)TGW_EVEAGER * sizeof(tgpu_sync_event_rec_dev) +
				(((size_t)p->max_slots + 31) / 32 + 4) * 4;
	const size_t big = nc * (size_t)TGW_EVCAP * sizeof(tgpu_sync_event_rec_dev);
	const size_t o_down = (up + 255) & ~(size_t)255, o_big = (o_down + down_max + 255) & ~(size_t)255;
	if (!p->d_walk) {
		HCHK(hipMalloc((void **)&p->d_walk, o_big + big));
		/* (mapped and coherent: the batch's outcome is stored into it by a kernel, tgk_copy16) */
		if (hipHostMalloc((void **)&p->h_walk, o_big, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
			p->h_walk = NULL;
			(void)hipFree(p->d_walk);
			p->d_walk = NULL;
			return TGPU_ENOMEM;
		}
	}
	if (!p->d_walk_recs) {
		/* (a channel has at most as many nodes as slots) */
		hipError_t e = hipMalloc(&p->d_walk_recs, nc * (size_t)((p->max_slots < TGW_NCAP ? p->max_slots : TGW_NCAP) + 1) * TGW_REC_BYTES);
		if (e != hipSuccess) {
			p->d_walk_recs = NULL;
			return (int)e;
		}
	}
	if (!p->d_walk_tmp) {
		hipError_t e = hipMalloc(&p->d_walk_tmp, TGW_TMP_BYTES);
		if (e != hipSuccess) {
			p->d_walk_tmp = NULL;
			return (int)e;
		}
	}
	memset(io, 0, sizeof(*io));
	io->d_tmp = p->d_walk_tmp;
	io->d_up0 = p->d_walk;
	io->h_up0 = p->h_walk;
	io->up_bytes = up;
	io->d_tab = (struct tg_chan_ent *)io->d_up0;
	io->h_tab = (struct tg_chan_ent *)io->h_up0;
	io->d_roots = (struct tg_walk_root *)(io->d_tab + 64);
	io->h_roots = (struct tg_walk_root *)(io->h_tab + 64);
	io->d_codes = (uint32_t *)(io->d_roots + 64);
	io->h_codes = (uint32_t *)(io->h_roots + 64);
	io->d_down0 = p->d_walk + o_down;
	io->h_down0 = p->h_walk + o_down;
	{
		void *dv = NULL;
		io->hd_down0 = (hipHostGetDevicePointer(&dv, p->h_walk, 0) == hipSuccess && dv) ? (uint8_t *)dv + o_down : NULL;
		io->hd_up0 = io->hd_down0 ? (uint8_t *)dv : NULL;
	}
	io->d_sums = (struct tg_walk_sum *)io->d_down0;
	io->h_sums = (struct tg_walk_sum *)io->h_down0;
	io->d_final = (uint32_t *)(io->d_sums + 64);
	io->h_final = (uint32_t *)(io->h_sums + 64);
	io->d_eager = (tgpu_sync_event_rec_dev *)(io->d_final + 68);
	io->h_eager = (tgpu_sync_event_rec_dev *)(io->h_final + 68);
	io->d_bits2 = (uint32_t *)(io->d_eager + (size_t)nchan * TGW_EVEAGER);
	io->h_bits2 = (uint32_t *)(io->h_eager + (size_t)nchan * TGW_EVEAGER);
	io->down_bytes = (size_t)((uint8_t *)(io->d_bits2 + ((size_t)ngrid + 31) / 32) - io->d_down0);
	io->d_evbig = (tgpu_sync_event_rec_dev *)(p->d_walk + o_big);
	io->d_recs = p->d_walk_recs;
	io->rec_stride = (p->max_slots < TGW_NCAP ? p->max_slots : TGW_NCAP) + 1;
	return TGPU_OK;
}

/* A channel short enough for the LDS form of the walk can still have more exceptions than that form holds (TGW_NCAP: a noisy
 * recording).  The batch it happens in goes through the host walks; the plan remembers the exception density it saw, and from
 * then on channels that would overflow at that density take the long form (k_walk_big) from the start.
 * tgpi_plan_walk_threshold(): channels of more than this many slots take the long form. */
void tgpi_plan_walk_overflow(struct tgpu_plan *p, uint32_t nslots, uint32_t nnodes)
{
	if (!p || !nslots || !nnodes)
		return;
	/* slots at which this density reaches three quarters of TGW_NCAP */
	uint64_t t = (uint64_t)nslots * (TGW_NCAP * 3 / 4) / nnodes;
	if (t < 4096)
		t = 4096;
	if (!p->walk_big_nodes || t < p->walk_big_nodes)
		p->walk_big_nodes = (uint32_t)t;
}

/* The LDS form's node arrays are laid out per launch: for twice the nodes the plan's channels have had so far (a workgroup
 * that asks for 25 KB of LDS finds a compute unit sooner than one that asks for 70), for the full TGW_NCAP before the first
 * batch has come back.  A batch whose channel outgrows the cap goes through the host walks (TGW_WHY_NODES), and the cap follows. */
uint32_t tgpi_plan_walk_ncap(const struct tgpu_plan *p)
{
	if (!p || !p->walk_nodes_seen)
		return TGW_NCAP;
	uint32_t n = 512;
	while (n < TGW_NCAP && n < 2 * p->walk_nodes_seen + 256)
		n <<= 1;
	return n;
}

void tgpi_plan_walk_seen(struct tgpu_plan *p, uint32_t nnodes)
{
	if (p && nnodes > p->walk_nodes_seen)
		p->walk_nodes_seen = nnodes;
}

uint32_t tgpi_plan_walk_threshold(const struct tgpu_plan *p)
{
	const uint32_t cap = TGW_WCAP * 32u;
	return p && p->walk_big_nodes && p->walk_big_nodes < cap ? p->walk_big_nodes : cap;
}

int tgpi_plan_walk_big(struct tgpu_plan *p, uint32_t nbig, struct tg_walk_io *io)
{
	if (!p || !io || !nbig || nbig > TGW_BIG_MAX)
		return TGPU_EINVAL;
	BIND(p->eng);
	/* caps from the plan's capacity: one such channel can be as long as the plan (the others are then short) */
	io->big.wcap = (p->max_slots + 31) / 32 + 1;
	io->big.ncap = tg_walk_big_ncap(p->max_slots);
	io->big.evcap = 4 * io->big.ncap;
	struct tg_walk_big_layout L;
	tg_walk_big_offsets(io->big.wcap, io->big.ncap, io->big.evcap, &L);
	if (p->walk_big_slots < nbig) {
		if (p->d_walk_big) {	/* (the plan is idle between collect and the next launch: nothing uses the old area) */
			(void)hipFree(p->d_walk_big);
			p->d_walk_big = NULL;
			p->walk_big_slots = 0;
		}
		hipError_t e = hipMalloc((void **)&p->d_walk_big, (size_t)nbig * L.slot_bytes);
		if (e != hipSuccess) {
			p->d_walk_big = NULL;
			return (int)e;
		}
		p->walk_big_slots = nbig;
	}
	io->d_big = p->d_walk_big;
	return TGPU_OK;
}

uint32_t *tgpi_plan_bits_dev(struct tgpu_plan *p)
{
	return p ? p->d_bits_dev : NULL;
}

int tgpu_plan_load(struct tgpu_plan *p, uint32_t nslots, const uint64_t *slot_off, const uint8_t *slot_type,
		   const uint32_t *slot_chan, uint32_t nchan, const uint32_t *chan_code)
{
	if (!p || (nslots && (!slot_off || !slot_type || !slot_chan)) || !nchan || !chan_code)
		return TGPU_EINVAL;
	return plan_load_strided(p, nslots, (const uint8_t *)slot_off, 8, slot_type, 1, (const uint8_t *)slot_chan, 4,
				 nchan, chan_code);
}

int tgpu_plan_load_slots(struct tgpu_plan *p, uint32_t nslots, const struct tgpu_sync_slot *slots, uint32_t scramb_init)
{
	static const uint32_t chan0 = 0;
	if (!p || (nslots && !slots))
		return TGPU_EINVAL;
	return plan_load_strided(p, nslots, (const uint8_t *)&slots->off, sizeof(*slots), &slots->type, sizeof(*slots),
				 (const uint8_t *)&chan0, 0, 1, &scramb_init);
}

struct tgpu_prof {
	uint32_t max_steps;
	hipEvent_t *ev;		/* (TGPU_NSTAGES + 1) per step */
};

int tgpu_prof_create(uint32_t max_steps, struct tgpu_prof **out)
{
	if (!out || !max_steps)
		return TGPU_EINVAL;
	struct tgpu_prof *pr = calloc(1, sizeof(*pr));
	if (!pr)
		return TGPU_ENOMEM;
	pr->max_steps = max_steps;
	const size_t n = (size_t)max_steps * (TGPU_NSTAGES + 1);
	pr->ev = calloc(n, sizeof(hipEvent_t));
	if (!pr->ev) {
		free(pr);
		return TGPU_ENOMEM;
	}
	for (size_t i = 0; i < n; i++) {
		hipError_t e = hipEventCreate(&pr->ev[i]);
		if (e != hipSuccess) {
			tgpu_prof_destroy(pr);
			return (int)e;
		}
	}
	*out = pr;
	return TGPU_OK;
}

void tgpu_prof_destroy(struct tgpu_prof *pr)
{
	if (!pr)
		return;
	if (pr->ev) {
		const size_t n = (size_t)pr->max_steps * (TGPU_NSTAGES + 1);
		for (size_t i = 0; i < n; i++)
			if (pr->ev[i])
				(void)hipEventDestroy(pr->ev[i]);
		free(pr->ev);
	}
	free(pr);
}

const char *tgpu_stage_name(int stage)
{
	static const char *const names[TGPU_NSTAGES] = { "k_front", "k_vit<SB1>", "k_fill", "k_masks", "k_vit<216>", "k_vit<432>" };
	return (stage >= 0 && stage < TGPU_NSTAGES) ? names[stage] : "?";
}

static uint32_t tgpi_burst_max(void)
{
	return (uint32_t)tgi_option(TGPU_OPT_BURST_MAX);
}

/* the traffic stage on the batch the plan holds (its packed slots, mask entries and item lists) */
static int plan_traffic(struct tgpu_plan *p, const uint8_t *d_traffic, uint8_t *d_rec, uint8_t *d_type4, int16_t *d_blocks,
			uint16_t *d_lens, void *stream)
{
	if (p->block_mode || p->last_burst)
		return TGPU_ESTATE;
	const int wo = p->wire_only && p->d_wire;
	return tgk_traffic(p->d_list_432, p->n432, p->d_list_216, p->n216, p->d_counts, d_traffic, p->d_packed, p->d_masks, p->d_maskidx,
			   wo ? NULL : d_rec, p->d_wire, p->nslots, d_type4, d_blocks, d_lens, stream);
}

int tgpu_plan_set_traffic(struct tgpu_plan *p, const uint8_t *d_traffic, uint8_t *d_type4, int16_t *d_blocks, uint16_t *d_lens)
{
	if (!p || (d_traffic && !d_lens) || ((uintptr_t)d_type4 & 3) || ((uintptr_t)d_blocks & 3) || ((uintptr_t)d_lens & 1))
		return TGPU_EINVAL;
	p->d_traffic = d_traffic;
	p->d_type4 = d_traffic ? d_type4 : NULL;
	p->d_tblocks = d_traffic ? d_blocks : NULL;
	p->d_tlens = d_traffic ? d_lens : NULL;
	return TGPU_OK;
}

int tgpu_plan_traffic(struct tgpu_plan *p, const uint8_t *d_traffic, uint8_t *d_rec, uint8_t *d_type4, int16_t *d_blocks,
		      uint16_t *d_lens, void *stream)
{
	if (!p || !d_traffic || !d_lens || ((uintptr_t)d_type4 & 3) || ((uintptr_t)d_blocks & 3) || ((uintptr_t)d_lens & 1))
		return TGPU_EINVAL;
	if (!p->loaded)
		return TGPU_ESTATE;
	if (!d_rec && !(p->wire_only && p->d_wire))
		return TGPU_EINVAL;
	BIND(p->eng);
	if (!p->nslots)
		return TGPU_OK;
	return plan_traffic(p, d_traffic, d_rec, d_type4, d_blocks, d_lens, stream);
}

/* soft: 0 = bits (1 per byte), 1 = int8 soft values, 2 = float phases (nfloats of them) */
static int plan_run(struct tgpu_plan *p, const uint8_t *d_stream, uint8_t *d_rec, void *stream, hipEvent_t *ev, int soft,
		    uint64_t nfloats)
{
	int rc;
#define MARK(i) do { if (ev) { hipError_t e_ = hipEventRecord(ev[i], (hipStream_t)stream); if (e_ != hipSuccess) return (int)e_; } } while (0)
	if (!p || !d_stream || !d_rec)
		return TGPU_EINVAL;
	BIND(p->eng);
	if (!p->loaded || (soft && p->packed_ready) || p->block_mode)
		return TGPU_ESTATE;
	if (p->early_pending) {		/* (TGPU_OPT_SLOT 3: the early trellises ran beside the walk on the plan's own stream; whatever decodes now --
					 * the slots the look-back hands back, or the whole batch again after a host walk -- writes the same records) */
		HCHK(hipStreamWaitEvent((hipStream_t)stream, p->ev_early1, 0));
		p->early_pending = 0;
	}
	/* float input: slot offsets count stream positions, two per symbol; a slot that starts past the input would make
	 * the kernel's clamp arithmetic wrap */
	if (soft == 2 && p->nslots && p->max_off + TG_SLOT_BITS > 2 * nfloats)
		return TGPU_EINVAL;
	/* small batches: one workgroup per burst, trellis states across lanes, two launches (k_burst).
	 * TGPU_BURST_MAX = largest batch that takes this path (0 = never) */
	if (!soft && !ev && !p->packed_ready && !p->rm_decode && !p->d_wire && !p->fastpath && !p->d_traffic && p->nslots &&
	    p->nslots <= tgpi_burst_max()) {
		if ((rc = tgk_burst(d_stream, p->d_slot_off, p->d_slot_chan, p->d_chan_code, p->nslots, p->nchan, p->nsb != 0, p->d_sb_ok,
				    p->d_sb_code, d_rec, p->d_maskidx, p->d_masks, p->marks, stream)))
			return rc;
		/* k_burst uses d_maskidx / d_masks as its own scratch (entry i = the code in force at slot i): a later execute
		 * of the same load on the batch kernels' path has to rebuild the static mask table and indices first */
		p->static_pending = p->static_masks;
		p->last_burst = 1;
		return TGPU_OK;
	}
	p->last_burst = 0;
	if (p->static_pending && p->nslots) {
		HCHK(hipMemcpyAsync(p->d_maskidx, p->d_idx_stage, (size_t)p->nslots * 4,
				    hipMemcpyDefault, (hipStream_t)stream));
		if ((rc = tgk_masks(p->d_chan_code, p->nchan, p->d_sb_ok, p->d_sb_code, 0, NULL, NULL, p->d_masks, stream)))
			return rc;
		p->static_pending = 0;
	}
	MARK(0);
	if (p->nslots) {
		if (soft) {
			if (!p->d_softarea) {
				hipError_t e_ = hipMalloc((void **)&p->d_softarea, (size_t)p->max_slots * TG_SOFT_SLOT_BYTES);
				if (e_ != hipSuccess)
					return (int)e_;
			}
			if (soft == 2)
				rc = tgk_front_soft_f32((const float *)d_stream, nfloats, p->d_slot_off, p->nslots, p->d_softarea,
							p->d_packed, d_rec, stream);
			else
				rc = tgk_front_soft((const int8_t *)d_stream, p->d_slot_off, p->nslots, p->d_softarea, p->d_packed,
						    d_rec, stream);
			if (rc)
				return rc;
		} else if (!p->packed_ready &&	/* stream mode: k_front_stream has already packed every grid slot */
			   (rc = tgk_front(d_stream, p->d_slot_off, p->nslots, p->d_packed, d_rec, stream)))
			return rc;
	}
	MARK(1);
	if (p->nslots && !p->static_masks && !p->dev_mid) {
		if ((rc = tgk_vit(TG_KIND_SB1, p->d_list_sb, p->nsb, p->d_packed, p->d_masks, p->d_maskidx, d_rec,
				  p->d_sb_ok, p->d_sb_code, p->d_wire, soft ? p->d_softarea : NULL, (p->rm_decode ? TGK_F_RM : 0) | (p->wire_only && p->d_wire ? TGK_F_WIREONLY : 0),
				  p->d_counts, stream)))
			return rc;
	}
	MARK(2);
	if (p->nslots && !p->static_masks && !p->dev_mid) {
		if ((rc = tgk_fill(p->d_slot_chan, p->d_slot_sbord, p->d_sb_ok, p->d_sb_code, p->d_list_sb, p->nchan, p->nslots, p->d_block_tmp,
				   p->d_maskidx, stream)))
			return rc;
	}
	MARK(3);
	if (p->nslots && !p->static_masks && !p->dev_mid) {
		if ((rc = tgk_masks_dev(p->d_chan_code, p->nchan, p->d_sb_ok, p->d_sb_code, p->nsb, p->d_counts, p->d_list_sb, p->d_slot_chan,
					p->d_masks, stream)))
			return rc;
	}
	MARK(4);
	/* the two trellis kernels do not depend on each other: unless per-stage timing was asked for,
	 * k_vit<432> goes to a side stream (fork/join with events, still capturable) so that the tails
	 * of the two launches overlap */
	const int fork = (ev == NULL) && !p->no_side && p->n216 && p->n432 && p->nslots > 4096 &&	/* (a small batch gains nothing from the side stream) */
			 !(tgi_option(TGPU_OPT_SLOT) && !soft && !(p->fastpath && !p->d_counts) && !p->rm_decode && p->d_list_all && (p->nall || p->nsync));
	if (fork) {
		hipError_t e_ = hipEventRecord(p->ev_fork, (hipStream_t)stream);
		if (e_ == hipSuccess)
			e_ = hipStreamWaitEvent(p->side, p->ev_fork, 0);
		if (e_ != hipSuccess)
			return (int)e_;
	}
	const int kf = (p->rm_decode ? TGK_F_RM : 0) | (p->wire_only && p->d_wire ? TGK_F_WIREONLY : 0);
	const int fast = p->fastpath && !soft && !p->d_counts;	/* (the clean-block pre-pass wants host-side counts) */
	const uint32_t *items216 = p->d_list_216, *items432 = p->d_list_432;
	const uint32_t *cnt216 = p->d_counts ? p->d_counts + 1 : NULL, *cnt432 = p->d_counts ? p->d_counts + 2 : NULL;
	void *s432 = fork ? (void *)p->side : stream;
	if (fast && p->nslots) {
		/* blocks that are code words are finished by k_clean; the trellis kernels get the rest, counted on the device */
		uint32_t *d216 = p->d_dirty + 2, *d432 = d216 + 2 * (size_t)p->max_slots;
		HCHK(hipMemsetAsync(p->d_dirty, 0, 8, (hipStream_t)stream));
		if (fork) {	/* the counters are cleared on the caller's stream: order the side stream behind that */
			HCHK(hipEventRecord(p->ev_fork, (hipStream_t)stream));
			HCHK(hipStreamWaitEvent(p->side, p->ev_fork, 0));
		}
		if ((rc = tgk_clean(TG_KIND_216, p->d_list_216, p->n216, p->d_packed, p->d_masks, p->d_maskidx, d_rec, p->d_sb_ok,
				    p->d_sb_code, p->d_wire, d216, p->d_dirty, kf, stream)))
			return rc;
		if ((rc = tgk_clean(TG_KIND_432, p->d_list_432, p->n432, p->d_packed, p->d_masks, p->d_maskidx, d_rec, p->d_sb_ok,
				    p->d_sb_code, p->d_wire, d432, p->d_dirty + 1, kf, s432)))
			return rc;
		items216 = d216;
		items432 = d432;
		cnt216 = p->d_dirty;
		cnt432 = p->d_dirty + 1;
	}
	/* round 6: one lane per SLOT (k_slot_t) in place of the two launches -- hard input, slot or device-walk batches (the ones that
	 * have the slot list), no option that only the lane-per-block kernels implement */
	const int by_slot = tgi_option(TGPU_OPT_SLOT) && !soft && !fast && !p->rm_decode && p->d_list_all && (p->nall || p->nsync);
	if (by_slot) {
		/* one launch over two lists: the NORM_1 / NORM_2 slots (no SYNC lane: their prologue and selects compiled away) and the SYNC slots
		 * (every lane one: the schedule starts at block slot 8) */
		if ((rc = tgk_slot_t(p->d_list_all, p->nall, p->d_counts ? p->d_counts + 3 : NULL, p->d_list_sync, p->nsync,
				     p->d_counts ? p->d_counts + 4 : NULL, p->d_counts ? p->nslots : 0, p->d_packed, p->d_masks, p->d_maskidx, d_rec,
				     p->d_wire, kf, stream)))
			return rc;
		MARK(5);
	} else {
	if (p->nslots) {
		if ((rc = tgk_vit(TG_KIND_216, items216, p->n216, p->d_packed, p->d_masks, p->d_maskidx, d_rec,
				  p->d_sb_ok, p->d_sb_code, p->d_wire, soft ? p->d_softarea : NULL, kf, cnt216, stream)))
			return rc;
	}
	MARK(5);
	if (p->nslots) {
		if ((rc = tgk_vit(TG_KIND_432, items432, p->n432, p->d_packed, p->d_masks, p->d_maskidx, d_rec,
				  p->d_sb_ok, p->d_sb_code, p->d_wire, soft ? p->d_softarea : NULL, kf, cnt432, s432)))
			return rc;
	}
	}
	if (fork) {
		hipError_t e_ = hipEventRecord(p->ev_join, p->side);
		if (e_ == hipSuccess)
			e_ = hipStreamWaitEvent((hipStream_t)stream, p->ev_join, 0);
		if (e_ != hipSuccess)
			return (int)e_;
	}
	MARK(6);
#undef MARK
	/* traffic bursts the caller marked (tgpu_plan_set_traffic): their SCH/F / second blocks as type-4 bits and dump blocks,
	 * their records marked -- behind both trellis kernels, in front of whatever packs the wire records */
	if (p->d_traffic && p->nslots)
		return plan_traffic(p, p->d_traffic, d_rec, p->d_type4, p->d_tblocks, p->d_tlens, stream);
	return TGPU_OK;
}

int tgpu_plan_set_fastpath(struct tgpu_plan *p, int on)
{
	if (!p)
		return TGPU_EINVAL;
	BIND(p->eng);
	if (on && !p->d_dirty) {
		hipError_t e = hipMalloc((void **)&p->d_dirty, 4 * (2 + 3 * (size_t)p->max_slots));
		if (e != hipSuccess)
			return (int)e;
	}
	p->fastpath = on != 0;
	return TGPU_OK;
}

int tgpu_plan_set_rm_decode(struct tgpu_plan *p, int on)
{
	if (!p)
		return TGPU_EINVAL;
	BIND(p->eng);
	if (on) {
		const uint32_t *t = tgi_rm_leader_table();
		if (!t)
			return TGPU_ENOMEM;
		int rc = tgk_rm_enable(t, tgi_rm_parity());
		if (rc)
			return rc;
	}
	p->rm_decode = on != 0;
	return TGPU_OK;
}

/* ---- block mode: items are type-5 blocks on their own (the unit tp_sap_udata_ind() receives) ---- */
static int cmp_u32(const void *a, const void *b)
{
	const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
	return x < y ? -1 : x > y;
}

int tgpu_plan_load_blocks(struct tgpu_plan *p, uint32_t nblocks, const uint64_t *blk_off, const uint8_t *blk_type,
			  const uint32_t *blk_code)
{
	if (!p || (nblocks && (!blk_off || !blk_type || !blk_code)))
		return TGPU_EINVAL;
	BIND(p->eng);
	if (nblocks > p->max_slots)
		return TGPU_ECAPACITY;
	/* the distinct scrambling codes become the mask-table entries 1 + u (what channels are in slot mode) */
	uint32_t *uniq = malloc(((size_t)nblocks + 1) * 4);
	if (!uniq)
		return TGPU_ENOMEM;
	memcpy(uniq, blk_code, (size_t)nblocks * 4);
	qsort(uniq, nblocks, 4, cmp_u32);
	uint32_t nu = 0;
	for (uint32_t i = 0; i < nblocks; i++)
		if (!nu || uniq[nu - 1] != uniq[i])
			uniq[nu++] = uniq[i];
	if (!nu)
		uniq[nu++] = 0;
	if (nu > p->max_chan) {
		free(uniq);
		return TGPU_ECAPACITY;
	}
	uint32_t cnt[TGPU_NKINDS + 1] = { 0 };
	for (uint32_t i = 0; i < nblocks; i++) {
		int x;
		switch (blk_type[i]) {
		case TPSAP_T_SB1: x = TG_KIND_SB1; break;
		case TPSAP_T_SB2: case TPSAP_T_NDB: x = TG_KIND_216; break;
		case TPSAP_T_SCH_F: x = TG_KIND_432; break;
		case TPSAP_T_SCH_HU: x = TG_KIND_168; break;
		case TPSAP_T_BBK: x = TGPU_NKINDS; break;
		default:
			free(uniq);
			return TGPU_EINVAL;
		}
		if (blk_off[i] >> 48) {
			free(uniq);
			return TGPU_EINVAL;
		}
		cnt[x]++;
	}
	size_t o = 0;
#define UP_PLACE(dptr, hptr, type, count) do { dptr = (type *)(p->d_up + o); hptr = (type *)(p->h_up + o); \
		o = (o + (size_t)(count) * sizeof(type) + UP_ALIGN - 1) & ~(size_t)(UP_ALIGN - 1); } while (0)
	uint64_t *h_desc;
	uint32_t *h_idx, *d_idx, *h_code, *h_list[TGPU_NKINDS + 1], *d_list[TGPU_NKINDS + 1];
	UP_PLACE(p->d_slot_off, h_desc, uint64_t, nblocks);
	UP_PLACE(d_idx, h_idx, uint32_t, nblocks);
	for (int x = 0; x <= TGPU_NKINDS; x++)
		UP_PLACE(d_list[x], h_list[x], uint32_t, cnt[x]);
	UP_PLACE(p->d_chan_code, h_code, uint32_t, nu);
#undef UP_PLACE
	if (o > p->up_bytes) {
		free(uniq);
		return TGPU_ECAPACITY;
	}
	uint32_t fill[TGPU_NKINDS + 1] = { 0 };
	for (uint32_t i = 0; i < nblocks; i++) {
		int x;
		switch (blk_type[i]) {
		case TPSAP_T_SB1: x = TG_KIND_SB1; break;
		case TPSAP_T_SB2: case TPSAP_T_NDB: x = TG_KIND_216; break;
		case TPSAP_T_SCH_F: x = TG_KIND_432; break;
		case TPSAP_T_SCH_HU: x = TG_KIND_168; break;
		default: x = TGPU_NKINDS; break;
		}
		h_desc[i] = blk_off[i] | ((uint64_t)x << 56) | ((uint64_t)blk_type[i] << 48);
		h_list[x][fill[x]++] = (x == TG_KIND_216) ? (i << 1) : i;	/* k_vit<216> items carry a block-select bit */
		const uint32_t *hit = bsearch(&blk_code[i], uniq, nu, 4, cmp_u32);
		h_idx[i] = 1 + (uint32_t)(hit - uniq);
	}
	memcpy(h_code, uniq, (size_t)nu * 4);
	free(uniq);
	if (!p->up_mapped)
		HCHK(hipMemcpy(p->d_up, p->h_up, o, hipMemcpyHostToDevice));
	HCHK(hipMemcpyAsync(p->d_maskidx, d_idx, (size_t)nblocks * 4, hipMemcpyDefault, NULL));
	int rc = tgk_masks(p->d_chan_code, nu, p->d_sb_ok, p->d_sb_code, 0, NULL, NULL, p->d_masks, NULL);
	if (rc)
		return rc;
	HCHK(hipDeviceSynchronize());
	p->d_list_sb = d_list[TG_KIND_SB1];
	p->d_list_216 = d_list[TG_KIND_216];
	p->d_list_432 = d_list[TG_KIND_432];
	p->d_list_all = p->d_list_sync = NULL;
	p->nall = p->nsync = 0;
	p->d_list_168 = d_list[TG_KIND_168];
	p->d_list_bbk = d_list[TGPU_NKINDS];
	p->nsb = cnt[TG_KIND_SB1];
	p->n216 = cnt[TG_KIND_216];
	p->n432 = cnt[TG_KIND_432];
	p->n168 = cnt[TG_KIND_168];
	p->nbbk = cnt[TGPU_NKINDS];
	p->nslots = nblocks;
	p->nchan = nu;
	p->static_masks = 1;
	p->static_pending = 0;
	p->d_counts = NULL;
	p->dev_mid = 0;
	p->have_final = 0;
	p->packed_ready = 0;
	p->block_mode = 1;
	p->loaded = 1;
	return TGPU_OK;
}

static int plan_run_blocks(struct tgpu_plan *p, const uint8_t *d_bits, uint8_t *d_rec, void *stream)
{
	int rc;
	BIND(p->eng);
	if (!p->nslots)
		return TGPU_OK;
	if ((rc = tgk_front_blocks(d_bits, p->d_slot_off, p->nslots, p->d_packed, stream)))
		return rc;
	const struct { int kind; const uint32_t *list; uint32_t n; } run[4] = {
		{ TG_KIND_432, p->d_list_432, p->n432 }, { TG_KIND_216, p->d_list_216, p->n216 },
		{ TG_KIND_168, p->d_list_168, p->n168 }, { TG_KIND_SB1, p->d_list_sb, p->nsb } };
	for (int i = 0; i < 4; i++)
		if ((rc = tgk_vit(run[i].kind, run[i].list, run[i].n, p->d_packed, p->d_masks, p->d_maskidx, d_rec, p->d_sb_ok,
				  p->d_sb_code, NULL, NULL, TGK_F_BLOCK, NULL, stream)))
			return rc;
	return tgk_bbk_blocks(p->d_list_bbk, p->nbbk, p->d_packed, p->d_masks, p->d_maskidx, d_rec,
			      (p->rm_decode ? TGK_F_RM : 0) | (p->wire_only && p->d_wire ? TGK_F_WIREONLY : 0), stream);
}

int tgpu_plan_execute(struct tgpu_plan *p, const uint8_t *d_stream, uint8_t *d_rec, void *stream)
{
	if (p && p->loaded && p->block_mode)
		return d_stream && d_rec ? plan_run_blocks(p, d_stream, d_rec, stream) : TGPU_EINVAL;
	return plan_run(p, d_stream, d_rec, stream, NULL, 0, 0);
}

int tgpu_plan_execute_soft(struct tgpu_plan *p, const int8_t *d_soft_stream, uint8_t *d_rec, void *stream)
{
	return plan_run(p, (const uint8_t *)d_soft_stream, d_rec, stream, NULL, 1, 0);
}

int tgpi_plan_last_burst(const struct tgpu_plan *p)
{
	return p && p->last_burst && p->marks;
}

/* the ring kernel (k_burst_ring) decodes batches in k_burst's place without a load: what it needs of the plan are the scratch
 * arrays k_burst uses (a later tgpu_plan_load() starts from scratch as always) */
int tgpi_plan_ring(struct tgpu_plan *p, uint32_t **sb_ok, uint32_t **sb_code, uint32_t **maskidx, uint32_t **masks)
{
	if (!p || p->rm_decode || p->d_wire || p->fastpath || p->block_mode)
		return TGPU_ESTATE;
	*sb_ok = p->d_sb_ok;
	*sb_code = p->d_sb_code;
	*maskidx = p->d_maskidx;
	*masks = p->d_masks;
	return TGPU_OK;
}

void tgpi_plan_set_marks(struct tgpu_plan *p, int on)
{
	if (p)
		p->marks = on;
}

int tgpu_plan_execute_float(struct tgpu_plan *p, const float *d_phi, uint64_t nfloats, uint8_t *d_rec, void *stream)
{
	if (!nfloats)
		return TGPU_EINVAL;
	return plan_run(p, (const uint8_t *)d_phi, d_rec, stream, NULL, 2, nfloats);
}

int tgpu_float_to_bits(struct tgpu_engine *eng, const float *d_in, uint64_t n, uint8_t *d_bits, int8_t *d_soft, void *stream)
{
	if (!eng || !d_in || !d_bits)
		return TGPU_EINVAL;
	BIND(eng);
	return tgk_float_to_bits(d_in, n, d_bits, d_soft, stream);
}

int tgpu_float_to_bits_afc(struct tgpu_engine *eng, const float *d_in, uint64_t n, uint8_t *d_bits, float filter_val,
			   float filter_goal, float *filter_state, void *stream)
{
	if (!eng || !d_in || !d_bits || !filter_state)
		return TGPU_EINVAL;
	BIND(eng);
	float *d_state = NULL;
	HCHK(hipMalloc((void **)&d_state, sizeof(float)));
	int rc = (int)hipMemcpyAsync(d_state, filter_state, sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
	if (!rc)
		rc = tgk_float_to_bits_afc(d_in, n, d_bits, filter_val, filter_goal, d_state, stream);
	if (!rc)
		rc = (int)hipMemcpyAsync(filter_state, d_state, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream);
	if (!rc)
		rc = (int)hipStreamSynchronize((hipStream_t)stream);
	(void)hipFree(d_state);
	return rc;
}

int tgpu_plan_execute_prof(struct tgpu_plan *p, const uint8_t *d_stream, uint8_t *d_rec, void *stream,
			   struct tgpu_prof *prof, uint32_t step)
{
	if (!prof || step >= prof->max_steps)
		return TGPU_EINVAL;
	return plan_run(p, d_stream, d_rec, stream, prof->ev + (size_t)step * (TGPU_NSTAGES + 1), 0, 0);
}

int tgpu_plan_execute_float_prof(struct tgpu_plan *p, const float *d_phi, uint64_t nfloats, uint8_t *d_rec, void *stream,
				 struct tgpu_prof *prof, uint32_t step)
{
	if (!prof || step >= prof->max_steps || !nfloats)
		return TGPU_EINVAL;
	return plan_run(p, (const uint8_t *)d_phi, d_rec, stream, prof->ev + (size_t)step * (TGPU_NSTAGES + 1), 2, nfloats);
}

int tgpu_prof_read(struct tgpu_prof *prof, uint32_t nsteps, float *ms)
{
	if (!prof || !ms || nsteps > prof->max_steps)
		return TGPU_EINVAL;
	for (uint32_t s = 0; s < nsteps; s++) {
		hipEvent_t *ev = prof->ev + (size_t)s * (TGPU_NSTAGES + 1);
		HCHK(hipEventSynchronize(ev[TGPU_NSTAGES]));
		for (int k = 0; k < TGPU_NSTAGES; k++)
			HCHK(hipEventElapsedTime(&ms[(size_t)s * TGPU_NSTAGES + k], ev[k], ev[k + 1]));
	}
	return TGPU_OK;
}

int tgpu_plan_set_side_stream(struct tgpu_plan *p, int on)
{
	if (!p)
		return TGPU_EINVAL;
	p->no_side = !on;
	return TGPU_OK;
}

int tgpu_plan_set_wire_only(struct tgpu_plan *p, int on)
{
	if (!p)
		return TGPU_EINVAL;
	p->wire_only = on ? 1 : 0;
	return TGPU_OK;
}

int tgpu_plan_set_cwire(struct tgpu_plan *p, uint8_t *d_cwire, size_t cap)
{
	if (!p || ((uintptr_t)d_cwire & 15) || (d_cwire && cap < 4096) || cap > 0xfffffff0u)
		return TGPU_EINVAL;
	p->d_cwire = d_cwire;
	p->cwire_cap = d_cwire ? cap : 0;
	return TGPU_OK;
}

/* device-walk batches (tg_stream.c): the compact form of the batch's wire records behind its decode, if asked for */
int tgpi_plan_cwire(struct tgpu_plan *p, const struct tg_cw_chans *ch, uint32_t *d_total, void *stream)
{
	if (!p || !ch)
		return TGPU_EINVAL;
	if (!p->d_cwire || !p->d_wire || !p->d_bits_dev || !p->nslots)
		return TGPU_OK;
	BIND(p->eng);
	/* the kernels write the header, the channel table, the bitmap and the block table without asking: a buffer that does not
	 * hold those (plus one record row) is not touched at all -- the batch reports the shortfall like any other (total 0 = not
	 * computed, marker 0xffffffff: tgpu_sync_multi_collect() returns TGPU_OK with cwire_bytes 0 and cwire_needed set -- the batch
	 * itself is valid, it has no compact form; include/tetra_gpu.h, tgpu_sync_dev_cwire_bytes).  The two words are written from
	 * the device side (two fills in stream order): a copy from pageable host memory goes through the runtime's staging path and
	 * can hold the calling thread until the stream drains. */
	struct tg_cw_layout L;
	tg_cw_offsets(ch->n, p->nslots, &L);
	if (p->cwire_cap < (size_t)L.o_rec + 16) {
		if (!d_total)
			return TGPU_ECAPACITY;
		HCHK(hipMemsetD32Async((hipDeviceptr_t)d_total, 0, 1, (hipStream_t)stream));
		HCHK(hipMemsetD32Async((hipDeviceptr_t)(d_total + 1), (int)0xffffffffu, 1, (hipStream_t)stream));
		return TGPU_OK;
	}
	return tgk_cwire(p->d_wire, p->d_bits_dev, p->nslots, ch, p->d_cwire, (uint32_t)p->cwire_cap, d_total, stream);
}

int tgpi_plan_is_early(const struct tgpu_plan *p)
{
	return p && p->early;
}

int tgpi_plan_has_cwire(const struct tgpu_plan *p)
{
	return p && p->d_cwire && p->d_wire;
}

int tgpu_plan_set_wire(struct tgpu_plan *p, uint8_t *d_wire)
{
	if (!p)
		return TGPU_EINVAL;
	p->d_wire = d_wire;
	return TGPU_OK;
}

/* wire record -> full record (host), inverse of the trellis kernels' packing */
static void unpack_bits(const uint8_t *src, int nbits, uint8_t *dst)
{
	for (int i = 0; i < nbits; i++)
		dst[i] = (src[i >> 3] >> (i & 7)) & 1;
}

static void pack_bits(const uint8_t *src, int nbits, uint32_t *dst)
{
	for (int i = 0; i < nbits; i++)
		if (src[i])
			dst[i >> 5] |= 1u << (i & 31);
}

/* full record -> wire record (host): what the trellis kernels write next to the record when a wire buffer is set */
int tgpu_wire_pack(const uint8_t *rec, uint8_t *wire)
{
	if (!rec || !wire)
		return TGPU_EINVAL;
	uint32_t w[TG_WIRE_WORDS];
	uint16_t crc[2];
	memset(w, 0, sizeof(w));
	memcpy(crc, rec + TG_REC_CRC, 4);
	const uint8_t type = rec[TG_REC_TYPE];
	uint32_t bbk = 0;
	pack_bits(rec + TG_REC_BBK, 14, &bbk);
	w[0] = type | ((uint32_t)rec[TG_REC_FLAGS] << 8) | (bbk << 16);
	switch (type) {
	case TETRA_TRAIN_NORM_1:
		pack_bits(rec + TG_REC_BITS1, 268, w + TG_WIRE_W_BITS1);
		w[TG_WIRE_W_CRC] |= (uint32_t)crc[0] << TG_WIRE_SCHF_CRC_SHIFT;
		break;
	case TETRA_TRAIN_NORM_2:
	case TETRA_TRAIN_SYNC:
		pack_bits(rec + TG_REC_BITS1, type == TETRA_TRAIN_SYNC ? 60 : 124, w + TG_WIRE_W_BITS1);
		pack_bits(rec + TG_REC_BITS2, 124, w + TG_WIRE_W_BITS2);
		w[TG_WIRE_W_CRC] = crc[0] | ((uint32_t)crc[1] << 16);
		break;
	default:
		memset(w, 0xff, sizeof(w));
		break;
	}
	memcpy(wire, w, sizeof(w));
	return TGPU_OK;
}

int tgpu_wire_unpack(const uint8_t *wire, uint32_t slot_id, uint32_t scrambling_code, uint8_t *rec)
{
	if (!wire || !rec)
		return TGPU_EINVAL;
	uint32_t w[TG_WIRE_WORDS];
	memcpy(w, wire, sizeof(w));
	memset(rec, 0, TG_REC_BYTES);
	const uint8_t type = (uint8_t)w[0];
	rec[TG_REC_TYPE] = type;
	if (type != TETRA_TRAIN_NORM_1 && type != TETRA_TRAIN_NORM_2 && type != TETRA_TRAIN_SYNC) {
		rec[TG_REC_TYPE] = TG_BURST_NONE;
		return TGPU_OK;
	}
	rec[TG_REC_FLAGS] = (uint8_t)(w[0] >> 8);
	memcpy(rec + TG_REC_CODE, &scrambling_code, 4);
	memcpy(rec + TG_REC_SLOT, &slot_id, 4);
	const uint32_t bbk = w[0] >> 16;
	unpack_bits((const uint8_t *)&bbk, 14, rec + TG_REC_BBK);
	uint16_t crc[2] = { 0, 0 };
	if (type == TETRA_TRAIN_NORM_1) {
		unpack_bits((const uint8_t *)(w + TG_WIRE_W_BITS1), 268, rec + TG_REC_BITS1);
		crc[0] = (uint16_t)(w[TG_WIRE_W_CRC] >> TG_WIRE_SCHF_CRC_SHIFT);
		rec[TG_REC_CRC_OK] = crc[0] == TG_CRC_OK;
	} else {
		unpack_bits((const uint8_t *)(w + TG_WIRE_W_BITS1), type == TETRA_TRAIN_SYNC ? 60 : 124, rec + TG_REC_BITS1);
		unpack_bits((const uint8_t *)(w + TG_WIRE_W_BITS2), 124, rec + TG_REC_BITS2);
		crc[0] = (uint16_t)w[TG_WIRE_W_CRC];
		crc[1] = (uint16_t)(w[TG_WIRE_W_CRC] >> 16);
		rec[TG_REC_CRC_OK] = crc[0] == TG_CRC_OK;
		rec[TG_REC_CRC_OK + 1] = crc[1] == TG_CRC_OK;
	}
	memcpy(rec + TG_REC_CRC, crc, 4);
	if (rec[TG_REC_FLAGS] & TG_FLAG_TRAFFIC)	/* the block that went to the traffic dump was not indicated (tg_traffic.hip) */
		rec[TG_REC_CRC_OK + (type == TETRA_TRAIN_NORM_1 ? 0 : 1)] = 0;
	if (type == TETRA_TRAIN_SYNC) {
		/* SYNC-PDU fields (lower_mac/tetra_lower_mac.c:284-297) from the SB1 bits */
		const uint8_t *b = rec + TG_REC_BITS1;
		uint32_t f[6];
		static const int pos[6][2] = { { 4, 6 }, { 10, 2 }, { 12, 5 }, { 17, 6 }, { 31, 10 }, { 41, 14 } };
		for (int k = 0; k < 6; k++) {
			f[k] = 0;
			for (int i = 0; i < pos[k][1]; i++)
				f[k] = (f[k] << 1) | b[pos[k][0] + i];
		}
		uint32_t f0 = f[0] | ((f[1] + 1) << 8) | (f[2] << 16) | (f[3] << 24), f1 = f[4] | (f[5] << 16);
		uint32_t code = ((((f[4] & 0x3ff) << 20) | ((f[5] & 0x3fff) << 6) | (f[0] & 0x3f)) << 2) | 3u;
		memcpy(rec + TG_REC_SBF0, &f0, 4);
		memcpy(rec + TG_REC_SBF1, &f1, 4);
		memcpy(rec + TG_REC_SBCODE, &code, 4);
	}
	return TGPU_OK;
}

int tgpu_plan_read_packed(struct tgpu_plan *p, uint32_t *out_words)
{
	if (!p || !out_words || !p->loaded)
		return TGPU_EINVAL;
	BIND(p->eng);
	HCHK(hipDeviceSynchronize());
	if (p->nslots)
		HCHK(hipMemcpy(out_words, p->d_packed, (size_t)p->nslots * TG_PACKED_WORDS * 4, hipMemcpyDeviceToHost));
	return TGPU_OK;
}

int tgpu_plan_final_codes(struct tgpu_plan *p, const uint8_t *d_rec, uint32_t *chan_code_out)
{
	if (!p || !chan_code_out || !p->loaded)
		return TGPU_EINVAL;
	BIND(p->eng);
	(void)d_rec;
	if (p->dev_mid) {	/* a device-walk batch: k_lb_scan computed them, tgpu_sync_multi_collect() brought them over */
		if (!p->have_final)
			return TGPU_ESTATE;
		memcpy(chan_code_out, p->h_final_code, (size_t)p->nchan * 4);
		return TGPU_OK;
	}
	/* code in effect after the batch = mask entry of the channel's last slot */
	HCHK(hipDeviceSynchronize());
	for (uint32_t c = 0; c < p->nchan; c++) {
		uint32_t last = p->h_last_slot_of_chan[c], idx, code;
		if (last == 0xffffffffu) {
			HCHK(hipMemcpy(&code, p->d_chan_code + c, 4, hipMemcpyDeviceToHost));
		} else {
			HCHK(hipMemcpy(&idx, p->d_maskidx + last, 4, hipMemcpyDeviceToHost));
			HCHK(hipMemcpy(&code, p->d_masks + (size_t)idx * TG_MASK_WORDS + TG_MW_CODE, 4, hipMemcpyDeviceToHost));
		}
		chan_code_out[c] = code;
	}
	return TGPU_OK;
}

/* ------------------------------------------------------------------------- */
/* record parsing (host)                                                      */
/* ------------------------------------------------------------------------- */
static void fill_block(struct tgpu_block *b, const uint8_t *rec, enum tp_sap_data_type t, int blk_num, int which,
		       uint16_t n1, uint32_t code)
{
	b->type = t;
	b->blk_num = blk_num;
	b->crc_ok = rec[TG_REC_CRC_OK + which];
	memcpy(&b->crc, rec + TG_REC_CRC + 2 * which, 2);
	b->scrambling_code = code;
	b->type1_len = n1;
	b->type1 = rec + (which ? TG_REC_BITS2 : TG_REC_BITS1);
}

int tgpu_record_blocks(const uint8_t *rec, struct tgpu_block out[3])
{
	uint32_t code;
	memcpy(&code, rec + TG_REC_CODE, 4);
	struct tgpu_block bbk = { TPSAP_T_BBK, 0, 1, 0, code, 14, rec + TG_REC_BBK };	/* crc_ok=1: tetra_lower_mac.c:270 */
	switch (rec[TG_REC_TYPE]) {
	case TETRA_TRAIN_SYNC:		/* phy/tetra_burst.c:350-352 */
		fill_block(&out[0], rec, TPSAP_T_SB1, BLK_1, 0, 60, SCRAMB_INIT);
		out[1] = bbk;
		fill_block(&out[2], rec, TPSAP_T_SB2, BLK_2, 1, 124, code);
		return 3;
	case TETRA_TRAIN_NORM_2:	/* :359-361 */
		out[0] = bbk;
		fill_block(&out[1], rec, TPSAP_T_NDB, BLK_1, 0, 124, code);
		fill_block(&out[2], rec, TPSAP_T_NDB, BLK_2, 1, 124, code);
		return 3;
	case TETRA_TRAIN_NORM_1:	/* :371-372 */
		out[0] = bbk;
		fill_block(&out[1], rec, TPSAP_T_SCH_F, 0, 0, 268, code);
		return 2;
	default:
		return 0;
	}
}

int tgpu_record_sync_info(const uint8_t *rec, struct tgpu_sync_info *out)
{
	if (!rec || !out || rec[TG_REC_TYPE] != TETRA_TRAIN_SYNC)
		return TGPU_EINVAL;
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
	return TGPU_OK;
}
