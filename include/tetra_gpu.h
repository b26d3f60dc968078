/*
 * tetra_gpu.h -- C ABI of the MI355X-native TETRA lower-MAC receive path.
 *
 * Drop-in boundary for osmocom/osmo-tetra's PHY + lower MAC (reference paths are
 * relative to its src/ directory).  Plain C types only; device buffers are raw
 * device pointers, the stream argument is a hipStream_t passed as void*.
 *
 * Three levels, lowest first:
 *
 *  1. plan API (device resident, what bench.py times):
 *       tgpu_plan_*        slots of a batch (offset, burst type, channel) -> one
 *                          320-byte record per slot in HBM.  Replaces the arithmetic
 *                          of tetra_burst_rx_cb() (phy/tetra_burst.c:341-379) and
 *                          tp_sap_udata_ind() (lower_mac/tetra_lower_mac.c:143-357)
 *                          for many bursts at once.  tgpu_plan_set_traffic() /
 *                          tgpu_plan_traffic(): the traffic-channel branch of
 *                          tp_sap_udata_ind() (:194-241) for a batch.
 *  2. channel API (host buffers in, callbacks out):
 *       tetra_burst_sync_in()   same symbol, same struct tetra_rx_state layout and
 *                          same return values as phy/tetra_burst_sync.c:54-154.
 *                          Bursts are queued, decoded on the GPU in batches and
 *                          delivered, in the reference's order, to a callback that
 *                          has upper_mac_prim_recv()'s contract
 *                          (tetra_upper_mac.c:549-566).
 *  3. helpers: record parsing, synthetic burst generation for benchmarks/tests.
 *
 * Error convention: functions return 0 on success, a negative TGPU_E* code for
 * argument/state errors and a positive hipError_t value when the HIP runtime
 * failed.  tetra_burst_sync_in() keeps the reference's own return values.
 * There is no CPU fallback: without a usable GPU tgpu_engine_create() fails.
 */
#ifndef TETRA_GPU_H
#define TETRA_GPU_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* types mirrored from the reference (values and layouts are ABI)             */
/* ------------------------------------------------------------------------- */

/* A translation unit that includes the reference's own headers next to this one (the adapter of INTEGRATION.md section 3,
 * tools/tgpu_adapter.c) takes these types from there: every mirror below stands behind the include guard of the reference
 * header it restates, so include the reference's headers FIRST. */
#ifndef TETRA_BURST_H
/* phy/tetra_burst.h:6-7 */
#define BLK_1 1
#define BLK_2 2

/* phy/tetra_burst.h:9-16 */
enum tp_sap_data_type {
	TPSAP_T_SB1,
	TPSAP_T_SB2,
	TPSAP_T_NDB,
	TPSAP_T_BBK,
	TPSAP_T_SCH_HU,
	TPSAP_T_SCH_F,
};

/* phy/tetra_burst.h:30-36 */
enum tetra_train_seq {
	TETRA_TRAIN_NORM_1,
	TETRA_TRAIN_NORM_2,
	TETRA_TRAIN_NORM_3,
	TETRA_TRAIN_SYNC,
	TETRA_TRAIN_EXT,
};
#endif

#ifndef TETRA_COMMON_H
/* tetra_common.h:22-39 */
enum tetra_log_chan {
	TETRA_LC_UNKNOWN,
	TETRA_LC_SCH_F,
	TETRA_LC_SCH_HD,
	TETRA_LC_SCH_HU,
	TETRA_LC_STCH,
	TETRA_LC_SCH_P8_F,
	TETRA_LC_SCH_P8_HD,
	TETRA_LC_SCH_P8_HU,
	TETRA_LC_AACH,
	TETRA_LC_TCH,
	TETRA_LC_BSCH,
	TETRA_LC_BNCH,
};
#define TETRA_CRC_OK 0x1d0f	/* tetra_common.h:69 */
#endif

#ifndef TETRA_TDMA_H
/* tetra_tdma.h:6-12 */
struct tetra_tdma_time {
	uint16_t hn;
	uint32_t sn;
	uint32_t tn;
	uint32_t fn;
	uint32_t mn;
};
#endif

#ifndef TETRA_BURST_SYNC_H
/* phy/tetra_burst_sync.h:6-20 */
enum rx_state {
	RX_S_UNLOCKED,
	RX_S_KNOW_FSTART,
	RX_S_LOCKED,
};

struct tetra_rx_state {
	enum rx_state state;
	unsigned int bits_in_buf;
	uint8_t bitbuf[4096];
	unsigned int bitbuf_start_bitnum;
	unsigned int next_frame_start_bitnum;
	void *burst_cb_priv;	/* must hold the struct tgpu_channel* (see tgpu_channel_create) */
};
#endif

#ifndef TETRA_SCRAMB_H
#define SCRAMB_INIT  3		/* lower_mac/tetra_scramb.h:14 */
#endif

/* ------------------------------------------------------------------------- */
/* error codes                                                                */
/* ------------------------------------------------------------------------- */
#define TGPU_OK         0
#define TGPU_EINVAL    -1	/* bad argument */
#define TGPU_ENOMEM    -2
#define TGPU_ENODEV    -3	/* no usable GPU: there is no CPU fallback */
#define TGPU_ECAPACITY -4	/* more slots/channels than the plan was created for */
#define TGPU_ESTATE    -5	/* call order violated (e.g. execute before load) */
#define TGPU_ENOSYS    -6	/* an optional component is not available in this process (no RCCL library to load) */
#define TGPU_ECOMM     -7	/* the communication library reported an error */

const char *tgpu_strerror(int err);

/* ------------------------------------------------------------------------- */
/* engine: one per process / per GPU                                          */
/* ------------------------------------------------------------------------- */
struct tgpu_engine;
int tgpu_engine_create(struct tgpu_engine **out, int device);
void tgpu_engine_destroy(struct tgpu_engine *eng);
/*
 * Process-wide switches: test aids and documented alternatives of the device path.  They are set by these calls only --
 * nothing in the library reads the environment, so a production process's environment cannot change which kernels run.
 * Results are the same bytes whatever the setting (the GPU suite runs its tests under each of them).
 */
enum tgpu_option {
	TGPU_OPT_BURST_MAX = 1,		/* largest batch (slots) that takes the workgroup-per-burst kernel k_burst; default 1024, 0 = never */
	TGPU_OPT_STREAM_EXACT,		/* 1: the stream front end runs its per-position form on every grid slot (default 0: on the slots the
					 * packed-bit form cannot settle) */
	TGPU_OPT_WALK_HOST,		/* 1: tgpu_sync_multi_collect() redoes every device-walk batch through the host walks (default 0: only
					 * where the device walk hands a channel over) */
	TGPU_OPT_WALK_MONO,		/* 1: the device walk as one launch per form instead of three (node pass over the whole chip) */
	TGPU_OPT_FRONT_BLOCKS,		/* > 0: cap on the front-end kernels' workgroups (tests run their loops' tails with 1 and 2) */
	TGPU_OPT_WALK_WIDE,		/* 1: the device walk's per-channel launches as 1024 threads with 128 KB of LDS each (the form of rounds
					 * 3 and 4; default 0: 256 threads, LDS for the batch's longest channel and twice the nodes seen so far) */
	TGPU_OPT_RING,			/* 1 (default since round 6; 0 turns it off): channels created from now on with a batch size of up to 4
					 * bursts decode their flushes through workgroups that stay on the device and take requests from mapped host
					 * memory (k_burst_ring) instead of a kernel launch per flush; they leave after 20 ms without a request and
					 * come back with the next one; at most 32 channels of a process hold workgroups, the others flush by
					 * launch, and so does any flush the ring does not answer.  While the workgroups are there, calls that wait
					 * for the whole device (hipDeviceSynchronize(), hipFree()) wait for them too: up to those 20 ms behind the
					 * channel's last flush -- a process that cannot have that sets the option to 0 before creating channels */
	TGPU_OPT_SLOT,			/* how the plan API's batches run their trellises (round 6; records are the same bytes either way):
					 * 0: k_vit<216> and k_vit<432>, one lane per BLOCK (rounds 1-5);
					 * 1 (default): k_slot_t, one lane per SLOT -- the batch's NORM_1 / NORM_2 slots, then its SYNC slots (SB1
					 *    included), both blocks of a burst on one schedule, every record written as whole 64-byte segments;
					 * 2: as 1, and device-walk batches (tgpu_sync_multi_launch) whose channels have a scrambling code to
					 *    decode on -- the caller's carry-in code, else the code the plan's last batch of the channel ended with --
					 *    run the stream front end AND the trellises in one launch (k_slot): a wave packs and classifies 64
					 *    neighbouring grid slots, keeps them in LDS and decodes them there and then; after the walk and the code
					 *    look-back every delivered slot whose code in force is not the one it was decoded under (or that the exact
					 *    pass settled) goes through k_slot_t.  Records of UNDELIVERED grid slots are unspecified in that form (they
					 *    may hold a decode nobody asked for); tgpu_sync_dev_fused() tells which form a batch took.  Same records
					 *    for delivered slots; measured slower than 1 on the metric's workload, so not the default.
					 * 3: as 1, and device-walk batches whose channels have a code to decode on (as for 2) run the trellises of
					 *    EVERY plain grid slot right behind the front end (k_slot_e), on a stream of the plan's own beside the
					 *    walk and the code look-back; k_slot_t then re-decodes what the look-back finds was decoded under another
					 *    code.  Meant for a caller who waits for every batch -- and measured to buy little: 0.603 against 0.620 ms
					 *    per 1 M-slot batch (the walk's small kernels do not get far beside a kernel that fills the chip, DESIGN.md
					 *    section 4); it decodes the slots the walk drops as well, so several batches in flight run 5 % slower than
					 *    under 1.  Undelivered slots' records are unspecified, as for 2.
					 * (soft input, block mode, the RM(30,14) option, the clean-block fast path and the traffic stage keep the
					 * earlier forms) */
	TGPU_OPT__COUNT
};
int tgpu_engine_set_option(struct tgpu_engine *eng, int /* enum tgpu_option */ option, long value);
long tgpu_engine_get_option(const struct tgpu_engine *eng, int option);
/* where the GPU sits in the host: its PCI address ("0000:bb:dd.f"), the NUMA node it hangs off (-1 = unknown) and that
 * node's CPUs as the kernel prints them ("64-127,192-255"; "" = unknown), from hipDeviceGetPCIBusId() and
 * /sys/bus/pci/devices/<address>/{numa_node,local_cpulist}.  A process that feeds the GPU from host buffers wants its
 * threads -- and with them its pinned buffers -- on these CPUs: on a two-socket box the far socket costs a third of the
 * host-to-device rate.  The library never changes anybody's affinity itself.  bdf >= 16 bytes, cpulist: n bytes. */
int tgpu_device_host_locality(int device, char bdf[16], int *numa_node, char *cpulist, size_t n);

/* ------------------------------------------------------------------------- */
/* 1. plan API                                                                */
/* ------------------------------------------------------------------------- */
#define TGPU_REC_BYTES    320	/* output record per slot, layout below */
#define TGPU_SLOT_BYTES   510

struct tgpu_plan;

/* capacity: the largest batch (slots) and channel count the plan will be loaded with */
int tgpu_plan_create(struct tgpu_engine *eng, uint32_t max_slots, uint32_t max_chan, struct tgpu_plan **out);
void tgpu_plan_destroy(struct tgpu_plan *plan);

/*
 * Describe a batch (host arrays, copied).  slot_off[i]: byte offset of slot i in the
 * stream buffer later given to tgpu_plan_execute().  slot_type[i]: enum tetra_train_seq
 * (NORM_1, NORM_2 or SYNC; anything else = skip).  slot_chan[i]: channel index < nchan,
 * NON-DECREASING in i (slots of one channel are contiguous and in stream order).
 * chan_code[c]: scrambling code in effect for channel c before its first slot
 * (tcd->scramb_init, lower_mac/tetra_lower_mac.c:104-113; 0 for a fresh channel).
 * A plan must be idle when it is loaded: plans of up to 256 slots keep the batch description in mapped host memory
 * that the kernels of an execute still in flight read in place (no copy at load time), so loading batch n + 1 while
 * batch n runs needs a second plan (which is also how larger batches are pipelined).
 */
int tgpu_plan_load(struct tgpu_plan *plan, uint32_t nslots, const uint64_t *slot_off,
		   const uint8_t *slot_type, const uint32_t *slot_chan,
		   uint32_t nchan, const uint32_t *chan_code);

/*
 * Run the loaded batch: d_stream (device, 1 bit per byte) -> d_rec (device,
 * nslots * TGPU_REC_BYTES).  Asynchronous on 'hip_stream'; launches only, no
 * allocation and no host synchronisation (hipGraph-capturable).  Of a record the fields its burst type has are
 * written (tgpu_record_blocks() reads only those): the SYNC fields in the header of a NORM burst and the bytes behind
 * the last block keep what the buffer held -- clear it once if records are to be compared byte for byte.
 */
int tgpu_plan_execute(struct tgpu_plan *plan, const uint8_t *d_stream, uint8_t *d_rec, void *hip_stream);

/*
 * Soft input (BASELINE config 5 -- an extension: the reference slices hard in float_to_bits and has no
 * soft path).  d_soft_stream has the layout of the bit stream (slot offsets of the loaded batch apply)
 * but holds int8 soft values: positive = bit 0, negative = bit 1, 0 = erasure, |v| <= 127.  The decoder
 * then maximises the correlation metric (what libosmocore's accelerated osmo_conv_decode does with soft
 * values) with the same tie rule; BBK bits are the signs.  Records come out in the same format.
 */
int tgpu_plan_execute_soft(struct tgpu_plan *plan, const int8_t *d_soft_stream, uint8_t *d_rec, void *hip_stream);

/*
 * The same decode straight from the float phase stream that float_to_bits.c (float_to_bits.c:33-72) would slice:
 * d_phi = nfloats float32 phase values (units of pi/4) on the device, symbol k = stream positions 2 k and 2 k + 1 (the
 * slot offsets of the loaded batch count those positions; every slot must lie inside 2 nfloats positions).  Equal, record
 * for record, to tgpu_float_to_bits(d_phi, nfloats, bits, soft) followed by tgpu_plan_execute_soft(soft): the soft
 * values are formed inside the gather kernel and neither the bit stream nor the soft stream is written to memory.
 */
int tgpu_plan_execute_float(struct tgpu_plan *plan, const float *d_phi, uint64_t nfloats, uint8_t *d_rec, void *hip_stream);

/*
 * float_to_bits.c on the device: n float32 phase values (units of pi/4) -> 2 n bits, 1 per byte, with
 * the slicer of float_to_bits.c:33-72 (bit-exact, NaN included).  d_soft (optional, 2 n int8):
 * soft0 = sat127(rint(64 phi)), soft1 = sat127(rint(64 (2 - |phi|))).
 * The _afc variant adds the pseudo-AFC of float_to_bits.c:142-146 (-a -f filter_val -F filter_goal): a
 * sequential IIR, run by one lane in the reference's operation order; *filter_state carries the
 * filter value across calls (start with 0).  It synchronises the stream.
 */
int tgpu_float_to_bits(struct tgpu_engine *eng, const float *d_in, uint64_t n, uint8_t *d_bits, int8_t *d_soft,
		       void *hip_stream);
int tgpu_float_to_bits_afc(struct tgpu_engine *eng, const float *d_in, uint64_t n, uint8_t *d_bits, float filter_val,
			   float filter_goal, float *filter_state, void *hip_stream);

/* scrambling code in effect per channel after the batch (device->host copy, synchronises) */
int tgpu_plan_final_codes(struct tgpu_plan *plan, const uint8_t *d_rec, uint32_t *chan_code_out);

/*
 * Transport form of the records: 40 bytes per slot = ten dwords -- header (burst type, flags, the 14 BBK bits), the
 * type-1 bits packed LSB first, the CRC words (crc_ok is not carried: it is crc == 0x1d0f); layout in
 * csrc/tg_layout.h.  When a device buffer of nslots * TGPU_WIRE_BYTES is attached to the plan the trellis kernels
 * write it alongside the full records; it is what a rank sends to the collecting rank over xGMI (RCCL gather).
 * Slots a batch does not decode are not written: clear the buffer to 0xff once.  tgpu_wire_unpack() rebuilds a full
 * TGPU_REC_BYTES record on the host (burst type 0xff for a slot that holds nothing); the slot id and the scrambling
 * code are not transported and are passed in.  tgpu_wire_pack() is the host form of what the kernels write.
 */
#define TGPU_WIRE_BYTES 40
int tgpu_plan_set_wire(struct tgpu_plan *plan, uint8_t *d_wire /* NULL: off */);
/* on: with a wire buffer attached the trellis kernels write ONLY the wire records -- the 320-byte records (one byte per
 * type-1 bit, the reference's format) are eight times the volume and are what a step's HBM writes consist of; a
 * consumer that takes the packed form (the gather to a collecting rank; a host that unpacks with tgpu_wire_unpack())
 * does not need them.  d_rec is still passed to execute: slots of an ignored burst type get their 0xff type byte. */
int tgpu_plan_set_wire_only(struct tgpu_plan *plan, int on);
/*
 * A plan owns a side stream: independent kernels of ONE batch run beside each other there (k_vit<432> beside k_vit<216>; in
 * device-walk batches the SB1 decode beside the synchroniser walk), which shortens a single batch's latency.  A caller that
 * keeps SEVERAL batches in flight on several streams is better off with off: each batch then lives on its caller's stream
 * alone, i.e. on ONE hardware queue (the HIP runtime multiplexes all streams of a process onto 4 of them, and two streams
 * that share a queue run one after the other) -- with the side streams in play the batches spread unevenly over the queues
 * (bench.py, four batches in flight: 0.46 ms per step on, 0.42 off).  Default: on.
 */
int tgpu_plan_set_side_stream(struct tgpu_plan *plan, int on);
int tgpu_wire_unpack(const uint8_t *wire, uint32_t slot_id, uint32_t scrambling_code, uint8_t *rec);
int tgpu_wire_pack(const uint8_t *rec, uint8_t *wire);

/*
 * Compact transport form of a batch ("cwire", csrc/tg_cwire.h): what a rank hands to the gather.  The 40-byte wire records
 * exist for every grid slot; the collecting rank's upper MAC is handed delivered bursts only (phy/tetra_burst_sync.c:113-150)
 * and acts on CRC-good blocks (tetra_upper_mac.c:480-488).  One buffer per batch: header, channel table, delivered bitmap,
 * block table, then the delivered bursts' records back to back in grid order -- 16 header bits (burst type, the 14 BBK bits)
 * + the type-1 bits: 36 bytes for NORM_1, 33 for NORM_2, 25 for SYNC; a burst with a flag or a failed CRC travels as its
 * 40-byte wire record behind an escape byte (41).  33.5 bytes per delivered burst of the SB+NDB mix instead of 40 per grid slot.
 *   tgpu_cwire_bound()      bytes a batch of ngrid slots in nchan channels can need at most (capacity for the calls below)
 *   tgpu_plan_set_cwire()   device-walk batches (tgpu_sync_multi_launch) of this plan leave the compact form of their wire
 *                           records (tgpu_plan_set_wire() must be set too) in d_cwire (16-byte aligned), enqueued behind
 *                           the decode; tgpu_sync_dev_cwire_bytes() after tgpu_sync_multi_collect() = the bytes to send
 *                           (0: no compact form was made -- no cwire buffer, an empty grid, a buffer smaller than the batch
 *                           needed (cap_bytes may be less than tgpu_cwire_bound(); tgpu_sync_dev_cwire_needed() then says how
 *                           much it takes), or the batch fell back to the host walks, whose decode does not redo it: use
 *                           tgpu_wire_compact() with the bitmap of the outcome)
 *   tgpu_wire_compact()     the same for any batch: d_wire = ngrid 40-byte records, d_grid_bits = its delivered bitmap
 *                           (device), channel c = grid slots gbase[c] .. gbase[c] + ncls[c] - 1 (gbase multiples of 32,
 *                           ascending, <= 64 channels); d_total (optional, device): two words -- the bytes the batch needs
 *                           and its delivered bursts (0xffffffff: cap was too small, the buffer holds no records).  Launches only.
 *   tgpu_cwire_pack()       the host form (same bytes); returns the size or a negative TGPU_E* code
 *   tgpu_cwire_info / _chan / _foreach / _expand: the reader the collecting rank runs on a received buffer (host): checks the
 *                           header, per-channel counts, every delivered burst handed to a callback as its 40-byte wire
 *                           record (grid order; tgpu_wire_unpack() makes the full record of it), or the whole grid's wire
 *                           records rebuilt (0xff for undelivered slots) + the bitmap.  _foreach returns the bursts handed
 *                           over or a negative code when the buffer does not parse.
 */
struct tgpu_cwire_info {
	uint32_t nchan, ngrid, ndelivered;
	uint64_t total_bytes;
};
typedef void (*tgpu_wire_cb)(const uint8_t *wire_rec, uint32_t grid_slot, void *priv);
uint64_t tgpu_cwire_bound(uint32_t ngrid, uint32_t nchan);
int tgpu_plan_set_cwire(struct tgpu_plan *plan, uint8_t *d_cwire /* NULL: off */, size_t cap_bytes);
int tgpu_wire_compact(struct tgpu_engine *eng, const uint8_t *d_wire, const uint32_t *d_grid_bits, uint32_t ngrid, uint32_t nchan,
		      const uint32_t *gbase, const uint32_t *ncls, uint8_t *d_cwire, size_t cap_bytes, uint32_t *d_total, void *hip_stream);
int64_t tgpu_cwire_pack(const uint8_t *wire, const uint32_t *grid_bits, uint32_t ngrid, uint32_t nchan, const uint32_t *gbase,
			const uint32_t *ncls, uint8_t *out, size_t cap_bytes);
int tgpu_cwire_info(const uint8_t *cwire, size_t nbytes, struct tgpu_cwire_info *out);
int tgpu_cwire_chan(const uint8_t *cwire, size_t nbytes, uint32_t chan, uint32_t *gbase, uint32_t *ncls, uint32_t *ndelivered);
int64_t tgpu_cwire_foreach(const uint8_t *cwire, size_t nbytes, tgpu_wire_cb cb, void *priv);
int tgpu_cwire_expand(const uint8_t *cwire, size_t nbytes, uint8_t *wire /* ngrid x 40 */, uint32_t *grid_bits /* optional */);

/*
 * The gather itself, in C (north star: "RCCL only for the final decoded-block gather over xGMI"; the
 * reference has no counterpart -- it runs one process per channel).  One process per GPU.  Rank 0 (or any one rank)
 * draws an id with tgpu_comm_unique_id() and hands its TGPU_COMM_ID_BYTES bytes to the other ranks by whatever means
 * the job has (a file, MPI, the launcher's store); every rank then calls tgpu_comm_create() with its rank.
 * tgpu_comm_gather(): every rank's nbytes at d_send (device) arrive at the root's d_recv + rank * nbytes; grouped
 * RCCL send / receive on hip_stream, asynchronous, peer -> root directly (one xGMI link per peer).  d_recv is read
 * on the root only.  RCCL is loaded on first use (an RCCL the process already holds is reused); TGPU_ENOSYS if there
 * is none.
 */
#define TGPU_COMM_ID_BYTES 128
struct tgpu_comm;
int tgpu_comm_unique_id(uint8_t id[TGPU_COMM_ID_BYTES]);
int tgpu_comm_create(struct tgpu_engine *eng, const uint8_t id[TGPU_COMM_ID_BYTES], int rank, int world, struct tgpu_comm **out);
int tgpu_comm_gather(struct tgpu_comm *comm, const void *d_send, size_t nbytes, void *d_recv, int root, void *hip_stream);
/* the same with a size per rank (the compact form above): rank r's nbytes[r] bytes arrive at the root's d_recv + offs[r]; every
 * rank passes its own size in nbytes[its rank], the root needs all of nbytes[] and offs[] (the other ranks may pass NULL offs).
 * How the sizes reach the root is the job's business, like the id (they are known after tgpu_sync_multi_collect()). */
int tgpu_comm_gatherv(struct tgpu_comm *comm, const void *d_send, const size_t *nbytes, void *d_recv, const size_t *offs, int root,
		      void *hip_stream);
/* nmsg of those in ONE exchange (the decoded blocks of nmsg steps gathered together: north star's "final" gather; one RCCL
 * group = one launch per rank, whatever nmsg): message m's rank-r share (nbytes[m * world + r] bytes at d_send[m] on rank r)
 * arrives at the root's d_recv + offs[m * world + r]. */
int tgpu_comm_gatherv_batch(struct tgpu_comm *comm, int nmsg, const void *const *d_send, const size_t *nbytes, void *d_recv,
			    const size_t *offs, int root, void *hip_stream);
void tgpu_comm_destroy(struct tgpu_comm *comm);

/* diagnostic: copy the front kernel's packed slots (20 dwords per slot, csrc/tg_layout.h) of the
 * last executed batch to the host; synchronises the device */
int tgpu_plan_read_packed(struct tgpu_plan *plan, uint32_t *out_words);

/*
 * Per-kernel timing with HIP events on the SAME stream the kernels are launched on.
 * tgpu_plan_execute_prof() is tgpu_plan_execute() plus one event record between stages
 * (no host synchronisation); tgpu_prof_read() synchronises on the last event and returns
 * the milliseconds of every stage of every recorded step: ms[step * TGPU_NSTAGES + stage].
 */
#define TGPU_NSTAGES 6
#define TGPU_STAGE_FRONT   0	/* k_front                */
#define TGPU_STAGE_SB1     1	/* k_vit<SB1>             */
#define TGPU_STAGE_FILL    2	/* k_fill_* (3 launches)  */
#define TGPU_STAGE_MASKS   3	/* k_masks                */
#define TGPU_STAGE_VIT216  4	/* k_vit<216>             */
#define TGPU_STAGE_VIT432  5	/* k_vit<432>             */
struct tgpu_prof;
int tgpu_prof_create(uint32_t max_steps, struct tgpu_prof **out);
void tgpu_prof_destroy(struct tgpu_prof *prof);
int tgpu_plan_execute_prof(struct tgpu_plan *plan, const uint8_t *d_stream, uint8_t *d_rec, void *hip_stream,
			   struct tgpu_prof *prof, uint32_t step);
/* the same for tgpu_plan_execute_float() (stage 0 = the fused slicer + soft gather kernel) */
int tgpu_plan_execute_float_prof(struct tgpu_plan *plan, const float *d_phi, uint64_t nfloats, uint8_t *d_rec, void *hip_stream,
				 struct tgpu_prof *prof, uint32_t step);
int tgpu_prof_read(struct tgpu_prof *prof, uint32_t nsteps, float *ms);
const char *tgpu_stage_name(int stage);

/* one decoded block of a record, in the reference's terms */
struct tgpu_block {
	enum tp_sap_data_type type;
	int blk_num;
	int crc_ok;
	uint16_t crc;
	uint32_t scrambling_code;
	uint16_t type1_len;
	const uint8_t *type1;	/* points into the record, 1 bit per byte */
};

/*
 * Split a (host copy of a) record into its blocks in tetra_burst_rx_cb() call order:
 * SYNC: SB1, BBK, SB2.  NORM_2: BBK, NDB(BLK_1), NDB(BLK_2).  NORM_1: BBK, SCH/F.
 * Returns the number of blocks (0 for a skipped slot).
 */
int tgpu_record_blocks(const uint8_t *rec, struct tgpu_block out[3]);

/* SYNC-PDU fields of a SYNC record (lower_mac/tetra_lower_mac.c:284-297) */
struct tgpu_sync_info {
	uint8_t cc, tn, fn, mn;
	uint16_t mcc, mnc;
	uint32_t scramb_init;
};
int tgpu_record_sync_info(const uint8_t *rec, struct tgpu_sync_info *out);

/* ------------------------------------------------------------------------- */
/* 2. channel API                                                             */
/* ------------------------------------------------------------------------- */

/* TMV-SAP UNITDATA indication: what the reference hands to upper_mac_prim_recv()
 * inside struct tetra_tmvsap_prim + msgb (tetra_prim.h:25-47). */
#define TGPU_BURST_UNKNOWN 0xff	/* tgpu_unitdata.burst_type: not known (never a value of enum tetra_train_seq) */
struct tgpu_unitdata {
	enum tp_sap_data_type type;
	int blk_num;
	enum tetra_log_chan lchan;
	int crc_ok;
	uint16_t crc;
	uint32_t scrambling_code;
	struct tetra_tdma_time tdma_time;
	uint32_t burst_seq;		/* ordinal of the LOCKED burst */
	enum tetra_train_seq burst_type;	/* of the burst the block came in; on the tp_sap_udata_ind() seam, where blocks
					 * arrive on their own, TGPU_BURST_UNKNOWN for a block no burst type is implied by
					 * (BBK, SCH/HU) unless tetra_burst_rx_cb() of this library handed it over */
	uint16_t type1_len;
	const uint8_t *type1;		/* msg->l1h: borrowed for the call */
	int traffic;			/* != 0: block is a traffic-channel block (cur_burst.is_traffic value);
					 * the reference dumps it instead of decoding it
					 * (lower_mac/tetra_lower_mac.c:198-241); type4 is set, type1 is NULL */
	const uint8_t *type4;		/* descrambled bits of a traffic block, type345 bits */
	uint16_t type4_len;
	struct tetra_tdma_time time_str;/* the PHY clock when the block came in (tetra_lower_mac.c:167-168): the time the
					 * reference's "<NAME> <time> type1: ..." line shows -- for an SB1 block the value
					 * from before its SYNC PDU moved the clock (tdma_time holds the one after) */
};

/*
 * Same contract as upper_mac_prim_recv(): return the number of type-1 bits parsed, or
 * -1 when done with this block.  'offset' = bits already consumed (msg->head advance,
 * lower_mac/tetra_lower_mac.c:326-352).  Traffic blocks are delivered once with
 * offset = UINT32_MAX and the return value is ignored.
 */
typedef int (*tgpu_unitdata_cb)(const struct tgpu_unitdata *ud, unsigned int offset, void *priv);

/* sync-layer notifications: what phy/tetra_burst_sync.c prints */
enum tgpu_sync_event {
	TGPU_EV_FOUND_SYNC = 1,
	TGPU_EV_BURST = 2,
	TGPU_EV_SYNC_MISPLACED = 3,
	TGPU_EV_NORM_MISPLACED = 4,
	TGPU_EV_NO_TRAIN = 5,
	TGPU_EV_ERROR = 6,	/* a queued batch could not be decoded: bitnum = the error, arg = bursts / blocks lost */
};
typedef void (*tgpu_event_cb)(int event, uint32_t bitnum, uint32_t arg, void *priv);

struct tgpu_channel;

/*
 * batch_slots: bursts queued before a GPU decode is triggered (1 = decode every burst
 * immediately, like the reference).  'priv' is passed to both callbacks (the reference
 * passes tms).  Store the returned pointer in trs->burst_cb_priv.
 */
int tgpu_channel_create(struct tgpu_engine *eng, uint32_t batch_slots, tgpu_unitdata_cb cb,
			tgpu_event_cb ev, void *priv, struct tgpu_channel **out);
void tgpu_channel_destroy(struct tgpu_channel *ch);

/* feedback flags of tms->cur_burst (tetra_common.h:51-55).  Either bind the upper MAC's
 * own variables (read at delivery time, blk1_stolen is also written) ... */
void tgpu_channel_bind_flags(struct tgpu_channel *ch, int *is_traffic, bool *blk1_stolen, bool *blk2_stolen);
/* ... or set them from the callback */
void tgpu_channel_set_traffic(struct tgpu_channel *ch, int is_traffic);
void tgpu_channel_set_blk2_stolen(struct tgpu_channel *ch, bool stolen);

/* decode and deliver everything queued so far.  A batch that cannot be decoded (HIP or plan error) is given up:
 * its time steps still advance the channel's clock, the error is kept (also when the flush was an automatic one
 * inside tetra_burst_sync_in(), whose return value has no room for it): tgpu_channel_flush() returns it from then
 * on, tgpu_channel_last_error() reads it, the event callback gets TGPU_EV_ERROR. */
int tgpu_channel_flush(struct tgpu_channel *ch);
int tgpu_channel_last_error(const struct tgpu_channel *ch);
void tgpu_channel_clear_error(struct tgpu_channel *ch);

/*
 * The lower-MAC seam under the reference's own signatures (phy/tetra_burst.h:18, phy/tetra_burst.c:341), priv = the
 * struct tgpu_channel * (the reference passes tms).  libosmo-tetra-phy.a links against this library unchanged: it
 * finds tp_sap_udata_ind and tetra_tdma_time_add_tn here and brings its own t_phy_state (ours is a weak definition
 * of the same object, phy/tetra_burst_sync.c:34, used when that file is not linked).  Blocks are queued and decoded
 * on the GPU in batches (SCH/HU included); an SB1 block is decoded before tp_sap_udata_ind() returns, so the code
 * and time it brings apply to the next block as in the reference; the rest is delivered when the queue fills, at
 * the next SB1 or by tgpu_channel_flush().  Delivery = the channel's tgpu_unitdata_cb, same order, same contents.
 */
#ifndef TETRA_COMMON_H	/* tetra_common.h:44-47 */
struct tetra_phy_state {
	struct tetra_tdma_time time;
};
extern struct tetra_phy_state t_phy_state;
#endif
void tp_sap_udata_ind(enum tp_sap_data_type type, int blk_num, const uint8_t *bits, unsigned int len, void *priv);
void tetra_burst_rx_cb(const uint8_t *burst, unsigned int len, enum tetra_train_seq type, void *priv);

/* phy/tetra_burst_sync.c:54 -- same symbol, same semantics */
int tetra_burst_sync_in(struct tetra_rx_state *trs, uint8_t *bits, unsigned int len);

/* phy/tetra_burst.c:269-339 -- same symbol, same semantics (host) */
int tetra_find_train_seq(const uint8_t *in, unsigned int end_of_in,
			 uint32_t mask_of_train_seq, unsigned int *offset);

/* tetra_tdma.c:77-81, lower_mac/tetra_scramb.c:87-99 */
void tetra_tdma_time_add_tn(struct tetra_tdma_time *tm, uint32_t tn_count);
uint32_t tetra_scramb_get_init(uint16_t mcc, uint16_t mnc, uint8_t colour);

/* ------------------------------------------------------------------------- */
/* 2b. whole-stream burst synchronisation, GPU-assisted (BASELINE config 3)    */
/* ------------------------------------------------------------------------- */
/*
 * For a recorded stream the per-call state machine of tetra_burst_sync_in() is a closed-form
 * function of byte positions.  tgpu_sync_stream() finds the first lock on the host, lets the GPU
 * front end (k_front_stream) search every grid slot's window for the first training sequence
 * (the reference's tetra_find_train_seq() scan, phy/tetra_burst.c:269-339), and walks the result:
 * the outcome -- which slots reach tetra_burst_rx_cb(), with which burst ordinal, and every sync
 * event -- is exactly what feeding the stream to tetra_burst_sync_in() 'chunk' bytes at a time gives.
 * d_stream must have TGPU_STREAM_SLACK readable bytes after 'len'.  The slots then go to the plan API.
 */
#define TGPU_STREAM_SLACK 192

struct tgpu_sync_slot {
	uint64_t off;		/* stream offset of the slot */
	uint32_t burst_seq;	/* ordinal of the LOCKED burst (1-based) */
	uint32_t tn_adds;	/* tetra_tdma_time_add_tn(,1) calls since the previously delivered burst */
	uint8_t type;		/* enum tetra_train_seq */
};

struct tgpu_sync_event_rec {
	int32_t ev;		/* enum tgpu_sync_event */
	uint32_t bitnum;
	uint32_t arg;
};

struct tgpu_sync_result {
	uint32_t nslots;
	struct tgpu_sync_slot *slots;
	uint32_t nevents;
	struct tgpu_sync_event_rec *events;
	int final_state;	/* enum rx_state after the last byte */
	uint32_t tail_tn_adds;	/* time steps after the last delivered burst */
	uint32_t burst_seq;	/* LOCKED bursts seen in total */
	uint64_t anchor;	/* start of the slot grid the GPU classified */
	/* grid mode (TGPU_SYNC_GRID / tgpu_sync_stream_grid): no slot table; bit g of grid_bits = the burst at
	 * anchor + 510 g is delivered; nslots = bits set; noffgrid = delivered bursts that are not grid slots
	 * (possible after a re-lock onto a shifted grid): if != 0 the caller falls back to tgpu_sync_stream() */
	uint32_t *grid_bits;
	uint32_t ngrid;
	uint32_t noffgrid;
	uint32_t grid_base;	/* multi-channel batches (tgpu_sync_multi_*): record index of this channel's grid slot 0 */
};

#define TGPU_SYNC_NO_BURST_EVENTS 1u	/* do not record one TGPU_EV_BURST per locked burst (throughput runs) */
#define TGPU_SYNC_GRID 2u		/* tgpu_sync_walk(): bitmap over the classified grid instead of a slot table */
#define TGPU_SYNC_PER_CALL 4u		/* tgpu_sync_walk(): the reference's state machine call by call on the bytes (slow, for checks) */
/* feed sizes (bytes per tetra_burst_sync_in() call being emulated) the synchroniser's closed form covers; tetra-rx.c
 * feeds 64.  Outside this range (1 .. 510 is accepted) tgpu_sync_walk() runs the per-call form instead. */
#define TGPU_SYNC_CHUNK_MIN 21u
#define TGPU_SYNC_CHUNK_MAX 296u
int tgpu_sync_stream(struct tgpu_engine *eng, const uint8_t *h_stream, const uint8_t *d_stream, uint64_t len,
		     uint32_t chunk, uint32_t flags, struct tgpu_sync_result *out, void *hip_stream);
/*
 * The same synchroniser feeding a plan without a host-built slot table: the classification pass packs every
 * grid slot into the plan's buffer, the host walk marks the delivered ones, and the plan's per-slot arrays and
 * item lists are built on the device.  Afterwards tgpu_plan_execute(plan, d_stream, d_rec, ...) decodes them:
 * record g (d_rec + 320 g, d_rec sized for out->ngrid records) belongs to the burst at out->anchor + 510 g and
 * is valid iff bit g of out->grid_bits is set.  scramb_init: the channel's code before its first slot.
 * plan: capacity >= (len - anchor) / 510 slots.  If out->noffgrid != 0 the plan is NOT loaded.
 */
int tgpu_sync_stream_grid(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *h_stream,
			  const uint8_t *d_stream, uint64_t len, uint32_t chunk, uint32_t flags, uint32_t scramb_init,
			  struct tgpu_sync_result *out, void *hip_stream);
/* the same in two halves, so that the classification of one stream (GPU, asynchronous) can run under the host walk
 * of another: _begin finds the first lock and launches classification + copy on hip_stream without waiting,
 * _finish (same plan, stream, h_stream, len, chunk and 'out') waits for them, walks and loads the plan */
int tgpu_sync_stream_grid_begin(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *h_stream,
				const uint8_t *d_stream, uint64_t len, uint32_t chunk, struct tgpu_sync_result *out,
				void *hip_stream);
int tgpu_sync_stream_grid_finish(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *h_stream, uint64_t len,
				 uint32_t chunk, uint32_t flags, uint32_t scramb_init, struct tgpu_sync_result *out,
				 void *hip_stream);
void tgpu_sync_result_free(struct tgpu_sync_result *r);
/*
 * Several recorded channels in one batch (BASELINE config 4: a GPU's share of the channels; the reference runs one
 * process per channel, src/receiver1:1-10).  The channels' streams lie in one device buffer (channel c at byte d_off,
 * at least 2304 readable bytes behind the last one) and are also in host memory for the synchroniser walks.
 *   begin : first lock of every channel (host), ONE classification launch over all grids, one copy back (async)
 *   finish: waits, walks every channel (up to nthreads host threads), builds one plan batch on the device: record
 *           index of grid slot i of channel c = out[c].grid_base + i, valid iff bit i of out[c].grid_bits;
 *           channel c's carry-in code = scramb_init; a channel that re-locks off its grid (out[c].noffgrid != 0)
 *           is left out of the batch (decode it through tgpu_sync_stream()).  out = nchan results, each to be
 *           released with tgpu_sync_result_free().  Then tgpu_plan_execute(plan, d_base, d_rec, stream).
 * The plan needs max_slots >= the padded grid total (tgpu_sync_multi_ngrid) and max_chan >= nchan (<= 64).
 */
struct tgpu_multi_chan {
	const uint8_t *h_stream;	/* host copy of the channel's stream */
	uint64_t d_off;			/* where it starts in the device buffer */
	uint64_t len;
	uint32_t scramb_init;		/* code in force before the first SYNC burst (0 for a fresh channel) */
};
struct tgpu_sync_multi;
int tgpu_sync_multi_begin(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			  const uint8_t *d_base, uint32_t chunk, struct tgpu_sync_multi **out, void *hip_stream);
int tgpu_sync_multi_finish(struct tgpu_sync_multi *st, uint32_t flags, unsigned int nthreads, struct tgpu_sync_result *out,
			   void *hip_stream);
uint32_t tgpu_sync_multi_ngrid(const struct tgpu_sync_multi *st);
void tgpu_sync_multi_free(struct tgpu_sync_multi *st);
/*
 * The same batch with the synchroniser walks on the DEVICE (round 3; k_walk, csrc/tg_walk_core.h): one call enqueues
 * everything on hip_stream -- classification, plain bitmap, the walk of every channel (delivered bitmap, events, counts),
 * the lists from that bitmap, the decode into d_rec (tgpu_plan_execute is part of it), and the copies of the outcomes
 * into pinned memory -- and returns without waiting; the host's share of a step is the first lock of every channel
 * (a few kB of each stream) and the launches.  chunk: a power of two inside the closed form's range (tetra-rx.c feeds 64).
 *   tgpu_sync_multi_collect() waits for the batch and fills out[nchan] like tgpu_sync_multi_finish() does (events without
 *   TGPU_EV_BURST; release each with tgpu_sync_result_free()).  Where the device walk cannot settle a channel -- only the
 *   bytes can decide (a byte other than 0 / 1 near an exception, a sequence in the first 21 bytes of a search buffer, a
 *   re-lock off the grid) -- the batch is redone through the
 *   host walks and decoded again before the call returns: same results (tgpu_sync_dev_fellback() tells).  A channel of
 *   more than 262 144 slots (an hour of one carrier) is walked on the device as well, with its working arrays in a scratch
 *   area of the plan instead of LDS (k_walk_big: up to 8 such channels per batch, exceptions up to an eighth of the plan's
 *   slots); so is a shorter one with more exceptions than the LDS form holds (8192: a noisy recording) -- from the second
 *   batch on: the first one that overflows goes through the host walks and leaves the density it saw with the plan.
 * Plan capacity as above; the plan must stay untouched between launch and collect; several batches are kept in flight
 * with several plans.  Records: valid for the delivered slots (bit set in grid_bits).  The SB1 decode of a device-walk batch
 * runs beside the walk, over every SYNC-classified slot, so the record (and wire record) of a SYNC-classified slot that the walk
 * then does NOT deliver may hold that slot's SB1 fields (bits, CRC word, SYNC-PDU fields; type byte untouched) -- the host-walk
 * path leaves such records untouched.  Nothing reads them: consumers go by the bitmap.
 */
struct tgpu_sync_dev;
int tgpu_sync_multi_launch(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			   const uint8_t *d_base, uint32_t chunk, uint8_t *d_rec, struct tgpu_sync_dev **out, void *hip_stream);
int tgpu_sync_multi_collect(struct tgpu_sync_dev *sd, struct tgpu_sync_result *out);
/*
 * Packed ingest (optional; the API's input format stays one bit per byte): a capture that sits in HOST memory crosses PCIe
 * as 510 bytes per burst, which bounds the end-to-end rate near 1e8 bursts/s whatever the kernels do.  tgpu_pack_bits()
 * packs it on the host -- bit i of packed byte k = bytes[8 k + i] & 1, the order of the kernels' own bit string; nthreads
 * host threads, AVX2 where the CPU has it -- and returns how many pieces held a byte other than 0 / 1 (0: the packed stream
 * is equivalent; otherwise the byte path is the one to use: such bytes break the reference's memcmp in tetra_find_train_seq()
 * and the packed form cannot show them), or a negative TGPU_E* code.  tgpu_sync_multi_launch_packed() is
 * tgpu_sync_multi_launch() on such a buffer: d_packed_base = the packed streams on the device, ch[c].d_off = the BIT offset
 * of channel c's position 0 in it (a multiple of 8), ch[c].h_stream / len = the unpacked host bytes as before (the first
 * lock of a channel is found on the host).  d_packed_base must be 16-byte aligned (TGPU_EINVAL otherwise: groups are fetched
 * from 16-byte aligned addresses counted from it) and needs 512 readable bytes behind the last channel's last bit.
 * Everything downstream is the same kernels on the same bits: records are byte-identical to the byte path's.
 */
int64_t tgpu_pack_bits(const uint8_t *bytes, uint64_t n, uint8_t *packed /* (n + 7) / 8 bytes */, unsigned int nthreads);
int tgpu_sync_multi_launch_packed(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
				  const uint8_t *d_packed_base, uint32_t chunk, uint8_t *d_rec, struct tgpu_sync_dev **out, void *hip_stream);
uint32_t tgpu_sync_dev_ngrid(const struct tgpu_sync_dev *sd);
int tgpu_sync_dev_fellback(const struct tgpu_sync_dev *sd);
/* 1: this batch's front end and trellises ran as ONE launch (k_slot, TGPU_OPT_SLOT 2) on hinted scrambling codes; 2: its trellises
 * ran early, beside the walk (k_slot_e, TGPU_OPT_SLOT 3); 0: neither (a plan's first batch without carry-in codes, another setting
 * of the option) -- same records for delivered slots either way */
int tgpu_sync_dev_fused(const struct tgpu_sync_dev *sd);
/* after collect: why the device walk handed channel c to the host walks (0: it did not).  1 a flagged slot (a byte other than
 * 0 / 1, a sequence below offset 21) on the walk's way, 2 a search window the kernel's view does not settle, 3 a SYNC sequence in
 * the first 21 bytes of a search buffer, 4 a lock beside the slot grid, 5 / 6 / 7 / 8 / 9 internal bounds (iterations, events or
 * deliveries per exception, SYNC summaries), 10 more exceptions than this launch's arrays hold, 11 a channel beyond the LDS
 * form's length that found no place in the long form */
int tgpu_sync_dev_why(const struct tgpu_sync_dev *sd, uint32_t chan);
uint64_t tgpu_sync_dev_cwire_bytes(const struct tgpu_sync_dev *sd);	/* after collect; tgpu_plan_set_cwire() */
/* after collect: non-zero when the buffer given to tgpu_plan_set_cwire() was too small for this batch -- the bytes it needs
 * (tgpu_sync_dev_cwire_bytes() is 0 then and the buffer holds no records; the batch itself is complete and collect returns OK) */
uint64_t tgpu_sync_dev_cwire_needed(const struct tgpu_sync_dev *sd);
void tgpu_sync_dev_free(struct tgpu_sync_dev *sd);
/* measurement aid: one such batch, synchronously, with HIP events between all of its stages on hip_stream: dev_ms[] =
 * the stages in front of the decode (names: tgpu_sync_dev_stage_name), the decode's stages in prof / step as
 * tgpu_plan_execute_prof() leaves them (read with tgpu_prof_read; stage 0, k_front, is empty in stream mode) */
#define TGPU_NDEVSTAGES 8
int tgpu_sync_multi_launch_prof(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
				const uint8_t *d_base, uint32_t chunk, uint8_t *d_rec, void *hip_stream, struct tgpu_prof *prof,
				uint32_t step, float dev_ms[TGPU_NDEVSTAGES]);
const char *tgpu_sync_dev_stage_name(int stage);
/* the consumer's end of a gathered batch: every delivered burst's 40-byte wire record (csrc/tg_layout.h) handed to a
 * callback, in grid order.  grid_bits = the delivered bitmap of the ngrid slots (NULL: every record that carries a burst
 * type); cb == NULL only counts.  Returns the number of records handed over.  tgpu_wire_noop_cb(): a callback that reads
 * the record's header and does nothing else (priv -> a uint64_t), for measurements. */
uint64_t tgpu_wire_foreach(const uint8_t *wire, const uint32_t *grid_bits, uint32_t ngrid, tgpu_wire_cb cb, void *priv);
tgpu_wire_cb tgpu_wire_noop_cb(void);
/* measurement aid, as tgpu_sync_front_prof() for a multi-channel batch */
int tgpu_sync_front_prof_multi(struct tgpu_engine *eng, struct tgpu_plan *plan, uint32_t nchan, const struct tgpu_multi_chan *ch,
			       const uint8_t *d_base, uint32_t chunk, uint32_t nrep, float us[2], void *hip_stream);

/* measurement aid: the stream front end of a grid (anchor + 510 n) alone, nrep times on hip_stream with HIP events
 * around its launches: us[0] = k_front_stream, us[1] = k_front_stream_fix (mean microseconds).  The plan's grid
 * buffers are the scratch; the plan is left unloaded. */
int tgpu_sync_front_prof(struct tgpu_engine *eng, struct tgpu_plan *plan, const uint8_t *d_stream, uint64_t len,
			 uint32_t chunk, uint64_t anchor, uint32_t nrep, float us[2], void *hip_stream);

/*
 * Optional, off by default: clean-block fast path.  A pre-pass (k_clean) recognises the blocks whose received bits
 * are exactly a code word -- for those the trellis search can only return that code word (see the kernel's
 * comment for the argument) -- and finishes them directly; only the other blocks go through the Viterbi kernels.
 * Records are bit-identical with the flag on or off; throughput becomes input dependent (all blocks clean: the
 * trellis kernels do nothing; no block clean: the pre-pass is overhead).  Applies to the 216- and 432-bit blocks
 * of slot / grid plans with hard input.
 */
int tgpu_plan_set_fastpath(struct tgpu_plan *plan, int on);

/*
 * Optional, off by default: decode the AACH's shortened (30,14) Reed-Muller word instead of keeping its first
 * 14 received bits like the reference (lower_mac/tetra_lower_mac.c:268-274; tetra_rm3014.c:88-96 is a stub).
 * Minimum-distance (syndrome / coset-leader) decoding, ties to the numerically smallest error pattern; up to
 * 3 bit errors are always corrected (d_min = 8).  The number of corrected bits is stored at byte 28 of the
 * record.  tgpu_rm3014_decode() is the same decoder on the host: rx30 bit 29 = first received bit.
 */
int tgpu_plan_set_rm_decode(struct tgpu_plan *plan, int on);
int tgpu_channel_set_rm_decode(struct tgpu_channel *ch, int on);
int tgpu_rm3014_decode(uint32_t rx30, uint16_t *data14, unsigned int *nerr);

/*
 * Block mode -- the unit tp_sap_udata_ind() receives (phy/tetra_burst.h:18, lower_mac/tetra_lower_mac.c:143):
 * blocks of type-5 bits on their own, without a burst around them.  blk_off[i] = byte offset of block i's
 * bits (1 bit per byte: 120 for SB1, 216 for SB2/NDB, 168 for SCH/HU, 432 for SCH/F, 30 for BBK) in the
 * buffer later given to tgpu_plan_execute(); blk_type[i] = enum tp_sap_data_type; blk_code[i] = the
 * scrambling code in force for it (tcd->scramb_init; ignored for SB1, which always uses 3).  The number of
 * distinct codes must not exceed the plan's max_chan.  tgpu_plan_execute(plan, d_bits, d_rec, stream) then
 * writes one 320-byte record per block (same layout as slot records): @0 block type, @2 crc_ok, @4 crc,
 * @8 code, @12 block index, type-1 bits @48 (BBK: 14 bits @32, crc_ok = 1), SB1 also the SYNC-PDU fields.
 * This is also the only way in for SCH/HU (lower_mac/tetra_lower_mac.c:80-87), which no downlink burst carries.
 */
int tgpu_plan_load_blocks(struct tgpu_plan *plan, uint32_t nblocks, const uint64_t *blk_off,
			  const uint8_t *blk_type, const uint32_t *blk_code);

/*
 * The remaining channel codings of the reference's tables: any of its seven RCPC
 * puncturers (enum tetra_rcpc_puncturer, lower_mac/tetra_conv_enc.h:17-25, same numbering: 0 = 2/3, 1 = 1/3,
 * 2 = 292/432, 3 = 148/432, 4 = 112/168, 5 = 72/162, 6 = 38/80) on the rate-1/4 mother code (mother_rate 4,
 * conv_cch_decode, lower_mac/viterbi_cch.c:58-66) or the rate-1/3 speech code (mother_rate 3, conv_tch_decode,
 * lower_mac/viterbi_tch.c:56-64), for batches of equally shaped blocks resident in HBM.
 *
 * tgpu_conv_execute() replaces, per block, what a caller of the reference writes (conv_enc_test.c:66-70,
 * tetra_lower_mac.c:249-253):  memset(dp, 0xff, ..); tetra_rcpc_depunct(punct, type3, type3_len, dp);
 * viterbi_dec_sb1_wrapper(dp, type2, type2_len)  -- start state 0, type2_len steps + 4 flush steps, ties to the
 * predecessor whose oldest bit is 0.  For the speech code the three de-punctured values of a step follow one
 * erased value (that code's struct says N = 4 with 3-bit outputs, viterbi_tch.c:49-54).
 *   d_type3: nblocks x type3_len received bytes, one per bit: 0 -> bit 0, 0xff -> erased, anything else -> bit 1
 *            (lower_mac/viterbi.c:12-22)
 *   d_type2: nblocks x type2_len decoded bits, one per byte
 * Launch only (graph-capturable).  tgpu_conv_create() returns TGPU_EINVAL for an unknown puncturer, a type3_len
 * whose positions run past type2_len x mother_rate, type2_len > 504 or type2_len mod 8 in {1,2,3}.
 */
struct tgpu_conv;
int tgpu_conv_create(struct tgpu_engine *eng, int punct, int mother_rate, uint32_t type3_len, uint32_t type2_len,
		     struct tgpu_conv **out);
int tgpu_conv_execute(struct tgpu_conv *cv, const void *d_type3, uint64_t nblocks, void *d_type2, void *hip_stream);
void tgpu_conv_destroy(struct tgpu_conv *cv);

/*
 * The lower MAC's intermediate bit strings of a batch of blocks of one type, one byte per bit, as the reference's DEBUGP
 * lines show them (lower_mac/tetra_lower_mac.c:175 type5, :188 type4, :246 type3, :251 type3dp, :254 type2) -- the steps
 * the product kernels fold into a mask, a gather order and a trellis on code words, here one after the other on the
 * device (csrc/tg_stages.c): for looking inside a block, and as a second formulation the tests hold the fused path
 * against.  type: TPSAP_T_SB1 / _SB2 / _NDB / _SCH_HU / _SCH_F (K = 120 / 216 / 216 / 168 / 432 received bits, the 2/3
 * puncturer on the rate-1/4 code) or TPSAP_T_BBK (30 bits: descrambled, the first 14 kept, :268-274).
 *   d_type5 : nblocks x K received bytes (0 = bit 0, anything else = bit 1), d_codes: a scrambling code per block (SB1
 *             ignores it: the fixed code 3, tetra_scramb.h:14)
 *   d_type4, d_type3: nblocks x K;  d_type3dp: nblocks x mother_len (0xff = punctured away);  d_type2: nblocks x
 *   type2_len decoded bits;  d_crc (optional): crc16_ccitt_bits() over the first type1_len + 16 of them, 0x1d0f = good.
 *   BBK: d_type3 / d_type3dp / d_crc are not written.  Lengths: tgpu_stages_lengths().  Launches only.
 */
struct tgpu_stages;
int tgpu_stages_create(struct tgpu_engine *eng, enum tp_sap_data_type type, struct tgpu_stages **out);
int tgpu_stages_lengths(const struct tgpu_stages *st, uint32_t *type345_len, uint32_t *mother_len, uint32_t *type2_len, uint32_t *type1_len);
int tgpu_stages_execute(struct tgpu_stages *st, const uint8_t *d_type5, const uint32_t *d_codes, uint64_t nblocks, uint8_t *d_type4,
			uint8_t *d_type3, uint8_t *d_type3dp, uint8_t *d_type2, uint16_t *d_crc, void *hip_stream);
void tgpu_stages_destroy(struct tgpu_stages *st);

/* the reference's puncturing entry points on host buffers, same names, arguments and -EINVAL behaviour
 * (lower_mac/tetra_conv_enc.c:201-248; 'pu' is enum tetra_rcpc_puncturer) */
int get_punctured_rate(int pu, uint8_t *in, int len, uint8_t *out);
int tetra_rcpc_depunct(int pu, const uint8_t *in, int len, uint8_t *out);

/*
 * ACELP bit re-ordering (lower_mac/tch_reordering.c:94-140) as an operation.  The class
 * position tables (EN 300 395-2 Table 4: three lists of 1-based positions inside one codec frame of
 * ncls[0] + ncls[1] + ncls[2] bits) are the CALLER's data -- none ships with this library.
 *   tgpu_acelp_build_map(): the index map of one direction, src_of_dst[2 * nbits]: to_codec != 0 = the reference's
 *       tetra_acelp_type2_to_codec() (class order -> two codec frames), 0 = tetra_acelp_codec_to_acelp().  Returns
 *       the block length 2 * nbits.  The reference's index arithmetic, also for tables that are no permutation (its
 *       own is not): later entries win, a destination nobody names gets -1 (left untouched when the map is applied),
 *       an entry 0 (out[-1] / in[-1] in the reference) is skipped.
 *   tgpu_reorder_*(): any such map applied to nblocks blocks of nbits bytes (1 bit per byte) resident in HBM,
 *       launch only.  Destinations with source -1 keep what d_out held.
 *   tetra_acelp_type2_to_codec() / tetra_acelp_codec_to_acelp(): the reference's entry points (same names and
 *       arguments) on host buffers, using the tables given to tgpu_acelp_set_tables(); they abort() with a
 *       message when none was given.
 */
struct tgpu_reorder;
int tgpu_acelp_build_map(const uint8_t *const cls[3], const unsigned int ncls[3], int to_codec, int32_t *src_of_dst);
int tgpu_reorder_create(struct tgpu_engine *eng, const int32_t *src_of_dst, uint32_t nbits, struct tgpu_reorder **out);
int tgpu_reorder_execute(struct tgpu_reorder *r, const uint8_t *d_in, uint64_t nblocks, uint8_t *d_out, void *hip_stream);
void tgpu_reorder_destroy(struct tgpu_reorder *r);
int tgpu_acelp_set_tables(const uint8_t *const cls[3], const unsigned int ncls[3]);
void tetra_acelp_type2_to_codec(const uint8_t *in, uint8_t *out);
void tetra_acelp_codec_to_acelp(const uint8_t *in, uint8_t *out);

/*
 * GSMTAP wire format of a decoded block: the message tetra_gsmtap_makemsg() builds
 * (tetra_gsmtap.c:31-63; called for every CRC-OK block with ts = tdma_time.tn - 1, ss = signal_dbm = snr = 0,
 * bitdata = msg->l1h, bitlen = msgb_l1len(msg), tetra_upper_mac.c:483-486) -- same arguments, the message goes to
 * 'out' instead of a msgb: 16-byte GSMTAP v2 header (type TETRA_I1, frame number = ((hn*60)+mn)*18+fn in network
 * order, channel sub-type) + the bits packed MSB first.  Returns the message length, TGPU_EINVAL if out_size
 * is too small.  In a tgpu_unitdata_cb: bitdata = ud->type1 + offset, bitlen = ud->type1_len - offset.
 */
int tgpu_gsmtap_makemsg(const struct tetra_tdma_time *tm, enum tetra_log_chan lchan, uint8_t ts, uint8_t ss,
			int8_t signal_dbm, uint8_t snr, const uint8_t *bitdata, unsigned int bitlen,
			uint8_t *out, size_t out_size);
/*
 * The same for a whole decoded batch on the device (k_gsmtap): d_rec = nslots 320-byte records as the plan left them,
 * d_times[i] = the PHY clock when burst i comes in (struct tetra_tdma_time, device memory; the host's replay has it: the
 * clock after tetra_tdma_time_add_tn() of the burst's time steps -- a SYNC burst with a good SB1 sets tn / fn / mn
 * itself), d_traffic (optional): byte per slot, bit 0 = the burst is a traffic burst (cur_burst.is_traffic: its SCH/F or
 * second block is dumped, not indicated), bit 1 = its second block was stolen (and is indicated after all).  Message k
 * (burst order: SB1 BBK SB2 / BBK BLK1 BLK2 / BBK SCH-F) of slot i at d_msgs + (3 i + k) * TGPU_GSMTAP_STRIDE, its length
 * in d_lens[3 i + k]; 0 = no message (CRC failed, no such block, slot not decoded).  ss = signal = snr = 0 and ts = tn - 1
 * as in the reference's call (tetra_upper_mac.c:483-486); the first indication of every block (further PDUs of one block
 * are the callback's: tgpu_gsmtap_makemsg() with the offset).
 */
#define TGPU_GSMTAP_STRIDE 52
int tgpu_gsmtap_batch(struct tgpu_engine *eng, const uint8_t *d_rec, const struct tetra_tdma_time *d_times, const uint8_t *d_traffic,
		      uint32_t nslots, uint8_t *d_msgs, uint8_t *d_lens, void *hip_stream);

/*
 * The reference's traffic-channel dump block (lower_mac/tetra_lower_mac.c:213-231, the input format of the
 * ETSI codec tools): 690 int16 = six frames of marker 0x6b21+i + 114 soft bits (bit 1 -> -127, bit 0 -> +127;
 * 432 bits in total, the rest 0), made from the descrambled type-4 bits a traffic block is delivered with
 * (struct tgpu_unitdata.type4 / type4_len).  For a 216-bit half-slot block the reference reads bits 216..431
 * from an uninitialised local array; here they are bit 0.
 */
void tgpu_traffic_block(const uint8_t *type4, unsigned int len, int16_t out[690]);
/*
 * The same for a whole decoded batch on the device (k_traffic, csrc/tg_traffic.hip): what tp_sap_udata_ind() does with the
 * blocks of a burst the upper MAC has marked as traffic (tetra_lower_mac.c:194-241).  d_traffic = byte per slot in the shape
 * tgpu_gsmtap_batch() takes: bit 0 = the burst is a traffic burst (cur_burst.is_traffic), bit 1 = its second block was stolen
 * (cur_burst.blk2_stolen: it is signalling after all).  For a flagged slot the batch decoded:
 *   NORM_1          the SCH/F block is dumped: 432 descrambled type-4 bits
 *   NORM_2 / SYNC   the second block (BLK2 / SB2: the reference's test is blk_num == BLK_2) is dumped unless bit 1 is set:
 *                   216 descrambled type-4 bits; the first block stays decoded (a stolen BLK1 is signalling)
 * d_type4 (optional): 432 bytes per slot, the dumped block's descrambled type-4 bits, one per byte (struct tgpu_unitdata.type4
 * of the callback path; bytes past the block's length are 0).  d_blocks (optional): 690 int16 per slot, the dump block of
 * tgpu_traffic_block().  d_lens: bits dumped per slot (432, 216, or 0: nothing of this slot was dumped) -- cleared and
 * written for all of the batch's slots.  The records are marked: TGPU_FLAG_TRAFFIC in the flags byte and crc_ok = 0 for
 * the dumped block (the reference does not decode or indicate it; the type-1 bits the batch decoded speculatively stay
 * where they are), TGPU_FLAG_BLK1_STOLEN for a traffic NORM_2 burst; wire records carry the same flags (tgpu_wire_unpack()
 * clears the dumped block's crc_ok).  Slot batches (tgpu_plan_load) and stream batches (tgpu_sync_*: slot = grid slot) alike;
 * not for block-mode plans (the tp_sap_udata_ind() seam has the flags at hand and hands traffic blocks to the callback).
 *   tgpu_plan_set_traffic()  every execute / stream batch of the plan ends with this stage (d_traffic NULL: off).  For a
 *                            caller that knows its traffic slots beforehand (an assigned timeslot of a call).
 *   tgpu_plan_traffic()      the stage on its own, on the batch the plan executed last -- for a caller that reads the
 *                            usage markers out of the batch's own AACH blocks first.  d_rec = the batch's records.
 */
#define TGPU_FLAG_NONBINARY   0x01	/* record flags byte: a byte other than 0 / 1 in the slot */
#define TGPU_FLAG_TRAFFIC     0x02
#define TGPU_FLAG_BLK1_STOLEN 0x04
int tgpu_plan_set_traffic(struct tgpu_plan *plan, const uint8_t *d_traffic, uint8_t *d_type4, int16_t *d_blocks, uint16_t *d_lens);
int tgpu_plan_traffic(struct tgpu_plan *plan, const uint8_t *d_traffic, uint8_t *d_rec, uint8_t *d_type4, int16_t *d_blocks,
		      uint16_t *d_lens, void *hip_stream);

/*
 * The tetra_burst_rx_cb() seam (phy/tetra_burst.c:341-379: void tetra_burst_rx_cb(const uint8_t *burst,
 * unsigned int len, enum tetra_train_seq type, void *priv)) for a host that keeps the reference's own
 * tetra_burst_sync.c: hand over the 510 bits of one burst and its training-sequence type; the burst is
 * queued, decoded with its batch and delivered like a burst found by this library's tetra_burst_sync_in().
 * tn_steps = tetra_tdma_time_add_tn(&t_phy_state.time, 1) calls since the previous hand-over (the
 * reference's synchroniser makes one per 510-bit step while LOCKED, phy/tetra_burst_sync.c:113, also for
 * steps that deliver no burst).  The library keeps t_phy_state.time itself (SYNC PDUs set it, later than in
 * the reference because decoding is deferred; the delivered tdma_time values are the reference's).
 * Types other than SYNC / NORM_1 / NORM_2 are ignored like the reference's switch, their steps count.
 */
int tgpu_channel_burst_rx(struct tgpu_channel *ch, const uint8_t *burst, unsigned int len,
			  int /* enum tetra_train_seq */ type, uint32_t tn_steps);

/* Deliver records decoded through the plan API (slot table from tgpu_sync_stream(), h_rec = host copy of
 * the nslots records, h_stream = host copy of the stream) to the channel's callback: the same in-order
 * replay as tgpu_channel_flush().  tgpu_channel_scramb_init() is the carry-in code for tgpu_plan_load(). */
int tgpu_channel_deliver(struct tgpu_channel *ch, uint32_t n, const struct tgpu_sync_slot *slots,
			 const uint8_t *h_stream, const uint8_t *h_rec);
int tgpu_channel_scramb_init(const struct tgpu_channel *ch, uint32_t *code);
/* tgpu_plan_load() for one channel, reading the slot table of tgpu_sync_stream() in place */
int tgpu_plan_load_slots(struct tgpu_plan *plan, uint32_t nslots, const struct tgpu_sync_slot *slots,
			 uint32_t scramb_init);

/* the two halves of tgpu_sync_stream(): the GPU classification of 'nslots' grid slots starting at
 * 'anchor' (one word per slot + optionally one uint16 SYNC-sequence summary per slot, layouts in
 * csrc/tg_layout.h) and the host walk (cls / ysum may be NULL: slots are then settled with
 * tetra_find_train_seq() on the bytes, re-lock searches scan the bytes -- no GPU needed) */
int tgpu_sync_classify(struct tgpu_engine *eng, const uint8_t *d_stream, uint64_t len, uint32_t chunk,
		       uint64_t anchor, uint32_t nslots, uint32_t *h_cls, uint16_t *h_ysum, void *hip_stream);
int tgpu_sync_walk(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor,
		   const uint32_t *cls, const uint16_t *ysum, uint32_t ncls, uint32_t flags,
		   struct tgpu_sync_result *out);
/* the same walk with the stream-mode kernels' third output (k_cls_plain, what the grid / multi-channel calls use
 * internally): plain = one bit per grid slot, bit i set iff (cls[i] & 0x03ffffff) is a training sequence of type
 * SYNC / NORM_1 / NORM_2 at its nominal offset (214 / 244 / 244) with no flag -- the steady state then reads 32 slots
 * per word of it.  plain == NULL: tgpu_sync_walk(). */
int tgpu_sync_walk_plain(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor, const uint32_t *cls,
			 const uint16_t *ysum, const uint32_t *plain, uint32_t ncls, uint32_t flags, struct tgpu_sync_result *out);

/*
 * The walk in the form the device runs it (k_walk: one lane per grid slot that is no plain delivery, reachability by
 * pointer doubling -- csrc/tg_walk_core.h), executed phase by phase on the host.  Test aid: same outputs as
 * tgpu_sync_walk_plain(..., TGPU_SYNC_GRID | TGPU_SYNC_NO_BURST_EVENTS) whenever *status comes back 0; *status = 1
 * where the device form hands the channel to the host walk (*why: the TGW_WHY_* reason).  chunk: a power of two
 * inside the closed form's range; anchor: the stream's first lock (as tgpu_sync_stream() reports it).
 */
int tgpu_sync_walk_emul(const uint8_t *h_stream, uint64_t len, uint32_t chunk, uint64_t anchor, const uint32_t *cls,
			const uint16_t *ysum, const uint32_t *plain, uint32_t ncls, struct tgpu_sync_result *out, int *status,
			int *why);

/* ------------------------------------------------------------------------- */
/* 3. synthetic downlink generator (TX side of the same chain; host, multi-threaded) */
/* ------------------------------------------------------------------------- */
struct tgpu_synth_cfg {
	uint64_t seed;
	uint32_t scramb_init;	/* code for every block except SB1 */
	uint16_t mcc, mnc;
	uint8_t cc;
	double ber;		/* i.i.d. flips inside the coded fields only */
	int null_pdu_header;	/* 1: first 16 payload bits = MAC-RESOURCE null-address header */
};

/* n slots of the given types (enum tetra_train_seq) into out[n*510]; payload bits of
 * slot i come from splitmix64(seed + i).  Returns 0. */
int tgpu_synth_slots(const struct tgpu_synth_cfg *cfg, const uint8_t *types, size_t n,
		     uint8_t *out, uint8_t *type1_out /* n*288 or NULL */);

#ifdef __cplusplus
}
#endif
#endif
