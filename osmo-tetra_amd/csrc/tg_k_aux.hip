/*
 * tg_k_aux.hip -- tables and init, k_cls_plain, k_reorder, k_fill_*, k_masks, k_grid_*, k_gsmtap, k_stages, the device-walk mid stages
 * (one of the four HIP units of the library: tg_dev.h has the map)
 */
#include "tg_dev.h"


/* one bit per grid slot: the classification word alone says "delivered" (a training sequence of the right type at its
 * nominal offset, no EARLY21 / NONBINARY flag) -- what the host walk's steady state tests, 32 slots to a word */
__global__ __launch_bounds__(256)
void k_cls_plain(const uint32_t *__restrict__ cls, uint32_t n, uint32_t *__restrict__ plain)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	const uint32_t v = i < n ? cls[i] & 0x03ffffffu : 0xffu;
	const bool ok = v == (TG_BURST_SYNC | TG_SYNC_TRAIN_OFF << 8) || v == (TG_BURST_NORM_1 | TG_NORM_TRAIN_OFF << 8) ||
			v == (TG_BURST_NORM_2 | TG_NORM_TRAIN_OFF << 8);
	const unsigned long long b = __ballot(ok);
	const uint32_t lane = threadIdx.x & 63, w = i >> 5;
	if (lane == 0 && 32 * w < n)
		plain[w] = (uint32_t)b;
	if (lane == 32 && 32 * w < n)
		plain[w] = (uint32_t)(b >> 32);
}

extern "C" int tgk_cls_plain(const uint32_t *d_cls, uint32_t n, uint32_t *d_plain, void *stream)
{
	if (!n)
		return 0;
	hipLaunchKernelGGL(k_cls_plain, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_cls, n, d_plain);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* k_reorder: a fixed index map applied to every block of a batch (ACELP re-ordering, tg_reorder.c)  */
/* ------------------------------------------------------------------------- */
/* out[b][j] = in[b][src[j]] where src[j] >= 0; destinations without a source keep what d_out held (the reference's
 * behaviour for a table that names a position never, lower_mac/tch_reordering.c:94-117).  One lane per output byte:
 * stores are consecutive, loads stay inside the block's row. */
__global__ __launch_bounds__(256)
void k_reorder(const uint8_t *__restrict__ in, unsigned long long nblocks, uint32_t nbits, const int32_t *__restrict__ src,
	       uint8_t *__restrict__ out)
{
	const unsigned long long total = nblocks * nbits;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
		const unsigned long long b = i / nbits;
		const uint32_t j = (uint32_t)(i - b * nbits);
		const int32_t s = src[j];
		if (s >= 0)
			out[i] = in[b * nbits + (uint32_t)s];
	}
}

extern "C" int tgk_reorder(const uint8_t *d_in, unsigned long long nblocks, uint32_t nbits, const int32_t *d_src, uint8_t *d_out,
			   void *stream)
{
	if (!nblocks)
		return 0;
	unsigned long long blocks = (nblocks * nbits + 255) / 256;
	if (blocks > 256 * 32)
		blocks = 256 * 32;
	hipLaunchKernelGGL(k_reorder, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, d_in, nblocks, nbits, d_src, d_out);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* scrambling-code forward fill: inclusive running max over (chan<<32 | entry) */
/* ------------------------------------------------------------------------- */
#define FILL_BLOCK 1024

/*
 * A decoded SYNC slot brings its own mask-table entry only when its code is news: the k-th SYNC slot of the batch is
 * "redundant" when the one before it (same channel) decoded to the same code -- the running maximum then keeps the
 * earlier entry, k_masks never computes this one, and all the slots of a cell share one 160-byte entry (a 1 M-slot
 * recording used to build and read 125 k identical ones).
 */
__device__ __forceinline__ bool sb_redundant(uint32_t k, const uint32_t *sb_ok, const uint32_t *sb_code,
					     const uint32_t *list_sb, const uint32_t *slot_chan)
{
	return k > 0 && sb_ok[k - 1] && sb_code[k] == sb_code[k - 1] && slot_chan[list_sb[k]] == slot_chan[list_sb[k - 1]];
}

__device__ __forceinline__ unsigned long long fill_key(uint32_t i, const uint32_t *slot_chan, const int32_t *slot_sbord,
						       const uint32_t *sb_ok, const uint32_t *sb_code, const uint32_t *list_sb,
						       uint32_t nchan)
{
	const uint32_t ch = slot_chan[i];
	const int32_t k = slot_sbord[i];
	/* entry ids: 0 = zero mask, 1+ch = channel carry-in, 1+nchan+k = k-th SYNC slot of the batch */
	uint32_t e = 1 + ch;
	if (k >= 0 && sb_ok[k] && !sb_redundant((uint32_t)k, sb_ok, sb_code, list_sb, slot_chan))
		e = 1 + nchan + (uint32_t)k;
	return ((unsigned long long)ch << 32) | e;
}

__device__ __forceinline__ unsigned long long wave_incl_max(unsigned long long v, uint32_t lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long o = __shfl_up(v, d);
		if (lane >= (uint32_t)d && o > v)
			v = o;
	}
	return v;
}

/* phase 1: per-block maximum */
__global__ __launch_bounds__(FILL_BLOCK)
void k_fill_reduce(const uint32_t *slot_chan, const int32_t *slot_sbord, const uint32_t *sb_ok, const uint32_t *sb_code,
		   const uint32_t *list_sb, uint32_t nchan, uint32_t nslots, unsigned long long *block_max)
{
	__shared__ unsigned long long sm[FILL_BLOCK / 64];
	const uint32_t i = blockIdx.x * FILL_BLOCK + threadIdx.x;
	unsigned long long v = (i < nslots) ? fill_key(i, slot_chan, slot_sbord, sb_ok, sb_code, list_sb, nchan) : 0ull;
	const uint32_t lane = threadIdx.x & 63;
	v = wave_incl_max(v, lane);
	if (lane == 63)
		sm[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long m = 0;
		for (int w = 0; w < FILL_BLOCK / 64; w++)
			m = sm[w] > m ? sm[w] : m;
		block_max[blockIdx.x] = m;
	}
}

/* phase 2: exclusive running max over the block maxima (one workgroup, 1024 maxima per pass) */
__global__ __launch_bounds__(FILL_BLOCK)
void k_fill_scan(unsigned long long *block_max, uint32_t nblocks)
{
	__shared__ unsigned long long sm[FILL_BLOCK / 64];
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long carry = 0;
	for (uint32_t base = 0; base < nblocks; base += FILL_BLOCK) {
		const uint32_t i = base + threadIdx.x;
		const unsigned long long v = (i < nblocks) ? block_max[i] : 0ull;
		const unsigned long long inc = wave_incl_max(v, lane);
		if (lane == 63)
			sm[w] = inc;
		__syncthreads();
		unsigned long long pre = carry, tot = carry;
		for (uint32_t q = 0; q < FILL_BLOCK / 64; q++) {
			if (q < w)
				pre = sm[q] > pre ? sm[q] : pre;
			tot = sm[q] > tot ? sm[q] : tot;
		}
		unsigned long long exc = __shfl_up(inc, 1);
		if (lane == 0)
			exc = 0;
		if (pre > exc)
			exc = pre;
		if (i < nblocks)
			block_max[i] = exc;
		carry = tot;
		__syncthreads();
	}
}

/* phase 3: in-block scan with carry-in, write the mask entry of every slot */
__global__ __launch_bounds__(FILL_BLOCK)
void k_fill_apply(const uint32_t *slot_chan, const int32_t *slot_sbord, const uint32_t *sb_ok, const uint32_t *sb_code,
		  const uint32_t *list_sb, uint32_t nchan, uint32_t nslots, const unsigned long long *block_excl, uint32_t *maskidx)
{
	__shared__ unsigned long long sm[FILL_BLOCK / 64];
	const uint32_t i = blockIdx.x * FILL_BLOCK + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	unsigned long long v = (i < nslots) ? fill_key(i, slot_chan, slot_sbord, sb_ok, sb_code, list_sb, nchan) : 0ull;
	v = wave_incl_max(v, lane);
	if (lane == 63)
		sm[w] = v;
	__syncthreads();
	unsigned long long pre = block_excl[blockIdx.x];
	for (uint32_t q = 0; q < w; q++)
		pre = sm[q] > pre ? sm[q] : pre;
	if (pre > v)
		v = pre;
	if (i < nslots)
		maskidx[i] = (uint32_t)v;	/* chan is monotone along the array, so the max stays inside the channel */
}

/* ------------------------------------------------------------------------- */
/* k_masks: one wavefront per mask-table entry                               */
/* ------------------------------------------------------------------------- */
__global__ __launch_bounds__(256)
void k_masks(const uint32_t *chan_code, uint32_t nchan, const uint32_t *sb_ok, const uint32_t *sb_code,
	     uint32_t nsb, const uint32_t *nsb_dev, const uint32_t *list_sb, const uint32_t *slot_chan, uint32_t *masks)
{
	if (nsb_dev)		/* the number of SYNC slots was counted on the device (nsb = its upper bound) */
		nsb = *nsb_dev;
	/* a wavefront keeps the linear-form masks of its 18 x 64 output bits in registers, looks at 64 entries at a
	 * time and builds the ones a slot can point at: entry 0, the channel carry-ins, and the SYNC slots that decoded
	 * (CRC) to a code other than their predecessor's (sb_redundant) -- per built entry 18 x (and, popcount, ballot)
	 * and one 160-byte store */
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t half = lane >> 5, bit = lane & 31;
	const uint32_t nent = 1 + nchan + nsb;
	uint32_t lin[TG_MW_ROUNDS];
#pragma unroll
	for (int r = 0; r < TG_MW_ROUNDS; r++) {
		const uint16_t pos = c_tab.mask_pos[2 * r + half][bit];
		lin[r] = (pos != 0xffff) ? c_tab.lfsr_lin[pos] : 0u;
	}
	for (uint32_t e0 = wave * 64; e0 < nent; e0 += nwaves * 64) {
		const uint32_t e = e0 + lane;
		uint32_t mycode = 0;
		bool need = e < nent;
		if (e >= 1 && e <= nchan)
			mycode = chan_code[e - 1];
		else if (e > nchan && e < nent) {
			const uint32_t k = e - 1 - nchan;
			need = sb_ok[k] && !sb_redundant(k, sb_ok, sb_code, list_sb, slot_chan);
			mycode = sb_code[k];
		}
		unsigned long long todo = __ballot(need);
		while (todo) {
			const uint32_t l = (uint32_t)__builtin_ctzll(todo);
			todo &= todo - 1;
			const uint32_t code = __builtin_amdgcn_readlane(mycode, l);
			uint32_t myword = 0;
#pragma unroll
			for (int r = 0; r < TG_MW_ROUNDS; r++) {
				const unsigned long long bal = __ballot(__popc(code & lin[r]) & 1);
				myword = (lane == (uint32_t)(2 * r)) ? (uint32_t)bal : myword;
				myword = (lane == (uint32_t)(2 * r + 1)) ? (uint32_t)(bal >> 32) : myword;
			}
			if (lane == TG_MW_CODE)
				myword = code;
			if (lane < TG_MASK_WORDS)
				masks[(size_t)(e0 + l) * TG_MASK_WORDS + lane] = myword;
		}
	}
}


/* ------------------------------------------------------------------------- */
/* k_copy16: a small block between device memory and mapped host memory        */
/* ------------------------------------------------------------------------- */
/* What a device-walk batch hands back (summaries, eager events, delivered bitmap: ~140 KB) goes to the host's mapped, coherent
 * buffer by stores of this kernel instead of hipMemcpyAsync: the runtime's copy path now and then holds the CALLING thread
 * for the 7 ms the stream's queued work takes (tools/experiments/bracket_times.py: one launch call in twenty, always the
 * copy down) -- a launch cannot. */
__global__ __launch_bounds__(256)
void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256)
		dst[i] = src[i];
}

extern "C" int tgk_copy16(const void *d_src, void *d_dst, size_t nbytes, void *stream)
{
	const uint32_t n16 = (uint32_t)((nbytes + 15) / 16);
	if (!n16)
		return 0;
	uint32_t blocks = (n16 + 255) / 256;
	if (blocks > 64)
		blocks = 64;
	hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_src, (uint4 *)d_dst, n16);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* the constant tables and the process-wide init                              */
/* ------------------------------------------------------------------------- */
static uint32_t lfsr_next(uint32_t *st)
{
	/* Fibonacci LFSR of lower_mac/tetra_scramb.c:34-50, taps 32 26 23 22 16 12 11 10 8 7 5 4 2 1 */
	static const int taps[14] = { 32, 26, 23, 22, 16, 12, 11, 10, 8, 7, 5, 4, 2, 1 };
	uint32_t s = *st, fb = 0;
	for (int i = 0; i < 14; i++)
		fb ^= s >> (32 - taps[i]);
	fb &= 1;
	*st = (s >> 1) | (fb << 31);
	return fb;
}

static void build_tables(tg_const_tables *t)
{
	memset(t, 0, sizeof(*t));
	const int btypes[3] = { TG_BURST_NORM_1, TG_BURST_NORM_2, TG_BURST_SYNC };
	for (int x = 0; x < 3; x++)
		for (int w = 0; w < TG_PACKED_WORDS; w++)
			for (int p = 0; p < 32; p++) {
				int o = tg_packed_src(btypes[x], w, p);
				t->front_src[x][w][p] = (o < 0) ? 0xffff : (uint16_t)o;
			}
	/* mask layout: which LFSR output index scrambles each packed bit */
	for (int w = 0; w < TG_MASK_WORDS; w++)
		for (int p = 0; p < 32; p++) {
			int pos = -1;
			if (w < TG_MW_216)
				pos = tg_codeword_src(TG_KIND_432, w - TG_MW_432, p);
			else if (w < TG_MW_BBK)
				pos = tg_codeword_src(TG_KIND_216, w - TG_MW_216, p);
			else if (w == TG_MW_BBK)
				pos = (p < 30) ? p : -1;
			else if (w >= TG_MW_168 && w < TG_MW_168 + 7)
				pos = tg_codeword_src(TG_KIND_168, w - TG_MW_168, p);
			t->mask_pos[w][p] = (pos < 0) ? 0xffff : (uint16_t)pos;
		}
	/* block mode: code-word bit -> type-5 bit of a block handed over on its own */
	for (int x = 0; x < TG_NBLKTYPES; x++)
		for (int w = 0; w < TG_PACKED_WORDS; w++)
			for (int p = 0; p < 32; p++) {
				int o = -1;
				if (x == TG_BLK_BBK)
					o = (w == TG_PW_BBK && p < 30) ? p : -1;
				else if (w < tg_kind_nblk(x) / 2)
					o = tg_codeword_src(x, w, p);
				t->blk_src[x][w][p] = (o < 0) ? 0xffff : (uint16_t)o;
			}
	/* linear form of the LFSR: run it on the 32 unit vectors */
	for (int b = 0; b < 32; b++) {
		uint32_t st = 1u << b;
		for (int n = 0; n < 432; n++)
			if (lfsr_next(&st))
				t->lfsr_lin[n] |= 1u << b;
	}
	/* SB1 mask for init = 3 (lower_mac/tetra_scramb.h:14) */
	{
		uint8_t seq[120];
		uint32_t st = 3;
		for (int n = 0; n < 120; n++)
			seq[n] = (uint8_t)lfsr_next(&st);
		for (int d = 0; d < 5; d++)
			for (int p = 0; p < 32; p++) {
				int j = tg_codeword_src(TG_KIND_SB1, d, p);
				if (j >= 0 && seq[j])
					t->sb1_mask[d] |= 1u << p;
			}
	}
	{
		static const int nblk_of[3] = { 10, 18, 36 };	/* TG_KIND_SB1, _216, _432 */
		for (int kind = 0; kind < 3; kind++) {
			const int nbits = 8 * (nblk_of[kind] - 1) + 4;
			uint16_t c = 0xffff;
			for (int i = 0; i < nbits; i++)
				c = tg_crc16_step_bits(c, 0, 1);
			t->crc_aff[kind] = c;
			for (int i = 0; i < 288; i++) {
				uint16_t v = 0;
				if (i < nbits) {
					v = tg_crc16_step_bits(0, 1, 1);
					for (int k = i + 1; k < nbits; k++)
						v = tg_crc16_step_bits(v, 0, 1);
				}
				t->crc_lin[kind][i] = v;
			}
		}
		t->crc_aff[3] = 0;
	}
	tg_crc16_make_table(t->crc_lsb);
	for (int x = 0; x < 256; x++) {
		int rv = 0;
		for (int i = 0; i < 8; i++)
			if (x & (1 << i))
				rv |= 0x80 >> i;
		t->crc_msb[x] = t->crc_lsb[rv];
	}
}

extern "C" int tgk_upload_aux(const tg_const_tables *host)
{
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), host, sizeof(*host)));
	return 0;
}

extern "C" int tgk_init(void)
{
	static tg_const_tables host;
	build_tables(&host);
	int rc = tgk_upload_front(&host);
	if (!rc)
		rc = tgk_upload_trellis(&host);
	if (!rc)
		rc = tgk_upload_aux(&host);
	if (!rc)
		rc = tgk_upload_slot(&host);
	return rc;
}

#ifdef TG_TRACE
extern "C" int tgk_trace_read(void *out, unsigned int *n, int reset)
{
	unsigned int a = 0, b = 0, c = 0;
	int rc = tgk_trace_read_front(out, &a, reset);
	if (!rc)
		rc = tgk_trace_read_trellis(out ? (tg_trace_rec *)out + a : NULL, &b, reset);
	if (!rc)
		rc = tgk_trace_read_slot(out ? (tg_trace_rec *)out + a + b : NULL, &c, reset);
	if (n)
		*n = a + b + c;
	return rc;
}
#endif

extern "C" int tgk_fill(const uint32_t *d_slot_chan, const int32_t *d_slot_sbord, const uint32_t *d_sb_ok, const uint32_t *d_sb_code,
			const uint32_t *d_list_sb, uint32_t nchan, uint32_t nslots, unsigned long long *d_block_tmp,
			uint32_t *d_maskidx, void *stream)
{
	if (!nslots)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const uint32_t nblocks = (nslots + FILL_BLOCK - 1) / FILL_BLOCK;
	hipLaunchKernelGGL(k_fill_reduce, dim3(nblocks), dim3(FILL_BLOCK), 0, s, d_slot_chan, d_slot_sbord, d_sb_ok, d_sb_code, d_list_sb,
			   nchan, nslots, d_block_tmp);
	hipLaunchKernelGGL(k_fill_scan, dim3(1), dim3(FILL_BLOCK), 0, s, d_block_tmp, nblocks);
	hipLaunchKernelGGL(k_fill_apply, dim3(nblocks), dim3(FILL_BLOCK), 0, s, d_slot_chan, d_slot_sbord, d_sb_ok, d_sb_code, d_list_sb,
			   nchan, nslots, d_block_tmp, d_maskidx);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* stream mode: the plan's per-slot arrays and item lists, built on the device  */
/* ------------------------------------------------------------------------- */
/*
 * Grid slot g is decoded iff the host walk marked it delivered (bit g of 'bits'); its burst type is the
 * classification word's.  Lists keep slot order (the forward fill of the scrambling code relies on the
 * SYNC ordinals growing with g): counts per 1024-slot block, exclusive scan over the blocks, emit.
 * blk[] : 3 words per block (sb, 216-items, 432-items), turned into exclusive bases in place; the three
 * totals follow at blk[3 * nblocks].
 */
#define GRID_BLOCK 1024

__device__ __forceinline__ uint32_t grid_type(const uint32_t *cls, const uint32_t *bits, uint32_t g, uint32_t n)
{
	if (g >= n || !((bits[g >> 5] >> (g & 31)) & 1))
		return TG_BURST_NONE;
	return cls[g] & 0xff;
}

__global__ __launch_bounds__(GRID_BLOCK)
void k_grid_count(const uint32_t *__restrict__ cls, const uint32_t *__restrict__ bits, uint32_t n, uint32_t *__restrict__ blk)
{
	__shared__ uint32_t sm[GRID_BLOCK / 64][3];
	const uint32_t g = blockIdx.x * GRID_BLOCK + threadIdx.x;
	const uint32_t t = grid_type(cls, bits, g, n);
	const uint32_t nsb = __builtin_popcountll(__ballot(t == TG_BURST_SYNC));
	const uint32_t nn2 = __builtin_popcountll(__ballot(t == TG_BURST_NORM_2));
	const uint32_t nn1 = __builtin_popcountll(__ballot(t == TG_BURST_NORM_1));
	if ((threadIdx.x & 63) == 0) {
		sm[threadIdx.x >> 6][0] = nsb;
		sm[threadIdx.x >> 6][1] = nsb + 2 * nn2;
		sm[threadIdx.x >> 6][2] = nn1;
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		uint32_t a = 0;
		for (int w = 0; w < GRID_BLOCK / 64; w++)
			a += sm[w][threadIdx.x];
		blk[3 * blockIdx.x + threadIdx.x] = a;
	}
}

__global__ __launch_bounds__(GRID_BLOCK)
void k_grid_scan(uint32_t *blk, uint32_t nblocks)
{
	/* exclusive prefix sums of the three per-block counts, 1024 blocks per pass, totals behind the last block */
	__shared__ uint32_t sm[GRID_BLOCK / 64][3];
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t carry[3] = { 0, 0, 0 };
	for (uint32_t base = 0; base < nblocks; base += GRID_BLOCK) {
		const uint32_t i = base + threadIdx.x;
		uint32_t v[3], inc[3];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			v[c] = (i < nblocks) ? blk[3 * i + c] : 0u;
			inc[c] = v[c];
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t o = __shfl_up(inc[c], d);
				if (lane >= (uint32_t)d)
					inc[c] += o;
			}
			if (lane == 63)
				sm[w][c] = inc[c];
		}
		__syncthreads();
#pragma unroll
		for (int c = 0; c < 3; c++) {
			uint32_t pre = carry[c], tot = carry[c];
			for (uint32_t q = 0; q < GRID_BLOCK / 64; q++) {
				if (q < w)
					pre += sm[q][c];
				tot += sm[q][c];
			}
			if (i < nblocks)
				blk[3 * i + c] = pre + inc[c] - v[c];
			carry[c] = tot;
		}
		__syncthreads();
	}
	if (threadIdx.x < 3)
		blk[3 * nblocks + threadIdx.x] = carry[threadIdx.x];
}

__global__ __launch_bounds__(GRID_BLOCK)
void k_grid_emit(const uint32_t *__restrict__ cls, const uint32_t *__restrict__ bits, uint32_t n,
		 const uint32_t *__restrict__ blk, uint32_t *__restrict__ slot_chan, int32_t *__restrict__ slot_sbord,
		 uint32_t *__restrict__ list_sb, uint32_t *__restrict__ list_216, uint32_t *__restrict__ list_432,
		 const tg_chan_ent *__restrict__ chan, uint32_t nchan)
{
	__shared__ uint32_t sm[GRID_BLOCK / 64][3];
	__shared__ uint32_t s_gbase[64];
	if (threadIdx.x < 64)
		s_gbase[threadIdx.x] = (chan && threadIdx.x < nchan) ? chan[threadIdx.x].gbase : 0xffffffffu;
	const uint32_t g = blockIdx.x * GRID_BLOCK + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t t = grid_type(cls, bits, g, n);
	const unsigned long long msb = __ballot(t == TG_BURST_SYNC), mn2 = __ballot(t == TG_BURST_NORM_2);
	const unsigned long long mn1 = __ballot(t == TG_BURST_NORM_1);
	const unsigned long long below = (1ull << lane) - 1;
	if (lane == 0) {
		sm[w][0] = __builtin_popcountll(msb);
		sm[w][1] = __builtin_popcountll(msb) + 2 * __builtin_popcountll(mn2);
		sm[w][2] = __builtin_popcountll(mn1);
	}
	__syncthreads();
	uint32_t bsb = blk[3 * blockIdx.x], b216 = blk[3 * blockIdx.x + 1], b432 = blk[3 * blockIdx.x + 2];
	for (uint32_t q = 0; q < w; q++) {
		bsb += sm[q][0];
		b216 += sm[q][1];
		b432 += sm[q][2];
	}
	const uint32_t psb = bsb + __builtin_popcountll(msb & below);
	const uint32_t p216 = b216 + __builtin_popcountll(msb & below) + 2 * __builtin_popcountll(mn2 & below);
	const uint32_t p432 = b432 + __builtin_popcountll(mn1 & below);
	if (g < n) {
		uint32_t c = 0;		/* channels own consecutive slot ranges: the last one that starts at or before g */
		for (uint32_t q = 1; q < nchan; q++)
			c += s_gbase[q] <= g;
		slot_chan[g] = c;
		slot_sbord[g] = (t == TG_BURST_SYNC) ? (int32_t)psb : -1;
	}
	if (t == TG_BURST_SYNC) {
		list_sb[psb] = g;
		list_216[p216] = (g << 1) | 1;	/* SB2 */
	} else if (t == TG_BURST_NORM_2) {
		list_216[p216] = g << 1;
		list_216[p216 + 1] = (g << 1) | 1;
	} else if (t == TG_BURST_NORM_1)
		list_432[p432] = g;
}

extern "C" int tgk_grid_lists(const uint32_t *d_cls, const uint32_t *d_bits, uint32_t n, uint32_t *d_blk,
			      uint32_t *d_slot_chan, int32_t *d_slot_sbord, uint32_t *d_list_sb, uint32_t *d_list_216,
			      uint32_t *d_list_432, const struct tg_chan_ent *d_chan, uint32_t nchan, void *stream)
{
	if (!n)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const uint32_t nblocks = (n + GRID_BLOCK - 1) / GRID_BLOCK;
	hipLaunchKernelGGL(k_grid_count, dim3(nblocks), dim3(GRID_BLOCK), 0, s, d_cls, d_bits, n, d_blk);
	hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(GRID_BLOCK), 0, s, d_blk, nblocks);
	hipLaunchKernelGGL(k_grid_emit, dim3(nblocks), dim3(GRID_BLOCK), 0, s, d_cls, d_bits, n, d_blk, d_slot_chan, d_slot_sbord,
			   d_list_sb, d_list_216, d_list_432, d_chan, nchan);
	return (int)hipGetLastError();
}

extern "C" int tgk_masks_dev(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_sb_ok,
			     const uint32_t *d_sb_code, uint32_t nsb, const uint32_t *d_nsb, const uint32_t *d_list_sb,
			     const uint32_t *d_slot_chan, uint32_t *d_masks, void *stream)
{
	const uint32_t nent = 1 + nchan + nsb;
	uint32_t blocks = ((nent + 63) / 64 + 3) / 4;	/* a wave per 64 entries */
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(k_masks, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_chan_code, nchan, d_sb_ok, d_sb_code, nsb,
			   d_nsb, d_list_sb, d_slot_chan, d_masks);
	return (int)hipGetLastError();
}

extern "C" int tgk_masks(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_sb_ok,
			 const uint32_t *d_sb_code, uint32_t nsb, const uint32_t *d_list_sb, const uint32_t *d_slot_chan,
			 uint32_t *d_masks, void *stream)
{
	return tgk_masks_dev(d_chan_code, nchan, d_sb_ok, d_sb_code, nsb, NULL, d_list_sb, d_slot_chan, d_masks, stream);
}

/* ------------------------------------------------------------------------- */
/* k_gsmtap: GSMTAP messages of a decoded batch          */
/* ------------------------------------------------------------------------- */
/*
 * What the reference's upper MAC sends for every CRC-OK block it is indicated (tetra_upper_mac.c:483-488 ->
 * tetra_gsmtap.c:31-63): 16-byte GSMTAP v2 header (type TETRA_I1, timeslot tn - 1, frame number ((hn 60) + mn) 18 + fn in
 * network order, channel sub-type) + the block's type-1 bits packed MSB first.  One thread per (slot, block of the
 * burst in the reference's order: SB1 BBK SB2 / BBK BLK1 BLK2 / BBK SCH-F); message k of slot i at msgs + (3 i + k) *
 * TG_GSMTAP_STRIDE, its length in lens[3 i + k] (0: none -- the block failed its CRC, the burst has no such block, the
 * slot was not decoded, or the caller marks the burst as traffic and the block is one the reference dumps instead).
 * times[i] = the PHY clock when the burst comes in (after the time steps of tetra_burst_sync_in()); a SYNC burst whose
 * SB1 passes its CRC sets tn / fn / mn from its PDU for all three of its blocks (tetra_lower_mac.c:291-302, 332).
 * Logical channels as tetra_lower_mac.c:170-173, 303, 315-319: SB1 BSCH, BBK AACH, SCH-F SCH_F, SB2 BNCH in the BNCH
 * frame, else (SB2, NDB) unknown (sub-type 0).  The first indication of a block only: further PDUs of the same block
 * (tetra_lower_mac.c:330-352) depend on the upper MAC's return value -- the host's tgpu_gsmtap_makemsg() with an offset.
 */
#define TG_GSMTAP_STRIDE 52
__global__ __launch_bounds__(256)
void k_gsmtap(const uint8_t *__restrict__ rec, const tg_tdma_time_dev *__restrict__ times, const uint8_t *__restrict__ traffic,
	      uint32_t nslots, uint8_t *__restrict__ msgs, uint8_t *__restrict__ lens)
{
	const uint32_t id = blockIdx.x * 256 + threadIdx.x;
	if (id >= 3 * nslots)
		return;
	const uint32_t i = id / 3, k = id % 3;
	const uint8_t *r = rec + (size_t)i * TG_REC_BYTES;
	uint8_t *m = msgs + (size_t)id * TG_GSMTAP_STRIDE;
	const uint32_t type = r[TG_REC_TYPE];
	/* block k of the burst: 0 SB1 / 1 BBK / 2 first block / 3 second block, 4 none */
	uint32_t what = 4;
	if (type == TG_BURST_SYNC)
		what = k == 0 ? 0u : k == 1 ? 1u : 3u;
	else if (type == TG_BURST_NORM_2)
		what = k == 0 ? 1u : k == 1 ? 2u : 3u;
	else if (type == TG_BURST_NORM_1)
		what = k == 0 ? 1u : k == 1 ? 2u : 4u;
	const uint8_t *bits = r + TG_REC_BBK;
	uint32_t nbits = 14, ok = 1, sub = 2 /* GSMTAP_TETRA_AACH */;
	tg_tdma_time_dev tm = times[i];
	if (type == TG_BURST_SYNC && r[TG_REC_CRC_OK]) {
		const uint32_t f0 = *(const uint32_t *)(r + TG_REC_SBF0);
		tm.tn = (f0 >> 8) & 0xff;
		tm.fn = (f0 >> 16) & 0xff;
		tm.mn = f0 >> 24;
	}
	const bool is_traffic = traffic && traffic[i];
	if (what == 0) {
		bits = r + TG_REC_BITS1;
		nbits = 60;
		ok = r[TG_REC_CRC_OK];
		sub = 1;	/* BSCH */
	} else if (what == 2) {
		bits = r + TG_REC_BITS1;
		nbits = type == TG_BURST_NORM_1 ? 268 : 124;
		ok = r[TG_REC_CRC_OK];
		sub = type == TG_BURST_NORM_1 ? 5u : 0u;	/* SCH_F; an NDB half has no channel yet (tetra_lower_mac.c:312) */
		if (is_traffic && type == TG_BURST_NORM_1)
			what = 4;			/* dumped, not indicated (tetra_lower_mac.c:198) */
	} else if (what == 3) {
		bits = r + TG_REC_BITS2;
		nbits = 124;
		ok = r[TG_REC_CRC_OK + 1];
		sub = (type == TG_BURST_SYNC && tm.fn == 18 && tm.tn == 4 - ((tm.mn + 3) % 4)) ? 6u : 0u;	/* BNCH (:122-127, 170) */
		if (is_traffic && (traffic[i] & 2) == 0)
			what = 4;			/* second block of a traffic slot that was not stolen */
	}
	if (what == 4 || !ok) {
		lens[id] = 0;
		return;
	}
	const uint32_t fn = ((tm.hn * 60u) + tm.mn) * 18u + tm.fn;
	const uint32_t nbytes = (nbits + 7) >> 3;
	m[0] = 2;	/* GSMTAP_VERSION */
	m[1] = 4;	/* header length in words */
	m[2] = 5;	/* GSMTAP_TYPE_TETRA_I1 */
	m[3] = (uint8_t)(tm.tn - 1);
	m[4] = m[5] = 0;
	m[6] = m[7] = 0;
	m[8] = (uint8_t)(fn >> 24);
	m[9] = (uint8_t)(fn >> 16);
	m[10] = (uint8_t)(fn >> 8);
	m[11] = (uint8_t)fn;
	m[12] = (uint8_t)sub;
	m[13] = m[14] = m[15] = 0;
	for (uint32_t b = 0; b < nbytes; b++) {
		uint32_t v = 0;
		for (uint32_t q = 0; q < 8; q++)
			if (8 * b + q < nbits && bits[8 * b + q])
				v |= 0x80u >> q;
		m[16 + b] = (uint8_t)v;
	}
	lens[id] = (uint8_t)(16 + nbytes);
}

extern "C" int tgk_gsmtap(const uint8_t *d_rec, const void *d_times, const uint8_t *d_traffic, uint32_t nslots, uint8_t *d_msgs,
			  uint8_t *d_lens, void *stream)
{
	if (!nslots)
		return 0;
	hipLaunchKernelGGL(k_gsmtap, dim3((3 * nslots + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_rec,
			   (const tg_tdma_time_dev *)d_times, d_traffic, nslots, d_msgs, d_lens);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* k_stages: the lower MAC's steps one by one (the reference's DEBUGP lines)   */
/* ------------------------------------------------------------------------- */
/*
 * What the product kernels fold into one gather table and a trellis, as the separate steps of
 * lower_mac/tetra_lower_mac.c:175-254 -- for looking inside a block, and as a second formulation the tests hold the fused
 * kernels against.  One workgroup per block, a thread per bit, one byte per bit:
 *   type4[i]   = type5[i] ^ seq(code)[i]                 (tetra_scramb_bits, :178-186; seq in its linear form)
 *   type3[i]   = type4[(a (i + 1)) mod K]                (block_deinterleave, :245)
 *   type3dp    = 0xff everywhere, then type3[j] at position 8 (j / 3) + {0, 1, 4}[j mod 3]
 *                                                        (tetra_rcpc_depunct with the 2/3 puncturer, :249-250)
 * type2 is the generic trellis' (tgpu_conv_execute on type3), the CRC k_stages_crc's.  a == 0 (BBK): type4 only.
 */
__global__ __launch_bounds__(256)
void k_stages(const uint8_t *__restrict__ type5, const uint32_t *__restrict__ codes, uint32_t fixed_code, uint32_t K, uint32_t a,
	      uint32_t mother_len, uint8_t *__restrict__ type4, uint8_t *__restrict__ type3, uint8_t *__restrict__ type3dp)
{
	__shared__ uint8_t s4[432];
	const size_t blk = blockIdx.x;
	const uint32_t code = codes ? codes[blk] : fixed_code;
	for (uint32_t i = threadIdx.x; i < K; i += 256) {
		const uint8_t b = (uint8_t)((type5[blk * K + i] != 0) ^ (__popc(code & c_tab.lfsr_lin[i]) & 1));
		s4[i] = b;
		type4[blk * K + i] = b;
	}
	if (!a)
		return;
	for (uint32_t i = threadIdx.x; i < mother_len; i += 256)
		type3dp[blk * mother_len + i] = 0xff;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < K; i += 256) {
		const uint8_t b = s4[(a * (i + 1)) % K];
		type3[blk * K + i] = b;
		const uint32_t r = i % 3;
		type3dp[blk * mother_len + 8 * (i / 3) + (r == 2 ? 4 : r)] = b;
	}
}

/* CRC-16 of a block's first n bits, bit by bit (x^16 + x^12 + x^5 + 1, register preset to ones, no final complement:
 * crc16_ccitt_bits() of lower_mac/crc_simple.c; a good block leaves 0x1d0f, crc_simple.h), a thread per block */
__global__ __launch_bounds__(256)
void k_stages_crc(const uint8_t *__restrict__ type2, unsigned long long nblocks, uint32_t type2_len, uint32_t n, uint16_t *__restrict__ crc)
{
	const unsigned long long blk = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	if (blk >= nblocks)
		return;
	uint32_t reg = 0xffff;
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t fb = ((reg >> 15) ^ type2[blk * type2_len + i]) & 1u;
		reg = (reg << 1) & 0xffff;
		if (fb)
			reg ^= 0x1021;
	}
	crc[blk] = (uint16_t)reg;
}

extern "C" int tgk_stages(const uint8_t *d_type5, const uint32_t *d_codes, uint32_t fixed_code, unsigned long long nblocks, uint32_t K,
			  uint32_t a, uint32_t mother_len, uint8_t *d_type4, uint8_t *d_type3, uint8_t *d_type3dp, void *stream)
{
	if (!nblocks)
		return 0;
	if (K > 432 || nblocks > 0x7fffffffull)
		return -1;	/* TGPU_EINVAL */
	hipLaunchKernelGGL(k_stages, dim3((uint32_t)nblocks), dim3(256), 0, (hipStream_t)stream, d_type5, d_codes, fixed_code, K, a,
			   mother_len, d_type4, d_type3, d_type3dp);
	return (int)hipGetLastError();
}

extern "C" int tgk_stages_crc(const uint8_t *d_type2, unsigned long long nblocks, uint32_t type2_len, uint32_t n, uint16_t *d_crc,
			      void *stream)
{
	if (!nblocks)
		return 0;
	hipLaunchKernelGGL(k_stages_crc, dim3((uint32_t)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_type2, nblocks,
			   type2_len, n, d_crc);
	return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* device-walk batches: everything between the front end and the trellis kernels in five small launches */
/* ------------------------------------------------------------------------- */
/*
 * The host-walk path builds ordered lists (count / scan / emit), decodes SB1 over the ordered SYNC list, forward-fills
 * the code over all slots (reduce / scan / apply) and builds mask entries per SYNC ordinal: ten launches, ~85 us of small
 * kernels per 1 M slots.  With the walk on the device none of that order is needed:
 *   k_cls_plain2  plain bitmap (as k_cls_plain) + the list of SYNC-classified slots (wave-aggregated append, any order)
 *                 + the channel of every 32-slot word                                    -- before the walk
 *   k_vit<SB1>    over that list, beside the walk (side stream): a block that passes its CRC sets its slot's bit in
 *                 'okbits' and takes a mask-table entry for its code from a small hash table (codes in play are few:
 *                 one per cell), remembered per slot                                     (tg_lb, vit_finish)
 *   k_masks2      the scrambling masks of the entries in use (carry-ins + hash table)
 *   k_lb_scan     per 32-slot word: the latest word at or before it (same channel) that holds a delivered SYNC slot
 *                 with a good SB1 -- one workgroup, running maximum; also every channel's code after the batch
 *   k_lists2      per delivered slot: the mask entry of the latest such SYNC slot at or before it (this slot included:
 *                 an SB1 sets the code for the BBK and SB2 of its own burst, tetra_lower_mac.c:179-186, 291-300), else
 *                 the channel's carry-in; and the slot's items appended to the 216 / 432 lists (wave-aggregated, any
 *                 order -- records are addressed by slot)
 * An undelivered SYNC slot may be decoded (its SB1 costs 84 trellis steps) but never counts: the look-back ANDs okbits
 * with the delivered bitmap.
 */
struct tg_lists_hints {
	uint32_t code[64];	/* k_slot batches: the code every channel's slots were decoded under there (0: none) */
};

/* the scrambling masks of up to 64 codes handed over by value (k_slot's hints: entries hint_base + c of the mask table) */
__global__ __launch_bounds__(64)
void k_masks_list(tg_lists_hints codes, uint32_t n, uint32_t *__restrict__ masks)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t half = lane >> 5, bit = lane & 31;
	uint32_t lin[TG_MW_ROUNDS];
#pragma unroll
	for (int r = 0; r < TG_MW_ROUNDS; r++) {
		const uint16_t pos = c_tab.mask_pos[2 * r + half][bit];
		lin[r] = (pos != 0xffff) ? c_tab.lfsr_lin[pos] : 0u;
	}
	for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
		uint32_t code = 0;
		for (uint32_t c = 0; c < 64; c++)
			code = (c == e) ? codes.code[c] : code;
		uint32_t myword = 0;
#pragma unroll
		for (int r = 0; r < TG_MW_ROUNDS; r++) {
			const unsigned long long bal = __ballot(__popc(code & lin[r]) & 1);
			myword = (lane == (uint32_t)(2 * r)) ? (uint32_t)bal : myword;
			myword = (lane == (uint32_t)(2 * r + 1)) ? (uint32_t)(bal >> 32) : myword;
		}
		if (lane == TG_MW_CODE)
			myword = code;
		if (lane < TG_MASK_WORDS)
			masks[(size_t)e * TG_MASK_WORDS + lane] = myword;
	}
}

extern "C" int tgk_masks_list(const uint32_t *codes, uint32_t n, uint32_t *d_masks_out, void *stream)
{
	if (!n || n > 64)
		return n ? -1 : 0;
	tg_lists_hints h;
	memset(&h, 0, sizeof(h));
	memcpy(h.code, codes, (size_t)n * 4);
	hipLaunchKernelGGL(k_masks_list, dim3(n), dim3(64), 0, (hipStream_t)stream, h, n, d_masks_out);
	return (int)hipGetLastError();
}

#define TG_MID_CHUNKS 4		/* 1024-slot chunks per workgroup of k_cls_plain2 / k_lists2: one atomic per 4096 slots and list */
__global__ __launch_bounds__(1024)
void k_cls_plain2(const uint32_t *__restrict__ cls, uint32_t n, uint32_t *__restrict__ plain, uint32_t *__restrict__ list_sb,
		  uint32_t *__restrict__ cnt_sb, uint8_t *__restrict__ word_chan, const tg_chan_ent *__restrict__ chan, uint32_t nchan,
		  const uint32_t *__restrict__ specbits /* or NULL; k_slot batches: SYNC slots whose SB1 that kernel decoded stay off the list */,
		  uint32_t *__restrict__ specbits_out /* or NULL; k_slot_e batches: bit per grid slot "plain, and its channel has a hint" = what that kernel decodes */,
		  tg_lists_hints hints)
{
	/* (a single word takes ~90 atomics per microsecond: one per wave would be 15 000 of them) */
	__shared__ uint32_t s_cnt[TG_MID_CHUNKS][16], s_base;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned long long sbm[TG_MID_CHUNKS];
	bool sb[TG_MID_CHUNKS];
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		const uint32_t i = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
		const uint32_t w = i >> 5;
		const uint32_t v = i < n ? cls[i] & 0x03ffffffu : 0xffu;
		const bool issb = v == (TG_BURST_SYNC | TG_SYNC_TRAIN_OFF << 8);
		sb[j] = issb && !(specbits && i < n && ((specbits[w] >> (i & 31)) & 1));
		const bool ok = issb || v == (TG_BURST_NORM_1 | TG_NORM_TRAIN_OFF << 8) || v == (TG_BURST_NORM_2 | TG_NORM_TRAIN_OFF << 8);
		const unsigned long long b = __ballot(ok);
		sbm[j] = __ballot(sb[j]);
		if (lane == 0 && 32 * w < n)
			plain[w] = (uint32_t)b;
		if (lane == 32 && 32 * w < n)
			plain[w] = (uint32_t)(b >> 32);
		if (specbits_out) {
			uint32_t c = 0, hc = 0;
			for (uint32_t q = 1; q < nchan; q++)
				c += chan[q].gbase <= i;
			for (uint32_t q = 0; q < nchan; q++)
				hc = (q == c) ? hints.code[q] : hc;
			const unsigned long long sp = __ballot(ok && hc != 0u);
			if (lane == 0 && 32 * w < n)
				specbits_out[w] = (uint32_t)sp;
			if (lane == 32 && 32 * w < n)
				specbits_out[w] = (uint32_t)(sp >> 32);
		}
		/* the wave's 64 slots lie in one or two channels (grids start at multiples of 32) */
		if ((lane == 0 || lane == 32) && 32 * w < n) {
			uint32_t c = 0;
			for (uint32_t q = 1; q < nchan; q++)
				c += chan[q].gbase <= i;
			word_chan[w] = (uint8_t)c;
		}
		if (lane == 0)
			s_cnt[j][wv] = (uint32_t)__builtin_popcountll(sbm[j]);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int q = 0; q < TG_MID_CHUNKS * 16; q++)
			tot += s_cnt[0][q];
		s_base = tot ? atomicAdd(cnt_sb, tot) : 0u;
	}
	__syncthreads();
	uint32_t pos = s_base;
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		uint32_t mine = pos;
		for (uint32_t q = 0; q < 16; q++) {
			if (q < wv)
				mine += s_cnt[j][q];
			pos += s_cnt[j][q];
		}
		if (sb[j])
			list_sb[mine + __builtin_popcountll(sbm[j] & ((1ull << lane) - 1))] = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
	}
}

__global__ __launch_bounds__(256)
void k_masks2(const uint32_t *__restrict__ chan_code, uint32_t nchan, const uint32_t *__restrict__ tbl, uint32_t *__restrict__ masks)
{
	/* as k_masks: a wavefront keeps the linear-form masks of its 18 x 64 output bits in registers and builds, of 64 entries
	 * at a time, the ones in use: entry 0, the channel carry-ins, the occupied slots of the code table */
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t half = lane >> 5, bit = lane & 31;
	const uint32_t nent = 1 + nchan + TG_LB_TBL;
	uint32_t lin[TG_MW_ROUNDS];
#pragma unroll
	for (int r = 0; r < TG_MW_ROUNDS; r++) {
		const uint16_t pos = c_tab.mask_pos[2 * r + half][bit];
		lin[r] = (pos != 0xffff) ? c_tab.lfsr_lin[pos] : 0u;
	}
	for (uint32_t e0 = wave * 64; e0 < nent; e0 += nwaves * 64) {
		const uint32_t e = e0 + lane;
		uint32_t mycode = 0;
		bool need = e < nent;
		if (e >= 1 && e <= nchan)
			mycode = chan_code[e - 1];
		else if (e > nchan && e < nent) {
			mycode = tbl[e - 1 - nchan];
			need = mycode != 0;
		}
		unsigned long long todo = __ballot(need);
		while (todo) {
			const uint32_t l = (uint32_t)__builtin_ctzll(todo);
			todo &= todo - 1;
			const uint32_t code = __builtin_amdgcn_readlane(mycode, l);
			uint32_t myword = 0;
#pragma unroll
			for (int r = 0; r < TG_MW_ROUNDS; r++) {
				const unsigned long long bal = __ballot(__popc(code & lin[r]) & 1);
				myword = (lane == (uint32_t)(2 * r)) ? (uint32_t)bal : myword;
				myword = (lane == (uint32_t)(2 * r + 1)) ? (uint32_t)(bal >> 32) : myword;
			}
			if (lane == TG_MW_CODE)
				myword = code;
			if (lane < TG_MASK_WORDS)
				masks[(size_t)(e0 + l) * TG_MASK_WORDS + lane] = myword;
		}
	}
}

/* latest delivered SYNC slot with a good SB1 at or before grid slot g in g's channel, or 0xffffffff */
__device__ __forceinline__ uint32_t tg_lb_find(uint32_t g, const uint32_t *__restrict__ okbits, const uint32_t *__restrict__ dbits,
					       const uint32_t *__restrict__ prevw, const uint8_t *__restrict__ word_chan)
{
	const uint32_t w = g >> 5;
	uint32_t m = okbits[w] & dbits[w] & (0xffffffffu >> (31 - (g & 31)));
	uint32_t ww = w;
	if (!m) {
		if (!w)
			return 0xffffffffu;
		const uint32_t p = prevw[w - 1];	/* 1 + the latest word <= w - 1 that has one, in that word's channel; 0: none */
		if (!p || word_chan[p - 1] != word_chan[w])
			return 0xffffffffu;
		ww = p - 1;
		m = okbits[ww] & dbits[ww];
	}
	return 32 * ww + 31 - __builtin_clz(m);
}

#define LBS_THREADS 1024
__global__ __launch_bounds__(LBS_THREADS)
void k_lb_scan(const uint32_t *__restrict__ okbits, const uint32_t *__restrict__ dbits, const uint8_t *__restrict__ word_chan,
	       uint32_t nwords, uint32_t *__restrict__ prevw, const tg_chan_ent *__restrict__ chan, uint32_t nchan,
	       const uint32_t *__restrict__ chan_code, const uint32_t *__restrict__ slot_entry, const uint32_t *__restrict__ masks,
	       const uint32_t *__restrict__ tbl, uint32_t *__restrict__ final_code)
{
	/* one workgroup per channel (a channel's words never look into another's): running maximum of (word + 1 if the word has
	 * a delivered good SYNC slot, else 0) over the channel's words, 1024 consecutive words per pass (coalesced loads, the
	 * next pass's in flight during the scan) */
	__shared__ uint32_t sm[LBS_THREADS / 64];
	const uint32_t c = blockIdx.x;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t w0 = chan[c].gbase >> 5, wn = (chan[c].ncls + 31) >> 5;
	(void)nwords;
	uint32_t carry = 0;
	uint32_t w = threadIdx.x;
	uint32_t has = w < wn ? (okbits[w0 + w] & dbits[w0 + w]) : 0u;
	for (uint32_t base = 0; base < wn; base += LBS_THREADS) {
		const uint32_t wnext = base + LBS_THREADS + threadIdx.x;
		const uint32_t hnext = wnext < wn ? (okbits[w0 + wnext] & dbits[w0 + wnext]) : 0u;
		uint32_t inc = has ? w0 + w + 1 : 0u;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t o = __shfl_up(inc, d);
			if (lane >= (uint32_t)d && o > inc)
				inc = o;
		}
		__syncthreads();
		if (lane == 63)
			sm[wv] = inc;
		__syncthreads();
		uint32_t pre = carry, tot = carry;
		for (uint32_t q = 0; q < LBS_THREADS / 64; q++) {
			if (q < wv)
				pre = sm[q] > pre ? sm[q] : pre;
			tot = sm[q] > tot ? sm[q] : tot;
		}
		if (w < wn)
			prevw[w0 + w] = inc > pre ? inc : pre;
		carry = tot;
		w = wnext;
		has = hnext;
	}
	(void)word_chan;
	/* the code in force after the batch: the channel's latest delivered good SYNC slot, else its carry-in */
	if (threadIdx.x == 0) {
		uint32_t code = chan_code[c];
		if (carry) {
			const uint32_t ww = carry - 1;
			const uint32_t m = okbits[ww] & dbits[ww];
			code = masks[(size_t)slot_entry[32 * ww + 31 - __builtin_clz(m)] * TG_MASK_WORDS + TG_MW_CODE];
		}
		final_code[c] = code;
		if (c == 0)
			final_code[64] = tbl[TG_LB_TBL];	/* != 0: the batch had more codes than the table holds */
	}
}

__global__ __launch_bounds__(1024)
void k_lists2(const uint32_t *__restrict__ cls, const uint32_t *__restrict__ dbits, uint32_t n, const uint32_t *__restrict__ okbits,
	      const uint32_t *__restrict__ prevw, const uint8_t *__restrict__ word_chan, const uint32_t *__restrict__ slot_entry,
	      uint32_t *__restrict__ maskidx, uint32_t *__restrict__ list_216, uint32_t *__restrict__ list_432,
	      uint32_t *__restrict__ list_all /* or NULL: the lane-per-slot kernel's items (tg_k_slot.hip): every listed NORM_1 / NORM_2 slot once */,
	      uint32_t *__restrict__ list_sync /* ... and every listed SYNC slot once (a list of their own: waves of one kind run a shorter schedule) */,
	      uint32_t *__restrict__ cnt /* [1]: 216 items, [2]: 432 items, [3]: slots on list_all, [4]: slots on list_sync */,
	      const uint32_t *__restrict__ specbits, tg_lists_hints hints, const uint32_t *__restrict__ chan_code, const uint32_t *__restrict__ tbl,
	      uint32_t nchan)
{
	/* k_slot batches (specbits != NULL): a delivered slot that kernel decoded under its channel's hint is DONE if the code in force at
	 * the slot -- the entry found here -- is the hint's; it then goes onto no list.  Every other delivered slot (decoded under another
	 * code: a channel's first batch, a cell change; or not decoded there at all: the exact pass's slots) is listed as before. */
	__shared__ uint32_t s_c216[TG_MID_CHUNKS][16], s_c432[TG_MID_CHUNKS][16], s_call[TG_MID_CHUNKS][16], s_csb[TG_MID_CHUNKS][16], s_b216, s_b432, s_ball, s_bsb;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long below = (1ull << lane) - 1;
	uint32_t t[TG_MID_CHUNKS], in216[TG_MID_CHUNKS], in432[TG_MID_CHUNKS], inall[TG_MID_CHUNKS], insb[TG_MID_CHUNKS];
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		const uint32_t g = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
		t[j] = TG_BURST_NONE;
		if (g < n && ((dbits[g >> 5] >> (g & 31)) & 1))
			t[j] = cls[g] & 0xff;
		if (t[j] != TG_BURST_NONE) {
			const uint32_t gs = tg_lb_find(g, okbits, dbits, prevw, word_chan);
			const uint32_t wc = word_chan[g >> 5];
			const uint32_t e = gs != 0xffffffffu ? slot_entry[gs] : 1u + wc;
			maskidx[g] = e;
			if (specbits && ((specbits[g >> 5] >> (g & 31)) & 1)) {
				const uint32_t code = e <= nchan ? chan_code[e - 1] : tbl[e - 1 - nchan];	/* (e >= 1: entry 0 is SB1's fixed code, never a slot's) */
				uint32_t hc = 0;
				for (uint32_t c = 0; c < nchan; c++)
					hc = (c == wc) ? hints.code[c] : hc;
				if (code == hc && hc != 0u)
					t[j] = TG_BURST_NONE;
			}
		}
		const unsigned long long msb = __ballot(t[j] == TG_BURST_SYNC), mn2 = __ballot(t[j] == TG_BURST_NORM_2);
		const unsigned long long mn1 = __ballot(t[j] == TG_BURST_NORM_1);
		in216[j] = __builtin_popcountll(msb & below) + 2 * __builtin_popcountll(mn2 & below);
		in432[j] = __builtin_popcountll(mn1 & below);
		inall[j] = __builtin_popcountll((mn2 | mn1) & below);
		insb[j] = __builtin_popcountll(msb & below);
		if (lane == 0) {
			s_c216[j][wv] = __builtin_popcountll(msb) + 2 * __builtin_popcountll(mn2);
			s_c432[j][wv] = __builtin_popcountll(mn1);
			s_call[j][wv] = __builtin_popcountll(mn2 | mn1);
			s_csb[j][wv] = __builtin_popcountll(msb);
		}
	}
	__syncthreads();
	if (threadIdx.x < 2) {		/* one atomic per workgroup and list */
		uint32_t tot = 0;
		for (int q = 0; q < TG_MID_CHUNKS * 16; q++)
			tot += threadIdx.x ? s_c432[0][q] : s_c216[0][q];
		const uint32_t base = tot ? atomicAdd(cnt + 1 + threadIdx.x, tot) : 0u;
		if (threadIdx.x)
			s_b432 = base;
		else
			s_b216 = base;
	}
	if (threadIdx.x == 64 || threadIdx.x == 128) {	/* the listed slots once each: NORM_1 / NORM_2 on one list, SYNC on another */
		const bool sbl = threadIdx.x == 128;
		uint32_t tot = 0;
		for (int q = 0; q < TG_MID_CHUNKS * 16; q++)
			tot += sbl ? s_csb[0][q] : s_call[0][q];
		const uint32_t base = (tot && list_all) ? atomicAdd(cnt + (sbl ? 4 : 3), tot) : 0u;
		if (sbl)
			s_bsb = base;
		else
			s_ball = base;
	}
	__syncthreads();
	uint32_t r216 = s_b216, r432 = s_b432, rall = s_ball, rsb = s_bsb;
#pragma unroll
	for (int j = 0; j < TG_MID_CHUNKS; j++) {
		const uint32_t g = (blockIdx.x * TG_MID_CHUNKS + j) * 1024 + threadIdx.x;
		uint32_t p216 = r216, p432 = r432, pall = rall, psb = rsb;
		for (uint32_t q = 0; q < 16; q++) {
			if (q < wv) {
				p216 += s_c216[j][q];
				p432 += s_c432[j][q];
				pall += s_call[j][q];
				psb += s_csb[j][q];
			}
			r216 += s_c216[j][q];
			r432 += s_c432[j][q];
			rall += s_call[j][q];
			rsb += s_csb[j][q];
		}
		p216 += in216[j];
		p432 += in432[j];
		pall += inall[j];
		psb += insb[j];
		if (list_all && (t[j] == TG_BURST_NORM_2 || t[j] == TG_BURST_NORM_1))
			list_all[pall] = g;
		if (list_all && t[j] == TG_BURST_SYNC)
			list_sync[psb] = g;
		if (t[j] == TG_BURST_SYNC)
			list_216[p216] = (g << 1) | 1;		/* SB2 */
		else if (t[j] == TG_BURST_NORM_2) {
			list_216[p216] = g << 1;
			list_216[p216 + 1] = (g << 1) | 1;
		} else if (t[j] == TG_BURST_NORM_1)
			list_432[p432] = g;
	}
}

extern "C" int tgk_cls_plain2(const uint32_t *d_cls, uint32_t n, uint32_t *d_plain, uint32_t *d_list_sb, uint32_t *d_cnt_sb,
			      uint8_t *d_word_chan, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_specbits,
			      uint32_t *d_specbits_out, const uint32_t *hints, void *stream)
{
	if (!n)
		return 0;
	tg_lists_hints h;
	memset(&h, 0, sizeof(h));
	if (d_specbits_out && hints)
		memcpy(h.code, hints, (size_t)(nchan < 64 ? nchan : 64) * 4);
	hipLaunchKernelGGL(k_cls_plain2, dim3((n + 1024 * TG_MID_CHUNKS - 1) / (1024 * TG_MID_CHUNKS)), dim3(1024), 0, (hipStream_t)stream, d_cls, n, d_plain, d_list_sb, d_cnt_sb,
			   d_word_chan, d_chan, nchan, d_specbits, (d_specbits_out && hints) ? d_specbits_out : NULL, h);
	return (int)hipGetLastError();
}

extern "C" int tgk_masks2(const uint32_t *d_chan_code, uint32_t nchan, const uint32_t *d_tbl, uint32_t *d_masks, void *stream)
{
	const uint32_t nent = 1 + nchan + TG_LB_TBL;
	hipLaunchKernelGGL(k_masks2, dim3(((nent + 63) / 64 + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_chan_code, nchan, d_tbl, d_masks);
	return (int)hipGetLastError();
}

extern "C" int tgk_lb_scan(const uint32_t *d_okbits, const uint32_t *d_dbits, const uint8_t *d_word_chan, uint32_t nwords,
			   uint32_t *d_prevw, const struct tg_chan_ent *d_chan, uint32_t nchan, const uint32_t *d_chan_code,
			   const uint32_t *d_slot_entry, const uint32_t *d_masks, const uint32_t *d_tbl, uint32_t *d_final_code, void *stream)
{
	if (!nwords)
		return 0;
	hipLaunchKernelGGL(k_lb_scan, dim3(nchan), dim3(LBS_THREADS), 0, (hipStream_t)stream, d_okbits, d_dbits, d_word_chan, nwords, d_prevw,
			   d_chan, nchan, d_chan_code, d_slot_entry, d_masks, d_tbl, d_final_code);
	return (int)hipGetLastError();
}

extern "C" int tgk_lists2(const uint32_t *d_cls, const uint32_t *d_dbits, uint32_t n, const uint32_t *d_okbits, const uint32_t *d_prevw,
			  const uint8_t *d_word_chan, const uint32_t *d_slot_entry, uint32_t *d_maskidx, uint32_t *d_list_216,
			  uint32_t *d_list_432, uint32_t *d_list_all, uint32_t *d_list_sync, uint32_t *d_cnt, const uint32_t *d_specbits, const uint32_t *hints,
			  const uint32_t *d_chan_code, const uint32_t *d_tbl, uint32_t nchan, void *stream)
{
	if (!n)
		return 0;
	tg_lists_hints h;
	memset(&h, 0, sizeof(h));
	if (d_specbits && hints)
		memcpy(h.code, hints, (size_t)(nchan < 64 ? nchan : 64) * 4);
	hipLaunchKernelGGL(k_lists2, dim3((n + 1024 * TG_MID_CHUNKS - 1) / (1024 * TG_MID_CHUNKS)), dim3(1024), 0, (hipStream_t)stream, d_cls, d_dbits, n, d_okbits, d_prevw,
			   d_word_chan, d_slot_entry, d_maskidx, d_list_216, d_list_432, d_list_sync ? d_list_all : NULL, d_list_sync, d_cnt, (d_specbits && hints) ? d_specbits : NULL, h, d_chan_code,
			   d_tbl, nchan);
	return (int)hipGetLastError();
}
