/*
 * tg_front_stream_body.h -- the body of the packed-bit stream front end ("k_front_stream"), included by
 *   tg_k_front.hip with TGS_FUSED 0: the kernel k_front_stream<PACKED> -- a wave takes groups wave, wave + nwaves, ... of the grid;
 *   tg_k_slot.hip  with TGS_FUSED 1: the device function slot_front_phase<PACKED> -- a wave takes the sixteen groups of ONE task
 *                  (64 neighbouring grid slots), keeps their packed slots in LDS for the trellis phase that follows in the same
 *                  wave, and notes type and channel of every slot there.
 * One text, so that the two cannot drift apart (the exact pass, the walk and every test read what either writes).
 */
#if !TGS_FUSED
#ifndef TG_STREAM_WPB
#define TG_STREAM_WPB 4	/* waves per workgroup (they share nothing: each has its own staging areas) */
#endif
#define TGS_SADR_LDS TGS_SYNC_LDS
template <bool PACKED>
__global__ __launch_bounds__(64 * TG_STREAM_WPB) __attribute__((amdgpu_waves_per_eu(TG_STREAM_WPE, TG_STREAM_WPE)))
void k_front_stream(const uint8_t *__restrict__ stream, tg_stream_params prm,
		    uint32_t *__restrict__ packed, uint32_t *__restrict__ cls, uint16_t *__restrict__ ysum,
		    uint32_t *__restrict__ defer)
{
	constexpr uint64_t PY = tsq_bits(TSQ_Y), PN = tsq_bits(TSQ_N), PP = tsq_bits(TSQ_P);
	__shared__ __attribute__((aligned(16))) uint32_t s_bits[TG_STREAM_WPB][72];	/* per wave: the group's bit string (68 dwords used; packed ingest: 72, 16 bytes per lane) */
	__shared__ uint32_t s_win[TG_STREAM_WPB][4 * TG_VER_SLOT];	/* per wave: four slots x eight shifted copies of the 512-bit window */
	__shared__ uint32_t s_out[TG_STREAM_WPB][160];	/* per wave: four packed slots on their way out, then their cls / ysum words (+ the idle lanes' dump) */
	/* per wave: the SYNC burst's eight gather addresses of every lane.  One slot in eight is a SYNC burst: its table waits
	 * here (two 16-byte reads in front of such a gather) instead of in eight of the 80 VGPRs six waves per SIMD allow */
	__shared__ __attribute__((aligned(16))) uint32_t s_sadr[TG_STREAM_WPB][TGS_SYNC_LDS ? 64 * 8 : 4];
#else
/* the fused form: one wave = one workgroup = one task of 64 neighbouring grid slots (sixteen groups); its LDS comes from the caller.
 * The sixteen groups' packed slots stay in s_out (80 dwords a group, 20 a slot: the trellis phase reads them as columns), the slots'
 * classification / summary words, burst types (TG_BURST_NONE: not this pass's to settle) and channels beside them.  The SYNC
 * burst's gather addresses stay in registers here (the kernel is compiled for the trellis phase's 168). */
#define TG_STREAM_WPB 1
#define TGS_SADR_LDS 0
struct tg_slot_front_lds {
	uint32_t s_bits[1][72];
	uint32_t s_win[1][4 * TG_VER_SLOT];
	uint32_t s_cls[64], s_ys[64];
	uint8_t s_dtype[64], s_chan[64];
};
template <bool PACKED>
__device__ __forceinline__ void slot_front_phase(const uint8_t *__restrict__ stream, const tg_stream_params &prm,
						  uint32_t *__restrict__ packed, uint32_t *__restrict__ cls, uint16_t *__restrict__ ysum,
						  uint32_t *__restrict__ defer, tg_slot_front_lds &F, uint32_t (&s_out)[1][16 * 80],
						  uint32_t task, uint32_t ntasks)
{
	constexpr uint64_t PY = tsq_bits(TSQ_Y), PN = tsq_bits(TSQ_N), PP = tsq_bits(TSQ_P);
	auto &s_bits = F.s_bits;
	auto &s_win = F.s_win;
	uint32_t s_sadr[1][4];
	(void)s_sadr;
#endif

#if !TGS_FUSED
#ifdef TGS_TIMING
	unsigned long long tgs_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tgs_last = __builtin_amdgcn_s_memtime();
#endif
	TG_TRACE_BEGIN;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wave = blockIdx.x * TG_STREAM_WPB + wib;
	const uint32_t nwaves = gridDim.x * TG_STREAM_WPB;
#else
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = 0;
	const uint32_t wave = task;		/* (the exact pass finds this task's deferred slots where a wave of k_front_stream leaves its own) */
	const uint32_t nwaves = ntasks;
#endif
	const uint32_t col = lane & 15;			/* 32-position column of the lane's slot */
	uint32_t *bits = s_bits[wib];
	uint32_t *win = s_win[wib];
	uint32_t *mo = s_out[wib];		/* (fused: moves on by 80 dwords per group) */

	/* the lane's byte of the packed slot: lanes 0..53 byte l % 3 of code word l / 3, 54 / 55 the lead-in bits of the two
	 * blocks (byte 3 of words 0 and 9), 56..59 the BBK word, 60..63 none; per burst type and round the LDS byte that
	 * carries the wanted bit at its bit 0 */
	const uint32_t ow = lane < 54 ? lane / 3 : lane == 54 ? 0u : lane == 55 ? (uint32_t)TG_PW_BLK2 : (uint32_t)TG_PW_BBK;
	const uint32_t ob = lane < 54 ? lane % 3 : lane < 56 ? 3u : lane - 56;
#if !TGS_FUSED
	const uint32_t obyte = lane < 60 ? 4 * ow + ob : 4 * 88 + (lane - 60);	/* (the idle lanes write behind the staged slots: < 640 with the last slot's offset) */
#else
	const uint32_t obyte = lane < 60 ? 4 * ow + ob : 4 * TG_PW_META + (lane - 60);	/* (fused: the next group's slots lie behind -- the idle lanes write into the slot's meta word, which its owner writes after the gathers) */
#endif
	uint32_t g_adr[3][8];
	/* (the asm block takes LDS addresses as the hardware sees them: the array's offset inside the workgroup's LDS) */
	const uint32_t ver0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)&s_win[0][0];
	{
		/* the lane's 24 table entries = 8 consecutive ushorts of three rows: three 16-byte loads in flight together (one
		 * load and one wait per entry cost every wave ~24 memory latencies before its first group: 186 -> 174 us) */
		uint4 row[3];
#pragma unroll
		for (int x = 0; x < 3; x++)
			row[x] = *(const uint4 *)&c_tab.front_src[x][lane < 60 ? ow : 0][lane < 60 ? 8 * ob : 0];
#pragma unroll
		for (int x = 0; x < 3; x++) {
			const uint32_t w4[4] = { row[x].x, row[x].y, row[x].z, row[x].w };
#pragma unroll
			for (int r = 0; r < 8; r++) {
				const uint32_t o = lane < 60 ? (w4[r >> 1] >> (16 * (r & 1))) & 0xffffu : 0xffffu;
				g_adr[x][r] = ver0 + wib * (4 * TG_VER_SLOT * 4) + (o == 0xffff ? 64u : (o & 7) * (TG_VER_STRIDE * 4) + (o >> 3));
			}
		}
#if TGS_SADR_LDS
		*(uint4 *)&s_sadr[wib][8 * lane] = make_uint4(g_adr[2][0], g_adr[2][1], g_adr[2][2], g_adr[2][3]);
		*(uint4 *)&s_sadr[wib][8 * lane + 4] = make_uint4(g_adr[2][4], g_adr[2][5], g_adr[2][6], g_adr[2][7]);
#endif
#pragma unroll
		for (int x = 0; x < (TGS_SADR_LDS ? 2 : 3); x++)
#pragma unroll
			for (int r = 0; r < 8; r++)
				asm volatile("" : "+v"(g_adr[x][r]));	/* the whole address in the register: the slot's offset is the immediate */
	}
	uint32_t zadr = ver0 + wib * (4 * TG_VER_SLOT * 4) + 64u;	/* the zero word of slot 0's window, as the asm blocks address LDS */
	asm volatile("" : "+v"(zadr));
	if (lane < 4)
		win[lane * TG_VER_SLOT + 16] = 0;	/* "no source" reads this */
	for (int i = lane; i < (TGS_FUSED ? 16 * 80 : 128); i += 64)
		mo[i] = 0;				/* bytes of the staged slots that nobody owns stay zero */
	/* which positions of the lane's column count: main search 21..472, "early" 0..20, SYNC summary 0..509 */
	const uint32_t vmain = (col == 0) ? 0xffe00000u : (col == 14) ? 0x01ffffffu : (col == 15) ? 0u : 0xffffffffu;
	const uint32_t vearly = (col == 0) ? 0x001fffffu : 0u;
	const uint32_t vys = (col == 15) ? 0x3fffffffu : 0xffffffffu;
	const uint32_t pos0 = (lane >> 4) * TG_SLOT_BITS + 32 * col;	/* first bit of the column inside the group */
#if TGS_PLAIN
	/* round 5, the "plain slot" form of search and outcome.  What this kernel may settle on its own is a slot that holds
	 * exactly ONE training sequence, of a downlink type, at its nominal offset (y at 214 = column 6 bit 22, n / p at 244 =
	 * column 7 bit 20) -- 99 % of a recording.  So it only has to VERIFY that: the expected hit is there, and nothing else
	 * is: no n / p at any other position 0..472, no y anywhere in the slot.  "No y" is checked on y's first 22 bits (a
	 * necessary condition: the three sequences then share 21 shifted copies of the string instead of 37) and the one
	 * expected y is confirmed on its last 16.  Every other slot -- a damaged or misplaced sequence, a second hit, a payload
	 * coincidence (3e-4 of the slots), anything below offset 21 -- goes to k_front_stream_fix, which evaluates
	 * tetra_find_train_seq()'s rule position by position as before.  The words this kernel does write are the exact
	 * pass's words for the same slot (test_stream_front_packed_bits_equals_per_position). */
	const uint32_t m_enp = (col == 7) ? (1u << 20) : 0u;			/* the expected n / p hit */
	const uint32_t m_ey = (col == 6) ? (1u << 22) : 0u;			/* the expected y hit */
	const uint32_t c_np = (vmain | vearly) & ~m_enp;			/* n / p hits that are not the expected one */
	const uint32_t c_y = vys & ~m_ey;					/* y (prefix) hits that are not the expected one */
#endif

	const uint32_t ngroups = (prm.nslots + 3) >> 2;
#if !TGS_FUSED
	if (wave >= ngroups) {
		if (lane == 0)
			defer[wave] = 0;
		return;
	}
#define TGS_G_FIRST wave
#define TGS_G_STEP  nwaves
#define TGS_G_END   ngroups
#define TGS_CLSW(i) mo[80 + (i)]
#define TGS_YSW(i)  mo[84 + (i)]
#else
	const uint32_t g_first = 16u * task, g_end = (g_first + 16u < ngroups) ? g_first + 16u : ngroups;
#define TGS_G_FIRST g_first
#define TGS_G_STEP  1u
#define TGS_G_END   g_end
#define TGS_CLSW(i) F.s_cls[4u * (g - g_first) + (i)]
#define TGS_YSW(i)  F.s_ys[4u * (g - g_first) + (i)]
	F.s_dtype[lane] = (uint8_t)TG_BURST_NONE;	/* (a task at the end of the grid has fewer than sixteen groups) */
	F.s_chan[lane] = 0;
	uint32_t cchan = 0;
#endif
	/* this wave's list of slots for the exact pass (k_front_stream_fix) and how many are on it */
	/* (wave-uniform, and kept in scalar registers by hand: the kernel sits at the 80 VGPRs that six waves per SIMD allow) */
	uint32_t dpos = __builtin_amdgcn_readfirstlane(TG_DEFER_L0(nwaves) + wave * (4u * ((ngroups + nwaves - 1) / nwaves)));
	const uint32_t dpos0 = dpos;

	/* request a group: 16 bytes per lane from the 16-byte aligned address below the group's first byte.  Groups the
	 * fast path may not touch (their windows or the exact form's 640-byte views reach past the stream) fetch group 0
	 * instead, so that every step issues the same loads */
	/* multi-channel batches: the channel a wave is in changes a handful of times over its groups, so its table entry
	 * is kept in scalar registers and looked up again only when a group falls outside [cg0, cg1) */
	uint32_t cg0 = 1, cg1 = 0, cncls = 0;
	uint64_t cfirst = 0, cspan = 0;		/* stream offset of the channel's grid slot 0; bytes from there to its end */
	auto fetch = [&](uint32_t g, tg_group_data &d) {
		uint64_t gb, first;
		if (prm.nchan) {
			const uint32_t s0 = 4u * g;
			if (s0 < cg0 || s0 >= cg1) {
				/* (readfirstlane: the values are wave-uniform and must live in scalar registers, so that the wait
				 * for these loads stays inside this rarely taken branch and does not drain the prefetch) */
				const uint32_t c = chan_of_slot(prm.chan, prm.nchan, s0, lane);
				const tg_chan_ent e = prm.chan[c];
				const uint32_t nxt = c + 1 < prm.nchan ? prm.chan[c + 1].gbase : prm.nslots;
#if TGS_FUSED
				cchan = __builtin_amdgcn_readfirstlane(c);
#endif
				cg0 = __builtin_amdgcn_readfirstlane(e.gbase);
				cg1 = __builtin_amdgcn_readfirstlane(nxt);
				cncls = __builtin_amdgcn_readfirstlane(e.ncls);
				const uint64_t f = (e.d_off & ~TG_CHAN_PACKED) + e.anchor, sp = e.len - e.anchor;
				cfirst = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)f) |
					 ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(f >> 32)) << 32);
				cspan = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)sp) |
					((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32);
			}
			const uint32_t i0 = s0 - cg0;
#if TGS_FUSED
			d.chan = cchan;
#endif
			first = cfirst;
			gb = first + (uint64_t)i0 * TG_SLOT_BITS;
			d.fast = i0 + 4u <= cncls && (uint64_t)i0 * TG_SLOT_BITS + TG_GROUP_BYTES + TG_VIEW_OF(prm.chunk) <= cspan;
		} else {
#if TGS_FUSED
			d.chan = 0;
#endif
			first = prm.anchor;
			gb = prm.anchor + (uint64_t)g * TG_GROUP_BYTES;
			d.fast = gb + TG_GROUP_BYTES + TG_VIEW_OF(prm.chunk) <= prm.len;
		}
		if (PACKED) {
			/* packed ingest: the stream lies in memory one bit per position, so a group is 255 bytes: eighteen lanes
			 * fetch 16 bytes each from the aligned address below its first bit, a0 = how many bits in the group starts */
			const uint64_t gbit = d.fast ? gb : first;
			const uint8_t *p = stream + (gbit >> 3);
			const uint32_t ab = (uint32_t)((uintptr_t)p & 15);
			d.a0 = 8 * ab + (uint32_t)(gbit & 7);
			d.a = *(const uint4 *)(p - ab + 16 * (lane < 18 ? lane : 17));
			d.b = d.c = make_uint4(0, 0, 0, 0);	/* (unused here; left unset they keep the whole struct in scratch memory) */
			return;
		}
#if TGS_ABLATE & 2
		const uint8_t *p = stream + first + 2040u * (wave & 1023u);
#else
		const uint8_t *p = stream + (d.fast ? gb : first);
#endif
		d.a0 = (uint32_t)((uintptr_t)p & 15);
		const uint8_t *base16 = p - d.a0;
#if TGS_TOUCH
		{	/* pull the lines of the group this wave takes TGS_TOUCH rounds after the one being fetched towards the L2 (a cold capture:
			 * DRAM page misses, translations), one dword per line, as long as that group lies in the same channel's bytes */
			const uint64_t adv = (uint64_t)TGS_TOUCH * nwaves * TG_GROUP_BYTES;
			const bool ahead = d.fast && (prm.nchan ? (gb - first) + adv + TG_GROUP_LOAD + 256 <= cspan : gb + adv + TG_GROUP_LOAD + 256 <= prm.len);
			d.touch = 0;
			if (ahead && lane < 18)
				d.touch = *(const volatile uint32_t *)(base16 + adv + 128 * lane);
		}
#endif
#if TGS_LOAD_NT	/* (A/B: the capture is read once -- non-temporal loads) */
		typedef uint32_t tgs_u4v __attribute__((ext_vector_type(4)));
		{
			const tgs_u4v va = __builtin_nontemporal_load((const tgs_u4v *)(base16 + 16 * lane));
			const tgs_u4v vb = __builtin_nontemporal_load((const tgs_u4v *)(base16 + 1024 + 16 * lane));
			const tgs_u4v vc = __builtin_nontemporal_load((const tgs_u4v *)(base16 + 2048 + 16 * (lane < 7 ? lane : 7)));
			d.a = make_uint4(va.x, va.y, va.z, va.w);
			d.b = make_uint4(vb.x, vb.y, vb.z, vb.w);
			d.c = make_uint4(vc.x, vc.y, vc.z, vc.w);
		}
#else
		d.a = *(const uint4 *)(base16 + 16 * lane);
		d.b = *(const uint4 *)(base16 + 1024 + 16 * lane);
		d.c = *(const uint4 *)(base16 + 2048 + 16 * (lane < 7 ? lane : 7));
#endif
	};

	auto work = [&](uint32_t g, const tg_group_data &cur) {
#if TGS_FUSED
		mo = s_out[0] + 80u * (g - g_first);	/* this group's four packed slots: they stay for the trellis phase */
#endif
#if TGS_TOUCH
		asm volatile("" :: "v"(cur.touch));	/* (the touch load's register stays its own until the group's own bytes are here) */
#endif
		TGS_MARK(0);	/* since the last mark: the next group's fetch issued */
		/* bytes other than 0 / 1 anywhere in the group: not for this kernel */
		bool defer_all;
		if (PACKED) {
			defer_all = !cur.fast;
			TGS_MARK(1);
			if (lane < 18)		/* the bits are the bit string: 288 bytes, as they came */
				((uint4 *)bits)[lane] = cur.a;
		} else {
			const uint32_t orall = cur.a.x | cur.a.y | cur.a.z | cur.a.w | cur.b.x | cur.b.y | cur.b.z | cur.b.w |
					       cur.c.x | cur.c.y | cur.c.z | cur.c.w;
			defer_all = !cur.fast || __ballot((orall & 0xfefefefeu) != 0) != 0;

			TGS_MARK(1);	/* the group's bytes are here */
			/* bytes -> bits -> LDS */
			tg_u16_alias *b16 = (tg_u16_alias *)bits;
			b16[lane] = (uint16_t)bytes16_to_bits(cur.a);
			b16[64 + lane] = (uint16_t)bytes16_to_bits(cur.b);
			if (lane < 8)
				b16[128 + lane] = (uint16_t)bytes16_to_bits(cur.c);
		}
		/* the lane's column of its slot: 96 bits from position pos0 + a0 of the string */
		uint32_t W0, W1, W2;
		{
			const uint32_t p = pos0 + cur.a0;
			const uint32_t *q = bits + (p >> 5);
			const uint32_t D0 = q[0], D1 = q[1], D2 = q[2], D3 = q[3];
			W0 = __builtin_amdgcn_alignbit(D1, D0, p);
			W1 = __builtin_amdgcn_alignbit(D2, D1, p);
			W2 = __builtin_amdgcn_alignbit(D3, D2, p);
		}
		TGS_MARK(2);	/* bits through LDS, the lane's column */
		{
			uint32_t *v = win + (lane >> 4) * TG_VER_SLOT + col;
			v[0] = W0;
#pragma unroll
			for (int sft = 1; sft < ((TGS_ABLATE & 16) ? 1 : 8); sft++)
				v[sft * TG_VER_STRIDE] = __builtin_amdgcn_alignbit(W1, W0, sft);
		}

		/* match masks of the three sequences at the column's 32 positions */
		/* one accumulator per sequence, two positions per step: acc & (t_j == p_j) & (t_j+1 == p_j+1) is one
		 * three-input logic instruction (v_bitop3_b32) whatever the two pattern bits are */
#if TGS_PLAIN
		uint32_t my = 0xffffffffu, mn = 0xffffffffu, mp = 0xffffffffu;	/* (my: the first 22 bits of y only) */
#pragma unroll
		for (int j = 0; j < 22; j += 2) {
			const uint32_t t0 = (j == 0) ? W0 : __builtin_amdgcn_alignbit(W1, W0, j);
			const int k = j + 1;
			const uint32_t t1 = __builtin_amdgcn_alignbit(W1, W0, k);
#define TSQ_STEP(acc, P) acc = tsq_and2(acc, t0, t1, 2 * (int)(((P) >> j) & 1) + (int)(((P) >> k) & 1))
			TSQ_STEP(my, PY);
			TSQ_STEP(mn, PN);
			TSQ_STEP(mp, PP);
#undef TSQ_STEP
		}
		const uint32_t any = my | mn | mp;
		(void)W2;
#else
		uint32_t my = vys, mn = 0xffffffffu, mp = 0xffffffffu;
#pragma unroll
		for (int j = 0; j < ((TGS_ABLATE & 8) ? 2 : 38); j += 2) {
			const uint32_t t0 = (j == 0) ? W0 : (j < 32) ? __builtin_amdgcn_alignbit(W1, W0, j)
					  : (j == 32) ? W1 : __builtin_amdgcn_alignbit(W2, W1, j - 32);
			const int k = j + 1;
			const uint32_t t1 = (k < 32) ? __builtin_amdgcn_alignbit(W1, W0, k)
					  : (k == 32) ? W1 : __builtin_amdgcn_alignbit(W2, W1, k - 32);
			/* truth table index = acc << 2 | t0 << 1 | t1: the one entry with acc = 1, t0 = p_j, t1 = p_k */
#define TSQ_STEP(acc, P) acc = tsq_and2(acc, t0, t1, 2 * (int)(((P) >> j) & 1) + (int)(((P) >> k) & 1))
			TSQ_STEP(my, PY);
			if (j < 22) {
				TSQ_STEP(mn, PN);
				TSQ_STEP(mp, PP);
			}
#undef TSQ_STEP
		}
		const uint32_t any = my | mn | mp;

#endif
		TGS_MARK(3);	/* shifted copies stored, match masks, ballots */
#if !(TGS_ABLATE & (8 | 32 | 256))
		/* per slot (= 16-lane row): the first hit and the SYNC summary by reductions inside the row -- every lane makes a
		 * key of its own first hit ((position << 2 | type) in the high half, first y position in the low half: one
		 * v_pk_min_u16 reduces both) and a count word (hit below 21 in the high half, number of y hits in the low), four
		 * rotate-and-combine steps (DPP row_ror 8 4 2 1) leave the row's result in all of its lanes.  Vector
		 * instructions only: the form with ballots, per-lane shifts of them and the LDS crossbar cost 24 us per 1 M
		 * slots in round trips between the vector unit, scalar registers and LDS (TGS_ABLATE), this one */
#if TGS_PLAIN
		/* per lane: "something that is not the expected hit" (bit 23) and the expected hits it holds (y 22, n 21, p 20); OR over
		 * the slot's 16-lane row in four DPP steps; the row's four bits decide: exactly one expected hit and nothing else,
		 * or the slot is the exact pass's */
		(void)any;
		uint32_t rest = __builtin_amdgcn_bitop3_b32(mn, mp, c_np, 0xa8);		/* (mn | mp) & c_np */
		rest = __builtin_amdgcn_bitop3_b32(my, c_y, rest, 0xea);			/* (my & c_y) | rest */
		/* y's last 16 bits behind the expected prefix hit: positions 236..251 = bits 12..27 of column 6's second word */
		const bool ytail = ((W1 >> 12) & 0xffffu) == (uint32_t)((PY >> 22) & 0xffffu);
		uint32_t ex = ((mn & m_enp) << 1) | (mp & m_enp);
		ex = (my & (ytail ? m_ey : 0u)) | ex;
		uint32_t rowc = ((rest != 0u ? 1u : 0u) << 23) | ex;
#define ROW_STEP(CTRL) rowc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rowc, (CTRL), 0xf, 0xf, true);
		ROW_STEP(0x128)	/* row_ror:8 */
		ROW_STEP(0x124)
		ROW_STEP(0x122)
		ROW_STEP(0x121)
#undef ROW_STEP
		/* 0b0001 p alone -> NORM_2, 0b0010 n alone -> NORM_1, 0b0100 y alone -> SYNC; anything else: not this kernel's slot */
		const uint32_t kk = rowc >> 20;
		const uint32_t rc = (0xfff3f01fu >> (4u * (kk < 8u ? kk : 7u))) & 0xfu;
		const bool dfr = defer_all || rc == 0xfu;
		const uint32_t offs = rc == TG_BURST_SYNC ? (uint32_t)TG_SYNC_TRAIN_OFF : (uint32_t)TG_NORM_TRAIN_OFF;
		const uint32_t dtype = dfr ? (uint32_t)TG_BURST_NONE : rc;
		uint32_t ys = rc == TG_BURST_SYNC ? (uint32_t)TG_SYNC_TRAIN_OFF : (uint32_t)TG_YS_NONE;
#else
		typedef unsigned short cls_us2 __attribute__((ext_vector_type(2)));
		const uint32_t hm = any & vmain;
		const uint32_t hb = (uint32_t)__builtin_ctz(hm | 0x80000000u);
		const uint32_t ht = ((my >> hb) & 1) ? (uint32_t)TG_BURST_SYNC : ((mn >> hb) & 1) ? (uint32_t)TG_BURST_NORM_1 : (uint32_t)TG_BURST_NORM_2;
		const uint32_t hkey = hm ? (((32u * col + hb) << 2) | ht) : 0xffffu;
		const uint32_t ykey = my ? (32u * col + (uint32_t)__builtin_ctz(my | 0x80000000u)) : 0xffffu;
		uint32_t rmin = (hkey << 16) | ykey;
		uint32_t rsum = (((any & vearly) != 0) ? 0x10000u : 0u) + (uint32_t)__builtin_popcount(my);	/* (<= 510 y hits: the halves do not meet) */
#define ROW_STEP(CTRL)													\
		{													\
			const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rmin, (CTRL), 0xf, 0xf, true);	\
			const uint32_t u_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rsum, (CTRL), 0xf, 0xf, true);	\
			const cls_us2 m_ = __builtin_elementwise_min(__builtin_bit_cast(cls_us2, rmin), __builtin_bit_cast(cls_us2, t_));	\
			rmin = __builtin_bit_cast(uint32_t, m_);							\
			rsum += u_;											\
		}
		ROW_STEP(0x128)	/* row_ror:8 */
		ROW_STEP(0x124)
		ROW_STEP(0x122)
		ROW_STEP(0x121)
#undef ROW_STEP
		const uint32_t k16 = rmin >> 16, yfirst = rmin & 0xffffu, ycnt = rsum & 0xffffu;
		const uint32_t offs = k16 >> 2, rc = k16 & 3u;
		uint32_t ys = ycnt ? (yfirst | (ycnt > 1 ? (uint32_t)TG_YS_MULTI : 0u)) : (uint32_t)TG_YS_NONE;
		/* a sequence below offset 21 is accepted or not by the reference's skewed look-ahead window: the exact pass
		 * evaluates that rule (rare: a payload coincidence, about ten slots in a million) */
		const bool dfr = defer_all || k16 == 0xffffu || (rsum >> 16) != 0;
		uint32_t dtype = TG_BURST_NONE;
		if (rc == TG_BURST_SYNC ? offs == TG_SYNC_TRAIN_OFF : offs == TG_NORM_TRAIN_OFF)
			dtype = rc;
		if (dfr)
			dtype = TG_BURST_NONE;
#endif
#define CLS_OWNER      ((lane & 15u) == 0u)	/* the lane that writes the slot's words */
#define CLS_SLOT       (lane >> 4)
#define CLS_LANE_OF(K) (16 * (K))
#endif
#if TGS_ABLATE & (8 | 32 | 256)
#define CLS_OWNER      (lane < 4u)
#define CLS_SLOT       lane
#define CLS_LANE_OF(K) (K)
		/* (measurement builds: every slot "a NORM_1 burst at its place", whatever the search said) */
		const bool dfr = false;
		const uint32_t dtype = TG_BURST_NORM_1;
		const uint32_t clsword = TG_BURST_NORM_1 | (TG_NORM_TRAIN_OFF << 8);
		const uint32_t meta = (dtype | (TG_NORM_TRAIN_OFF << 16)) ^ ((TGS_ABLATE & (32 | 256)) ? (any & 1u) : 0u);
		uint32_t ys = TG_YS_NONE;
#else
		const uint32_t clsword = dfr ? TG_CLS_DEFER : (rc | (offs << 8));
		const uint32_t meta = dfr ? 0u : (dtype | (offs << 16));
#endif

		const uint32_t first = 4u * g;
		const uint32_t cnt = (prm.nslots - first < 4u) ? prm.nslots - first : 4u;
#define STREAM_SLOT_K(K)												\
		{													\
			const uint32_t dt = (TGS_ABLATE & 128) ? (uint32_t)TG_BURST_NORM_1 :					\
					    (TGS_ABLATE & 256) ? (((g + (K)) & 1) ? (uint32_t)TG_BURST_NORM_1 : (uint32_t)TG_BURST_NORM_2) : \
					    (uint32_t)__builtin_amdgcn_readlane(dtype, CLS_LANE_OF(K));		\
			uint32_t mybyte = 0;										\
			if (TGS_ABLATE & 4)											\
				mybyte = dt;											\
			else if (dt == TG_BURST_NORM_1)									\
				mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 0>(g_adr[0]);			\
			else if (dt == TG_BURST_NORM_2)									\
				mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 1>(g_adr[1]);			\
			else if (dt == TG_BURST_SYNC) {									\
				if (TGS_SADR_LDS) {										\
					const uint4 a0_ = *(const uint4 *)&s_sadr[wib][8 * lane], a1_ = *(const uint4 *)&s_sadr[wib][8 * lane + 4];	\
					const uint32_t sa_[8] = { a0_.x, a0_.y, a0_.z, a0_.w, a1_.x, a1_.y, a1_.z, a1_.w };	\
					mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 2>(sa_);			\
				} else												\
					mybyte = front_gather_bytes<4 * TG_VER_SLOT * (K), 2>(g_adr[2]);		\
			}													\
			((uint8_t *)mo)[(K) * TG_PACKED_WORDS * 4 + obyte] = (uint8_t)mybyte;				\
		}
		TGS_MARK(4);	/* classification of the four slots */
#if TGS_GPIPE && !TGS_ABLATE
#define GP_ISSUE(K, T)													\
		{													\
			const uint32_t dt = (uint32_t)__builtin_amdgcn_readlane(dtype, CLS_LANE_OF(K));		\
			if (dt == TG_BURST_NORM_1)										\
				front_gather_issue<4 * TG_VER_SLOT * (K), 0>(g_adr[0], T);				\
			else if (dt == TG_BURST_NORM_2)									\
				front_gather_issue<4 * TG_VER_SLOT * (K), 1>(g_adr[1], T);				\
			else if (dt == TG_BURST_SYNC)									\
				front_gather_issue<4 * TG_VER_SLOT * (K), 2>(g_adr[2], T);				\
			else		/* not this kernel's slot: eight reads of the zero word (the waits count on eight) */	\
				front_gather_issue<0, 3>(g_zero, T);							\
		}
#define GP_TAKE(K, T, NEWER) ((uint8_t *)mo)[(K) * TG_PACKED_WORDS * 4 + obyte] = (uint8_t)front_gather_take<NEWER>(T);
		{
			uint32_t tA[8], tB[8];
			const uint32_t g_zero[8] = { zadr, zadr, zadr, zadr, zadr, zadr, zadr, zadr };
			GP_ISSUE(0, tA)
			GP_ISSUE(1, tB)
			GP_TAKE(0, tA, 8)
			GP_ISSUE(2, tA)
			GP_TAKE(1, tB, 8)
			GP_ISSUE(3, tB)
			GP_TAKE(2, tA, 8)
			GP_TAKE(3, tB, 0)
		}
#undef GP_ISSUE
#undef GP_TAKE
#else
		STREAM_SLOT_K(0)
		STREAM_SLOT_K(1)
		STREAM_SLOT_K(2)
		STREAM_SLOT_K(3)
#endif
#undef STREAM_SLOT_K
		TGS_MARK(5);	/* the four gathers */
		if (CLS_OWNER) {
			mo[CLS_SLOT * TG_PACKED_WORDS + TG_PW_META] = meta;
			TGS_CLSW(CLS_SLOT) = clsword;
			TGS_YSW(CLS_SLOT) = ys;
#if TGS_FUSED
			F.s_dtype[4u * (g - g_first) + CLS_SLOT] = (uint8_t)dtype;	/* what the trellis phase decodes this slot as (TG_BURST_NONE: not at all) */
			F.s_chan[4u * (g - g_first) + CLS_SLOT] = (uint8_t)cur.chan;
#endif
		}
		{	/* slots this pass could not settle: onto this wave's list for k_front_stream_fix */
#if TGS_SB & 1
			__builtin_amdgcn_sched_barrier(0);
#endif
#if TGS_SB & 2
			asm volatile("" ::: "memory");
#endif
			const bool mine = CLS_OWNER && CLS_SLOT < cnt && dfr;
			const unsigned long long dm = __ballot(mine);
			if (dm) {
#if TGS_DEFER_ATOMIC
				uint32_t pos = 0;
				if (lane == 0)
					pos = atomicAdd(defer + TG_DEFER_L0(nwaves) + (size_t)nwaves * capw, (uint32_t)__builtin_popcountll(dm));
				pos = __builtin_amdgcn_readfirstlane(pos);
				if (mine)
					defer[TG_DEFER_L0(nwaves) + pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u))] = first + CLS_SLOT;
#else
				if (mine)
					defer[dpos + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u))] = first + CLS_SLOT;
				if (!(TGS_ABLATE & 64))
					dpos = __builtin_amdgcn_readfirstlane(dpos + (uint32_t)__builtin_popcountll(dm));
#endif
			}
		}
#undef CLS_OWNER
#undef CLS_SLOT
#undef CLS_LANE_OF
		if (!(TGS_ABLATE & 1) || prm.nslots == 0xffffffffu)
			front_flush(mo, lane, first, cnt, packed);
		if (lane < cnt && (!(TGS_ABLATE & 1) || prm.nslots == 0xffffffffu)) {
			cls[first + lane] = TGS_CLSW(lane);
			if (ysum)
				ysum[first + lane] = (uint16_t)TGS_YSW(lane);
		}
		TGS_MARK(6);	/* staged stores */
	};

#if TGS_FUSED && TGS_FUSED_DEPTH == 4
	/* the fused form runs at the trellis phase's three waves per SIMD, with registers to spare in this phase: FOUR register sets with
	 * fixed roles, three groups requested ahead (one ahead, as below, left the phase waiting on its own chain of loads: 16 groups x one
	 * memory latency per task) */
	{
		tg_group_data dA, dB, dC, dD;
		auto cl = [&](uint32_t x) { return x < g_end ? x : g_end - 1u; };
		fetch(cl(g_first), dA);
		fetch(cl(g_first + 1u), dB);
		fetch(cl(g_first + 2u), dC);
		for (uint32_t g = g_first; g < g_end; g += 4u) {
			fetch(cl(g + 3u), dD);
			work(g, dA);
			if (g + 1u >= g_end)
				break;
			fetch(cl(g + 4u), dA);
			work(g + 1u, dB);
			if (g + 2u >= g_end)
				break;
			fetch(cl(g + 5u), dB);
			work(g + 2u, dC);
			if (g + 3u >= g_end)
				break;
			fetch(cl(g + 6u), dC);
			work(g + 3u, dD);
		}
	}
#else
	/* two register sets with fixed roles: the next group is requested before this one is worked on, no copies */
	tg_group_data dA, dB;
	uint32_t g = TGS_G_FIRST;
#if TGS_SB & 4
#define TGS_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define TGS_FENCE do { } while (0)
#endif
	fetch(g, dA);
	for (;;) {
		const uint32_t gB = g + TGS_G_STEP;
		TGS_FENCE;
		fetch(gB < TGS_G_END ? gB : g, dB);
		TGS_FENCE;
		work(g, dA);
		if (gB >= TGS_G_END)
			break;
		const uint32_t gA = gB + TGS_G_STEP;
		TGS_FENCE;
		fetch(gA < TGS_G_END ? gA : gB, dA);
		TGS_FENCE;
		work(gB, dB);
		if (gA >= TGS_G_END)
			break;
		g = gA;
	}
#undef TGS_FENCE
#endif
	if (lane == 0)
		defer[wave] = dpos - dpos0;
#undef TGS_G_FIRST
#undef TGS_G_STEP
#undef TGS_G_END
#undef TGS_CLSW
#undef TGS_YSW
#if !TGS_FUSED
	TG_TRACE_END(0u, (TG_STREAM_WPB <= 4 ? 4u / TG_STREAM_WPB : 1u));
#ifdef TGS_TIMING
	if (lane == 0)
		for (int i = 0; i < 8; i++)
			atomicAdd(&g_tgs_acc[i], tgs_acc[i]);
#endif
#endif
}
