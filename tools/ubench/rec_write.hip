// Microbenchmark (round 4): what does the memory system do with the trellis kernels' record writes?
//   records of PITCH bytes (320 = the product's, 384 = three whole 128-byte lines), every SEL-th record written by this
//   launch (2 = what either trellis kernel does in the SB+NDB mix: its records alternate with the other kernel's),
//   in pieces of PIECE bytes that complete together in one store instruction:
//     16  = one lane owns a record (or block) and stores it 16 bytes at a time: 64 different lines per instruction
//     64  = four lanes store one 64-byte segment  (k_vit<432>'s LDS transpose, round 2)
//     128 = eight lanes store one whole 128-byte line
//   NT = non-temporal stores.  After every write launch a streaming read of 510 MB (the next step's front end) is timed
//   on the same stream: what the dirty records cost the kernel that follows.
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this binary gives the fill traffic per variant (kernel names differ).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define NTSTORE(v, p) __builtin_nontemporal_store(*(const u32x4 *)&(v), (u32x4 *)(p))

template <int PITCH, int PIECE, int SEL, bool NT, int BYTES>
__global__ __launch_bounds__(256) void k_recw(uint8_t *rec, uint32_t nrec)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t r0 = wave * 64;		/* this wave's 64 records (ordinals among the selected ones) */
	if (r0 >= nrec)
		return;
	constexpr int LPP = PIECE / 16;		/* lanes per piece */
	constexpr int RPI = 64 / LPP;		/* records per store instruction */
	constexpr int NPIECE = BYTES / PIECE;
	const uint4 v = make_uint4(lane, wave, 0x01010101u, 0x00010001u);
#pragma unroll
	for (int c = 0; c < NPIECE; c++) {
#pragma unroll
		for (int i = 0; i < LPP; i++) {
			const uint32_t r = r0 + lane / LPP + RPI * i;
			uint4 *p = (uint4 *)(rec + (size_t)r * SEL * PITCH + (size_t)PIECE * c + 16 * (lane % LPP));
			if (NT)
				NTSTORE(v, p);
			else
				*p = v;
		}
	}
}

// the product's half-slot pattern: a lane owns one 128-byte block region at OFF0 / OFF1 of a record (two lanes per
// record), stored 16 bytes at a time + the primary lane's header pieces
template <int PITCH, int OFF0, int OFF1, int SEL, bool NT>
__global__ __launch_bounds__(256) void k_recw_blocks(uint8_t *rec, uint32_t nrec)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t r = wave * 32 + (lane >> 1);
	if (r >= nrec)
		return;
	uint8_t *base = rec + (size_t)r * SEL * PITCH;
	const uint4 v = make_uint4(lane, wave, 0x01010101u, 0x00010001u);
	uint4 *p = (uint4 *)(base + ((lane & 1) ? OFF1 : OFF0));
#pragma unroll
	for (int q = 0; q < 8; q++) {
		if (NT)
			NTSTORE(v, p + q);
		else
			p[q] = v;
	}
	if (!(lane & 1)) {
		*(uint4 *)(base + 32) = v;	/* BBK */
		*(uint4 *)base = v;		/* header */
	} else {
		base[3] = 1;			/* crc_ok[1], crc[1] */
		*(uint16_t *)(base + 6) = 0x1d0f;
	}
}

__global__ __launch_bounds__(256) void k_stream(const uint4 *in, size_t n16, uint32_t *out)
{
	uint32_t acc = 0;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
		const uint4 v = in[i];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

static uint8_t *d_rec, *d_in;
static uint32_t *d_o;
static const uint32_t NSEL = 500000;

template <typename F> static void run(const char *name, double bytes, F f)
{
	hipEvent_t e[3];
	for (auto &x : e)
		(void)hipEventCreate(&x);
	float tw = 0, tr = 0;
	const int reps = 8;
	for (int it = 0; it < reps + 2; it++) {
		(void)hipEventRecord(e[0]);
		f();
		(void)hipEventRecord(e[1]);
		hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const uint4 *)d_in, (size_t)510000000 / 16, d_o);
		(void)hipEventRecord(e[2]);
		(void)hipEventSynchronize(e[2]);
		float a, b;
		(void)hipEventElapsedTime(&a, e[0], e[1]);
		(void)hipEventElapsedTime(&b, e[1], e[2]);
		if (it >= 2) {
			tw += a;
			tr += b;
		}
	}
	printf("%-58s write %6.1f us (%5.2f TB/s)   following 510 MB read %6.1f us\n", name, tw / reps * 1e3,
	       bytes / (tw / reps) / 1e9, tr / reps * 1e3);
}

#define RECW(P, PC, S, NT, B) run("pitch " #P " piece " #PC " every " #S " nt " #NT " bytes " #B, (double)NSEL * B, [] { \
	hipLaunchKernelGGL((k_recw<P, PC, S, NT, B>), dim3((NSEL / 64 + 3) / 4), dim3(256), 0, 0, d_rec, NSEL); })
#define RECB(P, O0, O1, S, NT) run("pitch " #P " blocks at " #O0 "/" #O1 " every " #S " nt " #NT " (16 B per lane)", (double)NSEL * 304, [] { \
	hipLaunchKernelGGL((k_recw_blocks<P, O0, O1, S, NT>), dim3((NSEL / 32 + 3) / 4), dim3(256), 0, 0, d_rec, NSEL); })

int main()
{
	(void)hipMalloc(&d_rec, (size_t)1000064 * 384 + 4096);
	(void)hipMalloc(&d_in, 510000000 + 4096);
	(void)hipMalloc(&d_o, 4);
	(void)hipMemset(d_in, 1, 510000000);
	(void)hipMemset(d_rec, 0, (size_t)1000064 * 384);
	run("no write launch in front", 0, [] {});
	RECW(320, 16, 2, false, 320);
	RECW(320, 64, 2, false, 320);
	RECW(320, 64, 2, true, 320);
	RECW(320, 64, 1, false, 320);
	RECW(320, 64, 1, true, 320);
	RECW(384, 64, 2, false, 384);
	RECW(384, 128, 2, false, 384);
	RECW(384, 128, 2, true, 384);
	RECW(384, 64, 2, true, 384);
	RECW(384, 128, 1, false, 384);
	RECW(384, 128, 1, true, 384);
	RECW(384, 128, 2, false, 256);	/* two lines of three */
	RECW(384, 128, 2, true, 256);
	RECB(320, 48, 176, 2, false);
	RECB(320, 48, 176, 2, true);
	RECB(320, 64, 192, 2, false);
	RECB(384, 128, 256, 2, false);
	RECB(384, 128, 256, 2, true);
	return 0;
}
