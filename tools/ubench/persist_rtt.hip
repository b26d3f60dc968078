// What a persistent decoder workgroup on a mapped ring could buy a batch-1 flush (small batches):
//   (1) host -> device -> host round trip through mapped host memory with a kernel that is already running (no launch),
//   (2) the shader clock a single dependent chain runs at when the kernel is launched per request vs. when it spins.
// hipcc --offload-arch=gfx950 -O2 -o persist_rtt persist_rtt.hip && ./persist_rtt
// The spinning kernel ends on a stop request and, whatever the host does, after 3 s of its own clock.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

struct ctl {
	volatile uint32_t req;		// host -> device: request number (0xffffffff = stop)
	uint32_t pad0[15];
	volatile uint32_t resp;		// device -> host: last request served
	volatile uint32_t cycles;	// shader-clock cycles of the chain of the last request
	volatile uint32_t ticks;	// 100 MHz ticks of the same
	uint32_t pad1[13];
};

__device__ __forceinline__ uint32_t chain(uint32_t x, int n)
{
	for (int i = 0; i < n; i++)	// one dependent 32-bit multiply-add per step (quarter rate: 16 cycles)
		x = x * 1664525u + 1013904223u;
	return x;
}

__global__ void k_persist(ctl *c, int work, uint32_t *sink)
{
	if (threadIdx.x)
		return;
	const unsigned long long t_end = wall_clock64() + 300000000ull;	// 3 s at 100 MHz
	uint32_t last = 0;
	for (;;) {
		const uint32_t r = __atomic_load_n((uint32_t *)&c->req, __ATOMIC_RELAXED);
		if (r == 0xffffffffu || wall_clock64() > t_end)
			break;
		if (r == last)
			continue;
		last = r;
		const unsigned long long c0 = clock64(), w0 = wall_clock64();
		const uint32_t y = chain(r, work);
		const unsigned long long c1 = clock64(), w1 = wall_clock64();
		sink[0] = y;
		c->cycles = (uint32_t)(c1 - c0);
		c->ticks = (uint32_t)(w1 - w0);
		__threadfence_system();
		c->resp = r;
	}
}

__global__ void k_once(ctl *c, uint32_t r, int work, uint32_t *sink)
{
	if (threadIdx.x)
		return;
	const unsigned long long c0 = clock64(), w0 = wall_clock64();
	const uint32_t y = chain(r, work);
	const unsigned long long c1 = clock64(), w1 = wall_clock64();
	sink[0] = y;
	c->cycles = (uint32_t)(c1 - c0);
	c->ticks = (uint32_t)(w1 - w0);
	__threadfence_system();
	c->resp = r;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	ctl *h, *d;
	uint32_t *sink;
	(void)hipHostMalloc((void **)&h, sizeof(ctl), hipHostMallocMapped);
	(void)hipHostGetDevicePointer((void **)&d, h, 0);
	(void)hipMalloc(&sink, 64);
	hipStream_t s;
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	const int N = 20000;
	for (int work : { 0, 500, 2000 }) {
		// launch per request
		h->req = h->resp = 0;
		double t0 = 0, cyc = 0, tic = 0;
		for (int i = -500; i < N; i++) {
			if (i == 0) { t0 = now(); cyc = tic = 0; }
			const uint32_t v = (uint32_t)(i + 1000);
			hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, s, d, v, work, sink);
			while (h->resp != v) { }
			cyc += h->cycles; tic += h->ticks;
		}
		const double us1 = (now() - t0) / N * 1e6;
		printf("chain of %4d steps, a launch per request : %6.2f us per round trip; chain %6.2f us at %4.0f MHz\n", work, us1,
		       tic / N / 100.0, tic > 0 ? cyc / tic * 100.0 : 0.0);
		// one kernel that stays
		h->req = h->resp = 0;
		hipLaunchKernelGGL(k_persist, dim3(1), dim3(64), 0, s, d, work, sink);
		int ok = 1;
		for (int i = -500; i < N && ok; i++) {
			if (i == 0) { t0 = now(); cyc = tic = 0; }
			const uint32_t v = (uint32_t)(i + 1000);
			h->req = v;
			const double tw = now();
			while (h->resp != v)
				if (now() - tw > 0.5) { ok = 0; break; }	// the kernel's own time limit has ended it
			cyc += h->cycles; tic += h->ticks;
		}
		const double us2 = (now() - t0) / N * 1e6;
		h->req = 0xffffffffu;
		(void)hipStreamSynchronize(s);
		printf("chain of %4d steps, a kernel that stays    : %6.2f us per round trip; chain %6.2f us at %4.0f MHz%s\n", work, us2,
		       tic / N / 100.0, tic > 0 ? cyc / tic * 100.0 : 0.0, ok ? "" : "  (ended early)");
	}
	return 0;
}
