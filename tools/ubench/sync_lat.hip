// Round-trip latency of a tiny kernel: hipStreamSynchronize() against polling a flag the kernel writes to mapped host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k_flag(volatile uint32_t *flag, uint32_t v, uint32_t *sink)
{
	if (threadIdx.x == 0) {
		sink[blockIdx.x] = v;
		__threadfence_system();
		flag[blockIdx.x] = v;
	}
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	uint32_t *h_flag, *d_flag, *d_sink;
	(void)hipHostMalloc((void **)&h_flag, 4096, hipHostMallocMapped);
	(void)hipHostGetDevicePointer((void **)&d_flag, h_flag, 0);
	(void)hipMalloc(&d_sink, 4096);
	hipStream_t s;
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	const int N = 20000;
	for (int blocks = 1; blocks <= 64; blocks *= 8) {
		for (int mode = 0; mode < 3; mode++) {
			for (int b = 0; b < blocks; b++) h_flag[b] = 0;
			double t0 = 0;
			for (int i = -200; i < N; i++) {
				if (i == 0) t0 = now();
				const uint32_t v = (uint32_t)(i + 1000);
				hipLaunchKernelGGL(k_flag, dim3(blocks), dim3(64), 0, s, d_flag, v, d_sink);
				if (mode == 0) {
					(void)hipStreamSynchronize(s);
				} else if (mode == 1) {
					for (int b = 0; b < blocks; b++)
						while (((volatile uint32_t *)h_flag)[b] != v) { }
				} else {
					hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
					(void)hipEventRecord(e, s);
					while (hipEventQuery(e) == hipErrorNotReady) { }
					(void)hipEventDestroy(e);
				}
			}
			const double us = (now() - t0) / N * 1e6;
			printf("blocks %2d  %-28s %6.2f us per launch + wait\n", blocks,
			       mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "poll flag in mapped memory" : "hipEventQuery spin", us);
		}
	}
	(void)hipStreamSynchronize(s);
	return 0;
}
