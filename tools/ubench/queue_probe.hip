// queue_probe.hip -- which streams does a pending dispatch block?  One stream launches kernels with more workgroups than the device holds
// (its dispatch stays pending while it runs); a short kernel is then timed on each of N other streams, one at a time.
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench/queue_probe tools/ubench/queue_probe.hip && build/ubench/queue_probe [n_streams] [wgs_per_cu of the background]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_spin(unsigned long long cycles, unsigned long long *sink)
{
	const unsigned long long t0 = __builtin_readcyclecounter();
	unsigned long long x = threadIdx.x;
	while (__builtin_readcyclecounter() - t0 < cycles) x = x * 6364136223846793005ull + 1442695040888963407ull;
	if (x == 42) *sink = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	const int n_streams = argc > 1 ? atoi(argv[1]) : 16, bg_wgs = argc > 2 ? atoi(argv[2]) : 16, bg_threads = argc > 3 ? atoi(argv[3]) : 256;
	hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	unsigned long long *sink = nullptr; CHK(hipMalloc(&sink, 8));
	int lo = 0, hi = 0; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	hipStream_t bg; CHK(hipStreamCreateWithPriority(&bg, hipStreamNonBlocking, lo));
	std::vector<hipStream_t> st((size_t)n_streams);
	for (int i = 0; i < n_streams; ++i) CHK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, i & 1 ? hi : lo));
	// warm every queue
	for (int i = 0; i < n_streams; ++i) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], 1000ull, sink); CHK(hipStreamSynchronize(st[i])); }
	printf("background: %d workgroups of %d threads per CU (%d CUs), 25 ms each\n", bg_wgs, bg_threads, n_cu);
	for (int i = 0; i < n_streams; ++i) {
		for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(k_spin, dim3(n_cu * bg_wgs), dim3(bg_threads), 0, bg, 50000000ull, sink);
		const double t0 = now();
		hipLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, st[i], 100000ull, sink);   // 8 waves, 50 us
		CHK(hipStreamSynchronize(st[i]));
		const double t1 = now();
		CHK(hipStreamSynchronize(bg));
		printf("stream %2d (%s priority): short kernel done after %7.2f ms; background drained %6.1f ms after that\n", i, i & 1 ? "high" : "low ", (t1 - t0) * 1e3, (now() - t1) * 1e3);
	}
	return 0;
}
