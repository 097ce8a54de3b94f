// mask_probe.hip -- which compute units does a bit of hipExtStreamCreateWithCUMask's mask stand for?  A background stream with a mask fills every CU it
// may use with workgroups that live 25 ms, several rounds deep; a kernel of an UNMASKED stream (8, 64 or 2048 workgroups: they go round the XCDs in turn)
// is timed beside it.  If the CUs the mask leaves free are spread over all XCDs the kernel runs at once; if they are one whole XCD it waits for the others.
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench/mask_probe tools/ubench/mask_probe.hip && build/ubench/mask_probe
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
int main()
{
	setvbuf(stdout, NULL, _IONBF, 0);
	hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	unsigned long long *sink = nullptr; CHK(hipMalloc(&sink, 8));
	int lo = 0, hi = 0; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	hipStream_t fg; CHK(hipStreamCreateWithPriority(&fg, hipStreamNonBlocking, hi));
	struct { const char *name; int kind; } masks[] = { {"bit i set unless i % 8 == 7", 0}, {"bits 0 .. 223 set", 1}, {"bit i set unless i % 32 >= 28", 2}, {"bit i set unless (i / 8) % 4 == 3", 3} };
	for (auto &mk : masks) {
		std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
		int n_set = 0;
		for (int i = 0; i < n_cu; ++i) {
			const bool free_ = mk.kind == 0 ? i % 8 == 7 : mk.kind == 1 ? i >= n_cu - n_cu / 8 : mk.kind == 2 ? i % 32 >= 28 : (i / 8) % 4 == 3;
			if (!free_) { mask[i >> 5] |= 1u << (i & 31); ++n_set; }
		}
		hipStream_t bg; CHK(hipExtStreamCreateWithCUMask(&bg, (uint32_t)mask.size(), mask.data()));
		printf("mask %s: stream created\n", mk.name);
		hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, bg, 1000ull, sink); CHK(hipStreamSynchronize(bg));
		printf("warm\n");
		for (int wgs : {8, 64, 2048}) {
			for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(k_spin, dim3(n_set * 16), dim3(256), 0, bg, 50000000ull, sink);
			const double t0 = now();
			hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, fg, 100000ull, sink);
			CHK(hipStreamSynchronize(fg));
			const double t1 = now();
			CHK(hipStreamSynchronize(bg));
			printf("mask: %-34s (%d CUs) | unmasked kernel of %4d workgroups done after %7.2f ms; background drained %6.1f ms later\n", mk.name, n_set, wgs, (t1 - t0) * 1e3, (now() - t1) * 1e3);
		}
		CHK(hipStreamDestroy(bg));
	}
	return 0;
}
