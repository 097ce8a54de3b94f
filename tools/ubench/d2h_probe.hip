// d2h_probe.hip -- how long does a chunk's region download (1.4 GB in 8 MB pieces through a pinned buffer, as shim.hip's xfer does it) take
// while the device is full of long-lived workgroups of another stream?  Variants of the copying stream: plain, CU-masked (as the lanes' streams are),
// high priority; and of the copy: hipMemcpyAsync (the runtime's choice of engine) or a copy kernel writing the pinned buffer.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_probe tools/ubench/d2h_probe.hip && /tmp/d2h_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
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
__global__ void __launch_bounds__(256) k_copy(const uint4 *src, uint4 *dst, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
	const size_t total = (size_t)1400 << 20, piece = (size_t)8 << 20;
	if (argc > 1 && atoi(argv[1]) & 1) { if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) printf("hipSetDeviceFlags failed\n"); else printf("hipDeviceScheduleBlockingSync set\n"); }
	const bool kernel_first = argc > 1 && (atoi(argv[1]) & 2);   // a kernel on the copying stream ahead of every piece (as the product's streams have)
	hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	void *dev = nullptr, *pin = nullptr; unsigned long long *sink = nullptr;
	CHK(hipMalloc(&dev, total)); CHK(hipMemset(dev, 1, total)); CHK(hipMalloc(&sink, 8));
	CHK(hipHostMalloc(&pin, 2 * piece, hipHostMallocDefault));
	char *host = (char*)malloc(total);
	memset(host, 0, total);
	int lo = 0, hi = 0; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
	for (int cu = 0; cu < n_cu; ++cu) if (cu % 8 != 7) mask[cu >> 5] |= 1u << (cu & 31);
	hipStream_t s_plain, s_mask, s_hi, s_bg;
	CHK(hipStreamCreateWithPriority(&s_plain, hipStreamNonBlocking, lo));
	CHK(hipExtStreamCreateWithCUMask(&s_mask, (uint32_t)mask.size(), mask.data()));
	CHK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
	CHK(hipExtStreamCreateWithCUMask(&s_bg, (uint32_t)mask.size(), mask.data()));
	hipEvent_t ev[2]; CHK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
	struct { const char *name; hipStream_t st; int kern; } var[] = {
		{"hipMemcpyAsync, plain stream", s_plain, 0}, {"hipMemcpyAsync, CU-masked stream", s_mask, 0}, {"hipMemcpyAsync, high-priority stream", s_hi, 0},
		{"copy kernel, CU-masked stream", s_mask, 1}, {"copy kernel, high-priority stream", s_hi, 1},
	};
	// busy 1: background launches with more workgroups than the (masked) device holds at once -- a dispatch stays pending while its kernel runs;
	// busy 2: launches that fit (224 CUs x 4 workgroups of 256 threads: half the slots), one after the other; busy 3: 4x as many, 1/4 as long (bounded-life workgroups)
	const int n_masked = n_cu - n_cu / 8;
	for (int busy = 0; busy < 4; ++busy)
		for (auto &v : var) {
			if (busy == 1) for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(k_spin, dim3(n_cu * 8), dim3(256), 0, s_bg, 60000000ull, sink);
			if (busy == 2) for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(k_spin, dim3(n_masked * 4), dim3(256), 0, s_bg, 60000000ull, sink);
			if (busy == 3) for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(k_spin, dim3(n_masked * 8 * 8), dim3(256), 0, s_bg, 15000000ull, sink);
			const double t0 = now();
			size_t off = 0, prev_off = 0, prev_m = 0; int i = 0;
			for (; off < total; off += piece, ++i) {
				const size_t m = total - off < piece ? total - off : piece;
				char *p = (char*)pin + (size_t)(i & 1) * piece;
				if (v.kern) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, v.st, (const uint4*)((char*)dev + off), (uint4*)p, m >> 4);
				else { if (kernel_first) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, v.st, 100ull, sink); CHK(hipMemcpyAsync(p, (char*)dev + off, m, hipMemcpyDeviceToHost, v.st)); }
				CHK(hipEventRecord(ev[i & 1], v.st));
				if (i >= 1) { CHK(hipEventSynchronize(ev[(i - 1) & 1])); memcpy(host + prev_off, (char*)pin + (size_t)((i - 1) & 1) * piece, prev_m); }
				prev_off = off; prev_m = m;
			}
			CHK(hipStreamSynchronize(v.st));
			memcpy(host + prev_off, (char*)pin + (size_t)((i - 1) & 1) * piece, prev_m);
			const double t1 = now();
			CHK(hipStreamSynchronize(s_bg));
			printf("%-42s device %s: %7.1f ms for 1400 MB (%.1f GB/s)   [background drained %.0f ms later]\n", v.name, busy == 0 ? "idle" : busy == 1 ? "busy (oversubscribed launches)" : busy == 2 ? "busy (launches that fit)" : "busy (short-lived workgroups, oversubscribed)", (t1 - t0) * 1e3, 1.4 / (t1 - t0) * 1.048576, (now() - t1) * 1e3);
		}
	// one copy of the whole block into pinned memory (no staging)
	void *pin_all = nullptr;
	if (hipHostMalloc(&pin_all, total, hipHostMallocDefault) == hipSuccess) {
		for (int busy = 0; busy < 2; ++busy) {
			if (busy) for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(k_spin, dim3(n_cu * 8), dim3(256), 0, s_bg, 60000000ull, sink);
			const double t0 = now();
			CHK(hipMemcpyAsync(pin_all, dev, total, hipMemcpyDeviceToHost, s_mask));
			CHK(hipStreamSynchronize(s_mask));
			const double t1 = now();
			CHK(hipStreamSynchronize(s_bg));
			printf("%-42s device %s: %7.1f ms for 1400 MB (%.1f GB/s)\n", "one hipMemcpyAsync into pinned memory", busy ? "busy" : "idle", (t1 - t0) * 1e3, 1.4 / (t1 - t0) * 1.048576);
		}
	}
	return 0;
}
