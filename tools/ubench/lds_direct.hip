// global_load_lds_dwordx4 on gfx950: where does lane l's 16 bytes land when some lanes are masked off, and does the LDS pointer
// operand move the base?  (k_seed's block exchange wants to load its FM blocks straight into LDS.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4 *src, uint4 *dst, const int *idx, int base_slot)
{
	__shared__ uint4 buf[512];
	const int lane = threadIdx.x;
	for (int i = lane; i < 512; i += 64) buf[i] = make_uint4(0xdeadbeefu, 0, 0, 0);
	__syncthreads();
	if (idx[lane] >= 0) {
		const uint4 *p = src + idx[lane];
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(buf + base_slot), 16, 0, 0);
	}
	__builtin_amdgcn_s_waitcnt(0);
	__syncthreads();
	for (int i = lane; i < 512; i += 64) dst[i] = buf[i];
}
int main()
{
	std::vector<uint4> h(1024); for (int i = 0; i < 1024; ++i) h[i] = make_uint4(i, i * 3, i * 5, i * 7);
	std::vector<int> idx(64); for (int l = 0; l < 64; ++l) idx[l] = (l % 3 == 1) ? -1 : (l * 37 + 11) % 1024;
	uint4 *ds, *dd; int *di;
	hipMalloc(&ds, 1024 * 16); hipMalloc(&dd, 512 * 16); hipMalloc(&di, 256);
	hipMemcpy(ds, h.data(), 1024 * 16, hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), 256, hipMemcpyHostToDevice);
	for (int base : {0, 64, 200}) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, dd, di, base);
		std::vector<uint4> o(512); hipMemcpy(o.data(), dd, 512 * 16, hipMemcpyDeviceToHost);
		int ok = 1, touched = 0;
		for (int i = 0; i < 512; ++i) {
			const int l = i - base;
			const bool expect = l >= 0 && l < 64 && idx[l] >= 0;
			if (expect) { if (o[i].x != (unsigned)idx[l] || o[i].w != (unsigned)idx[l] * 7) ok = 0; }
			else if (o[i].x != 0xdeadbeefu) { ok = 0; ++touched; }
		}
		printf("base %d: lane l -> slot base + l, masked lanes leave their slot alone: %s (stray writes %d)\n", base, ok ? "yes" : "NO", touched);
	}
	return 0;
}
