// gather64.hip -- what HBM gives for the FM-index access pattern: dependent chains of random 64-byte block reads
// over a table of several GB.  Variants:
//   mode 0: one lane reads a whole block (4 x 16 B, what k_seed r01 does), `C` independent chains per lane
//   mode 1: four lanes read one block (16 B each: one 64-B request), the 16-lane group holds 4 chains; C chains per group slot
// Output: GB/s of 64-B blocks delivered.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int C>
__global__ void __launch_bounds__(256) k_lane(const uint4 *tab, uint64_t n_blocks, int steps, uint64_t *out)
{
	uint64_t idx[C], acc = 0;
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (int c = 0; c < C; ++c) idx[c] = mix(gid * C + c + 1) % n_blocks;
	for (int s = 0; s < steps; ++s) {
		uint4 v[C][4];
#pragma unroll
		for (int c = 0; c < C; ++c) { const uint4 *p = tab + idx[c] * 4; v[c][0] = p[0]; v[c][1] = p[1]; v[c][2] = p[2]; v[c][3] = p[3]; }
#pragma unroll
		for (int c = 0; c < C; ++c) {
			uint64_t h = v[c][0].x ^ v[c][1].y ^ v[c][2].z ^ v[c][3].w;
			acc += h;
			idx[c] = mix(idx[c] + h + s) % n_blocks;
		}
	}
	out[gid] = acc;
}

// four lanes per block: lane q of the quad loads bytes [16q, 16q+16); the next index is agreed inside the quad by xor-shuffle
template <int C>
__global__ void __launch_bounds__(256) k_quad(const uint4 *tab, uint64_t n_blocks, int steps, uint64_t *out)
{
	uint64_t idx[C], acc = 0;
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int q = threadIdx.x & 3;
	for (int c = 0; c < C; ++c) idx[c] = mix((gid >> 2) * C + c + 1) % n_blocks;
	for (int s = 0; s < steps; ++s) {
		uint4 v[C];
#pragma unroll
		for (int c = 0; c < C; ++c) v[c] = tab[idx[c] * 4 + q];
#pragma unroll
		for (int c = 0; c < C; ++c) {
			unsigned int h = v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
			h ^= __shfl_xor(h, 1); h ^= __shfl_xor(h, 2);
			acc += h;
			idx[c] = mix(idx[c] + h + s) % n_blocks;
		}
	}
	out[gid] = acc;
}

int main(int argc, char **argv)
{
	const double gb = argc > 1 ? atof(argv[1]) : 3.1;
	const uint64_t n_blocks = (uint64_t)(gb * 1e9 / 64);
	uint4 *tab; uint64_t *out;
	CHK(hipMalloc(&tab, n_blocks * 64));
	CHK(hipMemset(tab, 0x5a, n_blocks * 64));
	const int grid_max = 256 * 32;
	CHK(hipMalloc(&out, (size_t)grid_max * 256 * 8));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	printf("table %.2f GB (%llu blocks)\n", n_blocks * 64 / 1e9, (unsigned long long)n_blocks);
	for (int mode = 0; mode < 2; ++mode)
	for (int C = 1; C <= 4; C <<= 1)
	for (int wpc = 8; wpc <= 32; wpc <<= 1) {   // waves per CU (256-thread blocks: 4 waves each)
		const int grid = 256 * wpc / 4, steps = 400;
		float ms = 0;
		for (int rep = 0; rep < 2; ++rep) {
			CHK(hipEventRecord(e0));
			if (mode == 0) { if (C == 1) k_lane<1><<<grid, 256>>>(tab, n_blocks, steps, out); else if (C == 2) k_lane<2><<<grid, 256>>>(tab, n_blocks, steps, out); else k_lane<4><<<grid, 256>>>(tab, n_blocks, steps, out); }
			else { if (C == 1) k_quad<1><<<grid, 256>>>(tab, n_blocks, steps, out); else if (C == 2) k_quad<2><<<grid, 256>>>(tab, n_blocks, steps, out); else k_quad<4><<<grid, 256>>>(tab, n_blocks, steps, out); }
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
		}
		const double blocks = (double)grid * 256 * steps * C / (mode ? 4 : 1);
		printf("mode %s chains/lane-slot %d waves/CU %2d: %8.1f GB/s (%.0f M blocks/s, %.2f ms)\n", mode ? "quad(4 lanes x 16B)" : "lane(1 lane x 64B) ", C, wpc, blocks * 64 / ms / 1e6, blocks / ms / 1e3, ms);
	}
	return 0;
}
