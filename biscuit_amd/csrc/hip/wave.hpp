// wave.hpp -- 64-lane wavefront primitives used by the DP kernels (gfx950; wave64 hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
#define NEG_BIG (-(1 << 29))
// order LDS/global accesses of the lanes of one wavefront (no instruction beyond the waitcnt the fence implies)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off); v = v > o ? v : o; }
	return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off); v = v < o ? v : o; }
	return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	return v;
}
// inclusive max-scan across the wave
__device__ __forceinline__ int wave_scan_max_incl(int v)
{
	const int lane = wave_lane();
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(v, off); if (lane >= off) v = v > o ? v : o; }
	return v;
}
// segmented inclusive max-scan: `head` marks the first element of a segment.  On return `head`
// holds "a segment head occurred at or before this lane within the wave".
__device__ __forceinline__ int wave_segscan_max_incl(int v, int &head)
{
	const int lane = wave_lane();
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		int o = __shfl_up(v, off), oh = __shfl_up(head, off);
		if (lane >= off) { if (!head) v = v > o ? v : o; head |= oh; }
	}
	return v;
}
// value of lane `src` (uniform src)
__device__ __forceinline__ int wave_bcast(int v, int src) { return __shfl(v, src); }
// previous lane's value; lane 0 gets `first`
__device__ __forceinline__ int wave_prev(int v, int first)
{
	int o = __shfl_up(v, 1);
	return wave_lane() == 0 ? first : o;
}
