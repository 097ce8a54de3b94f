// wave.hpp -- 64-lane wavefront primitives used by the DP kernels (gfx950; wave64 hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
#define NEG_BIG (-(1 << 29))
// order LDS/global accesses of the lanes of one wavefront (no instruction beyond the waitcnt the fence implies)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63); }

// Cross-lane data movement goes through DPP (one VALU instruction, no LDS round trip) wherever the pattern allows:
// shifts inside a row of 16 lanes, the row broadcasts that stitch rows together, the whole-wave shift by one lane.
// __shfl_* compile to ds_bpermute, whose latency dominated the DP rows (a scan is six dependent steps).
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
// value of lane `src` (uniform src)
__device__ __forceinline__ int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }

// inclusive max-scan across the wave
__device__ __forceinline__ int wave_scan_max_incl(int v)
{
	// lanes without a source take INT_MIN, the identity of max: with it the compiler folds each step into one v_max_i32 with a
	// DPP operand instead of a DPP move followed by the max
	const int ID = (int)0x80000000;
	int t;
	t = __builtin_amdgcn_update_dpp(ID, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(ID, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(ID, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(ID, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(ID, v, DPP_ROW_BCAST15, 0xa, 0xf, false); v = v > t ? v : t;   // rows 1,3 take lane 15 of the row before
	t = __builtin_amdgcn_update_dpp(ID, v, DPP_ROW_BCAST31, 0xc, 0xf, false); v = v > t ? v : t;   // rows 2,3 take lane 31
	return v;
}
__device__ __forceinline__ int wave_scan_min_incl(int v)
{
	const int BIG = 0x7fffffff;
	int t;
	t = __builtin_amdgcn_update_dpp(BIG, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(BIG, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(BIG, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(BIG, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(BIG, v, DPP_ROW_BCAST15, 0xa, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(BIG, v, DPP_ROW_BCAST31, 0xc, 0xf, false); v = v < t ? v : t;
	return v;
}
__device__ __forceinline__ int wave_scan_sum_incl(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xa, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xc, 0xf, false);
	return v;
}
// reductions: the last lane of the inclusive scan holds the result; every lane gets it back as a scalar
__device__ __forceinline__ int wave_max_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_max_incl(v), 63); }
__device__ __forceinline__ int wave_min_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_min_incl(v), 63); }
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_sum_incl(v), 63); }
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
// previous lane's value; lane 0 gets `first`
__device__ __forceinline__ int wave_prev(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, DPP_WAVE_SHR1, 0xf, 0xf, false); }
