// k_extend.hip -- K4: banded affine-gap seed extension, ksw_extend2 (lib/aln/ksw.c:380-479).
//
// One wavefront per extension job; 64 lanes sweep the active band of one target row at a time.
// The reference opens gaps from the diagonal term M = H(i-1,j-1)+s and not from H(i,j)
// (ksw.c:433-447), so within a row
//     F(j) = max_{k<j} ( max(M(k)-oe_ins,0) - (j-1-k)*e_ins )
// is a pure max-plus prefix scan of previous-row data: a row needs one wave scan per 64 columns,
// rows stay sequential.  H/E live in LDS in exactly the reference's in-place eh[] layout
// (eh[j].h = H(i-1,j-1), eh[j].e = E(i,j)) because the adaptive band can re-grow over cells it
// skipped earlier and must then see what the reference sees there.  All reads of a row happen
// before any write of that row, which makes the parallel sweep equivalent to the sequential one.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"
#include "ext_dp.hpp"

template <int NC>
__global__ void __launch_bounds__(256)
k_extend(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order, long long n,
         bsx_ext_res_t *res, int qcap, int32_t *hbm_rows)
{
	extern __shared__ int32_t lds[];
	const int lane = wave_lane();
	const int waves_per_block = blockDim.x >> 6;
	const int wave = threadIdx.x >> 6;
	const int stride = 2 * (qcap + 2) + ((qcap + 3) >> 2) + 1;
	// hbm_rows: the rows of queries too long for LDS live in a per-wave slab in HBM (reads of tens of kilobases: rare, and slow)
	int32_t *H = hbm_rows ? hbm_rows + ((size_t)blockIdx.x * waves_per_block + wave) * (size_t)stride : lds + wave * stride;
	int32_t *E = H + (qcap + 2);
	uint8_t *qb = reinterpret_cast<uint8_t*>(E + (qcap + 2));

	for (long long jj = (long long)blockIdx.x * waves_per_block + wave; jj < n; jj += (long long)gridDim.x * waves_per_block) {
		const int job = order[jj];
		const bsx_ext_job_t J = jobs[job];
		const bsx_ext_res_t r = ext_dp<NC>(ix, sc, reads, J, H, E, qb, lane);
		if (lane == 0) res[job] = r;
		WAVE_SYNC();
	}
}

template <int NC>
static void launch_ext_nc(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs,
                          const int *order, long long n, bsx_ext_res_t *res, int qcap, int n_cu)
{
	const int stride = 2 * (qcap + 2) + ((qcap + 3) >> 2) + 1;
	int wpb = 4;
	while (wpb > 1 && (size_t)wpb * stride * 4 > 60 * 1024) wpb >>= 1;
	const size_t lds = (size_t)wpb * stride * 4;
	long long blocks = (n + wpb - 1) / wpb;
	long long cap = (long long)n_cu * 32;
	if (blocks > cap) blocks = cap;
	if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_extend<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(k_extend<NC>, dim3((unsigned)blocks), dim3(wpb * 64), lds, st, ix, sc, reads, jobs, order, n, res, qcap, (int32_t*)nullptr);
}
size_t extend_hbm_row_bytes(int qcap) { return (size_t)(2 * (qcap + 2) + ((qcap + 3) >> 2) + 1) * 4; }
void launch_extend_hbm(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order,
                       long long n, bsx_ext_res_t *res, int qcap, int blocks, void *rows)
{
	hipLaunchKernelGGL(k_extend<32>, dim3((unsigned)blocks), dim3(256), 0, st, ix, sc, reads, jobs, order, n, res, qcap, (int32_t*)rows);
}

void launch_extend(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order,
                   long long n, bsx_ext_res_t *res, int qcap, int nc, int n_cu)
{
	if (nc <= 4) launch_ext_nc<4>(st, ix, sc, reads, jobs, order, n, res, qcap, n_cu);
	else if (nc <= 8) launch_ext_nc<8>(st, ix, sc, reads, jobs, order, n, res, qcap, n_cu);
	else launch_ext_nc<32>(st, ix, sc, reads, jobs, order, n, res, qcap, n_cu);
}

// tests: plain jobs through the register window that follows the band (ext_dp_win, ext_dp.hpp: what the chains -> regions launch of chunks with
// long reads extends with), a wavefront per job; a job it cannot hold (a band of more than 256 columns, a score of 2^21) is answered X4_DECLINED
__global__ void __launch_bounds__(256)
k_extend_win(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_ext_job_t *jobs, long long n, bsx_ext_res_t *res)
{
	const int lane = wave_lane();
	const int waves_per_block = blockDim.x >> 6, wave = threadIdx.x >> 6;
	for (long long jj = (long long)blockIdx.x * waves_per_block + wave; jj < n; jj += (long long)gridDim.x * waves_per_block) {
		const bsx_ext_job_t J = jobs[jj];
		const int mx = J.parent ? sc.mx_ct : sc.mx_ga;
		bsx_ext_res_t r;
		if (2 * J.w + 1 <= 256 && (long long)J.h0 + (long long)J.qlen * mx < (1 << 21)) r = ext_dp_win<5>(ix, sc, reads, J, lane);
		else { r.score = X4_DECLINED; r.qle = r.tle = r.gtle = r.gscore = r.max_off = 0; }
		if (lane == 0) res[jj] = r;
	}
}
void launch_extwin_batch(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res, long long n)
{
	long long blocks = (n + 3) / 4;
	if (blocks > (long long)n_cu * 8) blocks = (long long)n_cu * 8;
	hipLaunchKernelGGL(k_extend_win, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, st, ix, sc, reads, jobs, n, res);
}
