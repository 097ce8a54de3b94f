// k_sw.hip -- K5: local Smith-Waterman with 2nd-best score and start recovery, ksw_align2
// (lib/aln/ksw.c:343-365) over ksw_u8 / ksw_i16 (ksw.c:111-334).  Used for mate rescue
// (lib/aln/mem_alnreg.c:395-493) and the long-read seed filter (lib/aln/memchain.c:501-535).
//
// One wavefront per job, H/E rows held in registers (lane l owns query columns l, l+64, ...), one
// target row per step.  The reference is Farrar's striped SSE2 kernel; what it computes is a
// row-wise DP over the query padded with zero-scoring columns to slen*p (p = 16 lanes for u8,
// 8 for i16) in which
//   * F(i,j) is a max-plus prefix scan of the row (the lazy-F loop converges to it), and
//   * E(i+1,j) opens from the H value the striped main loop had *before* lazy-F, i.e. with F
//     restricted to the stripe [k*slen,(k+1)*slen) containing j  (ksw.c:160-169).
// Both scans (full and stripe-segmented) are wave scans; u8 saturation/early stop (ksw.c:205,209),
// the b[] run merging behind score2/te2 (ksw.c:191-200,218-226) and the reverse pass with
// KSW_XSTOP (ksw.c:356-364) are reproduced exactly.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"

#include "sw_pass.hpp"

template <int NC, int TPB>
__global__ void __launch_bounds__(TPB)
k_sw(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order, long long n,
     bsx_sw_res_t *res, unsigned long long *bscratch, int bcap)
{
	const int lane = wave_lane();
	const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
	unsigned long long *b = bscratch + ((size_t)blockIdx.x * wpb + wave) * (size_t)bcap;
	for (long long jj = (long long)blockIdx.x * wpb + wave; jj < n; jj += (long long)gridDim.x * wpb) {
		const int job = order[jj];
		const bsx_sw_job_t J = jobs[job];
		const int8_t *mat = J.use_ct ? sc.ctmat : sc.gamat;
		const int is_u8 = (J.xtra & BSX_KSW_XBYTE) ? 1 : 0;
		int qv[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = (c << 6) + lane;
			int v = 5; // padding column: scores 0 against everything (ksw.c:96,105)
			if (j < J.qlen) { v = reads[(long long)J.qoff + (long long)j * J.qdir]; if (J.qcomp) v = v < 4 ? 3 - v : 4; }
			qv[c] = v;
		}
		SwPass r = sw_pass<NC>(ix, mat, is_u8, J.qlen, qv, J.tlen, J.tpos, J.tdir, -1, sc.o_del, sc.e_del, sc.o_ins, sc.e_ins, J.xtra, b, lane);
		bsx_sw_res_t o;
		o.score = r.score; o.te = r.te; o.qe = r.qe; o.score2 = r.score2; o.te2 = r.te2; o.tb = -1; o.qb = -1;
		const bool second = (J.xtra & BSX_KSW_XSTART) && !((J.xtra & BSX_KSW_XSUBO) && r.score < (J.xtra & 0xffff)) && r.qe >= 0;
		if (second) { // reverse pass on the prefixes ending at (qe, te) (ksw.c:357-364)
			int q2[NC];
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				const int j = (c << 6) + lane;
				int v = 5;
				if (j <= r.qe) { v = reads[(long long)J.qoff + (long long)(r.qe - j) * J.qdir]; if (J.qcomp) v = v < 4 ? 3 - v : 4; }
				q2[c] = v;
			}
			SwPass rr = sw_pass<NC>(ix, mat, is_u8, r.qe + 1, q2, J.tlen, J.tpos, J.tdir, r.te, sc.o_del, sc.e_del, sc.o_ins, sc.e_ins,
			                        BSX_KSW_XSTOP | r.score, b, lane);
			if (r.score == rr.score) { o.tb = r.te - rr.te; o.qb = r.qe - rr.qe; }
		}
		if (lane == 0) res[job] = o;
	}
}

template <int NC>
static void launch_sw_nc(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order,
                         long long n, bsx_sw_res_t *res, unsigned long long *bscratch, int bcap, int blocks)
{
	hipLaunchKernelGGL((k_sw<NC, 256>), dim3(blocks), dim3(256), 0, st, ix, sc, reads, jobs, order, n, res, bscratch, bcap);
}

void launch_sw(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order,
               long long n, bsx_sw_res_t *res, unsigned long long *bscratch, int bcap, int blocks, int nc)
{
	if (nc <= 4) launch_sw_nc<4>(st, ix, sc, reads, jobs, order, n, res, bscratch, bcap, blocks);
	else if (nc <= 16) launch_sw_nc<16>(st, ix, sc, reads, jobs, order, n, res, bscratch, bcap, blocks);
	else // queries up to 3072 columns (mates of paired reads longer than a kilobase): 48 register slots per lane, one wave per workgroup and SIMD
		hipLaunchKernelGGL((k_sw<48, 64>), dim3(blocks * 4), dim3(64), 0, st, ix, sc, reads, jobs, order, n, res, bscratch, bcap);
}
