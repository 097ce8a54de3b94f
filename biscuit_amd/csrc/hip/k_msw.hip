// k_msw.hip -- mate rescue's PLAN on the device (round 6): which of a pair's candidates need an alignment, and over which window.
//
// mem_alnreg_matesw (lib/aln/mem_alnreg.c:385-513) looks, for each of the up to max_matesw best regions of a read, whether the mate already has
// a region at a proper distance and, when it has none, aligns the mate against the window where one should be (ksw_align2), adding what it finds
// to the mate's list -- so what a candidate sees depends on the candidates before it.  The host runs that as plan / K5 batch / replay
// (csrc/host/pipeline.c): a first pass over the lists AS THEY ARE collects every alignment that could be needed, one batch computes them, a second
// pass replays the reference's loop with the results at hand.  The first pass is a pure function of the lists before rescue -- a lane per
// candidate here, over the regions the region kernels left in HBM in the order k_dedup / k_dedup_long gave them -- and its jobs never leave the
// device: they are binned by (stripes, window length) for k_swl's four-jobs-a-wavefront form and run from where they lie.  The host receives,
// per pair, where its jobs start and which candidates have one (a bit each), and the results; its replay finds them in its slots and is the
// only pass it runs.  A candidate the plan skipped and the replay does not (a region the skip rested on was removed by a later insertion's
// de-duplication) asks for its alignment the way it always did: a second, small round.
// Pairs left to the host's own plan: a read the device did not de-duplicate (dd_n < 0), lists not in score order (the candidates must be a
// prefix), more than 64 candidates a read.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "kernels.h"
#include "wave.hpp"

struct MswArgs {
	const bsx_region_t *regs; const long long *r_off; const int *r_n;        // the chunk's regions, per strand search
	const int *dd_n; const unsigned char *dd_idx; const long long *dd_off; const unsigned short *dd_pool; int dd_cap, per_read;
	const unsigned int *roff;                                                // reads in the chunk's read buffer: roff[r] .. roff[r + 1]
	int low, high, pen_unpaired, max_matesw, min_seed_len, a;
	int p0, n_pairs;
};
struct MswPair { int base; int n_c[2]; int pad; unsigned long long mask[2]; };   // base: the pair's first job (-1: left to the host's plan); mask[i]: candidates of read i with a job
#define MSW_NK (17 * 2048)

__device__ __forceinline__ int msw_pos2rid(const DevIndex &ix, long long pos_f)   // bns_pos2rid (bntseq.c:356-369) as csrc/host/index.c has it
{
	if (pos_f >= ix.l_pac) return -1;
	int left = 0, mid = 0, right = ix.n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= ix.ctg_off[mid]) {
			if (mid == ix.n_seqs - 1) break;
			if (pos_f < ix.ctg_off[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
// region k of read r in the order the de-duplication left
__device__ __forceinline__ const bsx_region_t *msw_reg(const MswArgs &A, int r, int k)
{
	const long long lo = A.dd_off ? A.dd_off[r] : -1;
	int li = lo >= 0 ? (int)A.dd_pool[lo + k] : (int)A.dd_idx[(size_t)r * A.dd_cap + k];
	int t = r * A.per_read;
	for (int u = 0; u < A.per_read - 1; ++u) { const int m = A.r_n[t]; if (li < m) break; li -= m; ++t; }
	return A.regs + A.r_off[t] + li;
}

__global__ void __launch_bounds__(256)
k_msw_plan(DevIndex ix, MswArgs A, bsx_sw_job_t *jobs, unsigned int job_cap, unsigned int *job_count, unsigned int *hist, MswPair *table)
{
	const int lane = wave_lane();
	const long long l_pac = ix.l_pac;
	const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * blockDim.x) >> 6);
	for (int pp = wave; pp < A.n_pairs; pp += n_waves) {
		const int p = A.p0 + pp;
		MswPair T; T.base = -1; T.n_c[0] = T.n_c[1] = 0; T.pad = 0; T.mask[0] = T.mask[1] = 0;
		const int n0 = A.dd_n[2 * p], n1 = A.dd_n[2 * p + 1];
		bool host = n0 < 0 || n1 < 0;
		bsx_sw_job_t J[2]; bool want[2] = {false, false};
		for (int i = 0; i < 2 && !host; ++i) {
			const int r = 2 * p + i, rm = 2 * p + (i ^ 1), n = i ? n1 : n0, nm = i ? n0 : n1;
			const int l_ms = (int)(A.roff[rm + 1] - A.roff[rm]);
			if (n == 0) continue;
			const int thr = msw_reg(A, r, 0)->score - A.pen_unpaired;
			// the candidates: the regions within pen_unpaired of the best, the first max_matesw of them -- a prefix of a list in score order
			const bool in = lane < n;
			bsx_region_t R; R.score = 0x7fffffff; R.rb = R.re = 0; R.qb = R.qe = 0; R.rid = -1; R.bss = 0;
			if (in) R = *msw_reg(A, r, lane);
			const unsigned long long qm = __ballot(in && R.score >= thr);
			int nc = __popcll(qm);
			if (qm != (nc >= 64 ? ~0ull : (1ull << nc) - 1)) { host = true; break; }     // not a prefix
			if (nc == 64 && n > 64 && A.max_matesw > 64) { host = true; break; }         // more candidates than lanes
			if (nc > A.max_matesw) nc = A.max_matesw;
			const bool cand = lane < nc;
			// does the mate have a region at a proper distance already? (mem_alnreg.c:395-401; bsx_reg_isize)
			bool skip = false;
			{
				const int isrev1 = R.rb > l_pac;
				const long long pos1 = isrev1 ? (l_pac << 1) - 1 - R.rb : R.rb;
				const int len1 = R.qe - R.qb;
				for (int m = 0; m < nm; ++m) {
					const bsx_region_t *Q = msw_reg(A, rm, m);
					const long long qrb = Q->rb; const int qrid = Q->rid, len2 = Q->qe - Q->qb;
					if (qrid != R.rid) continue;
					const int isrev2 = qrb > l_pac;
					const long long pos2 = isrev2 ? (l_pac << 1) - 1 - qrb : qrb;
					long long is; bool ok = false;
					if (isrev1 && !isrev2) { is = pos1 - pos2 + len1; ok = true; }
					else if (isrev2 && !isrev1) { is = pos2 - pos1 + len2; ok = true; }
					if (ok && is >= A.low && is <= A.high) skip = true;
				}
			}
			// the window (mem_alnreg.c:403-419) and the job
			bool job = false;
			if (cand && !skip && l_ms > 0) {
				long long rb = R.rb + A.low - l_ms, re = R.rb + A.high;
				rb = rb > 0 ? rb : 0; re = re < l_pac << 1 ? re : l_pac << 1;
				int rid = -1;
				if (rb < re) { // bns_fetch_seq's clamp to the contig of the window's middle (bntseq.c:428-452)
					const long long mid = (rb + re) >> 1;
					const int is_rev = mid >= l_pac;
					rid = msw_pos2rid(ix, is_rev ? (l_pac << 1) - 1 - mid : mid);
					long long far_beg = ix.ctg_off[rid], far_end = ix.ctg_off[rid + 1];
					if (is_rev) { const long long t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
					rb = rb > far_beg ? rb : far_beg; re = re < far_end ? re : far_end;
				}
				if (R.rid == rid && re - rb >= A.min_seed_len) {
					const int parent = R.bss ^ (R.rb < l_pac ? 1 : 0);
					job = true;
					J[i].tpos = rb; J[i].qoff = A.roff[rm] + (unsigned int)l_ms - 1; J[i].qlen = l_ms; J[i].tlen = (int)(re - rb);
					J[i].xtra = BSX_KSW_XSUBO | BSX_KSW_XSTART | (l_ms * A.a < 250 ? BSX_KSW_XBYTE : 0) | (A.min_seed_len * A.a);
					J[i].qdir = -1; J[i].tdir = 1; J[i].qcomp = 1; J[i].use_ct = (uint8_t)(parent ? 0 : 1);
				}
			}
			want[i] = job;
			T.n_c[i] = nc;
			T.mask[i] = __ballot(job);
		}
		if (host) { if (lane == 0) { T.base = -1; T.n_c[0] = T.n_c[1] = 0; T.mask[0] = T.mask[1] = 0; table[pp] = T; } continue; }
		const int c0 = __popcll(T.mask[0]), c1 = __popcll(T.mask[1]);
		unsigned int base = 0;
		if (lane == 0 && c0 + c1) base = atomicAdd(job_count, (unsigned int)(c0 + c1));
		base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
		if (c0 + c1 && base + (unsigned int)(c0 + c1) > job_cap) { if (lane == 0) { T.base = -1; T.n_c[0] = T.n_c[1] = 0; T.mask[0] = T.mask[1] = 0; table[pp] = T; } continue; }   // (no room: the host's plan)
		const unsigned long long lt = (1ull << lane) - 1;
		for (int i = 0; i < 2; ++i) if (want[i]) {
			const unsigned int at = base + (i ? (unsigned int)c0 : 0u) + (unsigned int)__popcll(T.mask[i] & lt);
			jobs[at] = J[i];
			atomicAdd(&hist[((J[i].qlen + 15) >> 4) * 2048 + (2047 - (J[i].tlen < 2047 ? J[i].tlen : 2047))], 1u);
		}
		if (lane == 0) { T.base = (int)base; table[pp] = T; }
	}
}

// the jobs in k_swl's order: by stripe count, then by window length, longest first (lane_sw_batch's counting sort; the order inside a bin is
// whatever the atomics make it: every job's result is its own)
__global__ void __launch_bounds__(1024)
k_msw_scan(unsigned int *hist)   // hist -> exclusive prefix sums, in place
{
	__shared__ unsigned int part[1024];
	const int per = (MSW_NK + 1023) / 1024, t = (int)threadIdx.x;
	unsigned int s = 0;
	for (int k = 0; k < per; ++k) { const int b = t * per + k; if (b < MSW_NK) s += hist[b]; }
	part[t] = s;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) { const unsigned int v = t >= off ? part[t - off] : 0u; __syncthreads(); part[t] += v; __syncthreads(); }
	unsigned int run = t ? part[t - 1] : 0u;
	for (int k = 0; k < per; ++k) { const int b = t * per + k; if (b < MSW_NK) { const unsigned int c = hist[b]; hist[b] = run; run += c; } }
}
__global__ void __launch_bounds__(256)
k_msw_scatter(const bsx_sw_job_t *jobs, const unsigned int *job_count, unsigned int job_cap, unsigned int *binpos, int *order)
{
	unsigned int n = *job_count; if (n > job_cap) n = job_cap;
	for (unsigned int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const bsx_sw_job_t J = jobs[j];
		order[atomicAdd(&binpos[((J.qlen + 15) >> 4) * 2048 + (2047 - (J.tlen < 2047 ? J.tlen : 2047))], 1u)] = (int)j;
	}
}

size_t msw_pair_bytes(void) { return sizeof(MswPair); }
int msw_hist_bins(void) { return MSW_NK; }
void launch_msw_plan(hipStream_t st, int n_cu, const DevIndex &ix, const bsx_region_t *regs, const long long *r_off, const int *r_n,
                     const int *dd_n, const unsigned char *dd_idx, const long long *dd_off, const unsigned short *dd_pool, int dd_cap, int per_read,
                     const unsigned int *roff, int low, int high, int pen_unpaired, int max_matesw, int min_seed_len, int a, int p0, int n_pairs,
                     bsx_sw_job_t *jobs, unsigned int job_cap, unsigned int *job_count, unsigned int *hist, void *table, int *order)
{
	MswArgs A;
	A.regs = regs; A.r_off = r_off; A.r_n = r_n; A.dd_n = dd_n; A.dd_idx = dd_idx; A.dd_off = dd_off; A.dd_pool = dd_pool; A.dd_cap = dd_cap; A.per_read = per_read;
	A.roff = roff; A.low = low; A.high = high; A.pen_unpaired = pen_unpaired; A.max_matesw = max_matesw; A.min_seed_len = min_seed_len; A.a = a; A.p0 = p0; A.n_pairs = n_pairs;
	const int blocks = (int)std::min<long long>(((long long)n_pairs + 3) / 4, (long long)n_cu * 8);
	hipLaunchKernelGGL(k_msw_plan, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, ix, A, jobs, job_cap, job_count, hist, (MswPair*)table);
	hipLaunchKernelGGL(k_msw_scan, dim3(1), dim3(1024), 0, st, hist);
	hipLaunchKernelGGL(k_msw_scatter, dim3(n_cu * 4), dim3(256), 0, st, (const bsx_sw_job_t*)jobs, (const unsigned int*)job_count, job_cap, hist, order);
}
