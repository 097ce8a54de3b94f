// k_extq.hip -- K4 in quarter-waves: ksw_extend2 (lib/aln/ksw.c:380-479), one extension job per ROW OF 16 LANES, four jobs per wavefront.
//
// Why: against an hg38-sized genome a strand search has a dozen chains after the filter and nearly all of them are one chance match of a
// 3-letter 19-mer; their extensions die after a handful of rows (measured: 16.5 M extensions per 1 M reads, 7.2 rows each) inside a band
// of a dozen columns.  A wavefront per extension (ext_dp.hpp) spends 64 lanes and a few hundred mostly scalar instructions on such a
// row, and the launch is bound by the one scalar unit a CU has.  Here a job owns one DPP row: the two DP rows of the reference's eh[]
// array live in registers, entry a in lane a & 15 of slot a >> 4; the max-plus prefix scan of F, the row maximum and the band
// bookkeeping are row_shr / row_ror DPP steps inside the 16 lanes, and everything the wave-wide form kept in scalar registers (band
// limits, scores, the row counter) is per-lane data that is equal across a job's lanes -- nothing is scalar, so four jobs in different
// rows of different bands advance with every trip of the loop.  Slots outside every job's band are skipped by wave-uniform branches.
//
// The recurrence, the band clamps and the stopping rules are those of ext_dp.hpp's ext_dp_reg (bit-exact with the reference:
// tests/test_gpu_golden.py runs this kernel too).  Needs qlen + 1 <= 16 * NCQ entries and scores below 2^21 (else the job is
// answered with score = EXTQ_DECLINED and the caller routes the strand search elsewhere).
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"

#ifndef EXTQ_OCC
#define EXTQ_OCC 2    // workgroups of four waves per CU the register allocation targets (2: no spills at 174 VGPRs; 4 spills 42 of them)
#endif
#define DPP_ROW_ROR(n) (0x120 + (n))
#define QID ((int)0x80000000)

// inclusive max-scan inside each row of 16 lanes
__device__ __forceinline__ int q_scan_max_incl(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v > t ? v : t;
	return v;
}
// maximum over the 16 lanes of a row, in every lane of it
__device__ __forceinline__ int q_allmax(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR(1), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR(2), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR(4), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR(8), 0xf, 0xf, false); v = v > t ? v : t;
	return v;
}
// the previous lane's value inside the row; lane 0 of the row gets `first`
__device__ __forceinline__ int q_prev(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, DPP_ROW_SHR(1), 0xf, 0xf, false); }
// lane 0 of the row gets lane 15's value (the others lane l - 1's)
__device__ __forceinline__ int q_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR(1), 0xf, 0xf, false); }

template <int NCQ>
__global__ void __launch_bounds__(256, EXTQ_OCC)
k_ext_q(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
        const unsigned int *n_ptr, unsigned int n_fixed, unsigned int *cursor, unsigned long long *prof)
{
	// scores of query base q against target bases 0..3, a byte each: [parent][q]
	__shared__ uint32_t s_sqp[2][8];
	if (threadIdx.x < 10) {
		const int p = threadIdx.x / 5, q = threadIdx.x % 5;
		const int8_t *mat = p ? sc.ctmat : sc.gamat;
		s_sqp[p][q] = (uint32_t)(uint8_t)mat[q] | (uint32_t)(uint8_t)mat[5 + q] << 8 | (uint32_t)(uint8_t)mat[10 + q] << 16 | (uint32_t)(uint8_t)mat[15 + q] << 24;
	}
	__syncthreads();
	const int lane = wave_lane(), l = lane & 15, gsh = lane & 48;
	const unsigned int n = n_ptr ? *n_ptr : n_fixed;
	const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
	bool run = false, done = false;
	unsigned int e = 0;
	int qlen = 0, tlen = 0, h0 = 0, w = 0, i = 0, beg = 0, end = 0, tdir = 1;
	long long tpos = 0;
	int max = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	int Hr[NCQ], Er[NCQ]; uint32_t sqp[NCQ];
	int tb_reg = 0;
#pragma unroll
	for (int c = 0; c < NCQ; ++c) { Hr[c] = Er[c] = 0; sqp[c] = 0; }
	unsigned int pf_rows = 0, pf_trips = 0, pf_jobs = 0;
	for (;;) {
		// ---- rows of 16 lanes without a job take the next ones of the queue (one atomic for all of them)
		const unsigned long long need = __ballot(!run && !done);
		if (need) {
			const unsigned long long heads = need & 0x0001000100010001ull;
			const int first = (int)__builtin_ctzll(need);
			unsigned int base = 0;
			if (lane == first) base = atomicAdd(cursor, (unsigned int)__popcll(heads));
			base = (unsigned int)__builtin_amdgcn_readlane((int)base, first);
			if (!run && !done) {
				e = base + (unsigned int)__popcll(heads & ((1ull << gsh) - 1));
				if (e >= n) done = true;
				else {
					const bsx_ext_job_t J = jobs[e];
					const int par = J.parent ? 1 : 0;
					qlen = J.qlen; tlen = J.tlen; h0 = J.h0; tdir = J.tdir; tpos = J.tpos;
					const int mx = par ? sc.mx_ct : sc.mx_ga;
					++pf_jobs;
					if (qlen + 1 > 16 * NCQ || qlen < 0 || (long long)h0 + (long long)qlen * mx >= (1 << 21)) { // not for this kernel's rows / packed maxima
						if (l == 0) { bsx_ext_res_t r; r.score = EXTQ_DECLINED; r.qle = r.tle = r.gtle = r.gscore = r.max_off = 0; res[e] = r; }
					} else if (tlen <= 0) { // no rows: what ksw_extend2 returns without entering its loop
						if (l == 0) { bsx_ext_res_t r; r.score = h0; r.qle = 0; r.tle = 0; r.gtle = 0; r.gscore = -1; r.max_off = 0; res[e] = r; }
					} else {
#pragma unroll
						for (int c = 0; c < NCQ; ++c) {
							const int a = (c << 4) + l;
							int q = a < qlen ? (int)reads[(long long)J.qoff + (long long)a * J.qdir] : 4;
							q = q < 4 ? q : 4;
							sqp[c] = s_sqp[par][q];
							const int v = a == 0 ? h0 : h0 - oe_ins - (a - 1) * e_ins;   // first row (ksw.c:395-397)
							Hr[c] = (a <= qlen && v > 0) ? v : 0;
							Er[c] = 0;
						}
						w = J.w;
						{ // band clamp (ksw.c:399-407)
							int max_ins = (int)((double)(qlen * mx + J.end_bonus - o_ins) / e_ins + 1.);
							max_ins = max_ins > 1 ? max_ins : 1;
							w = w < max_ins ? w : max_ins;
							int max_del = (int)((double)(qlen * mx + J.end_bonus - o_del) / e_del + 1.);
							max_del = max_del > 1 ? max_del : 1;
							w = w < max_del ? w : max_del;
						}
						max = h0; max_i = max_j = max_ie = -1; gscore = -1; max_off = 0;
						beg = 0; end = qlen; i = 0;
						run = true;
					}
				}
			}
		}
		if (__ballot(run) == 0) { if (__ballot(!done) == 0) break; continue; }
		++pf_trips;
		// ---- one row of every running job
		if (run && (i & 15) == 0) { // the reference bases of the next sixteen rows, one per lane
			const int r = i + l;
			tb_reg = r < tlen ? dev_ref_base(ix.pac, ix.l_pac, tpos + (long long)r * tdir) : 0;
		}
		const int t = __shfl(tb_reg, gsh | (i & 15));
		int h1_init = 0;
		if (run) {
			++pf_rows;
			if (beg < i - w) beg = i - w;
			if (end > i + w + 1) end = i + w + 1;
			if (end > qlen) end = qlen;
			if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		}
		int m = 0, mj = -1, h1_last = h1_init;
		const bool nonempty = run && beg < end;
		const int c_lo = beg >> 4, c_hi = end >> 4;   // slots holding entries beg .. end (entry `end` gets its E cleared)
		if (__ballot(nonempty)) {
			int hn[NCQ];
			int carry = NEG_BIG, lm = -1, lj = -1;
			const int tsh = (t & 3) << 3;
#pragma unroll
			for (int c = 0; c < NCQ; ++c) {
				hn[c] = 0;
				const bool in = nonempty && c >= c_lo && c <= c_hi;
				if (__ballot(in)) {
					if (in) {
						const int a = (c << 4) + l;
						const bool act = a >= beg && a < end;
						const int s = (int)(int8_t)(sqp[c] >> tsh);
						const int M = (act && Hr[c]) ? Hr[c] + s : 0;
						int tins = M - oe_ins; tins = tins > 0 ? tins : 0;
						const int g = act ? tins + a * e_ins : NEG_BIG;
						const int incl = q_scan_max_incl(g);
						int excl = q_prev(incl, NEG_BIG);
						excl = excl > carry ? excl : carry;                 // prefix max over all earlier columns
						{ const int tot = q_allmax(g); carry = carry > tot ? carry : tot; }
						int f = a == beg ? 0 : excl - (a - 1) * e_ins;
						if (f < 0) f = 0;
						if (act) {
							int h = M > Er[c] ? M : Er[c];
							h = h > f ? h : f;
							int tdel = M - oe_del; tdel = tdel > 0 ? tdel : 0;
							int ee = Er[c] - e_del; ee = ee > tdel ? ee : tdel;
							Er[c] = ee;
							hn[c] = h;
							if (h >= lm) { lm = h; lj = a; }
						} else if (a == end) Er[c] = 0;
					}
				}
			}
			// H: entry a takes h(i, a-1) for a-1 in the band, entry beg takes the first-column value
#pragma unroll
			for (int c = NCQ - 1; c >= 0; --c) {
				const bool in = nonempty && c >= c_lo && c <= c_hi;
				if (__ballot(in)) {
					if (in) {
						const int a = (c << 4) + l;
						int up = q_prev(hn[c], 0);
						const int edge = c > 0 ? q_ror1(hn[c > 0 ? c - 1 : 0]) : 0;
						if (l == 0) up = edge;
						if (a == beg) Hr[c] = h1_init;
						else if (a - 1 >= beg && a - 1 < end) Hr[c] = up;
					}
				}
			}
			if (nonempty) { // row maximum and the last column that attains it in one reduction: (h << 9 | column), columns < 512
				const int key = q_allmax(lm < 0 ? -1 : (lm << 9 | lj));
				m = key >> 9; mj = key < 0 ? -1 : (key & 511);
			}
			if (__ballot(nonempty && end == qlen)) { // h(i, end-1), only read for the to-the-end score
				int v = 0;
#pragma unroll
				for (int c = 0; c < NCQ; ++c) if ((c << 4) + l == end - 1) v = hn[c];
				v = q_allmax(v);   // h >= 0
				if (nonempty && end == qlen) h1_last = v;
			}
		}
		if (__ballot(run && !nonempty)) { // empty row: only the boundary cell is written (ksw.c:449)
#pragma unroll
			for (int c = 0; c < NCQ; ++c) if (run && !nonempty && (c << 4) + l == end) { Hr[c] = h1_init; Er[c] = 0; }
		}
		bool stop = false;
		if (run) {
			const int jfin = beg < end ? end : beg;
			if (jfin == qlen) { max_ie = gscore > h1_last ? max_ie : i; gscore = gscore > h1_last ? gscore : h1_last; }
			stop = m == 0;
			if (!stop) {
				if (m > max) {
					max = m; max_i = i; max_j = mj;
					int off = mj - i; off = off < 0 ? -off : off;
					max_off = max_off > off ? max_off : off;
				} else if (zdrop > 0) {
					if (i - max_i > mj - max_j) stop = max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop;
					else stop = max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop;
				}
			}
		}
		const bool shrink = run && !stop;
		if (__ballot(shrink)) { // the band of the next row: the non-zero cells (ksw.c:466-469)
			int fz = 0x7fffffff, lz = -1;
#pragma unroll
			for (int c = 0; c < NCQ; ++c) {
				const bool in = shrink && c >= c_lo && c <= c_hi;
				if (__ballot(in)) {
					const int a = (c << 4) + l;
					const bool nz = in && a >= beg && a <= end && (Hr[c] != 0 || Er[c] != 0);
					if (nz && a < end) fz = fz < a ? fz : a;
					if (nz) lz = lz > a ? lz : a;
				}
			}
			int nb = -q_allmax(-fz);
			int last = q_allmax(lz);
			if (shrink) {
				nb = nb < end ? nb : end;
				last = last > nb - 1 ? last : nb - 1;
				beg = nb;
				end = last + 2 < qlen ? last + 2 : qlen;
				++i;
				if (i >= tlen) stop = true;
			}
		}
		if (run && stop) {
			if (l == 0) { bsx_ext_res_t r; r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off; res[e] = r; }
			run = false;
		}
	}
	if (prof) { // tracing: jobs, rows and trips (a trip advances up to four rows)
		pf_rows = (unsigned int)wave_sum_i32(l == 0 ? (int)pf_rows : 0); pf_jobs = (unsigned int)wave_sum_i32(l == 0 ? (int)pf_jobs : 0);
		if (lane == 0) { atomicAdd(&prof[0], (unsigned long long)pf_jobs); atomicAdd(&prof[1], (unsigned long long)pf_rows); atomicAdd(&prof[2], (unsigned long long)pf_trips); }
	}
}

int ext_q_max_query(int ncq) { return 16 * ncq - 1; }

// jobs[0 .. n) -> res; n is *n_ptr if n_ptr is given (a device counter), else n_fixed; *cursor must be zero
void launch_ext_q(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                  const unsigned int *n_ptr, unsigned int n_upper, unsigned int *cursor, int max_qlen, unsigned long long *prof)
{
	// a workgroup = 4 waves = 16 jobs in flight; persistent rows (they take jobs until the queue is empty)
	const long long want = ((long long)n_upper + 15) / 16;
	const int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)n_cu * 8));
	if (max_qlen <= ext_q_max_query(10))
		hipLaunchKernelGGL(k_ext_q<10>, dim3(grid), dim3(256), 0, st, ix, sc, reads, jobs, res, n_ptr, n_upper, cursor, prof);
	else
		hipLaunchKernelGGL(k_ext_q<16>, dim3(grid), dim3(256), 0, st, ix, sc, reads, jobs, res, n_ptr, n_upper, cursor, prof);
}
