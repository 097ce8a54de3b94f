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
#include <stdlib.h>
#include <algorithm>
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
        const unsigned int *n_ptr, unsigned int n_fixed, unsigned int *cursor, const int *list, unsigned long long *prof)
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
					if (list) e = (unsigned int)list[e];   // the jobs named by the list (what k_ext_n left)
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


// ---------------------------------------------------------------------------------------------------------------------
// k_ext_n: ksw_extend2 for the NARROW jobs, a LANE per job.
//
// Most extensions of a strand search against an hg38-sized genome start from a chance match (h0 = 19..21): the scores decay by one
// per row, the band is a dozen columns wide, and the job is over after ~30 rows.  Such a job keeps no lane group busy, but 64 of
// them keep a wavefront busy: every lane runs the reference's own loops (rows, columns of the band, the two band-shrinking scans)
// as a state machine that does one cell per trip, so lanes in different rows of different bands advance together.  The eh[] row of a
// job is a ring of 32 entries in this wave's LDS (column j at slot j & 31, H and E 16 bits each): with h0 that small every entry
// outside the live band is zero -- the first row's non-zero values reach column (h0 - oe_ins) / e_ins, the band only ever leaves
// zero entries behind (ksw.c:466-469), and the cells of row 0 beyond the last non-zero value stay zero (they are not computed: see
// `trunc`) -- so a slot that comes round again reads what the reference reads there.  The first 64 query bases (2 bits each) and the
// pac bytes under the first 64 reference bases are loaded into registers when the job starts: nothing in the loop waits for HBM.
// A job that outgrows any of this (band wider than the ring, beyond column or row 63, an ambiguous base, a band clamp that cuts into
// live cells) is put on `wide_list` untouched and done by k_ext_q.
#define XN_RING 32
enum { XN_FETCH = 0, XN_ROW, XN_CELL, XN_ROWEND, XN_SHL, XN_SHR, XN_DONE };

__global__ void __launch_bounds__(64)
k_ext_n(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
        const unsigned int *n_ptr, unsigned int n_fixed, unsigned int *cursor, int *wide_list, unsigned int *wide_count, unsigned long long *prof, unsigned int cold_mask)
{
	__shared__ uint32_t eh_lds[XN_RING * 64];
	__shared__ uint32_t s_srow[2][4];   // [parent][target base]: scores against query bases 0..3, a byte each
	if (threadIdx.x < 8) {
		const int p = threadIdx.x >> 2, t = threadIdx.x & 3;
		const int8_t *mat = p ? sc.ctmat : sc.gamat;
		s_srow[p][t] = (uint32_t)(uint8_t)mat[t * 5] | (uint32_t)(uint8_t)mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)mat[t * 5 + 2] << 16 | (uint32_t)(uint8_t)mat[t * 5 + 3] << 24;
	}
	__syncthreads();
	const int lane = (int)threadIdx.x;
	uint32_t *eh = eh_lds + lane;
	const unsigned int n = n_ptr ? *n_ptr : n_fixed;
	const int o_del = sc.o_del, e_del = sc.e_del, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = sc.o_ins + e_ins, zdrop = sc.zdrop;
	int state = XN_FETCH;
	unsigned int e = 0;
	int qlen = 0, tlen = 0, h0 = 0, w = 0, par = 0;
	uint32_t q2a = 0, q2b = 0, q2c = 0, q2d = 0;           // query bases 0..63, 2 bits each
	uint32_t tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;  // pac bytes [tbyte0, tbyte0 + 20)
	long long tf0 = 0; int tfd = 1, tcomp = 0; long long tbyte0 = 0;   // forward coordinate of row 0's base, its step per row, complement?
	int i = 0, j = 0, beg = 0, end = 0, zhi = 0, f = 0, m = 0, mj = -1, h1 = 0, jtop = 0, trunc = 0;
	int max = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	uint32_t srow = 0;
	unsigned int pf_cells = 0, pf_rows = 0, pf_jobs = 0, pf_wide = 0, pf_trips = 0;
#define XN_DECLINE() do { wide_list[atomicAdd(wide_count, 1u)] = (int)e; ++pf_wide; state = XN_FETCH; } while (0)
	for (;;) {
		++pf_trips;
		// ---- every trip: a cell, or a step of one of the band scans
#define XN_ONE_CELL() do { \
			if (j >= zhi && f == 0 && h1 == 0) { trunc = 1; jtop = j; state = XN_ROWEND; }   /* the rest of the row is zero and stays zero */ \
			else if (j - beg >= XN_RING || j >= 64) XN_DECLINE(); \
			else { \
				const uint32_t x = eh[(j & (XN_RING - 1)) * 64]; \
				int M = (int)(x & 0xffffu), ee = (int)(x >> 16); \
				const uint32_t qw = j < 16 ? q2a : j < 32 ? q2b : j < 48 ? q2c : q2d; \
				const int q = (int)((qw >> ((j & 15) << 1)) & 3u); \
				const int s = (int)(int8_t)(srow >> (q << 3)); \
				M = M ? M + s : 0; \
				int h = M > ee ? M : ee; \
				h = h > f ? h : f; \
				mj = m > h ? mj : j; \
				m = m > h ? m : h; \
				int tt = M - oe_del; tt = tt > 0 ? tt : 0; \
				ee -= e_del; ee = ee > tt ? ee : tt; \
				eh[(j & (XN_RING - 1)) * 64] = (uint32_t)h1 | (uint32_t)ee << 16;   /* H(i,j-1) for the next row, E(i+1,j) */ \
				h1 = h; \
				tt = M - oe_ins; tt = tt > 0 ? tt : 0; \
				f -= e_ins; f = f > tt ? f : tt; \
				++j; ++pf_cells; \
				if (j >= end) { jtop = end; state = XN_ROWEND; } \
			} } while (0)
		if (state == XN_CELL) {
			XN_ONE_CELL();
			if (state == XN_CELL) XN_ONE_CELL();   // two cells a trip: a row of a dozen cells costs as many trips as its two ends
		} else if (state == XN_SHL) { // the band for the next row: the non-zero cells (ksw.c:466-469), from the left ...
			// (after a truncated row the entries from jtop on are zero and are not read: their slots may belong to lower columns)
			if (j < end && ((trunc && j >= jtop) || eh[(j & (XN_RING - 1)) * 64] == 0)) j = (trunc && j >= jtop) ? end : j + 1;
			else { beg = j; j = trunc ? jtop - 1 : end; state = XN_SHR; }
		} else if (state == XN_SHR) { // ... and from the right
			if (j >= beg && eh[(j & (XN_RING - 1)) * 64] == 0) --j;
			else { end = j + 2 < qlen ? j + 2 : qlen; zhi = j + 1; ++i; state = XN_ROW; }
		}
		// ---- the rest (end of a row, start of a row, next job) on the trips the mask names, or when a third of the wave waits for it:
		// a wave pays for every state one of its lanes is in, and with 64 lanes some lane always is in each of these
		const bool cold = state == XN_ROWEND || state == XN_ROW || state == XN_FETCH;
		const unsigned long long cm = __ballot(cold);
		if (cm && ((pf_trips & cold_mask) == 0 || __popcll(cm) > 20 || cm == __ballot(state != XN_DONE))) {
		// ---- the end of a row
		if (state == XN_ROWEND) {
			const int jfin = beg < end ? end : beg;
			if (!trunc) {
				if (end - beg >= XN_RING) XN_DECLINE();
				else eh[(end & (XN_RING - 1)) * 64] = (uint32_t)h1;   // eh[end] = {h1, 0}
			} else h1 = 0;
			if (state == XN_ROWEND) {
				if (jfin == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
				bool stop = m == 0;
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
				if (stop) { i = tlen; state = XN_ROW; }   // the row loop ends here (ksw.c:454,460-464)
				else { j = beg; state = XN_SHL; }
			}
		}
		// ---- the start of a row, or the end of the job
		if (state == XN_ROW) {
			if (i >= tlen) {
				bsx_ext_res_t r;
				r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
				res[e] = r;
				state = XN_FETCH;
			} else if (i >= 64 || beg < i - w) XN_DECLINE();   // beyond the bases in registers / a clamp that would leave live cells behind
			else {
				const long long tf = tf0 + (long long)i * tfd;
				const int bi = (int)((tf >> 2) - tbyte0);
				const uint32_t tw = bi < 4 ? tp0 : bi < 8 ? tp1 : bi < 12 ? tp2 : bi < 16 ? tp3 : tp4;
				const int t = (int)((tw >> (((bi & 3) << 3) + (int)((~tf & 3) << 1))) & 3u) ^ tcomp;
				srow = s_srow[par][t];
				if (end > i + w + 1) end = i + w + 1;
				if (end > qlen) end = qlen;
				if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
				f = 0; m = 0; mj = -1; j = beg; trunc = 0; jtop = beg;
				++pf_rows;
				state = beg < end ? XN_CELL : XN_ROWEND;
			}
		}
		// ---- next jobs, when a quarter of the wave waits (or nothing else is going on)
		{
			const unsigned long long fm = __ballot(state == XN_FETCH);
			if (fm && (__popcll(fm) >= 16 || fm == __ballot(state != XN_DONE))) {
				const int first = (int)__builtin_ctzll(fm);
				unsigned int base = 0;
				if (lane == first) base = atomicAdd(cursor, (unsigned int)__popcll(fm));
				base = (unsigned int)__builtin_amdgcn_readlane((int)base, first);
				if (state == XN_FETCH) {
					e = base + (unsigned int)__popcll(fm & ((1ull << lane) - 1));
					if (e >= n) state = XN_DONE;
					else {
						const bsx_ext_job_t J = jobs[e];
						++pf_jobs;
						par = J.parent ? 1 : 0;
						qlen = J.qlen; tlen = J.tlen; h0 = J.h0;
						const int mx = par ? sc.mx_ct : sc.mx_ga;
						// the first row's non-zero entries: 0 .. jmax0 (ksw.c:395-397)
						int jmax0 = h0 > oe_ins ? (h0 - oe_ins - 1) / e_ins + 1 : 0;
						jmax0 = jmax0 < qlen ? jmax0 : qlen;
						const bool fits = qlen >= 1 && tlen >= 1 && h0 >= 1 && jmax0 <= XN_RING - 6 && (long long)h0 + 64ll * mx < 32768;
						// the query's first 64 bases, 2 bits each; an ambiguous base among them sends the job to the wide kernel
						uint32_t amb = 0;
						if (fits) {
							const uint8_t *qp = reads + J.qoff;
							const int nq = qlen < 64 ? qlen : 64;
							uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
							for (int g = 0; g < 16; ++g) {
								uint32_t x = 0;
								if (g * 4 + 4 <= nq) {
									if (J.qdir > 0) { uint32_t v; __builtin_memcpy(&v, qp + g * 4, 4); x = v; }
									else { uint32_t v; __builtin_memcpy(&v, qp - g * 4 - 3, 4); x = __builtin_bswap32(v); }
								} else if (g * 4 < nq) { // the last bases one by one: nothing outside the query is touched
									for (int k = 0; k < 3; ++k) if (g * 4 + k < nq) x |= (uint32_t)qp[(long long)(g * 4 + k) * J.qdir] << (k << 3);
								}
								amb |= x & 0xfcfcfcfcu;
								const uint32_t p4 = (x & 3u) | ((x >> 6) & 0xcu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xc0u);
								pk[g >> 2] |= p4 << ((g & 3) << 3);
							}
							q2a = pk[0]; q2b = pk[1]; q2c = pk[2]; q2d = pk[3];
						}
						if (!fits || amb) XN_DECLINE();
						else {
							// the reference bases of rows 0..63: forward coordinates tf0 + r * tfd, complemented on the reverse strand
							const long long p0 = J.tpos;
							if (p0 >= ix.l_pac) { tf0 = (ix.l_pac << 1) - 1 - p0; tfd = -J.tdir; tcomp = 3; } else { tf0 = p0; tfd = J.tdir; tcomp = 0; }
							const int nr = tlen < 64 ? tlen : 64;
							const long long flo = tfd > 0 ? tf0 : tf0 - (nr - 1);
							tbyte0 = (flo >> 2) & ~3ll;
							const uint32_t *pp = (const uint32_t*)(ix.pac + tbyte0);   // pac is padded past its end
							tp0 = pp[0]; tp1 = pp[1]; tp2 = pp[2]; tp3 = pp[3]; tp4 = pp[4];
							w = J.w;
							{ // band clamp (ksw.c:399-407)
								int max_ins = (int)((double)(qlen * mx + J.end_bonus - sc.o_ins) / e_ins + 1.);
								max_ins = max_ins > 1 ? max_ins : 1;
								w = w < max_ins ? w : max_ins;
								int max_del = (int)((double)(qlen * mx + J.end_bonus - o_del) / e_del + 1.);
								max_del = max_del > 1 ? max_del : 1;
								w = w < max_del ? w : max_del;
							}
#pragma unroll
							for (int c = 0; c < XN_RING; ++c) { // first row (ksw.c:395-397); the ring holds columns 0..31 now
								const int v = c == 0 ? h0 : h0 - oe_ins - (c - 1) * e_ins;
								eh[c * 64] = (c <= qlen && v > 0) ? (uint32_t)v : 0u;
							}
							max = h0; max_i = max_j = max_ie = -1; gscore = -1; max_off = 0;
							beg = 0; end = qlen; zhi = jmax0 + 1; i = 0;
							state = XN_ROW;
						}
					}
				}
			}
		}
		}
		if (__ballot(state != XN_DONE) == 0) break;
	}
	if (prof) {
		const unsigned int c = (unsigned int)wave_sum_i32((int)pf_cells), r = (unsigned int)wave_sum_i32((int)pf_rows), jb = (unsigned int)wave_sum_i32((int)pf_jobs), wd = (unsigned int)wave_sum_i32((int)pf_wide);
		if (lane == 0) { atomicAdd(&prof[4], (unsigned long long)jb); atomicAdd(&prof[5], (unsigned long long)r); atomicAdd(&prof[6], (unsigned long long)c); atomicAdd(&prof[7], (unsigned long long)wd); atomicAdd(&prof[8], (unsigned long long)pf_trips); }
	}
}

int ext_q_max_query(int ncq) { return 16 * ncq - 1; }

// jobs[0 .. n) -> res; n is *n_ptr if n_ptr is given (a device counter), else n_upper; *cursor must be zero.  With `list`, the jobs are
// jobs[list[0 .. n)].
void launch_ext_q(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                  const unsigned int *n_ptr, unsigned int n_upper, unsigned int *cursor, int max_qlen, const int *list, unsigned long long *prof)
{
	// a workgroup = 4 waves = 16 jobs in flight; persistent rows (they take jobs until the queue is empty)
	const long long want = ((long long)n_upper + 15) / 16;
	const int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)n_cu * 8));
	if (max_qlen <= ext_q_max_query(10))
		hipLaunchKernelGGL(k_ext_q<10>, dim3(grid), dim3(256), 0, st, ix, sc, reads, jobs, res, n_ptr, n_upper, cursor, list, prof);
	else
		hipLaunchKernelGGL(k_ext_q<16>, dim3(grid), dim3(256), 0, st, ix, sc, reads, jobs, res, n_ptr, n_upper, cursor, list, prof);
}
// the narrow jobs of jobs[0 .. n) by k_ext_n (a lane per job); the others are listed in wide_list[0 .. *wide_count) for launch_ext_q
void launch_ext_n(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                  const unsigned int *n_ptr, unsigned int n_upper, unsigned int *cursor, int *wide_list, unsigned int *wide_count, unsigned long long *prof)
{
	// a workgroup = one wave = 64 jobs in flight and 8 KB of LDS; persistent lanes
	const long long want = ((long long)n_upper + 63) / 64;
	static const int wpc = getenv("BSX_EXTN_WPC") ? atoi(getenv("BSX_EXTN_WPC")) : 19;     // workgroups per CU: 19 rings of 8 KB fit its LDS
	static const unsigned int cold_mask = getenv("BSX_EXTN_COLD") ? (unsigned int)atoi(getenv("BSX_EXTN_COLD")) : 1u;   // cold states every (mask + 1)-th trip
	const int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)n_cu * wpc));
	hipLaunchKernelGGL(k_ext_n, dim3(grid), dim3(64), 0, st, ix, sc, reads, jobs, res, n_ptr, n_upper, cursor, wide_list, wide_count, prof, cold_mask);
}
