// k_extl.hip -- K4 a LANE per job, for the extensions that start from a short seed: ksw_extend2 (lib/aln/ksw.c:380-479) and around it the
// two-sided, two-band-width extension of a seed (memchain.c:613-730), as in k_ext4.hip.
//
// Why.  Six extensions in seven (406 M of 472 M rows per chunk of 1 M reads against an hg38-sized genome) start from a chance match of a
// 3-letter 19-mer: h0 = 19..31, the non-zero cells of a row span 2 (h0 - o - e) ~ 25-45 columns that move along the diagonal, and the scores
// are gone after ~30 rows.  In rows of 16 lanes (k_ext4) such a band straddles three or four 16-column slots, half of their cells outside
// it, and a row costs ~100 vector instructions whatever is in it.  Here a lane owns the whole extension: the eh[] array of the reference
// is a WINDOW OF 64 COLUMNS IN 64 REGISTERS (H and E 16 bits each), column base + p at register p, and the reference's own column loop runs
// over the window unrolled -- every register index is a constant, F is the plain sequential recurrence, nothing crosses lanes, and 64
// extensions in different rows of different bands advance with every trip (~30 instructions per window position, i.e. ~45 per row).  The
// window follows the band 16 columns at a time (registers shifted by selects); what lies right of it is provably zero (the first row's
// non-zero entries end before column 64, and a row whose F or H leaves the window alive sends the job to the wide kernel).
// Jobs this kernel cannot hold -- a band wider than the window, an ambiguous base in the query, the band beyond query column 128 -- are
// appended, untouched, to the wide queue that k_ext4 runs afterwards.  Results are bit-identical to ext_dp.hpp / k_ext4 (same tests).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"
#include "rgx.hpp"
#include "x4.hpp"

#define XL_W 64          // window positions (registers)
#define XL_QCOLS 128     // query columns whose bases a lane holds (2 bits each, 8 registers)
#ifndef XL_OCC
#define XL_OCC 2
#endif
enum { XL_IDLE = 0, XL_NEXT, XL_ROW, XL_AFTER, XL_REFILL, XL_BAIL, XL_DONE };

__device__ __forceinline__ int xl_max3(int a, int b, int c) { const int m = a > b ? a : b; return m > c ? m : c; }

// CHAIN: the narrow queue of launch_x4's pool; else plain ksw_extend2 jobs (bsx_ext_job_t -> bsx_ext_res_t at res_, n = jcap of them, for the
// kernel's own tests): a job this kernel cannot hold leaves its result slot as it was and counts in ctr32[0]
template <bool CHAIN>
__global__ void __launch_bounds__(64, XL_OCC)
k_extl(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, void *jobs_, void *res_, unsigned int jcap, unsigned int *ctr32, unsigned char *xbase, unsigned long long *prof)
{
	X4Job *jobs = (X4Job*)jobs_;
	// scores of target base t against query bases 0..3, a byte each: [parent][t]
	__shared__ uint32_t s_srow[2][4];
	if (threadIdx.x < 8) {
		const int p = threadIdx.x >> 2, t = threadIdx.x & 3;
		const int8_t *mat = p ? sc.ctmat : sc.gamat;
		s_srow[p][t] = (uint32_t)(uint8_t)mat[t * 5] | (uint32_t)(uint8_t)mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)mat[t * 5 + 2] << 16 | (uint32_t)(uint8_t)mat[t * 5 + 3] << 24;
	}
	__syncthreads();
	const int lane = (int)threadIdx.x;
	const unsigned int half = jcap / 2;
	const unsigned int n = !CHAIN ? jcap : ctr32[2] < half ? ctr32[2] : half;   // the narrow queue: jobs[half .. half + n)
	unsigned int e = 0;
	unsigned int bq = 0; int bqdir = 1, btdir = 1; long long btpos = 0;   // (plain jobs)
	const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
	const long long l_pac = ix.l_pac;
	int st = XL_IDLE, wait = 0;
	unsigned int wnext = 0, wend = 0; bool wdone = false;   // the wave's share of the queue (uniform)
	// the job
	long long s_rbeg = 0, rmax0 = 0, rmax1 = 0; unsigned long long ext_at = 0;
	unsigned int qoff0 = 0; int l_query = 0, s_qbeg = 0, s_len = 0, par = 0, si = 0;
	int side = 0, attempt = 0, prev = 0, sc0 = 0, aw = 0, clip = 0;
	int R_qb = 0, R_qe = 0, R_score = 0, R_truesc = 0, aw0 = 0, aw1 = 0; long long R_rb = 0, R_re = 0;
	// the extension under way
	int qlen = 0, tlen = 0, h0 = 0, w = 0, i = 0, beg = 0, end = 0, base = 0;
	long long tF = 0; int tfd = 1, tcomp = 0, yleft = 0;
	uint32_t y0 = 0, y1 = 0, y2 = 0;
	int max = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	uint32_t R[XL_W];                 // eh[base + p]: h | e << 16
	uint32_t QF[XL_QCOLS / 16];       // query bases of columns 0..127, 2 bits each
	uint32_t QW[XL_W / 16];           // those of the window's columns
	uint32_t M0 = 0, M1 = 0, M2 = 0, M3 = 0;   // the strand's score rows by target base
#pragma unroll
	for (int p = 0; p < XL_W; ++p) R[p] = 0;
#pragma unroll
	for (int g = 0; g < XL_QCOLS / 16; ++g) QF[g] = 0;
#pragma unroll
	for (int g = 0; g < XL_W / 16; ++g) QW[g] = 0;
	unsigned int pf_rows = 0, pf_trips = 0, pf_jobs = 0, pf_cold = 0, pf_bail = 0;
#define XL_QWIN() do { const int b16_ = base >> 4; _Pragma("unroll") for (int g_ = 0; g_ < XL_W / 16; ++g_) { uint32_t v_ = 0; \
		_Pragma("unroll") for (int k_ = 0; k_ < XL_QCOLS / 16; ++k_) v_ = b16_ + g_ == k_ ? QF[k_] : v_; QW[g_] = v_; } } while (0)
	for (;;) {
		// ---- between extensions (as in k_ext4): results, the next side or band, the next job; when a quarter of the lanes wait for it,
		// when a lane has waited for a few trips, or when no extension is under way
		{
			const bool cold = st != XL_ROW && st != XL_DONE;
			const unsigned long long cm = __ballot(cold), rm = __ballot(st == XL_ROW);
			if (cold) ++wait;
			if (cm && (rm == 0 || __popcll(cm) >= 16 || __ballot(wait >= 4))) {
				++pf_cold;
				if (st == XL_REFILL) { x4_bases(ix.pac, tF + (long long)i * tfd, tfd, tlen - i, y0, y1, y2); yleft = 48; st = XL_ROW; }
				if (st == XL_AFTER) { // an extension is over: the band loop (memchain.c:640-667,698-725), then the side's result
					const int r_score = max, r_qle = max_j + 1, r_tle = max_i + 1, r_gtle = max_ie + 1, r_gscore = gscore, r_off = max_off;
					R_score = r_score;
					if (!CHAIN) {
						bsx_ext_res_t r; r.score = r_score; r.qle = r_qle; r.tle = r_tle; r.gtle = r_gtle; r.gscore = r_gscore; r.max_off = r_off;
						((bsx_ext_res_t*)res_)[e] = r;
						st = XL_IDLE;
					} else
					if (attempt == 0 && !(R_score == prev || r_off < (aw >> 1) + (aw >> 2))) { attempt = 1; st = XL_NEXT; }
					else {
						const int local = r_gscore <= 0 || r_gscore <= R_score - clip;
						if (side == 0) {
							aw0 = aw;
							if (local) { R_qb = s_qbeg - r_qle; R_rb = s_rbeg - r_tle; R_truesc = R_score; }
							else { R_qb = 0; R_rb = s_rbeg - r_gtle; R_truesc = r_gscore; }
						} else {
							aw1 = aw;
							if (local) { R_qe = s_qbeg + s_len + r_qle; R_re = s_rbeg + s_len + r_tle; R_truesc += R_score - sc0; }
							else { R_qe = l_query; R_re = s_rbeg + s_len + r_gtle; R_truesc += r_gscore - sc0; }
						}
						++side; attempt = 0; st = XL_NEXT;
					}
				}
				// a job's next step: sides that need no extension (memchain.c:617-623,674-678), the end of the job
				auto xl_step = [&]() {
					if (CHAIN && st == XL_NEXT && attempt == 0) {
						if (side == 0 && s_qbeg == 0) { R_score = R_truesc = s_len * P.a; R_qb = 0; R_rb = s_rbeg; side = 1; }
						if (side == 1 && s_qbeg + s_len == l_query) { R_qe = l_query; R_re = s_rbeg + s_len; side = 2; }
						if (side >= 2) {
							RgXExt xe; xe.rb = R_rb; xe.re = R_re; xe.qb = R_qb; xe.qe = R_qe; xe.score = R_score; xe.truesc = R_truesc;
							xe.aw0 = aw0; xe.aw1 = aw1; xe.si = si; xe.status = 1;
							*(RgXExt*)(xbase + ext_at) = xe;
							st = XL_IDLE;
						}
					}
				};
				xl_step();
				{ // jobs this kernel cannot hold go to the wide queue as they came (k_ext4 runs it after this launch)
					const unsigned long long bm = __ballot(st == XL_BAIL);
					if (bm) {
						unsigned int b0 = 0;
						if (lane == 0) b0 = atomicAdd(&ctr32[0], (unsigned int)__popcll(bm));
						b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
						if (st == XL_BAIL) {
							const unsigned int at = b0 + (unsigned int)__popcll(bm & ((1ull << lane) - 1));
							if (CHAIN && at < half) { // (no room: the chain keeps status 0 and k_c2r extends it inline)
								X4Job J; J.s_rbeg = s_rbeg; J.rmax0 = rmax0; J.rmax1 = rmax1; J.ext_at = ext_at; J.qoff = qoff0; J.l_query = (short)l_query;
								J.s_qbeg = (short)s_qbeg; J.s_len = (short)s_len; J.parent = (unsigned char)par; J.pad = 0; J.si = si;
								jobs[at] = J;
							}
							++pf_bail;
							st = XL_IDLE;
						}
					}
				}
				const unsigned long long need = __ballot(st == XL_IDLE);
				if (need) { // idle lanes take the next jobs of the wave's share of the queue (refilled 64 at a time)
					if (wnext == wend) {
						unsigned int b0 = 0;
						if (lane == 0) b0 = atomicAdd(&ctr32[3], 64u);
						b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
						wnext = b0 < n ? b0 : n; wend = b0 + 64u < n ? b0 + 64u : n;
						if (wnext == wend) wdone = true;
					}
					const unsigned int e_new = wnext + (unsigned int)__popcll(need & ((1ull << lane) - 1));
					const bool served = e_new < wend;
					wnext = wnext + (unsigned int)__popcll(need) < wend ? wnext + (unsigned int)__popcll(need) : wend;
					if (st == XL_IDLE) {
						if (!served) { if (wdone) st = XL_DONE; }
						else if (!CHAIN) {
							e = e_new;
							const bsx_ext_job_t J = ((const bsx_ext_job_t*)jobs_)[e];
							par = J.parent ? 1 : 0;
							bq = J.qoff; bqdir = J.qdir; qlen = J.qlen; btpos = J.tpos; btdir = J.tdir; tlen = J.tlen; h0 = J.h0; aw = J.w; clip = J.end_bonus;
							++pf_jobs;
							st = XL_NEXT;
						} else {
							const X4Job J = jobs[half + e_new];
							s_rbeg = J.s_rbeg; rmax0 = J.rmax0; rmax1 = J.rmax1; ext_at = J.ext_at; qoff0 = J.qoff;
							l_query = J.l_query; s_qbeg = J.s_qbeg; s_len = J.s_len; par = J.parent; si = J.si;
							side = 0; attempt = 0; aw0 = aw1 = P.w; R_score = R_truesc = -1; R_qb = R_qe = 0; R_rb = R_re = 0;
							++pf_jobs;
							st = XL_NEXT;
						}
					}
				}
				xl_step();
				if (st == XL_NEXT) { // the next extension of the job
					const int qe = s_qbeg + s_len;
					unsigned int jq; int jqdir, jtdir; long long jtpos;
					if (!CHAIN) { jq = bq; jqdir = bqdir; jtdir = btdir; jtpos = btpos; }
					else {
					prev = R_score;
					if (attempt == 0) sc0 = R_score;
					aw = P.w << attempt;
					clip = side ? P.pen_clip3 : P.pen_clip5;
					if (side == 0) { jq = qoff0 + (unsigned int)s_qbeg - 1; jqdir = -1; qlen = s_qbeg; jtpos = s_rbeg - 1; jtdir = -1; tlen = (int)(s_rbeg - rmax0); h0 = s_len * P.a; }
					else { jq = qoff0 + (unsigned int)qe; jqdir = 1; qlen = l_query - qe; jtpos = s_rbeg + s_len; jtdir = 1; tlen = (int)(rmax1 - (s_rbeg + s_len)); h0 = sc0; }
					}
					const int mx = par ? sc.mx_ct : sc.mx_ga;
					// the first row's non-zero entries: 0 .. jmax0 (ksw.c:395-397); they have to end inside the first window
					int jmax0 = h0 > oe_ins ? (h0 - oe_ins - 1) / e_ins + 1 : 0;
					jmax0 = jmax0 < qlen ? jmax0 : qlen;
					if (qlen < 0 || h0 < 0 || jmax0 + 2 >= XL_W || (long long)h0 + (long long)qlen * mx >= 32768) st = XL_BAIL;
					else {
						max = h0; max_i = max_j = max_ie = -1; gscore = -1; max_off = 0;
						beg = 0; end = qlen; i = 0; base = 0;
						if (tlen <= 0) st = XL_AFTER;   // no rows: what ksw_extend2 returns without entering its loop
						else {
							// the query's first 128 bases, 2 bits each; an ambiguous one among them sends the job to the wide kernel
							const uint8_t *qp = reads + jq;
							const int nq = qlen < XL_QCOLS ? qlen : XL_QCOLS;
							uint32_t amb = 0;
#pragma unroll
							for (int k = 0; k < XL_QCOLS / 16; ++k) QF[k] = 0;
#pragma unroll
							for (int g = 0; g < XL_QCOLS / 4; ++g) {
								uint32_t x = 0;
								if (g * 4 + 4 <= nq) {
									if (jqdir > 0) { uint32_t v; __builtin_memcpy(&v, qp + g * 4, 4); x = v; }
									else { uint32_t v; __builtin_memcpy(&v, qp - g * 4 - 3, 4); x = __builtin_bswap32(v); }
								} else if (g * 4 < nq) { // the last bases one by one: nothing outside the query is touched
									for (int k = 0; k < 3; ++k) if (g * 4 + k < nq) x |= (uint32_t)qp[(long long)(g * 4 + k) * jqdir] << (k << 3);
								}
								amb |= x & 0xfcfcfcfcu;
								const uint32_t p4 = (x & 3u) | ((x >> 6) & 0xcu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xc0u);
								QF[g >> 2] |= p4 << ((g & 3) << 3);
							}
							if (amb) st = XL_BAIL;
							else {
								XL_QWIN();
								M0 = s_srow[par][0]; M1 = s_srow[par][1]; M2 = s_srow[par][2]; M3 = s_srow[par][3];
#pragma unroll
								for (int p = 0; p < XL_W; ++p) { // first row (ksw.c:395-397)
									const int v = p == 0 ? h0 : h0 - oe_ins - (p - 1) * e_ins;
									R[p] = (p <= qlen && v > 0) ? (uint32_t)v : 0u;
								}
								w = aw;
								{ // band clamp (ksw.c:399-407)
									int max_ins = (int)((double)(qlen * mx + clip - o_ins) / e_ins + 1.);
									max_ins = max_ins > 1 ? max_ins : 1;
									w = w < max_ins ? w : max_ins;
									int max_del = (int)((double)(qlen * mx + clip - o_del) / e_del + 1.);
									max_del = max_del > 1 ? max_del : 1;
									w = w < max_del ? w : max_del;
								}
								if (jtpos >= l_pac) { tF = (l_pac << 1) - 1 - jtpos; tfd = -jtdir; tcomp = 3; } else { tF = jtpos; tfd = jtdir; tcomp = 0; }
								x4_bases(ix.pac, tF, tfd, tlen, y0, y1, y2);
								yleft = 48;
								st = XL_ROW;
							}
						}
					}
				}
				if (st == XL_ROW || st == XL_DONE) wait = 0;
			}
			if (rm == 0 && __ballot(st == XL_ROW) == 0) { if (__ballot(st != XL_DONE) == 0) break; continue; }
		}
		++pf_trips;
		// ---- one row of every lane's extension (ksw.c:410-470); lanes without one compute on stale registers and throw the result away
		const bool run = st == XL_ROW;
		if (run) ++pf_rows;
		const int t = (int)(y0 & 3u) ^ tcomp;
		y0 = __builtin_amdgcn_alignbit(y1, y0, 2); y1 = __builtin_amdgcn_alignbit(y2, y1, 2); y2 >>= 2; --yleft;
		beg = beg > i - w ? beg : i - w;
		end = end < i + w + 1 ? end : i + w + 1; end = end < qlen ? end : qlen;
		int h1 = h0 - (o_del + e_del * (i + 1)); h1 = (beg == 0 && h1 > 0) ? h1 : 0;
		int f = 0, key = -1, first = XL_W, last = -1;
		const uint32_t srow = t == 0 ? M0 : t == 1 ? M1 : t == 2 ? M2 : M3;
		// window positions of the band [rb, re) and of column `end`; a lane that is not in a row (waiting for its next reference bases, say)
		// gets an empty band nowhere: its registers stay as they are
		const int rb = run ? beg - base : XL_W + 1, re_e = run ? end - base : -1, re = re_e < XL_W ? re_e : XL_W;
#pragma unroll
		for (int p = 0; p < XL_W; ++p) {
			const uint32_t x = R[p];
			const bool c_act = p >= rb && p < re;
			const bool c_end = p == re_e;
			const int Hd = (int)(x & 0xffffu), e = (int)(x >> 16);
			const int q2 = (int)((QW[p >> 4] >> ((p & 15) << 1)) & 3u);
			const int s = (int)(int8_t)(srow >> (q2 << 3));
			const int M = Hd ? Hd + s : 0;
			const int h = xl_max3(M, e, f);
			const int e2 = xl_max3(e - e_del, M - oe_del, 0);
			const int f2 = xl_max3(f - e_ins, M - oe_ins, 0);
			const uint32_t nr = (uint32_t)h1 | (c_act ? (uint32_t)e2 << 16 : 0u);   // eh[j] = {h(i, j-1), e(i+1, j)}; eh[end] = {h1, 0}
			const uint32_t xn = (c_act || c_end) ? nr : x;
			R[p] = xn;
			{ const int k = h << 8 | p; key = (c_act && k > key) ? k : key; }       // the row maximum and the last column that attains it
			h1 = c_act ? h : h1;
			f = c_act ? f2 : f;
			const bool nz = (c_act || c_end) && xn != 0u;                           // the non-zero cells the next row's band is made of (ksw.c:466-469)
			first = (nz && first == XL_W) ? p : first;
			last = nz ? p : last;
		}
		// a row that leaves the window alive (the band is wider than 64 columns): not for this kernel
		const bool wide = run && re_e >= XL_W && (h1 | f) != 0;
		const int m = key < 0 ? 0 : key >> 8, mj = key < 0 ? -1 : base + (key & 255);
		bool stop = false;
		if (run) {
			const int jfin = beg < end ? end : beg;
			if (jfin == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
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
			if (!stop) { // the band of the next row: the non-zero cells
				int nb = first < XL_W ? base + first : end;
				int lastc = last >= 0 ? base + last : nb - 1;
				lastc = lastc > nb - 1 ? lastc : nb - 1;
				beg = nb;
				end = lastc + 2 < qlen ? lastc + 2 : qlen;
				++i;
				if (i >= tlen) stop = true;
				else if (yleft == 0) st = XL_REFILL;
			}
			if (stop) st = XL_AFTER;
			if (wide) st = XL_BAIL;
		}
		// the window follows the band, 16 columns at a time
		while (__ballot((st == XL_ROW || st == XL_REFILL) && beg - base >= 16)) {
			const bool sh = (st == XL_ROW || st == XL_REFILL) && beg - base >= 16;
			if (sh && base + 16 + XL_W > XL_QCOLS) st = XL_BAIL;   // beyond the query bases a lane holds
			const bool go = sh && st != XL_BAIL;
#pragma unroll
			for (int p = 0; p < XL_W; ++p) { const uint32_t nx = p + 16 < XL_W ? R[p + 16 < XL_W ? p + 16 : p] : 0u; R[p] = go ? nx : R[p]; }
			base += go ? 16 : 0;
			if (go) XL_QWIN();
			if (sh && !go) base = beg;   // (leaves the loop: the job is on its way out)
		}
	}
	if (prof) {
		const unsigned int r = (unsigned int)wave_sum_i32((int)pf_rows), jb = (unsigned int)wave_sum_i32((int)pf_jobs), bl = (unsigned int)wave_sum_i32((int)pf_bail);
		if (lane == 0) { atomicAdd(&prof[6], (unsigned long long)jb); atomicAdd(&prof[7], (unsigned long long)r); atomicAdd(&prof[8], (unsigned long long)pf_trips); atomicAdd(&prof[9], (unsigned long long)pf_cold); atomicAdd(&prof[10], (unsigned long long)bl); }
	}
}

// the narrow queue of launch_x4's job pool (jobs[jcap / 2 ..], count ctr32[2], cursor ctr32[3]) a lane per job; what it cannot hold joins
// the wide queue (jobs[0 ..], count ctr32[0]) for k_ext4
void launch_extl(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, void *jobs, unsigned int jcap,
                 unsigned int *ctr32, unsigned char *xbase, long long n_upper, unsigned long long *prof)
{
	const int wpc = 4 * XL_OCC;
	const int grid = (int)std::max<long long>(1, std::min<long long>((n_upper + 63) / 64, (long long)n_cu * wpc));
	hipLaunchKernelGGL(k_extl<true>, dim3(grid), dim3(64), 0, st, ix, sc, P, reads, jobs, (void*)nullptr, jcap, ctr32, xbase, prof);
}

// plain jobs[0 .. n) through the same lanes (tests): res[i] is written for every job the kernel holds, ctr32[0] counts the others; ctr32[0..3] zero at launch
void launch_extl_batch(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                       unsigned int n, unsigned int *ctr32)
{
	RegParams P; memset(&P, 0, sizeof(P));
	const int grid = (int)std::max<long long>(1, std::min<long long>(((long long)n + 63) / 64, (long long)n_cu * 4 * XL_OCC));
	hipLaunchKernelGGL(k_extl<false>, dim3(grid), dim3(64), 0, st, ix, sc, P, reads, (void*)jobs, (void*)res, n, ctr32, (unsigned char*)nullptr, (unsigned long long*)nullptr);
}
