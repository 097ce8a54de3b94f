// k_ext4.hip -- K4 four to a wavefront: ksw_extend2 (lib/aln/ksw.c:380-479) with one extension per ROW OF 16 LANES, and around it the
// two-sided, two-band-width extension of a seed (left_extend_seed_set_align_beg / right_extend_seed_set_align_end, memchain.c:613-730).
//
// Why.  mem_chain2region1 (memchain.c:742-870) is a strictly ordered loop per strand search, but what its extensions RETURN does not depend
// on that order: the loop only decides whether a seed is extended at all (the containment tests), and the first seed it reaches in a
// chain's main list -- the best-scored one that passes asymmetric_flt_seed -- is extended unless an earlier chain's region already
// contains it.  Against an hg38-sized genome a strand search keeps a handful of chains, nearly all of them one chance match of a
// 3-letter 19-mer whose extensions die after a few rows inside a band of a dozen columns (16.5 M extensions per 1 M reads, 7 rows each).
// A wavefront per strand search (k_c2r) spent 64 lanes and ~4 k scalar instructions on each of them and was bound by the one scalar
// unit a CU has.  So the extensions are made AHEAD of the loop, every chain's best seed an independent job:
//   k_x4prep   a lane per exported chain: mem_chain_reference_span (memchain.c:585-605) + bns_fetch_seq's clamps, the seed the loop
//              reaches first; a seed that spans its read needs no extension and is answered here, the others become jobs
//   k_ext4     a row of 16 lanes per job, four jobs per wavefront, persistent rows: left extension, right extension, each with its
//              band retry; the DP rows live in registers (entry a of the reference's eh[] in lane a & 15 of slot a >> 4), F by a
//              max-plus prefix scan in DPP row shifts, the reference bases of 48 rows as 2-bit fields in three registers.  Nothing
//              is wave-uniform: band limits, scores and row counters are per-lane values equal across a job's 16 lanes, so four jobs
//              in different rows of different bands advance with every trip, and the work between extensions (set-up, results, the
//              next job) waits until two rows need it or nothing else is going on
// k_c2r then runs the reference's loop and takes a chain's extension from the record (RgXExt) when the seed it reaches is the one
// extended here; everything else (later seeds of a chain, the backup list, queries beyond this kernel's slots) it extends inline.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"
#include "tune.h"
#include "rgx.hpp"
#include "x4.hpp"

size_t x4_job_bytes(void) { return sizeof(X4Job); }

#define X4_GAPCAP 256
#define X4_XSEEDS 128    // = RG_XSEEDS: k_c2r hands longer lists to the next tier
#define X4P_WPB 4
// ---- k_x4prep: the exported strand searches 64 to a wave (a lane reads a header), then the wave's chains a lane each
__global__ void __launch_bounds__(64 * X4P_WPB)
k_x4prep(DevIndex ix, RegParams P, const bsx_seed_task_t *tasks, RgXPool X, X4Job *jobs, unsigned int jcap, unsigned int *jcount)
{
	__shared__ int gap_tab[X4_GAPCAP + 1];
	__shared__ int s_pre[X4P_WPB][64];
	__shared__ unsigned long long s_rec[X4P_WPB][64];
	__shared__ unsigned int s_qoff[X4P_WPB][64];
	__shared__ int s_lq[X4P_WPB][64];
	P.gap_cap = X4_GAPCAP;
	for (int q = threadIdx.x; q <= X4_GAPCAP; q += blockDim.x) gap_tab[q] = rg_cal_max_gap(P, q);
	__syncthreads();
	const int lane = wave_lane(), wv = (int)(threadIdx.x >> 6);
	const unsigned int n = *X.xcount;
	const unsigned int i0 = ((unsigned int)blockIdx.x * X4P_WPB + (unsigned int)wv) * 64u;
	if (i0 >= n) return;
	const long long l_pac = ix.l_pac;
	int nk = 0;
	if (i0 + lane < n) {
		const int t = X.xlist[i0 + lane];
		const unsigned long long at = (unsigned long long)X.xoff[t];
		const RgXHdr *H = (const RgXHdr*)(X.base + at);
		nk = H->has_ext ? H->n_chains : 0;   // (has_ext 2: a slot and a job per SEED of the main lists, below)
		s_rec[wv][lane] = at;
		s_qoff[wv][lane] = tasks[t].qoff;
		s_lq[wv][lane] = tasks[t].len << 1 | (tasks[t].parent & 1);
	}
	const int incl = wave_scan_sum_incl(nk);
	s_pre[wv][lane] = incl;
	const int tot = __builtin_amdgcn_readlane(incl, 63);
	WAVE_SYNC();
	for (int b = 0; b < tot; b += 64) {
		const int j = b + lane;
		// the lane's chain: its window, and which of its seeds get a job -- the one the loop reaches first (has_ext 1), or every seed of the main
		// list that passes asymmetric_flt_seed (has_ext 2); then the jobs, one per lane per round
		int n_it = 0, mode = 0, best = -1, ci = 0, l_query = 0, parent = 0, s = 0;
		long long rmax0 = 0, rmax1 = 0;
		const RgXSeed *XS = nullptr; RgXExt *XE = nullptr; RgXChain c; c.pos = 0; c.rid = 0; c.seed_off = 0; c.n_main = c.n_extra = 0; c.pad = 0;
		RgXExt xe0; xe0.rb = xe0.re = 0; xe0.qb = xe0.qe = 0; xe0.score = xe0.truesc = -1; xe0.aw0 = xe0.aw1 = P.w; xe0.si = -1; xe0.status = 0;
		if (j < tot) {
			int lo = 0, hi = 63;   // first s with s_pre[s] > j
			while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pre[wv][mid] > j) hi = mid; else lo = mid + 1; }
			s = lo;
			const unsigned long long at = s_rec[wv][s];
			const RgXHdr *H = (const RgXHdr*)(X.base + at);
			const int nch = H->n_chains, n_sd = H->n_seeds;
			mode = H->has_ext;
			ci = j - (s_pre[wv][s] - nch);
			l_query = s_lq[wv][s] >> 1; parent = s_lq[wv][s] & 1;
			const RgXChain *XC = (const RgXChain*)(H + 1);
			XS = (const RgXSeed*)(XC + nch);
			XE = (RgXExt*)(XS + n_sd);
			c = XC[ci];
			const int n_main = c.n_main;
			const int cap = mode == 2 ? 1024 : X4_XSEEDS;   // (k_c2r's own limits: lists it hands on are not worth extending for)
			if (n_main > 0 && n_main <= cap && c.n_extra <= cap && l_query <= X4_GAPCAP) {
				// mem_chain_reference_span (memchain.c:585-605) over the main list + bns_fetch_seq's contig clamp, as k_c2r does it;
				// the seed mem_chain2region1's loop reaches first: the largest (score, index) among those that pass asymmetric_flt_seed
				rmax0 = l_pac << 1; rmax1 = 0;
				long long bkey = -1;
				for (int o = 0; o < n_main; ++o) {
					const RgXSeed sd = XS[c.seed_off + o];
					const long long bb = sd.rbeg - (sd.qbeg + rg_gap(gap_tab, P, sd.qbeg));
					const long long ee = sd.rbeg + sd.len + ((l_query - sd.qbeg - sd.len) + rg_gap(gap_tab, P, l_query - sd.qbeg - sd.len));
					rmax0 = rmax0 < bb ? rmax0 : bb; rmax1 = rmax1 > ee ? rmax1 : ee;
					const long long key = (long long)((unsigned long long)(unsigned)XS_SCORE(sd) << 32 | (unsigned)o);
					if (!XS_BAD(sd) && key > bkey) { bkey = key; best = o; }
				}
				if (best >= 0) {
					rmax0 = rmax0 > 0 ? rmax0 : 0; rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
					if (rmax0 < l_pac && l_pac < rmax1) { if (c.pos < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
					{
						const int is_rev = c.pos >= l_pac;
						long long far_beg = ix.ctg_off[c.rid], far_end = ix.ctg_off[c.rid + 1];
						if (is_rev) { const long long tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
						rmax0 = rmax0 > far_beg ? rmax0 : far_beg; rmax1 = rmax1 < far_end ? rmax1 : far_end;
					}
					n_it = mode == 2 ? n_main : 1;
				}
			}
			if (n_it == 0) { // nothing to extend ahead: the slots say so
				if (mode == 2) { for (int o = 0; o < n_main; ++o) XE[c.seed_off + o] = xe0; }
				else XE[ci] = xe0;
			}
		}
		for (int it = 0; __ballot(it < n_it) != 0ull; ++it) {
		bool job = false;
		X4Job J;
		if (it < n_it) {
			const int o = mode == 2 ? it : best;
			const RgXSeed sd = XS[c.seed_off + o];
			RgXExt *slot = mode == 2 ? XE + c.seed_off + o : XE + ci;
			RgXExt xe = xe0;
			if (!XS_BAD(sd)) {
				xe.si = o;
				if (sd.qbeg == 0 && sd.qbeg + sd.len == l_query) { // the seed spans the read: no extension on either side (memchain.c:617,674)
					xe.score = xe.truesc = sd.len * P.a; xe.qb = 0; xe.rb = sd.rbeg; xe.qe = l_query; xe.re = sd.rbeg + sd.len;
					xe.status = 1;
				} else {
					job = true;
					J.s_rbeg = sd.rbeg; J.rmax0 = rmax0; J.rmax1 = rmax1;
					J.ext_at = (unsigned long long)((unsigned char*)slot - X.base);
					J.qoff = s_qoff[wv][s]; J.l_query = (short)l_query; J.s_qbeg = sd.qbeg; J.s_len = sd.len; J.parent = (unsigned char)parent; J.pad = 0;
					J.si = o;
				}
			}
			*slot = xe;
		}
		// Two queues, each in its half of the pool: a seed of fewer than X4_NARROW bases is nearly always a chance match whose extensions
		// stay inside a band of a dozen columns and die within thirty rows; the others (the read's own locus) run for as many rows as the
		// read has bases left, over all of its columns.  A wave pays for the widest band among its four rows, so the two kinds do not mix.
		const bool narrow = job && J.s_len < X4_NARROW;
		for (int q = 0; q < 2; ++q) {
			const unsigned long long jm = __ballot(job && narrow == (q == 1));
			if (jm) {
				unsigned int base = 0;
				if (lane == 0) base = atomicAdd(jcount + 2 * q, (unsigned int)__popcll(jm));
				base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
				const unsigned int at = base + (unsigned int)__popcll(jm & ((1ull << lane) - 1));
				if (job && narrow == (q == 1) && at < jcap / 2) jobs[(size_t)q * (jcap / 2) + at] = J;   // no room: the chain keeps status 0 and k_c2r extends it inline
			}
		}
		}
	}
}

// ---- k_ext4
#ifndef X4_OCC
#define X4_OCC 2     // workgroups of four waves per CU the register allocation targets
#endif
#define DPP_ROW_ROR(n) (0x120 + (n))
#define DPP_ROW_NEWBCAST(n) (0x150 + (n))
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
// maximum over the 16 lanes of a row, in every lane of it (a rotation has a source for every lane: `old` is never taken, and the
// identity there lets the compiler fold each step into one v_max_i32 with a DPP operand)
__device__ __forceinline__ int q_allmax(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_ROR(1), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_ROR(2), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_ROR(4), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(QID, v, DPP_ROW_ROR(8), 0xf, 0xf, false); v = v > t ? v : t;
	return v;
}
// the same for two unsigned 16-bit numbers packed in a word, each on its own (v_pk_max_u16)
typedef unsigned short x4_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t q_allmax_pk(uint32_t v)
{
#define X4_PK_STEP(n) do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_ROR(n), 0xf, 0xf, false); \
		const x4_us2 m_ = __builtin_elementwise_max(__builtin_bit_cast(x4_us2, v), __builtin_bit_cast(x4_us2, t_)); v = __builtin_bit_cast(uint32_t, m_); } while (0)
	X4_PK_STEP(1); X4_PK_STEP(2); X4_PK_STEP(4); X4_PK_STEP(8);
#undef X4_PK_STEP
	return v;
}
// the previous lane's value inside the row; lane 0 of the row gets `first`
__device__ __forceinline__ int q_prev(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, DPP_ROW_SHR(1), 0xf, 0xf, false); }
// lane 0 of the row gets lane 15's value (the others lane l - 1's)
__device__ __forceinline__ int q_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR(1), 0xf, 0xf, false); }
// lane 15's value in every lane of the row
__device__ __forceinline__ int q_last(int v) { return __builtin_amdgcn_update_dpp(v, v, DPP_ROW_NEWBCAST(15), 0xf, 0xf, false); }

enum { X4_IDLE = 0, X4_NEXT, X4_ROW, X4_AFTER, X4_REFILL, X4_DONE };
#define X4_ROWS 48

// CHAIN: jobs are X4Job (a chain's best seed, both sides, band retries; the result goes to the chain's RgXExt); else bsx_ext_job_t ->
// bsx_ext_res_t, one ksw_extend2 call each (the batch form, for the kernel's own tests; score = X4_DECLINED for a job that does not fit)
template <int NCQ, bool CHAIN>
__global__ void __launch_bounds__(256, X4_OCC)
k_ext4(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const void *jobs_, void *res_, unsigned char *xbase,
       const unsigned int *n_ptr, unsigned int n_fixed, unsigned int *cursor, unsigned long long *prof, int take_second)
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
	// CHAIN: two queues, n_ptr[0] jobs from jobs[0] and n_ptr[2] from jobs[n_fixed / 2] (n_fixed = the pool's capacity); taken in that order
	unsigned int n = n_ptr ? *n_ptr : n_fixed, n_first = n;
	if (CHAIN) {
		n_first = n < n_fixed / 2 ? n : n_fixed / 2;
		const unsigned int n_second = take_second ? (n_ptr[2] < n_fixed / 2 ? n_ptr[2] : n_fixed / 2) : 0u;   // (else k_extl has run that queue)
		n = n_first + n_second;
	}
	const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
	const long long l_pac = ix.l_pac;
	int st = X4_IDLE, wait = 0;
	unsigned int e = 0;
	unsigned int wnext = 0, wend = 0; bool wdone = false;   // the wave's share of the queue (uniform)
	// the job
	long long s_rbeg = 0, rmax0 = 0, rmax1 = 0; unsigned long long ext_at = 0;
	unsigned int qoff0 = 0; int l_query = 0, s_qbeg = 0, s_len = 0, par = 0, si = 0;
	int side = 0, attempt = 0, prev = 0, sc0 = 0, aw = 0, clip = 0;
	int R_qb = 0, R_qe = 0, R_score = 0, R_truesc = 0, aw0 = 0, aw1 = 0; long long R_rb = 0, R_re = 0;
	// the extension under way
	unsigned int jq = 0; int jqdir = 1, jtdir = 1, jbonus = 0; long long jtpos = 0;
	int qlen = 0, tlen = 0, h0 = 0, w = 0, i = 0, beg = 0, end = 0;
	long long tF = 0; int tfd = 1, tcomp = 0, yleft = 0;   // forward coordinate of row 0's base, its step per row, complement mask; rows left in y
	uint32_t y0 = 0, y1 = 0, y2 = 0;
	int max = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	int Hr[NCQ], Er[NCQ]; uint32_t sqp[NCQ];
#pragma unroll
	for (int c = 0; c < NCQ; ++c) { Hr[c] = Er[c] = 0; sqp[c] = 0; }
	unsigned int pf_rows = 0, pf_trips = 0, pf_jobs = 0, pf_cold = 0, pf_slots = 0, pf_nrows = 0;
	for (;;) {
		// ---- between extensions: results, the next side or band, the next job.  Run when two rows of lanes wait for it, when a row has
		// waited for a few trips, or when no extension is under way (a wave pays for this block whichever of its rows is in it)
		{
			const bool cold = st != X4_ROW && st != X4_DONE;
			const unsigned long long cm = __ballot(cold), rm = __ballot(st == X4_ROW);
			if (cold) ++wait;
			if (cm && (rm == 0 || __popcll(cm) >= 32 || __ballot(wait >= 3))) {
				++pf_cold;
				if (st == X4_REFILL) { // the next 48 rows' reference bases
					x4_bases(ix.pac, tF + (long long)i * tfd, tfd, tlen - i, y0, y1, y2);
					yleft = X4_ROWS; st = X4_ROW;
				}
				if (st == X4_AFTER) { // an extension is over
					const int r_score = max, r_qle = max_j + 1, r_tle = max_i + 1, r_gtle = max_ie + 1, r_gscore = gscore, r_off = max_off;
					if (!CHAIN) {
						if (l == 0) { bsx_ext_res_t r; r.score = r_score; r.qle = r_qle; r.tle = r_tle; r.gtle = r_gtle; r.gscore = r_gscore; r.max_off = r_off; ((bsx_ext_res_t*)res_)[e] = r; }
						st = X4_IDLE;
					} else {
						// the band loop (memchain.c:640-667,698-725): once more with twice the band when the score changed and the alignment
						// strayed beyond three quarters of it
						R_score = r_score;
						if (attempt == 0 && !(R_score == prev || r_off < (aw >> 1) + (aw >> 2))) { attempt = 1; st = X4_NEXT; }
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
							++side; attempt = 0; st = X4_NEXT;
						}
					}
				}
				// a job's next step: sides that need no extension (memchain.c:617-623,674-678), the end of the job
				auto x4_step = [&]() {
					if (CHAIN && st == X4_NEXT && attempt == 0) {
						if (side == 0 && s_qbeg == 0) { R_score = R_truesc = s_len * P.a; R_qb = 0; R_rb = s_rbeg; side = 1; }
						if (side == 1 && s_qbeg + s_len == l_query) { R_qe = l_query; R_re = s_rbeg + s_len; side = 2; }
						if (side >= 2) {
							if (l == 0) {
								RgXExt xe; xe.rb = R_rb; xe.re = R_re; xe.qb = R_qb; xe.qe = R_qe; xe.score = R_score; xe.truesc = R_truesc;
								xe.aw0 = aw0; xe.aw1 = aw1; xe.si = si; xe.status = 1;
								*(RgXExt*)(xbase + ext_at) = xe;
							}
							st = X4_IDLE;
						}
					}
				};
				x4_step();
				// rows of 16 lanes without a job take the next ones of the wave's share of the queue, which is refilled 64 jobs at a time (one
				// atomic on the queue's cursor per job saturates that word: ~88 per microsecond for the whole chip)
				const unsigned long long need = __ballot(st == X4_IDLE);
				if (need) {
					const unsigned long long heads = need & 0x0001000100010001ull;
					if (wnext == wend) {
						unsigned int base = 0;
						if (lane == 0) base = atomicAdd(cursor, 64u);
						base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
						wnext = base < n ? base : n; wend = base + 64u < n ? base + 64u : n;
						if (wnext == wend) wdone = true;
					}
					const unsigned int e_new = wnext + (unsigned int)__popcll(heads & ((1ull << gsh) - 1));
					const bool served = e_new < wend;
					wnext = wnext + (unsigned int)__popcll(heads) < wend ? wnext + (unsigned int)__popcll(heads) : wend;
					if (st == X4_IDLE) {
					if (!served) { if (wdone) st = X4_DONE; }   // (else: idle until the next pass refills the share)
					else if (CHAIN) {
						e = e_new;
						const X4Job J = ((const X4Job*)jobs_)[e < n_first ? e : n_fixed / 2 + (e - n_first)];
						s_rbeg = J.s_rbeg; rmax0 = J.rmax0; rmax1 = J.rmax1; ext_at = J.ext_at; qoff0 = J.qoff;
						l_query = J.l_query; s_qbeg = J.s_qbeg; s_len = J.s_len; par = J.parent; si = J.si;
						side = 0; attempt = 0; aw0 = aw1 = P.w; R_score = R_truesc = -1; R_qb = R_qe = 0; R_rb = R_re = 0;
						++pf_jobs;
						st = X4_NEXT;
					} else {
						e = e_new;
						const bsx_ext_job_t J = ((const bsx_ext_job_t*)jobs_)[e];
						par = J.parent ? 1 : 0;
						jq = J.qoff; jqdir = J.qdir; qlen = J.qlen; jtpos = J.tpos; jtdir = J.tdir; tlen = J.tlen; h0 = J.h0; aw = J.w; jbonus = J.end_bonus;
						++pf_jobs;
						st = X4_NEXT;
					}
					}
				}
				x4_step();
				if (st == X4_NEXT) { // the next extension of the job
					const bool go = true;
					if (CHAIN) {
						const int qe = s_qbeg + s_len;
						prev = R_score;
						if (attempt == 0) sc0 = R_score;
						aw = P.w << attempt;
						clip = side ? P.pen_clip3 : P.pen_clip5;
						jbonus = clip;
						if (side == 0) { jq = qoff0 + (unsigned int)s_qbeg - 1; jqdir = -1; qlen = s_qbeg; jtpos = s_rbeg - 1; jtdir = -1; tlen = (int)(s_rbeg - rmax0); h0 = s_len * P.a; }
						else { jq = qoff0 + (unsigned int)qe; jqdir = 1; qlen = l_query - qe; jtpos = s_rbeg + s_len; jtdir = 1; tlen = (int)(rmax1 - (s_rbeg + s_len)); h0 = sc0; }
					}
					if (go) {
						const int mx = par ? sc.mx_ct : sc.mx_ga;
						if (qlen + 1 > 16 * NCQ || qlen < 0 || (long long)h0 + (long long)qlen * mx >= (1 << 21)) { // not for this kernel's slots / packed maxima
							if (CHAIN) st = X4_IDLE;   // the chain keeps status 0 (k_x4prep): k_c2r extends it inline
							else { if (l == 0) { bsx_ext_res_t r; r.score = X4_DECLINED; r.qle = r.tle = r.gtle = r.gscore = r.max_off = 0; ((bsx_ext_res_t*)res_)[e] = r; } st = X4_IDLE; }
						} else {
							max = h0; max_i = max_j = max_ie = -1; gscore = -1; max_off = 0;
							beg = 0; end = qlen; i = 0;
							if (tlen <= 0) st = X4_AFTER;   // no rows: what ksw_extend2 returns without entering its loop
							else {
#pragma unroll
								for (int c = 0; c < NCQ; ++c) {
									const int a = (c << 4) + l;
									int q = a < qlen ? (int)reads[(long long)jq + (long long)a * jqdir] : 4;
									q = q < 4 ? q : 4;
									sqp[c] = s_sqp[par][q];
									const int v = a == 0 ? h0 : h0 - oe_ins - (a - 1) * e_ins;   // first row (ksw.c:395-397)
									Hr[c] = (a <= qlen && v > 0) ? v : 0;
									Er[c] = 0;
								}
								w = aw;
								{ // band clamp (ksw.c:399-407)
									int max_ins = (int)((double)(qlen * mx + jbonus - o_ins) / e_ins + 1.);
									max_ins = max_ins > 1 ? max_ins : 1;
									w = w < max_ins ? w : max_ins;
									int max_del = (int)((double)(qlen * mx + jbonus - o_del) / e_del + 1.);
									max_del = max_del > 1 ? max_del : 1;
									w = w < max_del ? w : max_del;
								}
								if (jtpos >= l_pac) { tF = (l_pac << 1) - 1 - jtpos; tfd = -jtdir; tcomp = 3; } else { tF = jtpos; tfd = jtdir; tcomp = 0; }
								x4_bases(ix.pac, tF, tfd, tlen, y0, y1, y2);
								yleft = X4_ROWS;
								st = X4_ROW;
							}
						}
					}
				}
				if (st == X4_ROW || st == X4_DONE) wait = 0;
			}
			if (rm == 0 && __ballot(st == X4_ROW) == 0) { if (__ballot(st != X4_DONE) == 0) break; continue; }
		}
		++pf_trips;
		// ---- one row of every extension under way.  Straight-line code: what a row of lanes without an extension computes is thrown
		// away by selects (a branch per 16 lanes costs the wave more than the instructions it skips)
		const bool run = st == X4_ROW;
		if (run) { ++pf_rows; if (CHAIN && s_len < X4_NARROW) ++pf_nrows; }
		// (what the lanes of a row without an extension do to these is of no consequence: the next set-up writes them all)
		const int t = (int)(y0 & 3u) ^ tcomp;
		y0 = __builtin_amdgcn_alignbit(y1, y0, 2); y1 = __builtin_amdgcn_alignbit(y2, y1, 2); y2 >>= 2; --yleft;
		beg = beg > i - w ? beg : i - w;
		end = end < i + w + 1 ? end : i + w + 1; end = end < qlen ? end : qlen;
		int h1_init = h0 - (o_del + e_del * (i + 1)); h1_init = (beg == 0 && h1_init > 0) ? h1_init : 0;
		int m = 0, mj = -1, h1_last = h1_init;
		const bool nonempty = run && beg < end;
		const int c_lo = beg >> 4, c_hi = end >> 4;   // slots holding entries beg .. end (entry `end` gets its E cleared and its H set)
		// the slots some row's band covers, as a scalar mask
		const unsigned int sm_l = nonempty ? (2u << c_hi) - (1u << c_lo) : 0u;
		const unsigned int smask = (unsigned int)__builtin_amdgcn_readlane((int)sm_l, 0) | (unsigned int)__builtin_amdgcn_readlane((int)sm_l, 16) |
		                           (unsigned int)__builtin_amdgcn_readlane((int)sm_l, 32) | (unsigned int)__builtin_amdgcn_readlane((int)sm_l, 48);
		uint32_t zpk = 0;   // the non-zero cells of the row: (0xffff - first) << 16 | last + 1, each the maximum over the columns
		pf_slots += (unsigned int)__builtin_popcount(smask);
		if (smask) {
			int carry = NEG_BIG, kmax = -1, hprev = 0, vlast = 0;
			const int tsh = t << 3;
#pragma unroll
			for (int c = 0; c < NCQ; ++c) {
				if (smask >> c & 1u) {
					const int a = (c << 4) + l;
					const bool in = nonempty && c >= c_lo && c <= c_hi;
					const bool act = in && a >= beg && a < end;
					const int hr = Hr[c], er = Er[c];
					const int s = (int)(int8_t)(sqp[c] >> tsh);
					const int M = (act && hr) ? hr + s : 0;
					int tins = M - oe_ins; tins = tins > 0 ? tins : 0;
					const int g = act ? tins + a * e_ins : NEG_BIG;
					const int incl = q_scan_max_incl(g);
					int excl = q_prev(incl, NEG_BIG);
					excl = excl > carry ? excl : carry;                 // prefix max over all earlier columns
					{ const int tot = q_last(incl); carry = carry > tot ? carry : tot; }
					int f = excl - (a - 1) * e_ins;
					f = (a == beg || f < 0) ? 0 : f;
					int h = M > er ? M : er; h = h > f ? h : f; h = act ? h : 0;
					int ee = M - oe_del; ee = ee > 0 ? ee : 0; ee = ee > er - e_del ? ee : er - e_del;
					const int en = act ? ee : ((in && a == end) ? 0 : er);
					Er[c] = en;
					// row maximum and the last column that attains it: (h << 9 | column), columns < 512
					{ const int key = act ? (h << 9 | a) : -1; kmax = kmax > key ? kmax : key; }
					vlast = (act && a == end - 1) ? h : vlast;
					// H: entry a takes h(i, a-1) for a-1 in the band, entry beg takes the first-column value
					int up = q_prev(h, 0);
					const int edge = q_ror1(hprev);
					up = l == 0 ? edge : up;
					const int hn = !in ? hr : a == beg ? h1_init : (a - 1 >= beg && a - 1 < end) ? up : hr;
					Hr[c] = hn;
					hprev = h;
					// the non-zero cells of the row as the next one finds them (ksw.c:466-469)
					const bool nz = in && a >= beg && a <= end && (hn | en) != 0;
					{ // (within a lane the columns grow with the slot: the first one seen stays, the last one seen wins)
						const uint32_t hi = (nz && a < end && (zpk >> 16) == 0) ? (uint32_t)(0xffff - a) << 16 : zpk & 0xffff0000u;
						const uint32_t lo = nz ? (uint32_t)(a + 1) : zpk & 0xffffu;
						zpk = hi | lo;
					}
				}
			}
			{
				const int key = q_allmax(kmax);
				m = nonempty ? key >> 9 : 0; mj = (nonempty && key >= 0) ? (key & 511) : -1;
			}
			if (__ballot(nonempty && end == qlen)) { // h(i, end-1), only read for the to-the-end score
				const int v = q_allmax(vlast);   // h >= 0
				h1_last = (nonempty && end == qlen) ? v : h1_last;
			}
		}
		if (__ballot(run && !nonempty)) { // empty row: only the boundary cell is written (ksw.c:449)
#pragma unroll
			for (int c = 0; c < NCQ; ++c) { const bool hit = run && !nonempty && (c << 4) + l == end; Hr[c] = hit ? h1_init : Hr[c]; Er[c] = hit ? 0 : Er[c]; }
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
		{ // the band of the next row: the non-zero cells (ksw.c:466-469)
			const uint32_t zr = q_allmax_pk(zpk);
			int nb = (zr >> 16) ? 0xffff - (int)(zr >> 16) : 0x7fffffff;
			int last = (int)(zr & 0xffffu) - 1;
			if (shrink) {
				nb = nb < end ? nb : end;
				last = last > nb - 1 ? last : nb - 1;
				beg = nb;
				end = last + 2 < qlen ? last + 2 : qlen;
				++i;
				if (i >= tlen) stop = true;
				else if (yleft == 0) st = X4_REFILL;
			}
		}
		if (run && stop) st = X4_AFTER;
	}
	if (prof) { // tracing: jobs, rows, trips (a trip advances up to four rows) and passes through the block between extensions
		pf_rows = (unsigned int)wave_sum_i32(l == 0 ? (int)pf_rows : 0); pf_jobs = (unsigned int)wave_sum_i32(l == 0 ? (int)pf_jobs : 0);
		if (lane == 0) { atomicAdd(&prof[0], (unsigned long long)pf_jobs); atomicAdd(&prof[1], (unsigned long long)pf_rows); atomicAdd(&prof[2], (unsigned long long)pf_trips); atomicAdd(&prof[3], (unsigned long long)pf_cold); atomicAdd(&prof[4], (unsigned long long)pf_slots); }
		pf_nrows = (unsigned int)wave_sum_i32(l == 0 ? (int)pf_nrows : 0);
		if (lane == 0) atomicAdd(&prof[5], (unsigned long long)pf_nrows);
	}
}

int x4_max_query(int ncq) { return 16 * ncq - 1; }

// jobs[0 .. n) of the batch form -> res; *cursor must be zero
void launch_ext4_batch(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                       unsigned int n, unsigned int *cursor, int max_qlen)
{
	const long long want = ((long long)n + 15) / 16;
	const int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)n_cu * 8));
	RegParams P; memset(&P, 0, sizeof(P));
	if (max_qlen <= x4_max_query(10))
		hipLaunchKernelGGL((k_ext4<10, false>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, (const void*)jobs, (void*)res, (unsigned char*)nullptr, (const unsigned int*)nullptr, n, cursor, (unsigned long long*)nullptr, 0);
	else
		hipLaunchKernelGGL((k_ext4<16, false>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, (const void*)jobs, (void*)res, (unsigned char*)nullptr, (const unsigned int*)nullptr, n, cursor, (unsigned long long*)nullptr, 0);
}

// The extensions of the chains the tiers exported (records with has_ext), ahead of launch_c2r: k_x4prep lists the jobs (ctr32[0], [2] = their
// numbers, ctr32[1] = k_ext4's cursor: zeroed here), k_ext4 runs them.  n_tasks bounds the number of exported strand searches.
void launch_x4(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, const bsx_seed_task_t *tasks,
               long long n_tasks, const RgXPoolArg &XA, void *jobs, unsigned long long job_cap, unsigned int *ctr32, unsigned long long *prof)
{
	RgXPool X = rgx_pool(&XA);
	const unsigned int jcap = (unsigned int)std::min<unsigned long long>(job_cap, 0xfffffff0ull);
	const int pgrid = (int)((n_tasks + 64 * X4P_WPB - 1) / (64 * X4P_WPB));
	(void)hipMemsetAsync(ctr32, 0, 16, st);
	hipLaunchKernelGGL(k_x4prep, dim3(std::max(1, pgrid)), dim3(64 * X4P_WPB), 0, st, ix, P, tasks, X, (X4Job*)jobs, jcap, ctr32);
	const int use_l = (int)bsx_tune_long("xl", 1);   // 0: the narrow queue through k_ext4 too
	if (use_l) launch_extl(st, n_cu, ix, sc, P, reads, jobs, jcap, ctr32, X.base, n_tasks * 4, prof);
	const int wpc = 3;   // (three waves per SIMD at 167 VGPRs)
	const int grid = (int)std::max<long long>(1, std::min<long long>((n_tasks * 4 + 15) / 16, (long long)n_cu * wpc));
	hipLaunchKernelGGL((k_ext4<10, true>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, (const void*)jobs, (void*)nullptr, X.base, (const unsigned int*)ctr32, jcap, ctr32 + 1, prof, use_l ? 0 : 1);
}
