// k_dedup.hip -- C5 on the device: mem_sort_deduplicate (lib/aln/mem_alnreg.c:112-202) over the regions the region tiers
// left in HBM, one LANE per read.
//
// The work of a read is a few dozen dependent steps over a handful of regions (two sorts with klib's introsort, whose
// order for equal keys matters; a backward scan per region with early exits): nothing in it spans lanes, and a chunk has a
// million reads, so each lane runs the reference's own loop nest for one read and the parallelism is the number of reads.
// The lane keeps its order array in LDS and reads the regions' fields where they lie (no per-lane copies: those were 528
// bytes of scratch memory and 256 registers).
// A read's regions are the regions of its strand searches concatenated in call order (bwamem.c:352-372); without a
// concatenation (below) the function only drops and reorders them, so the result is a list of indices into that
// concatenation: the host builds the read's mem_alnreg_v straight in its final order.
// Left to the host (out_n = -1): a read one of whose strand searches the device did not finish (declined, or being seeded
// again), more than DD_CAP regions or DD_PER_READ strand searches, and a read where mem_test_reg_concatenation (mem_alnreg.c:63-108) gets as far as its
// global alignment -- two collinear regions a gap apart: the score is a k_global job, and the host's merge rounds batch those.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "kernels.h"
#include "wave.hpp"
#include "parsort.hpp"

#define DD_CAP 32

// klib introsort (ksort.h:184-236) over region indices with the control flow of csrc/host/util.c:bsx_introsort (see
// rg_introsort_keys in k_regions.hip); LT(x, y) compares the regions the indices name
template <int STK = 4, typename Arr, typename LT>
__device__ __forceinline__ void dd_introsort(Arr a, int n, LT lt)
{
#define SWP(i, j) do { const auto t_ = a[i]; a[i] = a[j]; a[j] = t_; } while (0)
	if (n < 2) return;
	if (n == 2) { if (lt(a[1], a[0])) SWP(0, 1); return; }
	int d, s = 0, t = n - 1, i, j, k, top = 0;
	int stk_l[STK], stk_r[STK], stk_d[STK];   // a side is stacked only when longer than 16 (and it is the longer side: the stack stays below log2(n / 16) + 1)
	for (d = 2; (1 << d) < n; ++d);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { // comb sort fallback (ksort.h:162-183); not reached for n <= DD_CAP, kept for the control flow
				const double shrink = 1.2473309501039786540366528676643;
				int m = t - s + 1, gap = m, swapped;
				do {
					if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
					swapped = 0;
					for (i = 0; i + gap < m; ++i) if (lt(a[s + i + gap], a[s + i])) { SWP(s + i, s + i + gap); swapped = 1; }
				} while (swapped || gap > 2);
				if (gap != 1) for (i = s + 1; i <= t; ++i) for (j = i; j > s && lt(a[j], a[j - 1]); --j) SWP(j, j - 1);
				t = s;
				continue;
			}
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
			else k = lt(a[j], a[i]) ? i : j;
			const auto rp = a[k];
			if (k != t) SWP(k, t);
			for (;;) {
				do ++i; while (lt(a[i], rp));
				do --j; while (i <= j && lt(rp, a[j]));
				if (j <= i) break;
				SWP(i, j);
			}
			SWP(i, t);
			if (i - s > t - i) {
				if (i - s > 16) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { for (i = 1; i < n; ++i) for (j = i; j > 0 && lt(a[j], a[j - 1]); --j) SWP(j, j - 1); return; }
			--top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
		}
	}
#undef SWP
}

// a read's regions, numbered through the concatenation of its strand searches' lists; the records stay where the region
// tiers left them (32 of their 56 bytes are read here, a few times each and out of L2), so that a lane carries a handful
// of registers and an order array of DD_CAP bytes in LDS instead of six arrays of DD_CAP entries in scratch memory
#define DD_PER_READ 4
struct DdRead {
	const bsx_region_t *base[DD_PER_READ]; int cum[DD_PER_READ];   // list t holds entries cum[t-1] .. cum[t]-1
	__device__ __forceinline__ const bsx_region_t &at(int k) const
	{
		const bsx_region_t *r = base[0] + k;
#pragma unroll
		for (int t = 1; t < DD_PER_READ; ++t) if (k >= cum[t - 1]) r = base[t] + (k - cum[t - 1]);
		return *r;
	}
};
struct DdOrd {   // the order array of one lane: entry i at p[i * 256]
	unsigned char *p;
	__device__ __forceinline__ unsigned char &operator[](int i) const { return p[i * 256]; }
};

// reads with more regions than a lane holds go to k_dedup_long (below), a wavefront each: listed here by size class
#define DL_CAP_A 256
#define DL_CAP_B 1024
__global__ void __launch_bounds__(256)
k_dedup(const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
        long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, int *out_n, unsigned char *out_idx,
        int *long_list, unsigned int *long_count)   // long_list: two lists of n_reads entries (classes A, B), or null
{
	__shared__ unsigned char s_ord[DD_CAP * 256];
	const int rd = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (rd >= n_reads) return;
	if (per_read > DD_PER_READ) { out_n[rd] = -1; return; }
	DdRead R;
	int n = 0;
#pragma unroll
	for (int t = 0; t < DD_PER_READ; ++t) {
		R.base[t] = regs; R.cum[t] = n;
		if (t < per_read) {
			const int m = reg_n[rd * per_read + t];
			if (m < 0 || n + m > DD_CAP) {
				out_n[rd] = -1;
				if (m >= 0 && long_list) { // all of the read's strand searches finished on the device?  then a wavefront takes it
					int tot = n + m;
					for (int u = t + 1; u < per_read && tot >= 0; ++u) { const int mu = reg_n[rd * per_read + u]; tot = mu < 0 ? -1 : tot + mu; }
					if (tot >= 0 && tot <= DL_CAP_B) {
						const int cls = tot <= DL_CAP_A ? 0 : 1;
						long_list[(size_t)cls * (size_t)n_reads + atomicAdd(&long_count[cls], 1u)] = rd;
					}
				}
				return;
			}
			R.base[t] = regs + reg_off[rd * per_read + t];
			n += m; R.cum[t] = n;
		}
	}
	unsigned char *o = out_idx + (size_t)rd * DD_CAP;
	if (n <= 1) { if (n) o[0] = 0; out_n[rd] = n; return; }   // mem_alnreg.c:114
	DdOrd ord; ord.p = s_ord + threadIdx.x;
	for (int k = 0; k < n; ++k) ord[k] = (unsigned char)k;
	dd_introsort(ord, n, [&](int x, int y) { return R.at(x).re < R.at(y).re; });   // by END (alnreg_slt2)
	unsigned int dead = 0;   // stands for qe = qb (mem_alnreg.c:137,139)
	for (int a = 1; a < n; ++a) {
		const int p = ord[a];
		const bsx_region_t &P = R.at(p);
		const long long rb_p = P.rb, re_p = P.re; const int qb_p = P.qb, qe_p = P.qe, rid_p = P.rid, sc_p = P.score;
		for (int b = a - 1; b >= 0; --b) {
			const int q = ord[b];
			const bsx_region_t &Q = R.at(q);
			const long long rb_q = Q.rb, re_q = Q.re; const int qb_q = Q.qb, qe_q = Q.qe;
			if (!(rid_p == Q.rid && rb_p < re_q + max_chain_gap)) break;
			if (dead >> q & 1) continue;
			const long long or_ = re_q - rb_p;
			const long long oq = qb_q < qb_p ? qe_q - qb_p : qe_p - qb_q;
			const long long mr = re_q - rb_q < re_p - rb_p ? re_q - rb_q : re_p - rb_p;
			const long long mq = qe_q - qb_q < qe_p - qb_p ? qe_q - qb_q : qe_p - qb_p;
			if ((float)or_ > mask_level_redun * (float)mr && (float)oq > mask_level_redun * (float)mq) { // one of the two is redundant
				if (sc_p < Q.score) { dead |= 1u << p; break; }
				else dead |= 1u << q;
			} else if (rb_q < rb_p) { // mem_test_reg_concatenation(q, p) up to its alignment (mem_alnreg.c:63-91)
				if (rb_q < l_pac && rb_p >= l_pac) continue;
				if (qb_q >= qb_p || qe_q >= qe_p || re_q >= re_p) continue;
				long long w = (re_q - rb_p) - (long long)(qe_q - qb_p);
				int wi = (int)w; wi = wi > 0 ? wi : -wi;
				double r = (double)(re_q - rb_p) / (double)(re_p - rb_q) - (double)(qe_q - qb_p) / (double)(qe_p - qb_q);
				r = r > 0. ? r : -r;
				if (re_q < rb_p || qe_q < qb_p) { if (wi > opt_w << 1 || r >= (double)0.05f) continue; }
				else if (wi > opt_w << 2 || r >= (double)(0.05f * 2)) continue;
				out_n[rd] = -1;   // the two may be one alignment: scored and merged by the host's rounds
				return;
			}
		}
	}
	int m = 0;
	for (int k = 0; k < n; ++k) { const unsigned char v = ord[k]; if (!(dead >> v & 1)) ord[m++] = v; }
	dd_introsort(ord, m, [&](int x, int y) {   // alnreg_slt
		const bsx_region_t &X = R.at(x), &Y = R.at(y);
		return X.score > Y.score || (X.score == Y.score && (X.rb < Y.rb || (X.rb == Y.rb && X.qb < Y.qb)));
	});
	dead = 0;
	for (int k = 1; k < m; ++k) { // identical hits
		const int x = ord[k]; const bsx_region_t &X = R.at(x), &Y = R.at(ord[k - 1]);
		if (X.score == Y.score && X.rb == Y.rb && X.qb == Y.qb) dead |= 1u << x;
	}
	int m2 = 0;
	for (int k = 0; k < m; ++k) { const unsigned char v = ord[k]; if (k == 0 || !(dead >> v & 1)) o[m2++] = v; }
	out_n[rd] = m2;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same function for the reads a lane cannot hold (more than DD_CAP regions: on a genome with hg38's repeat content 15 % of the
// reads, which hold 60 % of a chunk's regions -- until round 6 the host's): one WAVEFRONT per read, the fields it looks at in LDS.
//  * the two sorts: klib's introsort is not stable, so for EQUAL keys its exact sequence of swaps decides the order -- but only then.
//    Every element's rank by counting (a lane per element, the other keys broadcast one after the other: how many are smaller) gives the
//    sorted order when no two keys are equal, whatever the algorithm.  A list with a tie goes through klib's own sequence of partitions as
//    the whole wavefront runs it (rg_introsort_par, parsort.hpp, the chain filter's sort), from the order the reference starts from, on
//    keys made of that rank: equal keys have equal ranks, smaller keys smaller ones.
//  * the redundancy scan (mem_alnreg.c:131-160): region p of the order by end against the earlier ones, nearest first, until one lies
//    outside p's reach.  What a q does to p (and p to q) depends on the two alone, and a q is met once per p: so the q's go a lane each,
//    64 at a time -- the loop's first exit is the lowest lane outside the reach, p dies at the lowest live lane whose q is redundant
//    with it and scores higher, the redundant q's below that lane die, and a live pair that mem_test_reg_concatenation would have to
//    align (below that lane) hands the read to the host's rounds, as k_dedup does.
// Output: the surviving regions' indices (16 bits each) in the order the reference leaves them, one list behind the other in `pool`.
template <int CAP>
struct DlStore {
	long long re[CAP], rb[CAP];
	int score[CAP], rid[CAP];
	unsigned int q[CAP];             // qb | qe << 16
	unsigned short ord[CAP], ord2[CAP];
	unsigned char dead[CAP];
	unsigned long long act[CAP / 64];   // the p's of the redundancy scan that have anybody within reach
	unsigned int keys[CAP], tmp[CAP];   // a list with tied keys: the packed keys of rg_introsort_par (parsort.hpp) and its partner lists
	short kk[CAP]; int cnt[64], stk[48];
};

template <int CAP>
__global__ void __launch_bounds__(64)
k_dedup_long(const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int per_read, const int *list, const unsigned int *list_n,
             long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun,
             int *out_n, long long *out_off, unsigned short *pool, unsigned long long pool_cap, unsigned long long *pool_cursor)
{
	__shared__ DlStore<CAP> S;
	constexpr int NS = CAP / 64;
	const int lane = (int)threadIdx.x;
	const unsigned int n_list = *list_n;
	for (unsigned int it = blockIdx.x; it < n_list; it += gridDim.x) {
		const int rd = list[it];
		int cum[DD_PER_READ + 1]; long long off[DD_PER_READ];
		int n = 0;
#pragma unroll
		for (int t = 0; t < DD_PER_READ; ++t) {
			cum[t] = n; off[t] = 0;
			if (t < per_read) { off[t] = reg_off[rd * per_read + t]; n += reg_n[rd * per_read + t]; }
		}
		cum[DD_PER_READ] = n;
		if (n > CAP || n < 2) continue;   // (not listed for this class)
		WAVE_SYNC();     // (the LDS of the read before is done with)
		bool wide = false;
		for (int k = lane; k < n; k += 64) {
			long long o = off[0] + k;
#pragma unroll
			for (int t = 1; t < DD_PER_READ; ++t) if (k >= cum[t]) o = off[t] + (k - cum[t]);
			const bsx_region_t R = regs[o];
			S.re[k] = R.re; S.rb[k] = R.rb; S.score[k] = R.score; S.rid[k] = R.rid;
			S.q[k] = (unsigned int)R.qb | (unsigned int)R.qe << 16;
			wide |= (unsigned int)R.qb > 0xffffu || (unsigned int)R.qe > 0xffffu;
			S.ord[k] = (unsigned short)k; S.dead[k] = 0;
		}
		if (__ballot(wide)) continue;     // (a read of 64 k bases or more: the host's)
		WAVE_SYNC();
		// ---- order by END (alnreg_slt2)
		{
			long long key[NS]; int rk[NS]; bool tie = false;
#pragma unroll
			for (int c = 0; c < NS; ++c) { key[c] = c * 64 + lane < n ? S.re[c * 64 + lane] : 0; rk[c] = 0; }
			for (int j = 0; j < n; ++j) {
				const long long kj = S.re[j];
#pragma unroll
				for (int c = 0; c < NS; ++c) if (c * 64 < n) { rk[c] += kj < key[c]; tie |= kj == key[c] && j != c * 64 + lane && c * 64 + lane < n; }
			}
			if (__ballot(tie) == 0) {
#pragma unroll
				for (int c = 0; c < NS; ++c) if (c * 64 + lane < n) S.ord[rk[c]] = (unsigned short)(c * 64 + lane);
			} else {
#pragma unroll
				for (int c = 0; c < NS; ++c) if (c * 64 + lane < n) S.keys[c * 64 + lane] = (unsigned int)(n - rk[c]) << RG_KEY_BITS | (unsigned int)(c * 64 + lane);
				WAVE_SYNC();
				rg_introsort_par<NS>(S.keys, n, S.tmp, S.tmp + CAP / 2, S.stk, lane, S.kk, S.cnt);
				for (int k = lane; k < n; k += 64) S.ord[k] = (unsigned short)(S.keys[k] & ((1u << RG_KEY_BITS) - 1));
			}
		}
		WAVE_SYNC();
		// ---- the redundancy scan.  The reference's inner loop leaves at once when the region just before p (in the order by end) is out of p's
		// reach -- on a repeat family's reads, whose regions lie all over the genome, that is most p: found for every p at once, a lane each,
		// and only the others are walked
		for (int c0 = 0; c0 < n; c0 += 64) {
			const int a = c0 + lane;
			bool act = false;
			if (a >= 1 && a < n) { const int p = S.ord[a], q = S.ord[a - 1]; act = S.rid[p] == S.rid[q] && S.rb[p] < S.re[q] + max_chain_gap; }
			const unsigned long long am = __ballot(act);
			if (lane == 0) S.act[c0 >> 6] = am;
		}
		WAVE_SYNC();
		bool bail = false;
		for (int c0 = 0; c0 < n && !bail; c0 += 64)
		for (unsigned long long am = (unsigned long long)(unsigned int)uni((int)(S.act[c0 >> 6] >> 32)) << 32 | (unsigned int)uni((int)S.act[c0 >> 6]); am && !bail; am &= am - 1) {
			const int a = c0 + (int)__builtin_ctzll(am);
			const int p = __builtin_amdgcn_readfirstlane((int)S.ord[a]);
			const long long rb_p = S.rb[p], re_p = S.re[p];
			const int qb_p = (int)(S.q[p] & 0xffffu), qe_p = (int)(S.q[p] >> 16), rid_p = S.rid[p], sc_p = S.score[p];
			for (int base = a - 1; base >= 0; base -= 64) {
				const int b = base - lane;
				const int q = b >= 0 ? (int)S.ord[b] : 0;
				const long long rb_q = S.rb[q], re_q = S.re[q];
				const int qb_q = (int)(S.q[q] & 0xffffu), qe_q = (int)(S.q[q] >> 16);
				const bool reach = b >= 0 && rid_p == S.rid[q] && rb_p < re_q + max_chain_gap;
				const unsigned long long stop = ~__ballot(reach);
				const int f = stop ? (int)__builtin_ctzll(stop) : 64;           // the loop's exit: the lowest lane outside p's reach
				const bool live = lane < f && !S.dead[q];
				const long long or_ = re_q - rb_p;
				const long long oq = qb_q < qb_p ? qe_q - qb_p : qe_p - qb_q;
				const long long mr = re_q - rb_q < re_p - rb_p ? re_q - rb_q : re_p - rb_p;
				const long long mq = qe_q - qb_q < qe_p - qb_p ? qe_q - qb_q : qe_p - qb_p;
				const bool red = live && (float)or_ > mask_level_redun * (float)mr && (float)oq > mask_level_redun * (float)mq;
				const bool kills = red && sc_p < S.score[q];
				bool cc = false;
				if (live && !red && rb_q < rb_p) { // mem_test_reg_concatenation(q, p) up to its alignment (mem_alnreg.c:63-91)
					if (!(rb_q < l_pac && rb_p >= l_pac) && !(qb_q >= qb_p || qe_q >= qe_p || re_q >= re_p)) {
						const long long w = (re_q - rb_p) - (long long)(qe_q - qb_p);
						int wi = (int)w; wi = wi > 0 ? wi : -wi;
						double r = (double)(re_q - rb_p) / (double)(re_p - rb_q) - (double)(qe_q - qb_p) / (double)(qe_p - qb_q);
						r = r > 0. ? r : -r;
						if (re_q < rb_p || qe_q < qb_p) cc = !(wi > opt_w << 1 || r >= (double)0.05f);
						else cc = !(wi > opt_w << 2 || r >= (double)(0.05f * 2));
					}
				}
				const unsigned long long km = __ballot(kills), cm = __ballot(cc);
				const int k = km ? (int)__builtin_ctzll(km) : 64;                 // p dies here
				const unsigned long long below = k >= 64 ? ~0ull : (1ull << k) - 1;
				if (cm & below) { bail = true; break; }
				if (red && !kills && lane < k) S.dead[q] = 1;
				if (k < 64) { if (lane == 0) S.dead[p] = 1; break; }
				if (f < 64) break;
			}
			WAVE_SYNC();
		}
		if (bail) continue;   // (out_n stays -1: the host's rounds score and merge the pair)
		// ---- what is left, still in the order by end
		int m = 0;
		for (int c0 = 0; c0 < n; c0 += 64) {
			const int pos = c0 + lane;
			const int v = pos < n ? (int)S.ord[pos] : 0;
			const bool keep = pos < n && !S.dead[v];
			const unsigned long long km = __ballot(keep);
			if (keep) S.ord2[m + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)km, 0u))] = (unsigned short)v;
			m += __popcll(km);
		}
		WAVE_SYNC();
		// ---- order by score, then start, then query start (alnreg_slt)
		auto lt2 = [&](int x, int y) {
			const int sx = S.score[x], sy = S.score[y];
			const long long bx = S.rb[x], by = S.rb[y];
			return sx > sy || (sx == sy && (bx < by || (bx == by && (S.q[x] & 0xffffu) < (S.q[y] & 0xffffu))));
		};
		{
			int el[NS], rk[NS]; bool tie = false;
#pragma unroll
			for (int c = 0; c < NS; ++c) { el[c] = c * 64 + lane < m ? (int)S.ord2[c * 64 + lane] : 0; rk[c] = 0; }
			for (int j = 0; j < m; ++j) {
				const int ej = __builtin_amdgcn_readfirstlane((int)S.ord2[j]);
#pragma unroll
				for (int c = 0; c < NS; ++c) if (c * 64 < m) {
					const bool a_lt = lt2(ej, el[c]);
					rk[c] += a_lt;
					tie |= !a_lt && !lt2(el[c], ej) && j != c * 64 + lane && c * 64 + lane < m;
				}
			}
			if (__ballot(tie) == 0) {
#pragma unroll
				for (int c = 0; c < NS; ++c) if (c * 64 + lane < m) S.ord[rk[c]] = (unsigned short)el[c];
			} else {   // (from the order the reference starts this sort from: what is left, by end)
#pragma unroll
				for (int c = 0; c < NS; ++c) if (c * 64 + lane < m) S.keys[c * 64 + lane] = (unsigned int)(m - rk[c]) << RG_KEY_BITS | (unsigned int)(c * 64 + lane);
				WAVE_SYNC();
				rg_introsort_par<NS>(S.keys, m, S.tmp, S.tmp + CAP / 2, S.stk, lane, S.kk, S.cnt);
				for (int k = lane; k < m; k += 64) S.ord[k] = S.ord2[S.keys[k] & ((1u << RG_KEY_BITS) - 1)];
			}
		}
		WAVE_SYNC();
		// ---- identical hits (mem_alnreg.c:183-189): every region equal to the one before it in (score, start, query start) goes
		int m2 = 0;
		for (int c0 = 0; c0 < m; c0 += 64) {
			const int pos = c0 + lane;
			bool keep = pos < m;
			if (keep && pos > 0) { const int x = S.ord[pos], y = S.ord[pos - 1]; keep = !(S.score[x] == S.score[y] && S.rb[x] == S.rb[y] && (S.q[x] & 0xffffu) == (S.q[y] & 0xffffu)); }
			const unsigned long long km = __ballot(keep);
			if (keep) S.ord2[m2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)km, 0u))] = S.ord[pos];
			m2 += __popcll(km);
		}
		WAVE_SYNC();
		unsigned long long at = 0;
		if (lane == 0) at = atomicAdd(pool_cursor, (unsigned long long)m2);
		at = (unsigned long long)__shfl((long long)at, 0);
		if (at + (unsigned long long)m2 > pool_cap) continue;   // (no room: the host's)
		for (int k = lane; k < m2; k += 64) pool[at + k] = S.ord2[k];
		if (lane == 0) { out_off[rd] = (long long)at; out_n[rd] = m2; }
	}
}

int dedup_cap(void) { return DD_CAP; }
int dedup_long_cap(void) { return DL_CAP_B; }
void launch_dedup_long(hipStream_t st, int n_cu, const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
                       long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, const int *long_list, const unsigned int *long_count,
                       int *out_n, long long *out_off, unsigned short *pool, unsigned long long pool_cap, unsigned long long *pool_cursor)
{
	hipLaunchKernelGGL(k_dedup_long<DL_CAP_A>, dim3(n_cu * 12), dim3(64), 0, st, regs, reg_off, reg_n, per_read, long_list, long_count,
	                   l_pac, max_chain_gap, opt_w, mask_level_redun, out_n, out_off, pool, pool_cap, pool_cursor);
	hipLaunchKernelGGL(k_dedup_long<DL_CAP_B>, dim3(n_cu * 4), dim3(64), 0, st, regs, reg_off, reg_n, per_read, long_list + n_reads, long_count + 1,
	                   l_pac, max_chain_gap, opt_w, mask_level_redun, out_n, out_off, pool, pool_cap, pool_cursor);
}
void launch_dedup(hipStream_t st, const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
                  long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, int *out_n, unsigned char *out_idx,
                  int *long_list, unsigned int *long_count)
{
	hipLaunchKernelGGL(k_dedup, dim3((n_reads + 255) / 256), dim3(256), 0, st, regs, reg_off, reg_n, n_reads, per_read, l_pac, max_chain_gap, opt_w, mask_level_redun, out_n, out_idx,
	                   long_list, long_count);
}
