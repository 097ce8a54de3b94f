// k_regions.hip -- the middle of the pipeline fused on the device: one wavefront per strand search takes
// the SA intervals of K1+K2 all the way to alignment regions:
//   K3  bwt_sa for every occurrence (lanes in parallel)                       lib/aln/bwt.c:87-97
//   C1  mem_chain: seeds clustered into chains in reference order             lib/aln/memchain.c:268-393
//   C2  mem_chain_flt (weights, klib introsort permutation, overlap filter)   lib/aln/memchain.c:406-488
//   C4  mem_chain2region(1): best-first seed extension with the containment   lib/aln/memchain.c:742-904
//       tests, left/right ksw_extend2 (wave-wide, ext_dp.hpp) and band retries
// Everything lives in this wave's LDS; nothing goes back to the host between seeding and regions.
//
// The kernel only takes the common case and says so per task (status != 0 => the host runs its own
// C1/C2/C4 for that task through the batch kernels): more than RG_SCAP occurrences / RG_CCAP chains /
// RG_RCAP regions, reads longer than RG_QCAP or long enough for the seed-SW filter (memchain.c:544),
// and two chains starting at the same reference position (there the reference's B-tree shape decides).
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"
#include "ext_dp.hpp"

#define RG_ICAP 48
#define RG_SCAP 96
#define RG_CCAP 96       // every occurrence may start its own chain
#define RG_RCAP 12
#define RG_QCAP 256
#define RG_NC 4          // band <= 2w+1 <= 201 columns with the default w; wider bands fall back

struct RgChain {         // mem_chain_t reduced to what chaining and the filter read: first seed = (pos, first_q), last seed
	long long pos, last_r;
	int rid, w;
	short first_q, last_q, last_len, first;
	signed char kept; unsigned char is_alt, has_extra, pad;
};

struct RgLds {
	// intervals (sorted by info)
	unsigned long long iv_x0[RG_ICAP];
	int iv_n[RG_ICAP]; short iv_beg[RG_ICAP], iv_end[RG_ICAP];
	// seeds in arrival order
	long long s_rbeg[RG_SCAP];
	int s_rid[RG_SCAP];
	short s_qbeg[RG_SCAP], s_len[RG_SCAP]; signed char s_chain[RG_SCAP], s_extra[RG_SCAP];
	RgChain ch[RG_CCAP];
	unsigned char ord[RG_CCAP];       // chain indices: by position, then in filter order
	unsigned char keep[RG_CCAP];      // mem_chain_flt's kept list (indices into ord)
	unsigned char lst[RG_SCAP];       // seed indices of the current chain / list
	unsigned long long srt[RG_SCAP];  // score<<32|i, ascending (memchain.c:748-752)
	bsx_region_t regs[RG_RCAP];
	int n_chains, n_regs, status;
	int32_t H[RG_QCAP + 2], E[RG_QCAP + 2];
	uint8_t qb[RG_QCAP + 4];
};

// wave-uniform values live in scalar registers: say so for what comes out of LDS, shuffles and reductions
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v)
{
	return (long long)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32 | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

__device__ __forceinline__ long long wave_max_i64(long long v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const long long o = (long long)((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), off) << 32 | (unsigned)__shfl_xor((int)v, off));
		v = v > o ? v : o;
	}
	return v;
}

__device__ __forceinline__ int rg_pos2rid(const DevIndex &ix, long long pos_f)   // bns_pos2rid, bntseq.c:356-369
{
	int left = 0, mid = 0, right = ix.n_seqs;
	if (pos_f >= ix.l_pac) return -1;
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
__device__ __forceinline__ long long rg_depos(long long l_pac, long long p) { return p >= l_pac ? (l_pac << 1) - 1 - p : p; }
__device__ __forceinline__ int rg_intv2rid(const DevIndex &ix, long long rb, long long re)   // bns_intv2rid, bntseq.c:371-379
{
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	const int a = rg_pos2rid(ix, rg_depos(ix.l_pac, rb));
	const int b = rb < re ? rg_pos2rid(ix, rg_depos(ix.l_pac, re - 1)) : a;
	return a == b ? a : -1;
}
__device__ __forceinline__ int rg_cal_max_gap(const RegParams &P, int qlen)   // memchain.c:576-582
{
	int l_del = (int)((double)(qlen * P.a - P.o_del) / P.e_del + 1.);
	int l_ins = (int)((double)(qlen * P.a - P.o_ins) / P.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < P.w << 1 ? l : P.w << 1;
}
#define RG_BSS(parent, l_pac, rb) ((((rb) > (l_pac)) == (parent)) ? 1 : 0)

// klib introsort (ksort.h:184-236) on chain indices ord[0..n) with "a before b" = w[a] > w[b]; n <= RG_CCAP.
// Same control flow as csrc/host/util.c:bsx_introsort so that equal weights end in the reference's order.
__device__ __noinline__ void rg_introsort_w(unsigned char *a, int n, const RgChain *ch, int *stk)
{
#define LT(x, y) (ch[(x)].w > ch[(y)].w)
#define SWP(i, j) do { unsigned char t_ = a[i]; a[i] = a[j]; a[j] = t_; } while (0)
	if (n < 2) return;
	if (n == 2) { if (LT(a[1], a[0])) SWP(0, 1); return; }
	int d, s = 0, t = n - 1, i, j, k, top = 0;
	int *stk_l = stk, *stk_r = stk + 8, *stk_d = stk + 16;   // a range is stacked only when longer than 16: depth <= 3 for n <= 96
	for (d = 2; (1 << d) < n; ++d);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { // comb sort fallback (ksort.h:162-183)
				const double shrink = 1.2473309501039786540366528676643;
				int m = t - s + 1, gap = m, swapped;
				do {
					if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
					swapped = 0;
					for (i = 0; i + gap < m; ++i) if (LT(a[s + i + gap], a[s + i])) { SWP(s + i, s + i + gap); swapped = 1; }
				} while (swapped || gap > 2);
				if (gap != 1) for (i = s + 1; i <= t; ++i) for (j = i; j > s && LT(a[j], a[j - 1]); --j) SWP(j, j - 1);
				t = s;
				continue;
			}
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (LT(a[k], a[i])) { if (LT(a[k], a[j])) k = j; }
			else k = LT(a[j], a[i]) ? i : j;
			const unsigned char rp = a[k];
			if (k != t) SWP(k, t);
			for (;;) {
				do ++i; while (LT(a[i], rp));
				do --j; while (i <= j && LT(rp, a[j]));
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
			if (top == 0) { for (i = 1; i < n; ++i) for (j = i; j > 0 && LT(a[j], a[j - 1]); --j) SWP(j, j - 1); return; }
			--top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
		}
	}
#undef LT
#undef SWP
}

__global__ void __launch_bounds__(256, 3)
k_regions(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks,
          const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
          bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
          unsigned int *task_cursor)
{
	__shared__ RgLds lds[4];
	const int lane = wave_lane();
	RgLds &S = lds[threadIdx.x >> 6];
	const long long l_pac = ix.l_pac;

	for (;;) {
		int t = 0;
		if (lane == 0) t = (int)atomicAdd(task_cursor, 1u);
		t = uni(t);
		if (t >= n_tasks) break;
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const uint8_t *query = reads + qoff;
		const int n_iv = uni(task_n[t]);
		int status = 0, n_seeds = 0;
		if (lane == 0) { S.n_chains = 0; S.n_regs = 0; S.status = 0; }
		// reads the host must take: seeding overflowed, too long for the LDS tile, or long enough for the seed-SW filter
		if (n_iv < 0 || n_iv > RG_ICAP || l_query > RG_QCAP) status = 1;
		else {
			const double min_l = P.min_chain_weight ? 1.1f * P.min_chain_weight : 5.5f * log((double)l_query);
			if (l_query >= 1 && !(min_l > 0.05f * l_query)) status = 1;
		}
		WAVE_SYNC();
		if (status == 0 && n_iv > 0 && l_query >= P.min_seed_len) {
			// ---- A. intervals, ordered by info (ks_introsort(mem_intv), memchain.c:105; equal keys are identical records)
			const DevIntv *src = seeds_dense + uni64(task_off[t]);
			DevIntv mine; mine.x0 = mine.x1 = mine.x2 = 0; mine.info = 0;
			if (lane < n_iv) mine = src[lane];
			int rank = 0;
			for (int k = 0; k < n_iv; ++k) {
				const unsigned long long oi = (unsigned long long)(unsigned)__shfl((int)(mine.info >> 32), k) << 32 | (unsigned)__shfl((int)mine.info, k);
				rank += (oi < mine.info) || (oi == mine.info && k < lane);
			}
			if (lane < n_iv) {
				S.iv_x0[rank] = mine.x0;
				S.iv_n[rank] = mine.x2 > 0x7fffffffull ? 0x7fffffff : (int)mine.x2;
				S.iv_beg[rank] = (short)(mine.info >> 32); S.iv_end[rank] = (short)(uint32_t)mine.info;
			}
			WAVE_SYNC();
			// ---- B. occurrences: every k < x[2] of every interval (the caps of memchain.c:325-326 cannot bind below RG_SCAP)
			int tot = 0, over = 0;
			for (int i = 0; i < n_iv; ++i) { const int c = uni(S.iv_n[i]); if (c > RG_SCAP || c > P.max_occ) over = 1; tot += c > RG_SCAP ? RG_SCAP : c; }
			if (over || tot > RG_SCAP) status = 2;
			if (status == 0) {
				n_seeds = tot;
				for (int o = lane; o < tot; o += 64) {
					int i = 0, acc = 0;
					while (acc + S.iv_n[i] <= o) { acc += S.iv_n[i]; ++i; }
					const unsigned long long k0 = S.iv_x0[i] + (unsigned long long)(o - acc);
					// bwt_sa (bwt.c:87-97) on this strand's own index
					const unsigned long long prim = dev_ix_primary(ix, parent);
					const uint32_t *bw = dev_ix_bwt(ix, parent);
					const uint64_t *sa = parent ? ix.fmi[1].sa : ix.fmi[0].sa;
					const uint32_t sa_mask = ix.fmi[0].sa_mask, sa_shift = ix.fmi[0].sa_shift;
					unsigned long long k = k0, steps = 0;
					while (k & sa_mask) {
						if (k == prim) { k = 0; ++steps; continue; }
						const unsigned long long x = k - (k > prim);
						const DevBlock B = dev_load_block4(bw, x);
						const uint32_t wsel = (uint32_t)((x & 127) >> 4);
						const uint32_t word = wsel == 0 ? B.v2.x : wsel == 1 ? B.v2.y : wsel == 2 ? B.v2.z : wsel == 3 ? B.v2.w :
						                      wsel == 4 ? B.v3.x : wsel == 5 ? B.v3.y : wsel == 6 ? B.v3.z : B.v3.w;
						const int c = (int)((word >> ((~x & 15) << 1)) & 3);
						uint32_t ca, cc, cg, ct;
						dev_block_count4(B, (int)(x & 127), ca, cc, cg, ct);
						const unsigned long long base = c == 0 ? ((unsigned long long)B.v0.y << 32 | B.v0.x) : c == 1 ? ((unsigned long long)B.v0.w << 32 | B.v0.z) :
						                                c == 2 ? ((unsigned long long)B.v1.y << 32 | B.v1.x) : ((unsigned long long)B.v1.w << 32 | B.v1.z);
						k = dev_ix_L2(ix, parent, c) + base + (c == 0 ? ca : c == 1 ? cc : c == 2 ? cg : ct);
						++steps;
					}
					const long long pos = (long long)(steps + sa[k >> sa_shift]);
					const int slen = S.iv_end[i] - S.iv_beg[i];
					S.s_rbeg[o] = pos; S.s_qbeg[o] = S.iv_beg[i]; S.s_len[o] = (short)slen;
					S.s_rid[o] = rg_intv2rid(ix, pos, pos + slen);
					S.s_chain[o] = -1; S.s_extra[o] = 0;
				}
				WAVE_SYNC();
				// ---- C. chaining in arrival order; chain c's start lives with lanes c and c-64 for the predecessor search
				int nc = 0;
				long long cpos0 = -1, cpos1 = -1;   // start of chain `lane` / `lane + 64`, -1 while unused
				for (int o = 0; o < tot && status == 0; ++o) {
					const int rid = uni(S.s_rid[o]);
					const long long rbeg = uni64(S.s_rbeg[o]);
					const int qbeg = uni(S.s_qbeg[o]), len = uni(S.s_len[o]);
					if (rid < 0) continue;
					if ((P.bsstrand & 1) && RG_BSS(parent, l_pac, rbeg) != P.bsstrand >> 1) continue;
					// kb_intervalp's `lower`: the chain with the largest start <= rbeg (starts are unique here)
					const long long c0 = cpos0 <= rbeg ? cpos0 : -1, c1 = cpos1 <= rbeg ? cpos1 : -1;
					const long long best = uni64(wave_max_i64(c0 > c1 ? c0 : c1));
					int lower = -1;
					if (best >= 0) {
						const unsigned long long b0 = __ballot(c0 == best), b1 = __ballot(c1 == best);
						lower = b0 ? __ffsll((long long)b0) - 1 : 64 + __ffsll((long long)b1) - 1;
					}
					int merged = 0;
					if (lower >= 0) { // merge_seed_to_chain, memchain.c:227-256
						const RgChain c = S.ch[lower];
						if (rid == c.rid) {
							if (qbeg >= c.first_q && qbeg + len <= c.last_q + c.last_len && rbeg >= c.pos && rbeg + len <= c.last_r + c.last_len) {
								if (lane == 0) { S.s_chain[o] = (signed char)lower; S.s_extra[o] = 1; S.ch[lower].has_extra = 1; }
								merged = 1;
							} else if (!((c.last_r < l_pac || c.pos < l_pac) && rbeg >= l_pac)) {
								const long long qdist = qbeg - c.last_q, rdist = rbeg - c.last_r;
								if (rdist >= 0 && qdist - rdist <= P.w && rdist - qdist <= P.w && qdist - c.last_len < P.max_chain_gap && rdist - c.last_len < P.max_chain_gap) {
									if (lane == 0) { S.s_chain[o] = (signed char)lower; S.ch[lower].last_q = (short)qbeg; S.ch[lower].last_r = rbeg; S.ch[lower].last_len = (short)len; }
									merged = 1;
								}
							}
						}
					}
					if (!merged) {
						if (nc == RG_CCAP) { status = 3; break; }
						if (best == rbeg) { status = 4; break; }   // duplicate key: the B-tree shape would matter
						if (lane == 0) {
							RgChain c;
							c.pos = c.last_r = rbeg; c.rid = rid; c.w = 0; c.first_q = c.last_q = (short)qbeg; c.last_len = (short)len; c.first = -1;
							c.kept = 0; c.is_alt = ix.ctg_alt[rid] ? 1 : 0; c.has_extra = 0; c.pad = 0;
							S.ch[nc] = c;
							S.s_chain[o] = (signed char)nc;
						}
						if (lane == (nc & 63)) { if (nc < 64) cpos0 = rbeg; else cpos1 = rbeg; }
						++nc;
					}
					WAVE_SYNC();
				}
				WAVE_SYNC();
				// ---- D. chain order = by start position; weights; filter (mem_chain_flt, memchain.c:406-488)
				if (status == 0 && nc > 0) {
					for (int c = lane; c < nc; c += 64) { // mem_chain_weight, memchain.c:158-180, one lane per chain
						long long end = 0; int w = 0, tmp;
						for (int o = 0; o < tot; ++o) if (S.s_chain[o] == c && !S.s_extra[o]) {
							const int qb = S.s_qbeg[o], ln = S.s_len[o];
							if (qb >= end) w += ln; else if (qb + ln > end) w += (int)(qb + ln - end);
							end = end > qb + ln ? end : qb + ln;
						}
						tmp = w; w = 0; end = 0;
						for (int o = 0; o < tot; ++o) if (S.s_chain[o] == c && !S.s_extra[o]) {
							const long long rb = S.s_rbeg[o]; const int ln = S.s_len[o];
							if (rb >= end) w += ln; else if (rb + ln > end) w += (int)(rb + ln - end);
							end = end > rb + ln ? end : rb + ln;
						}
						w = w < tmp ? w : tmp;
						S.ch[c].w = w < 1 << 30 ? w : (1 << 30) - 1;
						int r = 0;
						const long long mypos = S.ch[c].pos;
						for (int k = 0; k < nc; ++k) r += S.ch[k].pos < mypos;
						S.ord[r] = (unsigned char)c;   // in-order traversal of the tree (memchain.c:372-379)
					}
					WAVE_SYNC();
					int n = 0;
					if (lane == 0) {
						for (int i = 0; i < nc; ++i) { const int c = S.ord[i]; if (S.ch[c].w >= P.min_chain_weight) S.ord[n++] = (unsigned char)c; }
						rg_introsort_w(S.ord, n, S.ch, S.H);
					}
					n = uni(n);
					WAVE_SYNC();
					int nk = 0;
					if (n > 0) {
						if (lane == 0) { S.ch[S.ord[0]].kept = 3; S.keep[0] = 0; }
						nk = 1;
						WAVE_SYNC();
						for (int i = 1; i < n; ++i) {
							// chain i against every kept chain: lanes take kept entries lane and lane+64; the reference's loop stops at the first `drop`
							const RgChain ci = S.ch[S.ord[i]];
							const int ci_beg = ci.first_q, ci_end = ci.last_q + ci.last_len;
							int ov[2] = {0, 0}, dr[2] = {0, 0};
#pragma unroll
							for (int h = 0; h < 2; ++h) {
								const int k = lane + 64 * h;
								if (k < nk) {
									const RgChain ck = S.ch[S.ord[S.keep[k]]];
									const int ck_beg = ck.first_q, ck_end = ck.last_q + ck.last_len;
									const int b_max = ck_beg > ci_beg ? ck_beg : ci_beg, e_min = ck_end < ci_end ? ck_end : ci_end;
									if (e_min > b_max && (!ck.is_alt || ci.is_alt)) {
										const int li = ci_end - ci_beg, lj = ck_end - ck_beg, min_l = li < lj ? li : lj;
										if ((float)(e_min - b_max) >= (float)min_l * P.mask_level && min_l < P.max_chain_gap) {
											ov[h] = 1;
											if ((float)ci.w < (float)ck.w * P.drop_ratio && ck.w - ci.w >= P.min_seed_len << 1) dr[h] = 1;
										}
									}
								}
							}
							const unsigned long long d0 = __ballot(dr[0]), d1 = __ballot(dr[1]);
							const int stop = d0 ? __ffsll((long long)d0) - 1 : d1 ? 64 + __ffsll((long long)d1) - 1 : nk;   // first k that drops chain i
							int large = 0;
#pragma unroll
							for (int h = 0; h < 2; ++h) {
								const int k = lane + 64 * h;
								const int hit = ov[h] && k <= stop;
								if (hit) { RgChain &ck = S.ch[S.ord[S.keep[k]]]; if (ck.first < 0) ck.first = (short)i; }
								if (__ballot(hit)) large = 1;
							}
							if (stop == nk) {
								if (lane == 0) { S.keep[nk] = (unsigned char)i; S.ch[S.ord[i]].kept = large ? 2 : 3; }
								++nk;
							}
							WAVE_SYNC();
						}
						if (lane == 0) {
							for (int i = 0; i < nk; ++i) { const RgChain &c = S.ch[S.ord[S.keep[i]]]; if (c.first >= 0) S.ch[S.ord[c.first]].kept = 1; }
							int i; unsigned int k = 0;
							for (i = 0; i < n; ++i) { const int kp = S.ch[S.ord[i]].kept; if (kp == 0 || kp == 3) continue; if (++k >= P.max_chain_extend) break; }
							for (; i < n; ++i) if (S.ch[S.ord[i]].kept < 3) S.ch[S.ord[i]].kept = 0;
							int m = 0;
							for (i = 0; i < n; ++i) if (S.ch[S.ord[i]].kept) S.ord[m++] = S.ord[i];
							S.n_chains = m;   // ord[0..m) = surviving chains in processing order
						}
						WAVE_SYNC();
					}
				}
			}
			// ---- E. chains -> regions (mem_chain2region, memchain.c:873-904)
			if (status == 0) {
				const int nk = uni(S.n_chains), ns = n_seeds;
				for (int ci = 0; ci < nk && status == 0; ++ci) {
					const int c = uni(S.ord[ci]);
					const long long ch_pos = uni64(S.ch[c].pos);
					const int ch_has_extra = uni(S.ch[c].has_extra);
					// mem_chain_reference_span (memchain.c:585-605) + bns_fetch_seq's contig clamp; one lane per seed
					long long rmax0 = l_pac << 1, rmax1 = 0;
					for (int o = lane; o < ns; o += 64) if (S.s_chain[o] == c && !S.s_extra[o]) {
						const long long rb = S.s_rbeg[o]; const int qb = S.s_qbeg[o], ln = S.s_len[o];
						const long long b = rb - (qb + rg_cal_max_gap(P, qb));
						const long long e = rb + ln + ((l_query - qb - ln) + rg_cal_max_gap(P, l_query - qb - ln));
						rmax0 = rmax0 < b ? rmax0 : b; rmax1 = rmax1 > e ? rmax1 : e;
					}
					rmax0 = uni64(-wave_max_i64(-rmax0)); rmax1 = uni64(wave_max_i64(rmax1));
					rmax0 = rmax0 > 0 ? rmax0 : 0; rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
					if (rmax0 < l_pac && l_pac < rmax1) { if (ch_pos < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
					int rid;
					{
						const long long mid = ch_pos;
						const int is_rev = mid >= l_pac;
						rid = uni(rg_pos2rid(ix, rg_depos(l_pac, mid)));
						long long far_beg = uni64(ix.ctg_off[rid]), far_end = uni64(ix.ctg_off[rid + 1]);
						if (is_rev) { const long long tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
						rmax0 = rmax0 > far_beg ? rmax0 : far_beg; rmax1 = rmax1 < far_end ? rmax1 : far_end;
					}
					const int n0 = uni(S.n_regs);
					for (int pass = 0; pass < 2 && status == 0; ++pass) {
						if (pass == 1 && !(uni(S.n_regs) == n0 && ch_has_extra)) break;
						// the list (seeds or seeds_extra) in arrival order, and its best-first order
						int nl = 0;
						for (int base = 0; base < ns; base += 64) {
							const int o = base + lane;
							const bool in = o < ns && S.s_chain[o] == c && (int)S.s_extra[o] == pass;
							const unsigned long long b = __ballot(in);
							if (in) S.lst[nl + __popcll(b & ((1ull << lane) - 1))] = (unsigned char)o;
							nl += __popcll(b);
						}
						WAVE_SYNC();
						for (int i = lane; i < nl; i += 64) { // keys score<<32|i are unique: rank by counting
							const unsigned long long key = (unsigned long long)(unsigned)S.s_len[S.lst[i]] << 32 | (unsigned)i;
							int r = 0;
							for (int k = 0; k < nl; ++k) r += ((unsigned long long)(unsigned)S.s_len[S.lst[k]] << 32 | (unsigned)k) < key;
							S.srt[r] = key;
						}
						WAVE_SYNC();
						for (int k = nl - 1; k >= 0 && status == 0; --k) {
							const int si = uni((int)(uint32_t)S.srt[k]);
							const int o = uni(S.lst[si]);
							const long long s_rbeg = uni64(S.s_rbeg[o]); const int s_qbeg = uni(S.s_qbeg[o]), s_len = uni(S.s_len[o]);
							// asymmetric_flt_seed (memchain.c:138-149)
							int bad = 0;
							for (int base = 0; base < s_len; base += 64) {
								const int i = base + lane; int v = 0;
								if (i < s_len) { const int r = dev_ref_base(ix.pac, l_pac, s_rbeg + i), q = query[s_qbeg + i]; v = (r == 3 && q == 1) || (r == 0 && q == 2); }
								if (__ballot(v)) bad = 1;
							}
							if (bad) continue;
							// contained in a region of this strand search? (memchain.c:761-819)
							int u;
							const int nr = uni(S.n_regs);
							for (u = 0; u < nr; ++u) {
								const bsx_region_t &rg = S.regs[u];
								if (s_rbeg < rg.rb || s_rbeg + s_len > rg.re || s_qbeg < rg.qb || s_qbeg + s_len > rg.qe) continue;
								if (s_len - rg.seedlen0 > .1 * l_query) continue;
								int qd = s_qbeg - rg.qb; long long rd = s_rbeg - rg.rb;
								int max_gap = rg_cal_max_gap(P, (int)(qd < rd ? qd : rd));
								int w = max_gap < rg.w ? max_gap : rg.w;
								if (qd - rd < w && rd - qd < w) break;
								qd = rg.qe - (s_qbeg + s_len); rd = rg.re - (s_rbeg + s_len);
								max_gap = rg_cal_max_gap(P, (int)(qd < rd ? qd : rd));
								w = max_gap < rg.w ? max_gap : rg.w;
								if (qd - rd < w && rd - qd < w) break;
							}
							if (u < nr) {
								int i;
								for (i = k + 1; i < nl; ++i) {
									if (S.srt[i] == 0) continue;
									const int oo = S.lst[(int)(uint32_t)S.srt[i]];
									const long long t_rbeg = S.s_rbeg[oo]; const int t_qbeg = S.s_qbeg[oo], t_len = S.s_len[oo];
									if (t_len < s_len * .95) continue;
									if (s_qbeg <= t_qbeg && s_qbeg + s_len - t_qbeg >= s_len >> 2 && t_qbeg - s_qbeg != t_rbeg - s_rbeg) break;
									if (t_qbeg <= s_qbeg && t_qbeg + t_len - s_qbeg >= s_len >> 2 && s_qbeg - t_qbeg != s_rbeg - t_rbeg) break;
								}
								if (i == nl) { WAVE_SYNC(); if (lane == 0) S.srt[k] = 0; WAVE_SYNC(); continue; }
							}
							// extension (memchain.c:613-730): left then right, each with up to MAX_BAND_TRY band widths
							bsx_region_t R; memset(&R, 0, sizeof(R));
							int aw0 = P.w, aw1 = P.w;
							const int qe = s_qbeg + s_len;
							R.score = R.truesc = -1; R.rid = rid;
							for (int side = 0; side < 2; ++side) {
								if (side == 0 && s_qbeg == 0) { R.score = R.truesc = s_len * P.a; R.qb = 0; R.rb = s_rbeg; continue; }
								if (side == 1 && qe == l_query) { R.qe = l_query; R.re = s_rbeg + s_len; continue; }
								const int sc0 = R.score, clip = side ? P.pen_clip3 : P.pen_clip5;
								int aw = P.w;
								bsx_ext_res_t res; res.score = -1; res.qle = res.tle = res.gtle = 0; res.gscore = -1; res.max_off = 0;
								bsx_ext_job_t J;
								J.parent = (uint8_t)parent; J.pad = 0; J.end_bonus = clip;
								if (side == 0) { J.qoff = qoff + (uint32_t)s_qbeg - 1; J.qdir = -1; J.qlen = s_qbeg; J.tpos = s_rbeg - 1; J.tdir = -1; J.tlen = (int)(s_rbeg - rmax0); J.h0 = s_len * P.a; }
								else { J.qoff = qoff + (uint32_t)qe; J.qdir = 1; J.qlen = l_query - qe; J.tpos = s_rbeg + s_len; J.tdir = 1; J.tlen = (int)(rmax1 - (s_rbeg + s_len)); J.h0 = sc0; }
								for (int i = 0; i < 2; ++i) {
									const int prev = R.score;
									aw = P.w << i;
									J.w = aw;
									res = ext_dp<RG_NC>(ix, sc, reads, J, S.H, S.E, S.qb, lane);
									res.score = uni(res.score); res.qle = uni(res.qle); res.tle = uni(res.tle); res.gtle = uni(res.gtle);
									res.gscore = uni(res.gscore); res.max_off = uni(res.max_off);
									R.score = res.score;
									if (R.score == prev || res.max_off < (aw >> 1) + (aw >> 2)) break;
								}
								const int local = res.gscore <= 0 || res.gscore <= R.score - clip;
								if (side == 0) {
									aw0 = aw;
									if (local) { R.qb = s_qbeg - res.qle; R.rb = s_rbeg - res.tle; R.truesc = R.score; }
									else { R.qb = 0; R.rb = s_rbeg - res.gtle; R.truesc = res.gscore; }
								} else {
									aw1 = aw;
									if (local) { R.qe = qe + res.qle; R.re = s_rbeg + s_len + res.tle; R.truesc += R.score - sc0; }
									else { R.qe = l_query; R.re = s_rbeg + s_len + res.gtle; R.truesc += res.gscore - sc0; }
								}
							}
							R.bss = (uint8_t)RG_BSS(parent, l_pac, R.rb); R.parent = (uint8_t)parent;
							if (RG_BSS(parent, l_pac, R.re) != R.bss) continue;   // crosses the strand boundary (memchain.c:846-849)
							int cov = 0;
							for (int i = lane; i < nl; i += 64) {
								const int oo = S.lst[i];
								const long long t_rbeg = S.s_rbeg[oo]; const int t_qbeg = S.s_qbeg[oo], t_len = S.s_len[oo];
								if (t_qbeg >= R.qb && t_qbeg + t_len <= R.qe && t_rbeg >= R.rb && t_rbeg + t_len <= R.re) cov += t_len;
							}
							R.seedcov = uni(wave_sum_i32(cov));
							R.w = aw0 > aw1 ? aw0 : aw1; R.seedlen0 = s_len; R.frac_rep = 0.f;   // no over-represented seed reaches this kernel
							if (uni(S.n_regs) == RG_RCAP) { status = 6; break; }
							WAVE_SYNC();
							if (lane == 0) { S.regs[S.n_regs] = R; ++S.n_regs; }
							WAVE_SYNC();
						}
					}
				}
			}
		}
		// ---- F. publish
		WAVE_SYNC();
		if (lane == 0) {
			const int n = status ? 0 : S.n_regs;
			unsigned long long base = 0;
			if (n > 0) {
				base = atomicAdd(out_cursor, (unsigned long long)n);
				if (base + n <= out_cap) for (int k = 0; k < n; ++k) out[base + k] = S.regs[k];
				else status = 7;
			}
			reg_off[t] = (long long)base;
			reg_n[t] = status ? -status : n;
		}
		WAVE_SYNC();
	}
}

void launch_regions(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                    const bsx_seed_task_t *tasks, int n_tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                    bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                    unsigned int *task_cursor)
{
	hipLaunchKernelGGL(k_regions, dim3(grid), dim3(256), 0, st, ix, sc, P, reads, tasks, n_tasks, seeds_dense, task_off, task_n,
	                   out, out_cap, out_cursor, reg_off, reg_n, task_cursor);
}
