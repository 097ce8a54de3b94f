// parsort.hpp -- klib's introsort (ksort.h:184-236) run by a whole wavefront over packed 32-bit keys (weight << RG_KEY_BITS | index), larger
// weights first, reproducing the reference's order of EQUAL keys: every partition of a segment made at once from the stop masks of the
// original segment, klib's stack discipline and depth count, its comb-sort fallback, the closing insertion pass as a rank by counting.
// Used by the chain filter (k_regions.hip: chains by weight) and by the de-duplication of long region lists (k_dedup.hip: regions by the
// dense rank of their key, for the lists that have tied keys).  Checked against the sequential algorithm in tools/dbg/parsort_model.py and
// tests/test_parsort_model.py.
#pragma once
#include "wave.hpp"
#define RG_KEY_BITS 13   // chain indices < 8192 (RgHuge::CCAP)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The same sort by the whole wavefront with every partition made AT ONCE (round 4).  klib's partition loop (ksort.h:212-217) is Hoare's: i stops at the
// positions whose weight is <= the pivot's ("L stops", the pivot at a[t] among them), j at those in (s, t) whose weight is >= it ("R
// stops"), and round k swaps the k-th L stop from the left with the k-th R stop from the right while the first lies below the second -- the
// swaps never touch what later rounds scan, so which positions swap, and with whom, follows from the two stop masks of the ORIGINAL segment:
// L stop number k (from the left) takes part iff at least k R stops lie above it, R stop number k (from the right) iff at least k L stops lie
// below it, partners have equal numbers; the loop ends at the first L stop that takes no part or at the lowest R stop that does, whichever
// comes first.  A lane per position (segment-relative: only a segment longer than 64 spans register slots), partners meet through two
// lists in LDS.  The closing insertion pass (ksort.h:229) never carries an element across a pivot, and is stable: it is the order by
// (weight descending, position ascending) of the array as the partitions leave it -- a rank by counting.  Same stack discipline and depth
// count as klib, so the comb-sort case (pre-sorted input) meets the array klib would have: one lane runs it on that segment.  Checked
// against the sequential algorithm on random keys with ties in tools/dbg/parsort_model.py.  For n <= 256.
static __device__ void rg_combsort_keys(unsigned int *a, int m)   // ks_combsort (ksort.h:162-183) over a[0, m), by one lane
{
#define LT(x, y) (((x) >> RG_KEY_BITS) > ((y) >> RG_KEY_BITS))
#define SWP(i, j) do { const unsigned int t_ = a[i]; a[i] = a[j]; a[j] = t_; } while (0)
	const double shrink = 1.2473309501039786540366528676643;
	int gap = m, swapped, i, j;
	do {
		if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = 0;
		for (i = 0; i + gap < m; ++i) if (LT(a[i + gap], a[i])) { SWP(i, i + gap); swapped = 1; }
	} while (swapped || gap > 2);
	if (gap != 1) for (i = 1; i < m; ++i) for (j = i; j > 0 && LT(a[j], a[j - 1]); --j) SWP(j, j - 1);
#undef LT
#undef SWP
}
template <int NS>
__device__ __forceinline__ int rg_par_partition(unsigned int *a, unsigned int *tmpL, unsigned int *tmpR, int s, int t, unsigned int rp, int lane)
{
	const unsigned int wp = rp >> RG_KEY_BITS;
	const int len = t - s + 1;
	unsigned int x[NS]; unsigned long long Lm[NS], Rm[NS]; int kk[NS];
	int r_after = 0;
#pragma unroll
	for (int c = 0; c < NS; ++c) {
		const int r = c * 64 + lane;
		const bool in = r >= 1 && r < len;
		x[c] = in ? a[s + r] : 0u;
		const unsigned int w = x[c] >> RG_KEY_BITS;
		Lm[c] = __ballot(in && w <= wp);
		Rm[c] = __ballot(in && r < len - 1 && w >= wp);
		r_after += __popcll(Rm[c]);
	}
	int l_before = 0, first_free = 0x7fffffff, low_r = 0x7fffffff;
#pragma unroll
	for (int c = 0; c < NS; ++c) {
		const int rc = __popcll(Rm[c]);
		r_after -= rc;                                  // R stops in the slots above this one
		const int is_l = (int)((Lm[c] >> lane) & 1), is_r = (int)((Rm[c] >> lane) & 1);
		const int l_lt = l_before + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(Lm[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)Lm[c], 0u));
		const int r_lt = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(Rm[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)Rm[c], 0u));
		const int k_l = l_lt + 1;                       // my number among the L stops, from the left
		const int r_gt = r_after + rc - r_lt - is_r;    // R stops above me
		const int k_r = r_gt + 1;                       // my number among the R stops, from the right
		const bool pl = is_l && r_gt >= k_l, pr = is_r && l_lt >= k_r;
		if (pl) tmpL[k_l - 1] = x[c];
		if (pr) tmpR[k_r - 1] = x[c];
		kk[c] = pl ? k_l : pr ? -k_r : 0;
		const unsigned long long fm = Lm[c] & ~__ballot(pl), rm = __ballot(pr);
		if (fm && first_free == 0x7fffffff) first_free = c * 64 + (int)__builtin_ctzll(fm);
		if (rm && low_r == 0x7fffffff) low_r = c * 64 + (int)__builtin_ctzll(rm);
		l_before += __popcll(Lm[c]);
	}
	WAVE_SYNC();
#pragma unroll
	for (int c = 0; c < NS; ++c) {
		if (kk[c] > 0) a[s + c * 64 + lane] = tmpR[kk[c] - 1];
		else if (kk[c] < 0) a[s + c * 64 + lane] = tmpL[-kk[c] - 1];
	}
	const int i = s + (first_free < low_r ? first_free : low_r);
	WAVE_SYNC();
	if (i != t && lane == 0) { const unsigned int v = a[i]; a[t] = v; a[i] = rp; }
	WAVE_SYNC();
	return i;
}
// The same partition for a segment of any length (the kilobase-read tiers: a strand search on the strand its read does not come from has
// ~800 chains of weight 19-22), 64 positions at a time: a first sweep counts the stops of every block (cnt[0..31] L stops, cnt[32..63] R
// stops), a second one numbers them (L stops from the left, R stops from the right), decides who takes part and lists the partners, a third
// one puts the partners in place.  kk: which pair a position belongs to, kept in LDS between the sweeps (a short per position).
__device__ __forceinline__ int rg_par_partition_blk(unsigned int *a, unsigned int *tmpL, unsigned int *tmpR, short *kk, int *cnt, int s, int t, unsigned int rp, int lane)
{
	const unsigned int wp = rp >> RG_KEY_BITS;
	const int len = t - s + 1, nb = (len + 63) >> 6;
	int tot_r = 0;
	for (int c = 0; c < nb; ++c) {
		const int r = c * 64 + lane;
		const bool in = r >= 1 && r < len;
		const unsigned int w = (in ? a[s + r] : 0u) >> RG_KEY_BITS;
		const int nl = __popcll(__ballot(in && w <= wp)), nr = __popcll(__ballot(in && r < len - 1 && w >= wp));
		if (lane == 0) { cnt[c] = nl; cnt[32 + c] = nr; }
		tot_r += nr;
	}
	WAVE_SYNC();
	int l_before = 0, r_after = tot_r, first_free = 0x7fffffff, low_r = 0x7fffffff;
	for (int c = 0; c < nb; ++c) {
		const int r = c * 64 + lane;
		const bool in = r >= 1 && r < len;
		const unsigned int x = in ? a[s + r] : 0u, w = x >> RG_KEY_BITS;
		const unsigned long long Lm = __ballot(in && w <= wp), Rm = __ballot(in && r < len - 1 && w >= wp);
		const int rc = uni(cnt[32 + c]);
		r_after -= rc;                                  // R stops in the blocks above this one
		const int is_l = (int)((Lm >> lane) & 1), is_r = (int)((Rm >> lane) & 1);
		const int l_lt = l_before + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(Lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)Lm, 0u));
		const int r_lt = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(Rm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)Rm, 0u));
		const int k_l = l_lt + 1, r_gt = r_after + rc - r_lt - is_r, k_r = r_gt + 1;
		const bool pl = is_l && r_gt >= k_l, pr = is_r && l_lt >= k_r;
		if (pl) tmpL[k_l - 1] = x;
		if (pr) tmpR[k_r - 1] = x;
		if (r < len) kk[r] = (short)(pl ? k_l : pr ? -k_r : 0);
		const unsigned long long fm = Lm & ~__ballot(pl), rm = __ballot(pr);
		if (fm && first_free == 0x7fffffff) first_free = c * 64 + (int)__builtin_ctzll(fm);
		if (rm && low_r == 0x7fffffff) low_r = c * 64 + (int)__builtin_ctzll(rm);
		l_before += uni(cnt[c]);
	}
	WAVE_SYNC();
	for (int c = 0; c < nb; ++c) {
		const int r = c * 64 + lane;
		if (r < len) {
			const int k = kk[r];
			if (k > 0) a[s + r] = tmpR[k - 1];
			else if (k < 0) a[s + r] = tmpL[-k - 1];
		}
	}
	const int i = s + (first_free < low_r ? first_free : low_r);
	WAVE_SYNC();
	if (i != t && lane == 0) { const unsigned int v = a[i]; a[t] = v; a[i] = rp; }
	WAVE_SYNC();
	return i;
}
template <int MS>   // register slots of 64 keys: n <= 64 * MS; MS > 4: kk (n shorts) and cnt (64 ints) are LDS scratch for the partitions of long segments
__device__ __forceinline__ void rg_introsort_par(unsigned int *a, int n, unsigned int *tmpL, unsigned int *tmpR, int *stk, int lane, short *kk = nullptr, int *cnt = nullptr)
{
#define WGT(x) ((x) >> RG_KEY_BITS)
	if (n < 2) return;
	if (n == 2) { if (lane == 0) { const unsigned int a0 = a[0], a1 = a[1]; if (WGT(a1) > WGT(a0)) { a[0] = a1; a[1] = a0; } } WAVE_SYNC(); return; }
	int d, s = 0, t = n - 1, top = 0;
	int *stk_l = stk, *stk_r = stk + 16, *stk_d = stk + 32;
	for (d = 2; (1 << d) < n; ++d);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { if (lane == 0) rg_combsort_keys(a + s, t - s + 1); WAVE_SYNC(); t = s; continue; }
			const int k0 = s + ((t - s) >> 1) + 1;
			const unsigned int ai = (unsigned int)uni((int)a[s]), ak = (unsigned int)uni((int)a[k0]), aj = (unsigned int)uni((int)a[t]);
			int k = k0;
			if (WGT(ak) > WGT(ai)) { if (WGT(ak) > WGT(aj)) k = t; }
			else k = WGT(aj) > WGT(ai) ? s : t;
			const unsigned int rp = k == k0 ? ak : k == s ? ai : aj;
			if (k != t) { WAVE_SYNC(); if (lane == 0) { a[k] = aj; a[t] = rp; } WAVE_SYNC(); }
			const int len = t - s + 1;
			int i;
			if (len <= 64) i = rg_par_partition<1>(a, tmpL, tmpR, s, t, rp, lane);
			else if (MS <= 2 || len <= 128) i = rg_par_partition<2>(a, tmpL, tmpR, s, t, rp, lane);
			else if (MS <= 4 || len <= 256) i = rg_par_partition<MS <= 2 ? 2 : 4>(a, tmpL, tmpR, s, t, rp, lane);
			else i = rg_par_partition_blk(a, tmpL, tmpR, kk, cnt, s, t, rp, lane);
			if (i - s > t - i) {
				if (i - s > 16) { if (lane == 0) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; } ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { if (lane == 0) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; } ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) break;
			--top;
			WAVE_SYNC();
			s = uni(stk_l[top]); t = uni(stk_r[top]); d = uni(stk_d[top]);
		}
	}
	// the insertion pass: every key to its rank by (weight descending, position ascending)
	WAVE_SYNC();
	unsigned int x[MS], u[MS]; int rk[MS];
#pragma unroll
	for (int c = 0; c < MS; ++c) {
		const int p = c * 64 + lane;
		x[c] = (c * 64 < n && p < n) ? a[p] : 0u;
		u[c] = p < n ? (MS <= 4 ? (WGT(x[c]) << 9 | (unsigned int)(256 - p)) : (WGT(x[c]) << 11 | (unsigned int)(2047 - p))) : 0u;   // unique, heavier and earlier = larger; 0: no key
		rk[c] = 0;
	}
#pragma unroll
	for (int cq = 0; cq < MS; ++cq) {
		if (cq * 64 < n) {
			const int lim = n - cq * 64 < 64 ? n - cq * 64 : 64;
			for (int q = 0; q < lim; ++q) {
				const unsigned int uq = (unsigned int)__builtin_amdgcn_readlane((int)u[cq], q);
#pragma unroll
				for (int c = 0; c < MS; ++c) if (c * 64 < n) rk[c] += uq > u[c];
			}
		}
	}
	WAVE_SYNC();
#pragma unroll
	for (int c = 0; c < MS; ++c) if (c * 64 + lane < n) a[rk[c]] = x[c];
	WAVE_SYNC();
#undef WGT
}

