/* oracle/port.c -- TEST INFRASTRUCTURE: CPU restatement ("port") of the device kernels.
 *
 * Plain scalar C restatements of the reference algorithms that the product runs as HIP kernels,
 * behind the same batch seams (bsx_backend_t).  Used ONLY by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker / the timed CPU baseline -- never by the product
 * library, which has no CPU path.  Each function cites the reference lines it follows and is
 * pinned against the real reference functions (oracle/_ref, built from /root/reference) by
 * tests/test_oracle_vs_ref.py:
 *   K1  bwt_occ4/bwt_2occ4/bwt_extend   lib/aln/bwt.c:173-236,278-293
 *   K2  bwt_smem1a/bwt_seed_strategy1   lib/aln/bwt.c:307-396 ; driver mem_collect_intv
 *       lib/aln/memchain.c:50-106 (static there: restated, pinned through its two callees)
 *   K3  bwt_sa/bwt_invPsi/bwt_occ       lib/aln/bwt.c:54-60,87-130
 *   K4  ksw_extend2                     lib/aln/ksw.c:380-479
 *   K5  ksw_align2 (ksw_u8/ksw_i16)     lib/aln/ksw.c:63-365
 *   K6  ksw_global2 + band set-up/retry lib/aln/ksw.c:504-606, bwa.c:314-340,
 *                                       mem_alnreg_format.c:63-77
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <assert.h>
#include "../biscuit_amd/csrc/host/bsx_core.h"

/* Optional: the reference's OWN kernels (oracle/_ref/libbiscuit_ref.so: bwt.c and ksw.c compiled where they lie, SSE2 ksw_u8/i16
 * included) behind the same seams, for timing: the restatements above are plain scalar C and 2-3x slower than the code they restate.
 * oracle_port_use_reference_kernels() switches a context over; results are the same (tests/test_oracle_vs_ref.py pins the
 * restatements against these very functions). */
#include <dlfcn.h>
typedef struct {
	void *lib;
	void *bwt[2];
	void *(*bwt_wrap)(uint64_t, const uint64_t*, uint64_t, uint64_t, uint32_t*, int, uint64_t, uint64_t*);
	void (*bwt_unwrap)(void*);
	void *(*smem_ctx_new)(void);
	void (*smem_ctx_free)(void*);
	int (*smem1a_ctx)(void*, void*, void*, int, const uint8_t*, int, int, const uint64_t**, int*);
	int (*seed_strategy1)(void*, void*, int, const uint8_t*, int, int, int, uint64_t*);
	uint64_t (*sa)(void*, uint64_t);
	void (*extend2)(int, const uint8_t*, int, const uint8_t*, const int8_t*, int, int, int, int, int, int, int, int, int*);
	void (*align2)(int, uint8_t*, int, uint8_t*, const int8_t*, int, int, int, int, int, int*);
	int (*global2)(int, const uint8_t*, int, const uint8_t*, const int8_t*, int, int, int, int, int, int, int*, uint32_t*, int);
} ref_kernels_t;

typedef struct {
	const bsx_index_t *idx;
	bsx_opt_t opt;
	const uint8_t *reads; size_t n_reads;
	int n_threads;
	uint64_t counters[4]; /* occ4 calls, same-block 2occ4 calls, occ calls, sa calls */
	ref_kernels_t R;
} port_ctx_t;

/* ============================== FM index ============================== */

/* symbols 0..upto (inclusive) of one 128-symbol block: per-symbol counts */
static inline void block_count(const uint32_t *sym, int upto, uint64_t cnt[4])
{
	int w, last = upto >> 4;
	for (w = 0; w <= last; ++w) {
		uint32_t x = sym[w], lo, hi, valid = 0x55555555u;
		int nv = 16, t, g, c;
		if (w == last) { /* keep only the top (upto&15)+1 symbols */
			nv = (upto & 15) + 1;
			valid = nv == 16 ? 0x55555555u : (0x55555555u & ~((1u << ((16 - nv) << 1)) - 1));
		}
		lo = x & valid; hi = (x >> 1) & valid;
		t = __builtin_popcount(hi & lo);
		g = __builtin_popcount(hi & ~lo);
		c = __builtin_popcount(~hi & lo);
		cnt[3] += t; cnt[2] += g; cnt[1] += c; cnt[0] += nv - t - g - c;
	}
}

/* bwt_occ4: number of each symbol in B[0..k] (k in "with-$" coordinates), bwt.c:173-200 */
static void fm_occ4(const bsx_fmi_t *f, uint64_t k, uint64_t cnt[4])
{
	const uint32_t *p;
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= f->primary);
	p = f->bwt + ((k >> 7) << 4);
	memcpy(cnt, p, 32);
	block_count(p + 8, (int)(k & 127), cnt);
}

/* bwt_2occ4, bwt.c:204-236: same values as two bwt_occ4; reports which path the reference takes */
static int fm_2occ4(const bsx_fmi_t *f, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	uint64_t k_ = k - (k >= f->primary), l_ = l - (l >= f->primary);
	fm_occ4(f, k, ck); fm_occ4(f, l, cl);
	return !(l_ >> 7 != k_ >> 7 || k == (uint64_t)-1 || l == (uint64_t)-1); /* 1 = one-block fast path */
}

/* bwt_extend, bwt.c:278-293 */
static void fm_extend(const bsx_fmi_t *f, const bsx_intv_t *ik, bsx_intv_t ok[4], int is_back, uint64_t *ctr)
{
	uint64_t tk[4], tl[4];
	int i, fast;
	fast = fm_2occ4(f, ik->x[!is_back] - 1, ik->x[!is_back] - 1 + ik->x[2], tk, tl);
	if (ctr) { if (fast) ++ctr[1]; else ctr[0] += 2; }
	for (i = 0; i != 4; ++i) {
		ok[i].x[!is_back] = f->L2[i] + 1 + tk[i];
		ok[i].x[2] = tl[i] - tk[i];
	}
	ok[3].x[is_back] = ik->x[is_back] + (ik->x[!is_back] <= f->primary && ik->x[!is_back] + ik->x[2] - 1 >= f->primary);
	ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
	ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
	ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

/* bwt_set_intv, bwt.h:105 */
static inline void fm_set_intv(const bsx_fmi_t *f, const bsx_fmi_t *fc, int c, bsx_intv_t *ik)
{
	ik->x[0] = f->L2[c] + 1; ik->x[2] = f->L2[c + 1] - f->L2[c]; ik->x[1] = fc->L2[3 - c] + 1; ik->info = 0;
}

typedef BSX_VEC(bsx_intv_t) intv_v;

static void intv_reverse(intv_v *v)
{
	size_t i;
	for (i = 0; i < v->n >> 1; ++i) { bsx_intv_t t = v->a[i]; v->a[i] = v->a[v->n - 1 - i]; v->a[v->n - 1 - i] = t; }
}

/* bwt_smem1a with max_intv == 0 (the only value bwt_smem1 passes), bwt.c:307-374 */
static int fm_smem1(const bsx_fmi_t *f, const bsx_fmi_t *fc, int len, const uint8_t *q, int x, int min_intv,
                    intv_v *mem, intv_v tmp[2], uint64_t *ctr)
{
	int i, c, ret;
	size_t j;
	bsx_intv_t ik, ok[4];
	intv_v *prev = &tmp[0], *curr = &tmp[1], *swap;

	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	fm_set_intv(f, fc, q[x], &ik);
	ik.info = x + 1;
	for (i = x + 1, curr->n = 0; i < len; ++i) { /* forward: through the complementary index */
		if (q[i] < 4) {
			c = 3 - q[i];
			fm_extend(fc, &ik, ok, 0, ctr);
			if (ok[c].x[2] != ik.x[2]) {
				bsx_vec_push(*curr, ik);
				if (ok[c].x[2] < (uint64_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else { bsx_vec_push(*curr, ik); break; }
	}
	if (i == len) bsx_vec_push(*curr, ik);
	intv_reverse(curr);
	ret = (int)curr->a[0].info;
	swap = curr; curr = prev; prev = swap;
	for (i = x - 1; i >= -1; --i) { /* backward: through the own index */
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < prev->n; ++j) {
			bsx_intv_t *p = &prev->a[j];
			if (c >= 0) fm_extend(f, p, ok, 1, ctr);
			if (c < 0 || ok[c].x[2] < (uint64_t)min_intv) {
				if (curr->n == 0) {
					if (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n - 1].info >> 32) {
						ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
						bsx_vec_push(*mem, ik);
					}
				}
			} else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2]) {
				ok[c].info = p->info;
				bsx_vec_push(*curr, ok[c]);
			}
		}
		if (curr->n == 0) break;
		swap = curr; curr = prev; prev = swap;
	}
	intv_reverse(mem);
	return ret;
}

/* bwt_seed_strategy1, bwt.c:376-396 */
static int fm_seed_strategy1(const bsx_fmi_t *f, const bsx_fmi_t *fc, int len, const uint8_t *q, int x, int min_len, int max_intv,
                             bsx_intv_t *mem, uint64_t *ctr)
{
	int i, c;
	bsx_intv_t ik, ok[4];
	memset(mem, 0, sizeof(*mem));
	if (q[x] > 3) return x + 1;
	fm_set_intv(f, fc, q[x], &ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			fm_extend(fc, &ik, ok, 0, ctr);
			if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) {
				*mem = ok[c];
				mem->info = (uint64_t)x << 32 | (uint64_t)(i + 1);
				return i + 1;
			}
			ik = ok[c];
		} else return i + 1;
	}
	return len;
}

static int intv_lt(const void *a, const void *b) { return ((const bsx_intv_t*)a)->info < ((const bsx_intv_t*)b)->info; }

/* mem_collect_intv (memchain.c:50-106) over the reference's own bwt_smem1a / bwt_seed_strategy1 */
static void ref_collect_intv(const ref_kernels_t *R, void *sctx, const bsx_opt_t *opt, int parent, int len, const uint8_t *seq, intv_v *mem)
{
	int k, x = 0, old_n, n, j;
	const uint64_t *v;
	void *b = R->bwt[parent], *bc = R->bwt[!parent];
	int start_width = (opt->flag & BSX_F_SELF_OVLP) ? 2 : 1;
	int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	mem->n = 0;
#define REF_TAKE() do { for (j = 0; j < n; ++j) if ((uint32_t)v[4 * j + 3] - (v[4 * j + 3] >> 32) >= (uint64_t)opt->min_seed_len) { bsx_intv_t t_; memcpy(&t_, v + 4 * j, 32); bsx_vec_push(*mem, t_); } } while (0)
	while (x < len) {
		if (seq[x] < 4) { x = R->smem1a_ctx(sctx, b, bc, len, seq, x, start_width, &v, &n); REF_TAKE(); }
		else ++x;
	}
	old_n = (int)mem->n;
	for (k = 0; k < old_n; ++k) {
		bsx_intv_t p = mem->a[k];
		int start = (int)(p.info >> 32), end = (int32_t)p.info;
		if (end - start < split_len || p.x[2] > (uint64_t)opt->split_width) continue;
		R->smem1a_ctx(sctx, b, bc, len, seq, (start + end) >> 1, (int)(p.x[2] + 1), &v, &n); REF_TAKE();
	}
	if (opt->max_mem_intv > 0) {
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				bsx_intv_t m;
				x = R->seed_strategy1(b, bc, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, (uint64_t*)&m);
				if (m.x[2] > 0) bsx_vec_push(*mem, m);
			} else ++x;
		}
	}
#undef REF_TAKE
	bsx_introsort(mem->a, mem->n, sizeof(bsx_intv_t), intv_lt);
}

/* mem_collect_intv, memchain.c:50-106 */
static void fm_collect_intv(const bsx_opt_t *opt, const bsx_fmi_t *f, const bsx_fmi_t *fc, int len, const uint8_t *seq,
                            intv_v *mem, intv_v *mem1, intv_v tmp[2], uint64_t *ctr)
{
	int k, x = 0, old_n;
	size_t i;
	int start_width = (opt->flag & BSX_F_SELF_OVLP) ? 2 : 1;
	int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	mem->n = 0;
	while (x < len) { /* pass 1: all SMEMs */
		if (seq[x] < 4) {
			x = fm_smem1(f, fc, len, seq, x, start_width, mem1, tmp, ctr);
			for (i = 0; i < mem1->n; ++i)
				if ((uint32_t)mem1->a[i].info - (mem1->a[i].info >> 32) >= (uint64_t)opt->min_seed_len) bsx_vec_push(*mem, mem1->a[i]);
		} else ++x;
	}
	old_n = (int)mem->n; /* pass 2: re-seed inside long, rare SMEMs */
	for (k = 0; k < old_n; ++k) {
		bsx_intv_t p = mem->a[k];
		int start = (int)(p.info >> 32), end = (int32_t)p.info;
		if (end - start < split_len || p.x[2] > (uint64_t)opt->split_width) continue;
		fm_smem1(f, fc, len, seq, (start + end) >> 1, (int)(p.x[2] + 1), mem1, tmp, ctr);
		for (i = 0; i < mem1->n; ++i)
			if ((uint32_t)mem1->a[i].info - (mem1->a[i].info >> 32) >= (uint64_t)opt->min_seed_len) bsx_vec_push(*mem, mem1->a[i]);
	}
	if (opt->max_mem_intv > 0) { /* pass 3: LAST-like */
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				bsx_intv_t m;
				x = fm_seed_strategy1(f, fc, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, &m, ctr);
				if (m.x[2] > 0) bsx_vec_push(*mem, m);
			} else ++x;
		}
	}
	/* ks_introsort(mem_intv): equal keys (same read span) are identical records, any sort will do */
	bsx_introsort(mem->a, mem->n, sizeof(bsx_intv_t), intv_lt);
}

/* bwt_occ: symbol c in B[0..k], bwt.c:108-130 */
static uint64_t fm_occ(const bsx_fmi_t *f, uint64_t k, int c)
{
	uint64_t cnt[4];
	if (k == f->seq_len) return f->L2[c + 1] - f->L2[c];
	if (k == (uint64_t)-1) return 0;
	fm_occ4(f, k, cnt);
	return cnt[c];
}
/* bwt_invPsi + bwt_sa, bwt.c:54-60,87-97 */
static uint64_t fm_sa(const bsx_fmi_t *f, uint64_t k, uint64_t *ctr)
{
	uint64_t sa = 0, mask = f->sa_intv - 1;
	while (k & mask) {
		uint64_t x = k - (k > f->primary);
		int c = f->bwt[((x >> 7) << 4) + 8 + ((x & 0x7f) >> 4)] >> ((~x & 0xf) << 1) & 3;
		++sa;
		if (ctr) ++ctr[2];
		k = k == f->primary ? 0 : f->L2[c] + fm_occ(f, k, c);
	}
	if (ctr) ++ctr[3];
	return sa + f->sa[k / f->sa_intv];
}

/* ============================== views ============================== */
static inline void read_view(const uint8_t *buf, uint32_t qoff, int qlen, int dir, int comp, uint8_t *out)
{
	int i;
	for (i = 0; i < qlen; ++i) {
		int b = buf[(int64_t)qoff + (int64_t)i * dir];
		out[i] = (uint8_t)(comp ? (b < 4 ? 3 - b : 4) : b);
	}
}
static inline void ref_view(const bsx_index_t *idx, int64_t tpos, int tlen, int dir, uint8_t *out)
{
	int i;
	for (i = 0; i < tlen; ++i) out[i] = (uint8_t)bsx_ref_base(idx->ref.l_pac, idx->pac, tpos + (int64_t)i * dir);
}

/* ============================== K4: ksw_extend2 ============================== */
static int dp_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                     int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0, bsx_ext_res_t *r)
{
	int32_t *H = (int32_t*)calloc(qlen + 1, 4), *E = (int32_t*)calloc(qlen + 1, 4);
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
	/* first row (ksw.c:395-397) */
	H[0] = h0; H[1] = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && H[j - 1] > e_ins; ++j) H[j] = H[j - 1] - e_ins;
	/* band clamp (ksw.c:399-407) */
	for (i = 0, max = 0, k = 25; i < k; ++i) max = max > mat[i] ? max : mat[i];
	max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
	max_ins = max_ins > 1 ? max_ins : 1;
	w = w < max_ins ? w : max_ins;
	max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
	max_del = max_del > 1 ? max_del : 1;
	w = w < max_del ? w : max_del;
	max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
	beg = 0; end = qlen;
	for (i = 0; i < tlen; ++i) {
		int f = 0, h1, m = 0, mj = -1;
		const int8_t *srow = mat + target[i] * 5;
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
		for (j = beg; j < end; ++j) {
			/* H[j] holds H(i-1,j-1), E[j] holds E(i,j); gaps open from the diagonal term M only */
			int h, M = H[j], e = E[j], t;
			H[j] = h1;
			M = M ? M + srow[query[j]] : 0;
			h = M > e ? M : e;
			h = h > f ? h : f;
			h1 = h;
			mj = m > h ? mj : j;
			m = m > h ? m : h;
			t = M - oe_del; t = t > 0 ? t : 0;
			e -= e_del; e = e > t ? e : t;
			E[j] = e;
			t = M - oe_ins; t = t > 0 ? t : 0;
			f -= e_ins; f = f > t ? f : t;
		}
		H[end] = h1; E[end] = 0;
		if (j == qlen) { max_ie = gscore > h1 ? max_ie : i; gscore = gscore > h1 ? gscore : h1; }
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			max_off = max_off > abs(mj - i) ? max_off : abs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j);
		beg = j;
		for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j);
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	free(H); free(E);
	r->score = max; r->qle = max_j + 1; r->tle = max_i + 1; r->gtle = max_ie + 1; r->gscore = gscore; r->max_off = max_off;
	return max;
}

/* ============================== K5: ksw_align2 ==============================
 * The reference runs Farrar's striped SW on 16 (u8) or 8 (i16) SSE lanes.  Its result equals a
 * row-by-row DP over the query padded with zero-scoring columns to slen*p, with two properties
 * that a plain SW does not have and that are kept here:
 *  - E(i+1,j) is opened from the H value *before* the cross-stripe lazy-F correction, i.e. from
 *    max(diag, E, F restricted to the stripe [l*slen,(l+1)*slen) that contains j)  (ksw.c:160-169);
 *  - u8 scores saturate at 255-shift and the scan stops there (ksw.c:205,209).
 */
typedef struct { int score, te, qe, score2, te2; } sw1_t;

static sw1_t dp_sw_pass(int is_u8, int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                        int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	int p = is_u8 ? 16 : 8, slen = (qlen + p - 1) / p, Q = slen * p;
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int minsc = (xtra & BSX_KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	int endsc = (xtra & BSX_KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	int i, j, a, shift = 127, mx = 0, gmax = 0, te = -1, n_b = 0, m_b = 0;
	int32_t *H0 = (int32_t*)calloc(Q + 1, 4), *H1 = (int32_t*)calloc(Q + 1, 4), *E = (int32_t*)calloc(Q + 1, 4), *Hmax = (int32_t*)calloc(Q + 1, 4);
	uint64_t *b = 0;
	sw1_t r;
	r.score = 0; r.te = -1; r.qe = -1; r.score2 = -1; r.te2 = -1;
	for (a = 0; a < 25; ++a) { if (mat[a] < shift) shift = mat[a]; if (mat[a] > mx) mx = mat[a]; }
	shift = (uint8_t)(256 - (uint8_t)shift); /* ksw.c:84-88 */
	for (i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * 5;
		int f = 0, imax = 0, fseg = 0;
		int32_t *S;
		for (j = 0; j < Q; ++j) {
			int s = j < qlen ? srow[query[j]] : 0, h, hpre, e = E[j], t;
			int diag = j ? H0[j - 1] : 0;
			if (j % slen == 0) fseg = 0;
			if (is_u8) { h = diag + s + shift; if (h > 255) h = 255; h -= shift; if (h < 0) h = 0; }
			else { h = diag + s; if (h > 32767) h = 32767; }
			h = h > e ? h : e;
			hpre = h > fseg ? h : fseg; /* what the striped main loop has before lazy-F */
			h = h > f ? h : f;          /* after lazy-F: full prefix scan */
			H1[j] = h;
			imax = imax > h ? imax : h;
			e -= e_del; if (e < 0) e = 0;
			t = hpre - oe_del; if (t < 0) t = 0;
			E[j] = e > t ? e : t;
			t = hpre - oe_ins; if (t < 0) t = 0;
			fseg -= e_ins; if (fseg < 0) fseg = 0;
			fseg = fseg > t ? fseg : t;
			t = h - oe_ins; if (t < 0) t = 0;  /* opening from a lazy-F corrected H never beats extending that F */
			f -= e_ins; if (f < 0) f = 0;
			f = f > t ? f : t;
		}
		if (imax >= minsc) {
			if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
				if (n_b == m_b) { m_b = m_b ? m_b << 1 : 8; b = (uint64_t*)realloc(b, 8 * m_b); }
				b[n_b++] = (uint64_t)imax << 32 | i;
			} else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (uint64_t)imax << 32 | i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H1, sizeof(int32_t) * Q);
			if ((is_u8 && gmax + shift >= 255) || gmax >= endsc) break;
		}
		S = H1; H1 = H0; H0 = S;
	}
	r.score = is_u8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!is_u8 || r.score != 255) {
		int max = -1, low, high;
		for (j = 0; j < Q; ++j) if (Hmax[j] > max) { max = Hmax[j]; r.qe = j; } /* smallest query index among maxima */
		if (b) {
			i = (r.score + mx - 1) / mx;
			low = te - i; high = te + i;
			for (i = 0; i < n_b; ++i) {
				int e = (int32_t)b[i];
				if ((e < low || e > high) && (int)(b[i] >> 32) > r.score2) { r.score2 = (int)(b[i] >> 32); r.te2 = e; }
			}
		}
	}
	free(b); free(H0); free(H1); free(E); free(Hmax);
	return r;
}

static void dp_sw(int qlen, uint8_t *query, int tlen, uint8_t *target, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int xtra, bsx_sw_res_t *out)
{
	int is_u8 = (xtra & BSX_KSW_XBYTE) ? 1 : 0, i;
	sw1_t r = dp_sw_pass(is_u8, qlen, query, tlen, target, mat, o_del, e_del, o_ins, e_ins, xtra), rr;
	out->score = r.score; out->te = r.te; out->qe = r.qe; out->score2 = r.score2; out->te2 = r.te2; out->tb = -1; out->qb = -1;
	if ((xtra & BSX_KSW_XSTART) == 0 || ((xtra & BSX_KSW_XSUBO) && r.score < (xtra & 0xffff))) return;
	/* second pass on the reversed prefixes (ksw.c:357-364); the target keeps its unreversed tail */
	{
		uint8_t *q2 = (uint8_t*)malloc(r.qe + 2), *t2 = (uint8_t*)malloc(tlen + 1);
		for (i = 0; i <= r.qe; ++i) q2[i] = query[r.qe - i];
		memcpy(t2, target, tlen);
		for (i = 0; i <= r.te; ++i) t2[i] = target[r.te - i];
		rr = dp_sw_pass(is_u8, r.qe + 1, q2, tlen, t2, mat, o_del, e_del, o_ins, e_ins, BSX_KSW_XSTOP | r.score);
		free(q2); free(t2);
	}
	if (r.score == rr.score) { out->tb = r.te - rr.te; out->qb = r.qe - rr.qe; }
}

/* ============================== K6: ksw_global2 ============================== */
#define NEG_INF (-0x40000000)

static int dp_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                     int o_del, int e_del, int o_ins, int e_ins, int w, int want_cigar, uint32_t *cigar, int cigar_cap, int *n_cigar_)
{
	int32_t *H = (int32_t*)calloc(qlen + 1, 4), *E = (int32_t*)calloc(qlen + 1, 4);
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
	uint8_t *z = 0;
	n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	if (want_cigar) z = (uint8_t*)malloc((size_t)n_col * tlen + 1);
	H[0] = 0; E[0] = NEG_INF;
	for (j = 1; j <= qlen && j <= w; ++j) { H[j] = -(o_ins + e_ins * j); E[j] = NEG_INF; }
	for (; j <= qlen; ++j) H[j] = E[j] = NEG_INF;
	for (i = 0; i < tlen; ++i) {
		int32_t f = NEG_INF, h1, beg, end, t;
		const int8_t *srow = mat + target[i] * 5;
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		for (j = beg; j < end; ++j) {
			int32_t h, m = H[j], e = E[j];
			uint8_t d;
			H[j] = h1;
			m += srow[query[j]];
			d = m >= e ? 0 : 1;
			h = m >= e ? m : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = m - oe_del;
			e -= e_del;
			d |= e > t ? 1 << 2 : 0;
			e = e > t ? e : t;
			E[j] = e;
			t = m - oe_ins;
			f -= e_ins;
			d |= f > t ? 2 << 4 : 0;
			f = f > t ? f : t;
			if (z) z[(size_t)i * n_col + (j - beg)] = d;
		}
		H[end] = h1; E[end] = NEG_INF;
	}
	score = H[qlen];
	if (want_cigar) { /* traceback (ksw.c:587-604) */
		int n = 0, which = 0;
		BSX_VEC(uint32_t) cg; bsx_vec_init(cg);
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		#define PUSH(op, len) do { if (cg.n == 0 || (uint32_t)(op) != (cg.a[cg.n - 1] & 0xf)) bsx_vec_push(cg, (uint32_t)(len) << 4 | (op)); else cg.a[cg.n - 1] += (uint32_t)(len) << 4; } while (0)
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { PUSH(0, 1); --i; --k; }
			else if (which == 1) { PUSH(2, 1); --i; }
			else { PUSH(1, 1); --k; }
		}
		if (i >= 0) PUSH(2, i + 1);
		if (k >= 0) PUSH(1, k + 1);
		#undef PUSH
		n = (int)cg.n;
		if (n <= cigar_cap) for (i = 0; i < n; ++i) cigar[i] = cg.a[n - 1 - i];
		*n_cigar_ = n <= cigar_cap ? n : -n;
		bsx_vec_free(cg);
	}
	free(H); free(E); free(z);
	return score;
}

/* one K6 job: band set-up of bis_bwa_gen_cigar2 (bwa.c:314-340) inside the retry loop of
 * mem_alnreg_setSAM (mem_alnreg_format.c:63-77) */
static void glb_job(const port_ctx_t *c, const bsx_glb_job_t *jb, bsx_glb_res_t *res, uint32_t *pool)
{
	const bsx_opt_t *o = &c->opt;
	const int8_t *mat = jb->use_ct ? o->ctmat : o->gamat;
	uint8_t *q = (uint8_t*)malloc(jb->qlen + 1), *t = (uint8_t*)malloc(jb->tlen + 1);
	int it, w_ = jb->w0, score = 0, last_sc = -(1 << 30), n_cigar = 0, w_used = 0;
	read_view(c->reads, jb->qoff, jb->qlen, jb->qdir, 0, q);
	ref_view(c->idx, jb->tpos, jb->tlen, jb->tdir, t);
	for (it = 0; it < jb->n_try; ++it, w_ <<= 1, last_sc = score) {
		w_ = w_ < jb->w_max ? w_ : jb->w_max;
		w_used = w_;
		if (jb->qlen == jb->tlen && w_ == 0) { /* no gap: bwa.c:314-322 */
			int i;
			for (i = 0, score = 0; i < jb->qlen; ++i) score += mat[t[i] * 5 + q[i]];
			if (jb->want_cigar) { if (jb->cigar_cap >= 1) { pool[jb->cigar_off] = (uint32_t)jb->qlen << 4; n_cigar = 1; } else n_cigar = -1; }
		} else {
			int w, max_gap, max_ins, max_del, min_w, dl = jb->tlen - jb->qlen;
			if (dl < 0) dl = -dl;
			max_ins = (int)((double)(((jb->qlen + 1) >> 1) * mat[0] - o->o_ins) / o->e_ins + 1.);
			max_del = (int)((double)(((jb->qlen + 1) >> 1) * mat[0] - o->o_del) / o->e_del + 1.);
			max_gap = max_ins > max_del ? max_ins : max_del;
			max_gap = max_gap > 1 ? max_gap : 1;
			w = (max_gap + dl + 1) >> 1;
			w = w < w_ ? w : w_;
			min_w = dl + 3;
			w = w > min_w ? w : min_w;
			if (c->R.lib) {
				score = c->R.global2(jb->qlen, q, jb->tlen, t, mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w, jb->want_cigar, &n_cigar, jb->want_cigar ? pool + jb->cigar_off : 0, (int)jb->cigar_cap);
				if (n_cigar > (int)jb->cigar_cap) n_cigar = -n_cigar;   /* the room it needs (pipeline.c redoes the job) */
			} else
			score = dp_global(jb->qlen, q, jb->tlen, t, mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w,
			                  jb->want_cigar, pool + jb->cigar_off, (int)jb->cigar_cap, &n_cigar);
		}
		if (jb->n_try == 1) break;
		if (score == last_sc) break;
		if (w_ == jb->w_max) break;
		if (score >= jb->truesc - o->a) break;
	}
	res->score = score; res->n_cigar = n_cigar; res->w_used = w_used; res->pad = 0;
	free(q); free(t);
}

/* ============================== batch seams ============================== */
static int port_set_opt(void *ctx, const bsx_opt_t *opt) { ((port_ctx_t*)ctx)->opt = *opt; return BSX_OK; }
static int port_set_reads(void *ctx, const uint8_t *buf, size_t n) { port_ctx_t *c = (port_ctx_t*)ctx; c->reads = buf; c->n_reads = n; return BSX_OK; }

typedef struct {
	port_ctx_t *c; const bsx_opt_t *opt; const bsx_seed_task_t *tasks;
	intv_v *per_task; uint64_t (*ctr)[4];
	struct seed_tls { intv_v mem1, tmp[2]; uint8_t *conv; int m_conv; void *sctx; } *tls;
} seed_par_t;

static void seed_worker(void *data, long i, int tid)
{
	seed_par_t *P = (seed_par_t*)data;
	const bsx_seed_task_t *t = &P->tasks[i];
	struct seed_tls *L = &P->tls[tid];
	const bsx_index_t *idx = P->c->idx;
	const uint8_t *raw = P->c->reads + t->qoff;
	int j;
	if (L->m_conv < t->len + 1) { L->m_conv = t->len * 2 + 16; L->conv = (uint8_t*)realloc(L->conv, L->m_conv); }
	/* bseq_bsconvert, bwamem.c:161-178 */
	for (j = 0; j < t->len; ++j) L->conv[j] = t->parent ? (raw[j] == 1 ? 3 : raw[j]) : (raw[j] == 2 ? 0 : raw[j]);
	bsx_vec_init(P->per_task[i]);
	if (t->len < P->opt->min_seed_len) return; /* mem_chain's early return, memchain.c:279 */
	if (P->c->R.lib) {
		if (!L->sctx) L->sctx = P->c->R.smem_ctx_new();
		ref_collect_intv(&P->c->R, L->sctx, P->opt, t->parent, t->len, L->conv, &P->per_task[i]);
	} else
	fm_collect_intv(P->opt, &idx->fmi[t->parent], &idx->fmi[!t->parent], t->len, L->conv, &P->per_task[i], &L->mem1, L->tmp, P->ctr[tid]);
}

static int port_seed_batch(void *ctx, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
                           bsx_intv_t **out, int64_t *out_cap, int64_t *out_off)
{
	port_ctx_t *c = (port_ctx_t*)ctx;
	seed_par_t P;
	int nt = c->n_threads > 0 ? c->n_threads : 1, t;
	int64_t i, tot = 0;
	P.c = c; P.opt = opt; P.tasks = tasks;
	P.per_task = (intv_v*)calloc(n ? n : 1, sizeof(intv_v));
	P.ctr = (uint64_t(*)[4])calloc(nt, sizeof(uint64_t[4]));
	P.tls = (struct seed_tls*)calloc(nt, sizeof(*P.tls));
	bsx_parallel_for(nt, seed_worker, &P, (long)n);
	for (i = 0; i < n; ++i) { out_off[i] = tot; tot += (int64_t)P.per_task[i].n; }
	out_off[n] = tot;
	if (*out_cap < tot) { *out_cap = tot + (tot >> 2) + 16; *out = (bsx_intv_t*)realloc(*out, sizeof(bsx_intv_t) * *out_cap); }
	for (i = 0; i < n; ++i) {
		if (P.per_task[i].n) memcpy(*out + out_off[i], P.per_task[i].a, sizeof(bsx_intv_t) * P.per_task[i].n);
		bsx_vec_free(P.per_task[i]);
	}
	for (t = 0; t < nt; ++t) {
		int k; for (k = 0; k < 4; ++k) c->counters[k] += P.ctr[t][k];
		bsx_vec_free(P.tls[t].mem1); bsx_vec_free(P.tls[t].tmp[0]); bsx_vec_free(P.tls[t].tmp[1]); free(P.tls[t].conv);
		if (P.tls[t].sctx) c->R.smem_ctx_free(P.tls[t].sctx);
	}
	free(P.per_task); free(P.ctr); free(P.tls);
	return BSX_OK;
}

typedef struct { port_ctx_t *c; const void *jobs; void *res; uint32_t *pool; uint64_t (*ctr)[4]; } job_par_t;

static void sa_worker(void *d, long i, int tid)
{
	job_par_t *P = (job_par_t*)d;
	const bsx_sa_job_t *j = (const bsx_sa_job_t*)P->jobs + i;
	if (P->c->R.lib) ((uint64_t*)P->res)[i] = P->c->R.sa(P->c->R.bwt[j->parent], j->k);
	else
	((uint64_t*)P->res)[i] = fm_sa(&P->c->idx->fmi[j->parent], j->k, P->ctr[tid]);
}
static int port_sa_batch(void *ctx, int64_t n, const bsx_sa_job_t *jobs, uint64_t *pos)
{
	port_ctx_t *c = (port_ctx_t*)ctx;
	int nt = c->n_threads > 0 ? c->n_threads : 1, t, k;
	job_par_t P; P.c = c; P.jobs = jobs; P.res = pos; P.pool = 0;
	P.ctr = (uint64_t(*)[4])calloc(nt, sizeof(uint64_t[4]));
	bsx_parallel_for(nt, sa_worker, &P, (long)n);
	for (t = 0; t < nt; ++t) for (k = 0; k < 4; ++k) c->counters[k] += P.ctr[t][k];
	free(P.ctr);
	return BSX_OK;
}

static void ext_worker(void *d, long i, int tid)
{
	job_par_t *P = (job_par_t*)d;
	const bsx_ext_job_t *j = (const bsx_ext_job_t*)P->jobs + i;
	const bsx_opt_t *o = &P->c->opt;
	uint8_t *q = (uint8_t*)malloc(j->qlen + 1), *t = (uint8_t*)malloc(j->tlen + 1);
	(void)tid;
	read_view(P->c->reads, j->qoff, j->qlen, j->qdir, 0, q);
	ref_view(P->c->idx, j->tpos, j->tlen, j->tdir, t);
	if (P->c->R.lib) P->c->R.extend2(j->qlen, q, j->tlen, t, j->parent ? o->ctmat : o->gamat, o->o_del, o->e_del, o->o_ins, o->e_ins,
	                                 j->w, j->end_bonus, o->zdrop, j->h0, (int*)((bsx_ext_res_t*)P->res + i));   /* {score,qle,tle,gtle,gscore,max_off} */
	else
	dp_extend(j->qlen, q, j->tlen, t, j->parent ? o->ctmat : o->gamat, o->o_del, o->e_del, o->o_ins, o->e_ins,
	          j->w, j->end_bonus, o->zdrop, j->h0, (bsx_ext_res_t*)P->res + i);
	free(q); free(t);
}
static int port_extend_batch(void *ctx, int64_t n, const bsx_ext_job_t *jobs, bsx_ext_res_t *res)
{
	port_ctx_t *c = (port_ctx_t*)ctx;
	job_par_t P; P.c = c; P.jobs = jobs; P.res = res; P.pool = 0; P.ctr = 0;
	bsx_parallel_for(c->n_threads > 0 ? c->n_threads : 1, ext_worker, &P, (long)n);
	return BSX_OK;
}

static void sw_worker(void *d, long i, int tid)
{
	job_par_t *P = (job_par_t*)d;
	const bsx_sw_job_t *j = (const bsx_sw_job_t*)P->jobs + i;
	const bsx_opt_t *o = &P->c->opt;
	uint8_t *q = (uint8_t*)malloc(j->qlen + 1), *t = (uint8_t*)malloc(j->tlen + 1);
	(void)tid;
	read_view(P->c->reads, j->qoff, j->qlen, j->qdir, j->qcomp, q);
	ref_view(P->c->idx, j->tpos, j->tlen, j->tdir, t);
	if (P->c->R.lib) P->c->R.align2(j->qlen, q, j->tlen, t, j->use_ct ? o->ctmat : o->gamat, o->o_del, o->e_del, o->o_ins, o->e_ins, j->xtra, (int*)((bsx_sw_res_t*)P->res + i));   /* {score,te,qe,score2,te2,tb,qb} */
	else
	dp_sw(j->qlen, q, j->tlen, t, j->use_ct ? o->ctmat : o->gamat, o->o_del, o->e_del, o->o_ins, o->e_ins, j->xtra, (bsx_sw_res_t*)P->res + i);
	free(q); free(t);
}
static int port_sw_batch(void *ctx, int64_t n, const bsx_sw_job_t *jobs, bsx_sw_res_t *res)
{
	port_ctx_t *c = (port_ctx_t*)ctx;
	job_par_t P; P.c = c; P.jobs = jobs; P.res = res; P.pool = 0; P.ctr = 0;
	bsx_parallel_for(c->n_threads > 0 ? c->n_threads : 1, sw_worker, &P, (long)n);
	return BSX_OK;
}

static void glb_worker(void *d, long i, int tid)
{
	job_par_t *P = (job_par_t*)d;
	(void)tid;
	glb_job(P->c, (const bsx_glb_job_t*)P->jobs + i, (bsx_glb_res_t*)P->res + i, P->pool);
}
static int port_global_batch(void *ctx, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res, uint32_t *pool, size_t pool_len)
{
	port_ctx_t *c = (port_ctx_t*)ctx;
	job_par_t P; P.c = c; P.jobs = jobs; P.res = res; P.pool = pool; P.ctr = 0;
	(void)pool_len;
	bsx_parallel_for(c->n_threads > 0 ? c->n_threads : 1, glb_worker, &P, (long)n);
	return BSX_OK;
}

/* ---- public (test-only) entry points ---- */
BSX_API void *oracle_port_new(const bsx_index_t *idx, int n_threads)
{
	port_ctx_t *c = (port_ctx_t*)calloc(1, sizeof(*c));
	c->idx = idx; c->n_threads = n_threads;
	bsx_opt_init(&c->opt);
	return c;
}
BSX_API void oracle_port_free(void *c_)
{
	port_ctx_t *c = (port_ctx_t*)c_;
	if (c && c->R.lib) { c->R.bwt_unwrap(c->R.bwt[0]); c->R.bwt_unwrap(c->R.bwt[1]); dlclose(c->R.lib); }
	free(c);
}
/* the batch seams of this context over the reference's own kernels (oracle/_ref/libbiscuit_ref.so); 0 on success */
BSX_API int oracle_port_use_reference_kernels(void *c_, const char *so_path)
{
	port_ctx_t *c = (port_ctx_t*)c_;
	ref_kernels_t R;
	int i;
	memset(&R, 0, sizeof(R));
	if ((R.lib = dlopen(so_path, RTLD_NOW | RTLD_LOCAL)) == 0) return -1;
#define REF_SYM(field, name) do { *(void**)&R.field = dlsym(R.lib, name); if (!R.field) { dlclose(R.lib); return -2; } } while (0)
	REF_SYM(bwt_wrap, "ref_bwt_wrap"); REF_SYM(bwt_unwrap, "ref_bwt_unwrap"); REF_SYM(smem_ctx_new, "ref_smem_ctx_new"); REF_SYM(smem_ctx_free, "ref_smem_ctx_free");
	REF_SYM(smem1a_ctx, "ref_bwt_smem1a_ctx"); REF_SYM(seed_strategy1, "ref_bwt_seed_strategy1"); REF_SYM(sa, "ref_bwt_sa");
	REF_SYM(extend2, "ref_ksw_extend2"); REF_SYM(align2, "ref_ksw_align2"); REF_SYM(global2, "ref_ksw_global2");
#undef REF_SYM
	for (i = 0; i < 2; ++i) {
		const bsx_fmi_t *f = &c->idx->fmi[i];
		if (!f->bwt || !f->sa) { dlclose(R.lib); return -3; }
		R.bwt[i] = R.bwt_wrap(f->primary, f->L2, f->seq_len, f->bwt_size, f->bwt, f->sa_intv, f->n_sa, f->sa);
	}
	c->R = R;
	return 0;
}
BSX_API void oracle_port_backend(void *c, bsx_backend_t *be)
{
	memset(be, 0, sizeof(*be));   /* no regions_batch: the CPU checker always runs the host chaining path */
	be->ctx = c; be->name = "oracle-port-cpu";
	be->set_opt = port_set_opt; be->set_reads = port_set_reads;
	be->seed_batch = port_seed_batch; be->sa_batch = port_sa_batch; be->extend_batch = port_extend_batch;
	be->sw_batch = port_sw_batch; be->global_batch = port_global_batch;
}
BSX_API void oracle_port_counters(void *c_, uint64_t out[4], int reset)
{
	port_ctx_t *c = (port_ctx_t*)c_;
	memcpy(out, c->counters, 32);
	if (reset) memset(c->counters, 0, 32);
}
/* direct, ctypes-friendly forms of the batch seams (same semantics as include/bsx.h) */
BSX_API int oracle_port_set_opt(void *c, const bsx_opt_t *o) { return port_set_opt(c, o); }
BSX_API int oracle_port_set_reads(void *c, const uint8_t *b, size_t n) { return port_set_reads(c, b, n); }
BSX_API int oracle_port_seed_batch(void *c, const bsx_opt_t *o, int64_t n, const bsx_seed_task_t *t, bsx_intv_t **out, int64_t *cap, int64_t *off)
{ return port_seed_batch(c, o, n, t, out, cap, off); }
BSX_API int oracle_port_sa_batch(void *c, int64_t n, const bsx_sa_job_t *j, uint64_t *pos) { return port_sa_batch(c, n, j, pos); }
BSX_API int oracle_port_extend_batch(void *c, int64_t n, const bsx_ext_job_t *j, bsx_ext_res_t *r) { return port_extend_batch(c, n, j, r); }
BSX_API int oracle_port_sw_batch(void *c, int64_t n, const bsx_sw_job_t *j, bsx_sw_res_t *r) { return port_sw_batch(c, n, j, r); }
BSX_API int oracle_port_global_batch(void *c, int64_t n, const bsx_glb_job_t *j, bsx_glb_res_t *r, uint32_t *pool, size_t len)
{ return port_global_batch(c, n, j, r, pool, len); }

/* single-call forms on explicit arrays, for pinning against oracle/_ref */
BSX_API void oracle_extend1(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                            int w, int end_bonus, int zdrop, int h0, int out[6])
{
	bsx_ext_res_t r;
	dp_extend(qlen, q, tlen, t, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, &r);
	out[0] = r.score; out[1] = r.qle; out[2] = r.tle; out[3] = r.gtle; out[4] = r.gscore; out[5] = r.max_off;
}
BSX_API void oracle_sw1(int qlen, uint8_t *q, int tlen, uint8_t *t, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, int out[7])
{
	bsx_sw_res_t r;
	dp_sw(qlen, q, tlen, t, mat, o_del, e_del, o_ins, e_ins, xtra, &r);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}
BSX_API int oracle_global1(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                           int w, int want_cigar, int *n_cigar, uint32_t *cigar, int cap)
{
	int n = 0, sc = dp_global(qlen, q, tlen, t, mat, o_del, e_del, o_ins, e_ins, w, want_cigar, cigar, cap, &n);
	if (n_cigar) *n_cigar = n;
	return sc;
}
BSX_API int oracle_smem1(const bsx_index_t *idx, int parent, int len, const uint8_t *q, int x, int min_intv, uint64_t *out, int cap, int *ret)
{
	intv_v mem, tmp[2]; size_t i; int n;
	bsx_vec_init(mem); bsx_vec_init(tmp[0]); bsx_vec_init(tmp[1]);
	*ret = fm_smem1(&idx->fmi[parent], &idx->fmi[!parent], len, q, x, min_intv, &mem, tmp, 0);
	for (i = 0; i < mem.n && (int)i < cap; ++i) memcpy(out + i * 4, &mem.a[i], 32);
	n = (int)mem.n;
	bsx_vec_free(mem); bsx_vec_free(tmp[0]); bsx_vec_free(tmp[1]);
	return n;
}
BSX_API int oracle_seed_strategy1(const bsx_index_t *idx, int parent, int len, const uint8_t *q, int x, int min_len, int max_intv, uint64_t out[4])
{
	bsx_intv_t m;
	int r = fm_seed_strategy1(&idx->fmi[parent], &idx->fmi[!parent], len, q, x, min_len, max_intv, &m, 0);
	memcpy(out, &m, 32);
	return r;
}
BSX_API void oracle_occ4(const bsx_index_t *idx, int parent, uint64_t k, uint64_t cnt[4]) { fm_occ4(&idx->fmi[parent], k, cnt); }
BSX_API void oracle_extend_intv(const bsx_index_t *idx, int parent, const uint64_t ik[4], int is_back, uint64_t ok[16])
{
	bsx_intv_t i, o[4]; int c;
	memcpy(&i, ik, 32);
	fm_extend(&idx->fmi[parent], &i, o, is_back, 0);
	for (c = 0; c < 4; ++c) { ok[c * 4] = o[c].x[0]; ok[c * 4 + 1] = o[c].x[1]; ok[c * 4 + 2] = o[c].x[2]; ok[c * 4 + 3] = 0; }
}
BSX_API uint64_t oracle_sa(const bsx_index_t *idx, int parent, uint64_t k) { return fm_sa(&idx->fmi[parent], k, 0); }

/* `biscuit align` on the CPU restatement: the timed CPU baseline and the SAM-level checker */
/* where the reference's kernels live: $ORACLE_REF_KERNELS names the library, or "1" = _ref/libbiscuit_ref.so next to this one */
BSX_API int oracle_port_use_reference_kernels_env(void *c)
{
	const char *e = getenv("ORACLE_REF_KERNELS");
	char path[4096];
	Dl_info di;
	if (!e || !*e || !strcmp(e, "0")) return 1;
	if (strcmp(e, "1")) return oracle_port_use_reference_kernels(c, e);
	if (!dladdr((void*)oracle_port_use_reference_kernels_env, &di) || !di.dli_fname) return -1;
	snprintf(path, sizeof(path), "%s", di.dli_fname);
	{ char *sl = strrchr(path, '/'); if (!sl) return -1; snprintf(sl + 1, sizeof(path) - (size_t)(sl + 1 - path), "_ref/libbiscuit_ref.so"); }
	return oracle_port_use_reference_kernels(c, path);
}
static int port_open(int ordinal, const bsx_index_t *idx, void **ud)
{
	(void)ordinal;
	*ud = oracle_port_new(idx, 1);
	if (oracle_port_use_reference_kernels_env(*ud) < 0) { fprintf(stderr, "[oracle_align] $ORACLE_REF_KERNELS is set but oracle/_ref/libbiscuit_ref.so could not be used\n"); return BSX_E_ARG; }
	return BSX_OK;
}
static int port_process(void *ud, const bsx_opt_t *opt, const bsx_index_t *idx, int64_t np, int n, bsx_read_t *reads, const bsx_pestat_t *pes0)
{
	bsx_backend_t be;
	((port_ctx_t*)ud)->n_threads = bsx_host_threads(opt);
	oracle_port_backend(ud, &be);
	return bsx_process_seqs_backend(&be, opt, idx, np, n, reads, pes0);
}
BSX_API int oracle_align_main(int argc, char **argv) { return bsx_align_main_ranks_with(argc, argv, port_process, 0, port_open, 0); }   /* (several processes: over sockets) */
