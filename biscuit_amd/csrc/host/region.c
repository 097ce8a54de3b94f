/* region.c -- C5/C6/C7/R1: region de-duplication and merging, primary marking, insert-size
 * statistics, pairing and MAPQ.  All of it is cheap, branchy and order dependent, and the
 * floating-point parts (libm log/erfc/sqrt followed by +.499 truncation) must run on the host's
 * glibc to match the reference bit for bit, so it stays on the CPU.
 */
#include <math.h>
#include <limits.h>
#include "align_types.h"
#include "sort_tmpl.h"

#define PATCH_MAX_R_BW 0.05f       /* mem_alnreg.c:51-52 */
#define PATCH_MIN_SC_RATIO 0.90f

/* ------------------------------------------------------------------ sort / dedup / merge */
static int reg_re_lt(const void *a, const void *b) { return ((const reg_t*)a)->re < ((const reg_t*)b)->re; }          /* alnreg_slt2 */
static int reg_score_lt(const void *a_, const void *b_)                                                           /* alnreg_slt */
{
	const reg_t *a = (const reg_t*)a_, *b = (const reg_t*)b_;
	return a->score > b->score || (a->score == b->score && (a->rb < b->rb || (a->rb == b->rb && a->qb < b->qb)));
}

/* The two sorts of mem_sort_deduplicate on small proxies (the key and the record's index) instead of the 150-byte records: the same
 * introsort over the same keys in the same starting order takes the same decisions, so the records end in the same order -- ties
 * included -- and are moved once instead of three copies per swap.  (Mate rescue re-sorts a read's list after every hit it adds: on a
 * repeat-rich genome that was most of the host's time.) */
typedef struct { int64_t re; int idx; } prox_re_t;
typedef struct { int64_t rb; int score, qb, idx; } prox_sc_t;
#define PROX_RE_LT(a, b) ((a)->re < (b)->re)
#define PROX_SC_LT(a, b) ((a)->score > (b)->score || ((a)->score == (b)->score && ((a)->rb < (b)->rb || ((a)->rb == (b)->rb && (a)->qb < (b)->qb))))
BSX_SORT_DEFINE(sort_prox_re, prox_re_t, PROX_RE_LT)
BSX_SORT_DEFINE(sort_prox_sc, prox_sc_t, PROX_SC_LT)
static void regs_sort(reg_v *regs, int by_score)
{
	size_t n = regs->n, i;
	char stackbuf[4096];
	const size_t w = by_score ? sizeof(prox_sc_t) : sizeof(prox_re_t);
	char *px;
	reg_t *tmp;
	int moved = 0;
	if (n < 2) return;
	if (n < 4) { bsx_introsort(regs->a, n, sizeof(reg_t), by_score ? reg_score_lt : reg_re_lt); return; }
	px = n * w <= sizeof(stackbuf) ? stackbuf : (char*)malloc(n * w);
	if (by_score) { prox_sc_t *q = (prox_sc_t*)px; for (i = 0; i < n; ++i) { q[i].rb = regs->a[i].rb; q[i].score = regs->a[i].score; q[i].qb = regs->a[i].qb; q[i].idx = (int)i; } }
	else { prox_re_t *q = (prox_re_t*)px; for (i = 0; i < n; ++i) { q[i].re = regs->a[i].re; q[i].idx = (int)i; } }
	if (by_score) sort_prox_sc(n, (prox_sc_t*)px); else sort_prox_re(n, (prox_re_t*)px);
	for (i = 0; i < n; ++i) if ((by_score ? ((prox_sc_t*)px)[i].idx : ((prox_re_t*)px)[i].idx) != (int)i) { moved = 1; break; }
	if (moved) {
		tmp = (reg_t*)malloc(sizeof(reg_t) * n);
		for (i = 0; i < n; ++i) tmp[i] = regs->a[by_score ? ((prox_sc_t*)px)[i].idx : ((prox_re_t*)px)[i].idx];
		memcpy(regs->a, tmp, sizeof(reg_t) * n);
		free(tmp);
	}
	if (px != stackbuf) free(px);
}

/* mem_test_reg_concatenation, mem_alnreg.c:63-108 */
static int try_concat(const bsx_opt_t *opt, const bsx_refmeta_t *ref, const reg_t *a, const reg_t *b, int *_w,
                      bsx_glb_score_fn score_fn, void *ud, int *missing)
{
	int w, score, q_s, r_s;
	double r;
	if (a->rb < ref->l_pac && b->rb >= ref->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0 ? w : -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0. ? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
	} else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
	w += a->w + b->w;
	w = w < opt->w << 2 ? w : opt->w << 2;
	if (score_fn(ud, a, b, w, &score)) { *missing = 1; return 0; }
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
	*_w = w;
	return score;
}

/* mem_sort_deduplicate, mem_alnreg.c:112-202 */
void bsx_regs_sort_dedup(const bsx_opt_t *opt, const bsx_refmeta_t *ref, int can_merge, reg_v *regs,
                         bsx_glb_score_fn score_fn, void *ud, int *missing)
{
	int i, m;
	if (regs->n <= 1) return;
	regs_sort(regs, 0);   /* by END, not start (ks_introsort(mem_ars2)) */
	for (i = 0; (size_t)i < regs->n; ++i) regs->a[i].n_comp = 1;
	for (i = 1; (size_t)i < regs->n; ++i) {
		reg_t *p = &regs->a[i];
		int j;
		for (j = i - 1; j >= 0 && p->rid == regs->a[j].rid && p->rb < regs->a[j].re + opt->max_chain_gap; --j) {
			reg_t *q = &regs->a[j];
			int64_t or_, oq, mr, mq;
			int score, w;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) { /* one of the two is redundant */
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (can_merge && q->rb < p->rb && (score = try_concat(opt, ref, q, p, &w, score_fn, ud, missing)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->truesc = p->score = score;
				p->qb = q->qb; p->rb = q->rb;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; (size_t)i < regs->n; ++i)
		if (regs->a[i].qe > regs->a[i].qb) { if (m != i) regs->a[m++] = regs->a[i]; else ++m; }
	regs->n = m;
	regs_sort(regs, 1);
	for (i = 1; (size_t)i < regs->n; ++i)
		if (regs->a[i].score == regs->a[i - 1].score && regs->a[i].rb == regs->a[i - 1].rb && regs->a[i].qb == regs->a[i - 1].qb)
			regs->a[i].qe = regs->a[i].qb;
	for (i = 1, m = 1; (size_t)i < regs->n; ++i)
		if (regs->a[i].qe > regs->a[i].qb) { if (m != i) regs->a[m++] = regs->a[i]; else ++m; }
	regs->n = m;
}

/* mem_alnreg_matesw_core's "put the rescued hit into the mate's list by score, then mem_sort_deduplicate" (mem_alnreg.c:478-488), one
 * hit after another onto the same list.  A read inside a repeat family gets a hundred such hits onto a list of two hundred regions,
 * and sorting the list twice per hit was most of the host's time on a repeat-rich genome.  Between two hits nothing but the new
 * region is new: the list is already in the second sort's order and free of redundant pairs.  So the order by end is kept as an
 * index array beside the list, the new region is put into both orders by search, and the redundancy scan -- the reference's loop,
 * line by line, over the index array -- is all that runs.  That is the reference's result whenever the two sorts have no ties to
 * break (equal ends, or equal (score, start, query start) with the new region): with ties the outcome depends on klib's introsort
 * and on the order it starts from, so those calls go through bsx_regs_sort_dedup on the list as the reference would hold it. */
void bsx_regs_inc_free(bsx_regs_inc_t *st) { free(st->ord); st->ord = 0; st->valid = st->m = 0; }
static void inc_rebuild(reg_v *regs, bsx_regs_inc_t *st)
{
	size_t n = regs->n, i;
	prox_re_t *px = (prox_re_t*)malloc(sizeof(prox_re_t) * (n ? n : 1));
	if (st->m < (int)n + 8) { st->m = (int)n * 2 + 16; st->ord = (int*)realloc(st->ord, sizeof(int) * (size_t)st->m); }
	for (i = 0; i < n; ++i) { px[i].re = regs->a[i].re; px[i].idx = (int)i; }
	sort_prox_re(n, px);   /* (ties make the state invalid: their order does not matter) */
	st->valid = 1;
	for (i = 0; i < n; ++i) { st->ord[i] = px[i].idx; if (i && px[i].re == px[i - 1].re) st->valid = 0; }
	/* (the list comes out of mem_sort_deduplicate: in the second sort's order, identical (score, rb, qb) already removed) */
	free(px);
}
void bsx_regs_insert_dedup(const bsx_opt_t *opt, const bsx_refmeta_t *ref, reg_v *regs, const reg_t *b, bsx_regs_inc_t *st,
                           bsx_glb_score_fn no_score_fn)
{
	int ins, pos, i, n, lo, hi, at, m, any_dead = 0, missing = 0;
	if (regs->n == regs->m) { size_t m2 = regs->m ? regs->m << 1 : 4; regs->a = (reg_t*)bsx_crealloc(regs->a, sizeof(reg_t) * regs->n, sizeof(reg_t) * m2); regs->m = m2; }
	n = (int)regs->n;
	for (ins = 0; ins < n; ++ins) if (regs->a[ins].score < b->score) break;   /* where the reference puts it: ahead of the first lower score */
	if (st->valid && n >= 1) {
		/* ties?  equal end with a region of the list; equal (score, rb, qb) among the regions of b's score (they start at ins - run) */
		lo = 0; hi = n;
		while (lo < hi) { const int mid = (lo + hi) >> 1; if (regs->a[st->ord[mid]].re < b->re) lo = mid + 1; else hi = mid; }
		at = lo;
		if (at < n && regs->a[st->ord[at]].re == b->re) st->valid = 0;
		for (i = ins - 1; st->valid && i >= 0 && regs->a[i].score == b->score; --i)
			if (regs->a[i].rb == b->rb && regs->a[i].qb == b->qb) st->valid = 0;
	} else st->valid = 0;
	if (!st->valid) { /* the reference's own sequence on the reference's own arrangement, then the orders are set up afresh */
		for (pos = n; pos > ins; --pos) regs->a[pos] = regs->a[pos - 1];
		regs->a[ins] = *b; ++regs->n;
		bsx_regs_sort_dedup(opt, ref, 0, regs, no_score_fn, 0, &missing);
		inc_rebuild(regs, st);
		return;
	}
	/* b's place in the second sort's order: among its own score by (rb, qb) */
	for (pos = ins; pos > 0 && regs->a[pos - 1].score == b->score && (b->rb < regs->a[pos - 1].rb || (b->rb == regs->a[pos - 1].rb && b->qb < regs->a[pos - 1].qb)); --pos);
	for (i = n; i > pos; --i) regs->a[i] = regs->a[i - 1];
	regs->a[pos] = *b; ++regs->n; ++n;
	if (st->m < n + 8) { st->m = n * 2 + 16; st->ord = (int*)realloc(st->ord, sizeof(int) * (size_t)st->m); }
	for (i = 0; i < n - 1; ++i) if (st->ord[i] >= pos) ++st->ord[i];
	memmove(st->ord + at + 1, st->ord + at, sizeof(int) * (size_t)(n - 1 - at));
	st->ord[at] = pos;
	/* the redundancy scan of mem_sort_deduplicate (mem_alnreg.c:131-160, no merging here) over the order by end */
	for (i = 0; i < n; ++i) regs->a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		reg_t *p = &regs->a[st->ord[i]];
		int j;
		for (j = i - 1; j >= 0 && p->rid == regs->a[st->ord[j]].rid && p->rb < regs->a[st->ord[j]].re + opt->max_chain_gap; --j) {
			reg_t *q = &regs->a[st->ord[j]];
			int64_t or_, oq, mr, mq;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
				any_dead = 1;
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			}
		}
	}
	if (any_dead) { /* drop them from both orders */
		int *nw = (int*)malloc(sizeof(int) * (size_t)n);
		for (i = 0, m = 0; i < n; ++i) { if (regs->a[i].qe > regs->a[i].qb) { nw[i] = m; if (m != i) regs->a[m] = regs->a[i]; ++m; } else nw[i] = -1; }
		regs->n = (size_t)m;
		for (i = 0, m = 0; i < n; ++i) if (nw[st->ord[i]] >= 0) st->ord[m++] = nw[st->ord[i]];
		free(nw);
	}
}

/* ------------------------------------------------------------------ primary marking */
static int reg_hash_lt(const void *a_, const void *b_)   /* alnreg_hlt: score desc, is_alt, hash */
{
	const reg_t *a = (const reg_t*)a_, *b = (const reg_t*)b_;
	return a->score > b->score || (a->score == b->score && (a->is_alt < b->is_alt || (a->is_alt == b->is_alt && a->hash < b->hash)));
}
static int reg_hash_lt2(const void *a_, const void *b_)  /* alnreg_hlt2: is_alt, score desc, hash */
{
	const reg_t *a = (const reg_t*)a_, *b = (const reg_t*)b_;
	return a->is_alt < b->is_alt || (a->is_alt == b->is_alt && (a->score > b->score || (a->score == b->score && a->hash < b->hash)));
}

/* ... on proxies as well (regs_sort above has the argument): the keys and the record's index */
typedef struct { uint64_t hash; int score, is_alt, idx; } prox_h_t;
#define PROX_H_LT(a, b) ((a)->score > (b)->score || ((a)->score == (b)->score && ((a)->is_alt < (b)->is_alt || ((a)->is_alt == (b)->is_alt && (a)->hash < (b)->hash))))
#define PROX_H_LT2(a, b) ((a)->is_alt < (b)->is_alt || ((a)->is_alt == (b)->is_alt && ((a)->score > (b)->score || ((a)->score == (b)->score && (a)->hash < (b)->hash))))
BSX_SORT_DEFINE(sort_prox_h, prox_h_t, PROX_H_LT)
BSX_SORT_DEFINE(sort_prox_h2, prox_h_t, PROX_H_LT2)
static void regs_sort_hash(reg_v *regs, int second)
{
	size_t n = regs->n, i;
	prox_h_t stackbuf[128], *px;
	reg_t *tmp;
	int moved = 0;
	if (n < 2) return;
	if (n < 4) { bsx_introsort(regs->a, n, sizeof(reg_t), second ? reg_hash_lt2 : reg_hash_lt); return; }
	px = n <= sizeof(stackbuf) / sizeof(stackbuf[0]) ? stackbuf : (prox_h_t*)malloc(n * sizeof(prox_h_t));
	for (i = 0; i < n; ++i) { px[i].hash = regs->a[i].hash; px[i].score = regs->a[i].score; px[i].is_alt = regs->a[i].is_alt; px[i].idx = (int)i; }
	if (second) sort_prox_h2(n, px); else sort_prox_h(n, px);
	for (i = 0; i < n; ++i) if (px[i].idx != (int)i) { moved = 1; break; }
	if (moved) {
		tmp = (reg_t*)malloc(sizeof(reg_t) * n);
		for (i = 0; i < n; ++i) tmp[i] = regs->a[px[i].idx];
		memcpy(regs->a, tmp, sizeof(reg_t) * n);
		free(tmp);
	}
	if (px != stackbuf) free(px);
}

typedef BSX_VEC(int) int_v;

/* mem_mark_primary_se_core, mem_alnreg.c:252-288 */
static void mark_core(const bsx_opt_t *opt, int n_mark, reg_v *regs, int_v *z)
{
	int tmp = opt->a + opt->b, i;
	tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
	tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
	z->n = 0;
	bsx_vec_push(*z, 0);
	for (i = 1; i < n_mark; ++i) {
		reg_t *a = &regs->a[i];
		size_t k;
		for (k = 0; k < z->n; ++k) {
			reg_t *b = &regs->a[z->a[k]];
			int b_max = a->qb > b->qb ? a->qb : b->qb;
			int e_min = a->qe < b->qe ? a->qe : b->qe;
			if (e_min > b_max) {
				int min_l = a->qe - a->qb < b->qe - b->qb ? a->qe - a->qb : b->qe - b->qb;
				if (e_min - b_max >= min_l * opt->mask_level) {
					if (b->sub == 0) b->sub = a->score;
					if (b->score - a->score <= tmp && (b->is_alt || !a->is_alt)) ++b->sub_n;
					break;
				}
			}
		}
		if (k == z->n) bsx_vec_push(*z, i);
		else a->secondary = z->a[k];
	}
}

/* mem_mark_primary_se, mem_alnreg.c:290-380 */
void bsx_mark_primary(const bsx_opt_t *opt, reg_v *regs, int64_t id)
{
	int i;
	int_v z;
	regs->n_pri = 0;
	if (regs->n == 0) return;
	for (i = 0; (size_t)i < regs->n; ++i) {
		reg_t *p = &regs->a[i];
		p->sub = p->alt_sc = 0;
		p->secondary = -1;
		p->secondary_all = -1;
		p->hash = bsx_hash64((uint64_t)(id + i));
		if (!p->is_alt) ++regs->n_pri;
	}
	regs_sort_hash(regs, 0);
	bsx_vec_init(z);
	mark_core(opt, (int)regs->n, regs, &z);
	for (i = 0; (size_t)i < regs->n; ++i) {
		reg_t *p = &regs->a[i];
		p->secondary_all = i;
		if (!p->is_alt && p->secondary >= 0 && regs->a[p->secondary].is_alt) p->alt_sc = regs->a[p->secondary].score;
	}
	if (regs->n_pri > 0 && regs->n_pri < regs->n) {
		bsx_vec_reserve(z, regs->n);
		regs_sort_hash(regs, 1);
		for (i = 0; (size_t)i < regs->n; ++i) z.a[regs->a[i].secondary_all] = i;
		for (i = 0; (size_t)i < regs->n; ++i) {
			if (regs->a[i].secondary >= 0) {
				regs->a[i].secondary_all = z.a[regs->a[i].secondary];
				if (regs->a[i].is_alt) regs->a[i].secondary = INT_MAX;
			} else regs->a[i].secondary_all = -1;
		}
		for (i = 0; (size_t)i < regs->n_pri; ++i) { regs->a[i].sub = 0; regs->a[i].secondary = -1; }
		mark_core(opt, (int)regs->n_pri, regs, &z);
	} else {
		for (i = 0; (size_t)i < regs->n; ++i) regs->a[i].secondary_all = regs->a[i].secondary;
	}
	bsx_vec_free(z);
}

/* ------------------------------------------------------------------ insert size */
/* mem_infer_isize / mem_alnreg_isize, mem_alnreg.h:75-93 */
static int infer_isize(int64_t pos1, int64_t pos2, int isrev1, int isrev2, int len1, int len2, int64_t *isize)
{
	if (isrev1 && !isrev2) { *isize = pos1 - pos2 + len1; return 1; }
	else if (isrev2 && !isrev1) { *isize = pos2 - pos1 + len2; return 1; }
	return 0;
}
int bsx_reg_isize(const bsx_refmeta_t *ref, const reg_t *r1, const reg_t *r2, int64_t *isize)
{
	int isrev1, isrev2;
	int64_t pos1, pos2;
	if (r1->rid != r2->rid) return 0;
	isrev1 = r1->rb > ref->l_pac; isrev2 = r2->rb > ref->l_pac;
	pos1 = isrev1 ? (ref->l_pac << 1) - 1 - r1->rb : r1->rb;
	pos2 = isrev2 ? (ref->l_pac << 1) - 1 - r2->rb : r2->rb;
	return infer_isize(pos1, pos2, isrev1, isrev2, r1->qe - r1->qb, r2->qe - r2->qb, isize);
}

#define MIN_RATIO     0.8    /* mem_pair.c:35-40 */
#define MIN_DIR_CNT   10
#define OUTLIER_BOUND 2.0
#define MAPPING_BOUND 3.0
#define MAX_STDDEV    4.0

static int cal_sub(const bsx_opt_t *opt, const reg_v *regs)   /* mem_pair.c:42-57 */
{
	const reg_t *best = &regs->a[0], *p = 0;
	size_t j;
	for (j = 1; j < regs->n; ++j) {
		int b_max, e_min;
		p = &regs->a[j];
		b_max = p->qb > best->qb ? p->qb : best->qb;
		e_min = p->qe < best->qe ? p->qe : best->qe;
		if (e_min > b_max) {
			int min_l = p->qe - p->qb < best->qe - best->qb ? p->qe - p->qb : best->qe - best->qb;
			if (e_min - b_max >= min_l * opt->mask_level) break;
		}
	}
	return j < regs->n ? p->score : opt->min_seed_len * opt->a;
}

/* mem_pestat, mem_pair.c:60-144 (messages go to stderr at verbosity >= 3 like the reference) */
typedef struct { const bsx_opt_t *opt; const bsx_refmeta_t *ref; const reg_v *regs; int32_t *is; } pes_par_t;
#define PES_NONE INT32_MIN
static void pes_worker(void *data, long i, int tid)   /* the pair's insert size if both ends are unique enough (mem_pair.c:68-84) */
{
	pes_par_t *P = (pes_par_t*)data;
	const reg_v *r0 = &P->regs[i << 1 | 0], *r1 = &P->regs[i << 1 | 1];
	const reg_t *b0, *b1;
	int64_t is;
	(void)tid;
	P->is[i] = PES_NONE;
	if (r0->n == 0 || r1->n == 0) return;
	b0 = &r0->a[0]; b1 = &r1->a[0];
	if (cal_sub(P->opt, r0) > MIN_RATIO * b0->score) return;
	if (cal_sub(P->opt, r1) > MIN_RATIO * b1->score) return;
	if (b0->rid != b1->rid) return;
	if (b0->bss != b1->bss) return;
	if (bsx_reg_isize(P->ref, b0, b1, &is))
		if (is <= P->opt->max_ins && is >= -P->opt->max_ins) P->is[i] = (int32_t)is;
}

/* mem_pestat.  The reference sorts the insert sizes and walks the sorted array; they are integers in
 * [-max_ins, max_ins], so a histogram gives the same sorted sequence, and the sums below are taken over it
 * value by value in ascending order exactly as the reference's loops do. */
BSX_API void (*bsx_pes_hist_hook)(void *ud, int64_t *hist, int n_bins) = 0;
BSX_API void *bsx_pes_hist_ud = 0;
/* a rank whose slice of a chunk is empty still takes part in the exchange */
BSX_API void bsx_pestat_sync_empty(const bsx_opt_t *opt)
{
	int nb = 2 * opt->max_ins + 1;
	int64_t *hist;
	if (!bsx_pes_hist_hook || opt->max_ins < 0) return;
	hist = (int64_t*)calloc((size_t)nb, sizeof(int64_t));
	bsx_pes_hist_hook(bsx_pes_hist_ud, hist, nb);
	free(hist);
}
bsx_pestat_t bsx_pestat(const bsx_opt_t *opt, const bsx_refmeta_t *ref, int n, const reg_v *regs)
{
	bsx_pestat_t pes;
	pes_par_t P;
	int64_t *hist, tot = 0, c, want[3], cum;
	int i, x, p25 = 0, p50 = 0, p75 = 0, np = n >> 1, nb = 2 * opt->max_ins + 1, q;
	long j;
	memset(&pes, 0, sizeof(pes));
	if (opt->max_ins < 0) { pes.failed = 1; return pes; }
	P.opt = opt; P.ref = ref; P.regs = regs; P.is = (int32_t*)malloc(sizeof(int32_t) * (size_t)(np ? np : 1));
	bsx_parallel_for(bsx_host_threads(opt), pes_worker, &P, np);
	hist = (int64_t*)calloc((size_t)nb, sizeof(int64_t));
	for (i = 0; i < np; ++i) if (P.is[i] != PES_NONE) { ++hist[P.is[i] + opt->max_ins]; ++tot; }
	free(P.is);
	/* Several GPUs sharing one chunk (each aligns a slice of its pairs, cli.c): mem_pestat is the one step of a chunk that looks at all
	 * of its pairs (bwamem.c:464-467), and everything it computes is a function of the histogram of insert sizes -- so the ranks add their
	 * histograms (an all-reduce of 2 * max_ins + 1 counts through the hook) and each gets the statistics of the whole chunk. */
	if (bsx_pes_hist_hook) { bsx_pes_hist_hook(bsx_pes_hist_ud, hist, nb); for (i = 0, tot = 0; i < nb; ++i) tot += hist[i]; }
	if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] # candidate unique pairs: %ld\n", "mem_pestat", (long)tot);
	if (tot < MIN_DIR_CNT) {
		fprintf(stderr, "[M:%s] There are not enough pairs for insert size inference\n", "mem_pestat");
		free(hist);
		pes.failed = 1;
		return pes;
	}
	/* element k of the sorted array, for the three percentile positions */
	want[0] = (int)(.25 * tot + .499); want[1] = (int)(.50 * tot + .499); want[2] = (int)(.75 * tot + .499);
	for (i = 0, cum = 0, q = 0; i < nb && q < 3; ++i) {
		cum += hist[i];
		while (q < 3 && want[q] < cum) { int v = i - opt->max_ins; if (q == 0) p25 = v; else if (q == 1) p50 = v; else p75 = v; ++q; }
	}
	pes.low  = (int)(p25 - OUTLIER_BOUND * (p75 - p25) + .499);
	pes.high = (int)(p75 + OUTLIER_BOUND * (p75 - p25) + .499);
	if (bsx_verbose >= 3) {
		fprintf(stderr, "[M::%s] (25, 50, 75) percentile: (%d, %d, %d)\n", "mem_pestat", p25, p50, p75);
		fprintf(stderr, "[M::%s] low and high boundaries for computing mean and std.dev: (%d, %d)\n", "mem_pestat", pes.low, pes.high);
	}
	for (i = 0, x = 0, pes.avg = 0; i < nb; ++i) {
		int v = i - opt->max_ins;
		if (v >= pes.low && v <= pes.high) for (c = 0; c < hist[i]; ++c) { pes.avg += v; ++x; }
	}
	pes.avg /= x;
	for (i = 0, pes.std = 0; i < nb; ++i) {
		int v = i - opt->max_ins;
		if (v >= pes.low && v <= pes.high) for (j = 0; j < hist[i]; ++j) pes.std += (v - pes.avg) * (v - pes.avg);
	}
	pes.std = sqrt(pes.std / x);
	if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] mean and std.dev: (%.2f, %.2f)\n", "mem_pestat", pes.avg, pes.std);
	pes.low  = (int)(p25 - MAPPING_BOUND * (p75 - p25) + .499);
	pes.high = (int)(p75 + MAPPING_BOUND * (p75 - p25) + .499);
	if (pes.low > pes.avg - MAX_STDDEV * pes.std) pes.low = (int)(pes.avg - MAX_STDDEV * pes.std + .499);
	if (pes.high < pes.avg + MAX_STDDEV * pes.std) pes.high = (int)(pes.avg + MAX_STDDEV * pes.std + .499);
	if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] low and high boundaries for proper pairs: (%d, %d)\n", "mem_pestat", pes.low, pes.high);
	free(hist);
	return pes;
}

/* ------------------------------------------------------------------ pairing */
typedef struct { uint64_t x, y; } pair64_t;
typedef struct { uint64_t x, y, z; } trio64_t;
#define PAIR64_LT(a, b) ((a)->x < (b)->x || ((a)->x == (b)->x && (a)->y < (b)->y))   /* utils.c:46: the 192-bit sort ignores z */
BSX_SORT_DEFINE(sort_pair64, pair64_t, PAIR64_LT)
BSX_SORT_DEFINE(sort_trio64, trio64_t, PAIR64_LT)

static int region_depos(const bsx_refmeta_t *ref, const reg_t *reg)   /* mem_alnreg.h:135-140 */
{
	int is_rev;
	int64_t rpos = bsx_depos(ref->l_pac, reg->rb < ref->l_pac ? reg->rb : reg->re - 1, &is_rev);
	return (int)(rpos - ref->anns[reg->rid].offset);
}

/* mem_pair, mem_pair.c:147-270 */
void bsx_pair(const bsx_opt_t *opt, const bsx_refmeta_t *ref, const bsx_pestat_t *pes, reg_v pair[2], int id,
              int *score, int *sub, int *n_sub, int z[2])
{
	int64_t l_pac = ref->l_pac;
	BSX_VEC(trio64_t) v;
	BSX_VEC(pair64_t) pp;
	int i, r, k;
	bsx_vec_init(v); bsx_vec_init(pp);
	for (r = 0; r < 2; ++r) {
		for (i = 0; (size_t)i < pair[r].n_pri; ++i) {
			trio64_t key;
			const reg_t *p = &pair[r].a[i];
			key.x = (uint64_t)p->bss << 63 | (uint64_t)p->rid << 32 | region_depos(ref, p);
			key.y = (uint64_t)p->score << 32 | i << 2 | (p->rb >= l_pac) << 1 | r;
			key.z = (uint64_t)(p->qe - p->qb);
			bsx_vec_push(v, key);
		}
	}
	sort_trio64(v.n, v.a);
	for (i = 0; (size_t)i < v.n; ++i) {
		for (k = i - 1; k >= 0; --k) {
			int64_t is = 0;
			int hi = pes->low > pes->high ? pes->low : pes->high;
			if (v.a[i].x >> 32 != v.a[k].x >> 32) break;
			if (v.a[i].x >> 63 != v.a[k].x >> 63) break;
			if ((int64_t)(v.a[i].x & 0xffffffffU) - (int64_t)(v.a[k].x & 0xffffffffU) > hi) break;
			if ((v.a[i].y & 1) == (v.a[k].y & 1)) break;
			if (infer_isize((int64_t)v.a[k].x, (int64_t)v.a[i].x, (v.a[k].y >> 1) & 1, (v.a[i].y >> 1) & 1, (int)v.a[k].z, (int)v.a[i].z, &is) &&
			    is >= pes->low && is <= pes->high) {
				double zscore = (is - pes->avg) / pes->std;
				int _score = (int)((v.a[i].y >> 32) + (v.a[k].y >> 32) + .721 * log(2. * erfc(fabs(zscore) * M_SQRT1_2)) * opt->a + .499);
				pair64_t *p;
				if (_score < 0) _score = 0;
				p = bsx_vec_pushp(pp);
				p->y = (uint64_t)k << 32 | i;
				p->x = (uint64_t)_score << 32 | (bsx_hash64(p->y ^ id << 8) & 0xffffffffU);
			}
		}
	}
	if (pp.n) {
		int tmp;
		sort_pair64(pp.n, pp.a);
		i = (int)(pp.a[pp.n - 1].y >> 32);
		k = (int)(pp.a[pp.n - 1].y << 32 >> 32);
		z[v.a[i].y & 1] = (int)(v.a[i].y << 32 >> 34);
		z[v.a[k].y & 1] = (int)(v.a[k].y << 32 >> 34);
		*score = (int)(pp.a[pp.n - 1].x >> 32);
		*sub = pp.n > 1 ? (int)(pp.a[pp.n - 2].x >> 32) : 0;
		tmp = opt->a + opt->b;
		tmp = tmp > opt->o_del + opt->e_del ? tmp : opt->o_del + opt->e_del;
		tmp = tmp > opt->o_ins + opt->e_ins ? tmp : opt->o_ins + opt->e_ins;
		for (i = (int)((long)pp.n - 2), *n_sub = 0; i >= 0; --i)
			if (*sub - (int)(pp.a[i].x >> 32) <= tmp) ++*n_sub;
	} else { *score = 0; *sub = 0; *n_sub = 0; z[0] = -1; z[1] = -1; }
	bsx_vec_free(v); bsx_vec_free(pp);
}

/* ------------------------------------------------------------------ MAPQ */
#define MEM_MAPQ_COEF 30.0
/* mem_approx_mapq_se, bwamem.c:134-157 */
int bsx_approx_mapq_se(const bsx_opt_t *opt, const reg_t *a)
{
	int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
	if (a->score == 0) mapq = 0;
	else if (opt->mapQ_coef_len > 0) {
		double tmp;
		tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
	} else {
		mapq = (int)(MEM_MAPQ_COEF * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

/* ---- test hooks (tests/test_oracle_golden.py): the functions above that restate header-inline functions of the reference, against the
 * reference's own, recorded in tests/golden/ref_vectors.npz.  a / b: rid, rb, re, qb, qe of a region */
static void hook_mk_pair(bsx_refmeta_t *ref, reg_t r[2], int64_t l_pac, const int64_t a[5], const int64_t b[5])
{
	memset(ref, 0, sizeof(*ref)); memset(r, 0, 2 * sizeof(reg_t));
	ref->l_pac = l_pac;
	r[0].rid = (int)a[0]; r[0].rb = a[1]; r[0].re = a[2]; r[0].qb = (int)a[3]; r[0].qe = (int)a[4];
	r[1].rid = (int)b[0]; r[1].rb = b[1]; r[1].re = b[2]; r[1].qb = (int)b[3]; r[1].qe = (int)b[4];
}
BSX_API int bsx_hook_infer_isize(int64_t pos1, int64_t pos2, int isrev1, int isrev2, int len1, int len2, int64_t *isize)
{ return infer_isize(pos1, pos2, isrev1, isrev2, len1, len2, isize); }
BSX_API int bsx_hook_reg_isize(int64_t l_pac, const int64_t a[5], const int64_t b[5], int64_t *isize)
{ bsx_refmeta_t ref; reg_t r[2]; hook_mk_pair(&ref, r, l_pac, a, b); return bsx_reg_isize(&ref, &r[0], &r[1], isize); }
BSX_API int bsx_hook_region_depos(int64_t l_pac, int64_t contig_offset, int64_t rb, int64_t re)
{
	bsx_refmeta_t ref; bsx_ann_t ann; reg_t r;
	memset(&ref, 0, sizeof(ref)); memset(&ann, 0, sizeof(ann)); memset(&r, 0, sizeof(r));
	ref.l_pac = l_pac; ref.n_seqs = 1; ref.anns = &ann; ann.offset = contig_offset;
	r.rid = 0; r.rb = rb; r.re = re;
	return region_depos(&ref, &r);
}
BSX_API int64_t bsx_hook_depos(int64_t l_pac, int64_t pos, int *is_rev) { return bsx_depos(l_pac, pos, is_rev); }
