/* hooks.c -- small exported entry points used by the Python plumbing (multi-GPU launcher, tests):
 * flat wrappers over internal host functions, no logic of their own. */
#include "align_types.h"
#include "sort_tmpl.h"
#include "hook_types.h"
#include "fastq.h"

/* ---- chunked FASTQ access for the chunk-sharded multi-GPU launcher ---- */
BSX_API void *bsx_hook_fq_open(const char *fn) { return bsx_fq_open(fn); }
BSX_API int bsx_hook_fq_seek(void *f, int64_t off) { return bsx_fq_seek((bsx_fq_t*)f, off); }
BSX_API void bsx_hook_fq_close(void *f) { bsx_fq_close((bsx_fq_t*)f); }
BSX_API int bsx_hook_fq_error(void *f) { return bsx_fq_error((const bsx_fq_t*)f); }
BSX_API int bsx_hook_fq_skip_chunk(void *f1, void *f2, int chunk_size) { return bsx_fq_skip_chunk((bsx_fq_t*)f1, (bsx_fq_t*)f2, chunk_size); }
BSX_API bsx_read_t *bsx_hook_fq_chunk(void *f1, void *f2, int chunk_size, int has_bc, int *n) { return bsx_fq_read_chunk((bsx_fq_t*)f1, (bsx_fq_t*)f2, chunk_size, has_bc, n); }
BSX_API void bsx_hook_reads_free(bsx_read_t *r, int n) { int i; if (!r) return; for (i = 0; i < n; ++i) bsx_read_free(&r[i]); free(r); }
/* concatenated SAM text of a processed chunk; returns bytes written (or needed when cap is too small) */
BSX_API int64_t bsx_hook_chunk_sam(const bsx_read_t *r, int n, char *buf, int64_t cap)
{
	int64_t tot = 0; int i;
	for (i = 0; i < n; ++i) if (r[i].sam) { size_t l = strlen(r[i].sam); if (buf && tot + (int64_t)l <= cap) memcpy(buf + tot, r[i].sam, l); tot += (int64_t)l; }
	return tot;
}

/* ---- test hooks ---- */
typedef struct { int64_t key, id; } kv_t;
static int kv_lt(const void *a, const void *b) { return ((const kv_t*)a)->key < ((const kv_t*)b)->key; }
static int kv_gt(const void *a, const void *b) { return ((const kv_t*)a)->key > ((const kv_t*)b)->key; }
BSX_API void bsx_hook_sort_kv(int64_t n, int64_t *kv, int desc) { bsx_introsort(kv, (size_t)n, sizeof(kv_t), desc ? kv_gt : kv_lt); }
/* the same through sort_tmpl.h (the introsort compiled per element type that the back half's sorts use) */
#define KV_LT(a, b) ((a)->key < (b)->key)
#define KV_GT(a, b) ((a)->key > (b)->key)
BSX_SORT_DEFINE(sort_kv_asc, kv_t, KV_LT)
BSX_SORT_DEFINE(sort_kv_desc, kv_t, KV_GT)
BSX_API void bsx_hook_sort_kv_typed(int64_t n, int64_t *kv, int desc) { if (desc) sort_kv_desc((size_t)n, (kv_t*)kv); else sort_kv_asc((size_t)n, (kv_t*)kv); }

BSX_API int bsx_hook_mapq(int a, int b, int min_seed_len, float coef_len, int coef_fac, int score, int sub, int csub, int sub_n,
                          int qb, int qe, int64_t rb, int64_t re, int seedcov, float frac_rep)
{
	bsx_opt_t o; reg_t r;
	bsx_opt_init(&o);
	o.a = a; o.b = b; o.min_seed_len = min_seed_len; o.mapQ_coef_len = coef_len; o.mapQ_coef_fac = coef_fac;
	memset(&r, 0, sizeof(r));
	r.score = score; r.sub = sub; r.csub = csub; r.sub_n = sub_n; r.qb = qb; r.qe = qe; r.rb = rb; r.re = re; r.seedcov = seedcov; r.frac_rep = frac_rep;
	return bsx_approx_mapq_se(&o, &r);
}
BSX_API uint64_t bsx_hook_hash64(uint64_t k) { return bsx_hash64(k); }
BSX_API int bsx_hook_opt_defaults(char *buf, int cap)
{
	bsx_opt_t O, *o = &O;
	bsx_opt_init(o);
	return snprintf(buf, cap,
		"a=%d b=%d o_del=%d e_del=%d o_ins=%d e_ins=%d pen_unpaired=%d pen_clip5=%d pen_clip3=%d w=%d zdrop=%d "
		"max_mem_intv=%llu T=%d flag=%d min_seed_len=%d min_chain_weight=%d max_chain_extend=%u split_factor=%.9g "
		"split_width=%d max_occ=%u max_chain_gap=%d n_threads=%d chunk_size=%d mask_level=%.9g drop_ratio=%.9g "
		"XA_drop_ratio=%.9g mask_level_redun=%.9g mapQ_coef_len=%.9g mapQ_coef_fac=%d max_ins=%d max_matesw=%d "
		"max_XA_hits=%d max_XA_hits_alt=%d parent=%d bsstrand=%d clip5=%d clip3=%d min_base_qual=%d has_bc=%d",
		o->a, o->b, o->o_del, o->e_del, o->o_ins, o->e_ins, o->pen_unpaired, o->pen_clip5, o->pen_clip3, o->w, o->zdrop,
		(unsigned long long)o->max_mem_intv, o->T, o->flag, o->min_seed_len, o->min_chain_weight, o->max_chain_extend, o->split_factor,
		o->split_width, o->max_occ, o->max_chain_gap, o->n_threads, o->chunk_size, o->mask_level, o->drop_ratio,
		o->XA_drop_ratio, o->mask_level_redun, o->mapQ_coef_len, o->mapQ_coef_fac, o->max_ins, o->max_matesw,
		o->max_XA_hits, o->max_XA_hits_alt, o->parent, o->bsstrand, o->clip5, o->clip3, o->min_base_qual, o->has_bc);
}
BSX_API void *bsx_hook_fq_pair_open(void *f1, void *f2, int has_bc) { return bsx_fq_pair_open((bsx_fq_t*)f1, (bsx_fq_t*)f2, has_bc); }
BSX_API void bsx_hook_fq_pair_close(void *p) { bsx_fq_pair_close((bsx_fq_pair_t*)p); }
BSX_API bsx_read_t *bsx_hook_fq_pair_chunk(void *p, int chunk_size, int *n) { return bsx_fq_pair_read_chunk((bsx_fq_pair_t*)p, chunk_size, n); }
const uint8_t *bsx_nt4_table(void);
BSX_API const uint8_t *bsx_hook_nt4_table(void) { return bsx_nt4_table(); }

/* mem_sort_deduplicate as the host pipeline runs it (region.c) on one read's regions, given as the device hands them over:
 * keep[k] = index in a[] of the k-th region kept; returns their number, or -1 when a concatenation score would be needed */
static int hook_no_score(void *ud, const reg_t *a, const reg_t *b, int w, int *score) { (void)ud; (void)a; (void)b; (void)w; *score = 0; return 1; }
BSX_API int bsx_hook_regs_sort_dedup(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_region_t *a, int n, int *keep)
{
	reg_v v;
	int k, missing = 0, m;
	v.n = v.m = (size_t)n; v.n_pri = 0;
	v.a = (reg_t*)calloc(n ? n : 1, sizeof(reg_t));
	for (k = 0; k < n; ++k) {
		reg_t *r = &v.a[k];
		const bsx_region_t *d = &a[k];
		r->rb = d->rb; r->re = d->re; r->qb = d->qb; r->qe = d->qe; r->rid = d->rid; r->score = d->score; r->truesc = d->truesc;
		r->w = d->w; r->seedcov = d->seedcov; r->seedlen0 = d->seedlen0; r->frac_rep = d->frac_rep; r->bss = d->bss; r->parent = d->parent;
		r->hash = (uint64_t)k;   /* not touched by the function: carries the index through its sorts */
	}
	bsx_regs_sort_dedup(opt, &idx->ref, 1, &v, hook_no_score, 0, &missing);
	m = missing ? -1 : (int)v.n;
	for (k = 0; k < m; ++k) keep[k] = (int)v.a[k].hash;
	free(v.a);
	return m;
}

/* mate rescue's "add the hit by score, then mem_sort_deduplicate" (mem_alnreg.c:478-488) for hits a[n0..n) added one after the other to the
 * list a[0..n0) as it stands (bsx_regs_insert_dedup: the incremental form mate rescue uses); keep[] = the ids left, in their final order */
BSX_API int bsx_hook_regs_insert_seq(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_region_t *a, int n0, int n, int *keep)
{
	reg_v v;
	bsx_regs_inc_t st;
	int k, m;
	memset(&st, 0, sizeof(st));
	v.n = (size_t)n0; v.m = (size_t)n0 + 1; v.n_pri = 0;
	v.a = (reg_t*)calloc(v.m, sizeof(reg_t));
	for (k = 0; k < n; ++k) {
		reg_t r;
		const bsx_region_t *d = &a[k];
		memset(&r, 0, sizeof(r));
		r.rb = d->rb; r.re = d->re; r.qb = d->qb; r.qe = d->qe; r.rid = d->rid; r.score = d->score; r.truesc = d->truesc;
		r.hash = (uint64_t)k;
		if (k < n0) v.a[k] = r;
		else bsx_regs_insert_dedup(opt, &idx->ref, &v, &r, &st, hook_no_score);
	}
	m = (int)v.n;
	for (k = 0; k < m; ++k) keep[k] = (int)v.a[k].hash;
	bsx_regs_inc_free(&st);
	free(v.a);
	return m;
}

/* ---- the per-read / per-pair functions of the back half, callable on plain arrays (tests/test_oracle_backhalf.py compares them with
 * oracle/backhalf.py, an independent restatement of the reference's functions) */
static void hook_vec(const bsx_hook_reg_t *a, int n, int n_pri, reg_v *v)
{
	int k;
	v->n = v->m = (size_t)n; v->n_pri = (size_t)n_pri;
	v->a = (reg_t*)calloc(n ? n : 1, sizeof(reg_t));
	for (k = 0; k < n; ++k) bsx_hook_to_reg(&a[k], &v->a[k]);
}

/* mem_mark_primary_se in place; returns n_pri */
BSX_API int bsx_hook_mark_primary(const bsx_opt_t *opt, bsx_hook_reg_t *a, int n, int64_t id)
{
	reg_v v;
	int k, n_pri;
	hook_vec(a, n, 0, &v);
	bsx_mark_primary(opt, &v, id);
	for (k = 0; k < n; ++k) bsx_hook_from_reg(&v.a[k], &a[k]);
	n_pri = (int)v.n_pri;
	free(v.a);
	return n_pri;
}

/* mem_pestat over n_reads reads (pairs 2i, 2i+1); read i's regions are a[off[i] .. off[i+1]) */
BSX_API void bsx_hook_pestat(const bsx_opt_t *opt, const bsx_index_t *idx, int n_reads, const bsx_hook_reg_t *a, const int64_t *off, bsx_pestat_t *out)
{
	reg_v *regs = (reg_v*)calloc(n_reads ? n_reads : 1, sizeof(reg_v));
	int i;
	for (i = 0; i < n_reads; ++i) hook_vec(a + off[i], (int)(off[i + 1] - off[i]), 0, &regs[i]);
	*out = bsx_pestat(opt, &idx->ref, n_reads, regs);
	for (i = 0; i < n_reads; ++i) free(regs[i].a);
	free(regs);
}

/* mem_pair: out = score, sub, n_sub, z[0], z[1] */
BSX_API void bsx_hook_pair(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_pestat_t *pes, const bsx_hook_reg_t *a0, int n0, int n_pri0,
                           const bsx_hook_reg_t *a1, int n1, int n_pri1, int id, int out[5])
{
	reg_v pair[2];
	int z[2] = {-1, -1};
	hook_vec(a0, n0, n_pri0, &pair[0]);
	hook_vec(a1, n1, n_pri1, &pair[1]);
	bsx_pair(opt, &idx->ref, pes, pair, id, &out[0], &out[1], &out[2], z);
	out[3] = z[0]; out[4] = z[1];
	free(pair[0].a); free(pair[1].a);
}

/* mem_reg2sam_pe up to the text: the planning pass of the pipeline on one pair.  The region records come back as the decision left
 * them (flag, mapq, sub, secondary, secondary_all); trace[k] = the k-th record that would be written (see samctx_t); returns their number */
#include "pipeline.h"
BSX_API int bsx_hook_reg2sam_pe_plan(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_pestat_t *pes, uint64_t id, const int l_seq[2],
                                     bsx_hook_reg_t *a0, int n0, int n_pri0, bsx_hook_reg_t *a1, int n1, int n_pri1, int (*trace)[6], int m_trace)
{
	reg_v pair[2];
	bsx_read_t s[2];
	samctx_t ctx;
	bsx_hook_reg_t *a[2] = {a0, a1};
	int i, k, n[2] = {n0, n1};
	char name[] = "r";
	memset(s, 0, sizeof(s)); memset(&ctx, 0, sizeof(ctx));
	hook_vec(a0, n0, n_pri0, &pair[0]);
	hook_vec(a1, n1, n_pri1, &pair[1]);
	for (i = 0; i < 2; ++i) {
		s[i].l_seq = s[i].l_seq0 = l_seq[i]; s[i].name = name;
		for (k = 0; k < n[i]; ++k) pair[i].a[k].hash = (uint64_t)k;   /* lets the trace name a mate's region */
	}
	ctx.plan = 1; ctx.trace = trace; ctx.m_trace = m_trace;
	bsx_reg2sam_pe(opt, idx, id, s, pair, pes, &ctx, 0);
	for (i = 0; i < 2; ++i) {
		for (k = 0; k < n[i]; ++k) bsx_hook_from_reg(&pair[i].a[k], &a[i][k]);
		free(pair[i].a);
		bsx_cvec_free(ctx.want[i]);
	}
	return ctx.n_trace;
}
