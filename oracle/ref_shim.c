/* oracle/ref_shim.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Thin ctypes-friendly entry points onto the *real* reference functions of
 * zhou-lab/biscuit, compiled from the sources where they lie under
 * /root/reference/lib/aln (see oracle/Makefile).  Only reference files that
 * build without any stand-in header are used:
 *   bwt.c ksw.c utils.c is.c bwa.c bwamem.c  (+ ksort.h / kbtree.h templates)
 * memchain.c, mem_alnreg.c, mem_pair.c, mem_alnreg_format.c, bntseq.c,
 * bwtindex.c and align.c include un-vendored headers (wzmisc.h / encode.h,
 * huishenlab/utils@5f4aeab) and are therefore NOT buildable here; nothing in
 * this shim stands in for them.
 *
 * Nothing here computes anything itself: every function forwards to a
 * reference symbol (cited) and flattens its result into plain buffers.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <zlib.h>
#include "bwt.h"
#include "ksw.h"
#include "utils.h"
#include "ksort.h"
#include "kbtree.h"
#include "bntseq.h"
#include "bwa.h"
#include "bwamem.h"
#include "mem_alnreg.h"

#define API __attribute__((visibility("default")))

/* ---------------- FM index (bwt.c) ---------------- */

/* bwt_restore_bwt (bwt.c:456) + bwt_restore_sa (bwt.c:435) */
API void *ref_bwt_load(const char *fn_bwt, const char *fn_sa)
{
	bwt_t *b = bwt_restore_bwt(fn_bwt);
	if (fn_sa && fn_sa[0]) bwt_restore_sa(fn_sa, b);
	return b;
}
API void ref_bwt_free(void *h) { bwt_destroy((bwt_t*)h); }
API void ref_bwt_meta(void *h, uint64_t out[8])
{
	bwt_t *b = (bwt_t*)h;
	out[0] = b->primary; out[1] = b->L2[1]; out[2] = b->L2[2]; out[3] = b->L2[3]; out[4] = b->L2[4];
	out[5] = b->seq_len; out[6] = b->bwt_size; out[7] = b->n_sa;
}
/* bwt_occ4 (bwt.c:173) */
API void ref_bwt_occ4(void *h, uint64_t k, uint64_t cnt[4]) { bwt_occ4((bwt_t*)h, k, cnt); }
/* bwt_2occ4 (bwt.c:204) */
API void ref_bwt_2occ4(void *h, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4]) { bwt_2occ4((bwt_t*)h, k, l, ck, cl); }
/* bwt_occ (bwt.c:108) */
API uint64_t ref_bwt_occ(void *h, uint64_t k, int c) { return bwt_occ((bwt_t*)h, k, (ubyte_t)c); }
/* bwt_sa (bwt.c:87) */
API uint64_t ref_bwt_sa(void *h, uint64_t k) { return bwt_sa((bwt_t*)h, k); }
API void ref_bwt_sa_batch(void *h, int64_t n, const uint64_t *k, uint64_t *out)
{
	int64_t i;
	for (i = 0; i < n; ++i) out[i] = bwt_sa((bwt_t*)h, k[i]);
}
/* bwt_cal_sa (bwt.c:63): recompute sampled SA from the BWT; copies into out (n_sa entries) */
API uint64_t ref_bwt_cal_sa(void *h, int intv, uint64_t *out, uint64_t cap)
{
	bwt_t *b = (bwt_t*)h;
	uint64_t i;
	bwt_cal_sa(b, intv);
	for (i = 0; i < b->n_sa && i < cap; ++i) out[i] = b->sa[i];
	return b->n_sa;
}
/* bwt_extend (bwt.c:278); ik/ok flattened as 4 x u64 {x0,x1,x2,info} */
API void ref_bwt_extend(void *h, const uint64_t ik[4], int is_back, uint64_t ok[16])
{
	bwtintv_t i, o[4];
	int c;
	i.x[0] = ik[0]; i.x[1] = ik[1]; i.x[2] = ik[2]; i.info = ik[3];
	memset(o, 0, sizeof(o));
	bwt_extend((bwt_t*)h, &i, o, is_back);
	for (c = 0; c < 4; ++c) { ok[c*4] = o[c].x[0]; ok[c*4+1] = o[c].x[1]; ok[c*4+2] = o[c].x[2]; ok[c*4+3] = 0; }
}
/* bwt_smem1a (bwt.c:307). out: up to cap intervals x 4 u64; returns n, *ret = return value */
API int ref_bwt_smem1a(void *hb, void *hc, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv, uint64_t *out, int cap, int *ret)
{
	bwtintv_v mem = {0,0,0};
	size_t i;
	int n;
	*ret = bwt_smem1a((bwt_t*)hb, (bwt_t*)hc, len, q, x, min_intv, max_intv, &mem, 0);
	n = (int)mem.n;
	for (i = 0; i < mem.n && (int)i < cap; ++i) {
		out[i*4] = mem.a[i].x[0]; out[i*4+1] = mem.a[i].x[1]; out[i*4+2] = mem.a[i].x[2]; out[i*4+3] = mem.a[i].info;
	}
	free(mem.a);
	return n;
}
/* bwt_seed_strategy1 (bwt.c:376) */
API int ref_bwt_seed_strategy1(void *hb, void *hc, int len, const uint8_t *q, int x, int min_len, int max_intv, uint64_t out[4])
{
	bwtintv_t m;
	int r = bwt_seed_strategy1((bwt_t*)hb, (bwt_t*)hc, len, q, x, min_len, max_intv, &m);
	out[0] = m.x[0]; out[1] = m.x[1]; out[2] = m.x[2]; out[3] = m.info;
	return r;
}

/* a bwt_t over arrays that live elsewhere (nothing is copied): the fields bwt_restore_bwt / bwt_restore_sa fill from the files
 * (bwt.c:412-491) plus the count table bwt_restore_bwt builds.  For timing the reference's kernels on an index that was never
 * written to disk (bench.py's cpu_baseline). */
API void *ref_bwt_wrap(uint64_t primary, const uint64_t *L2, uint64_t seq_len, uint64_t bwt_size, uint32_t *bwt_, int sa_intv, uint64_t n_sa, uint64_t *sa)
{
	bwt_t *b = (bwt_t*)calloc(1, sizeof(bwt_t));
	int i;
	b->primary = primary; for (i = 0; i < 5; ++i) b->L2[i] = L2[i];
	b->seq_len = seq_len; b->bwt_size = bwt_size; b->bwt = bwt_;
	b->sa_intv = sa_intv; b->n_sa = n_sa; b->sa = (bwtint_t*)sa;
	bwt_gen_cnt_table(b);
	return b;
}
API void ref_bwt_unwrap(void *h) { free(h); }
/* bwt_smem1a with the caller's vectors kept between calls, as mem_collect_intv keeps them in its bwtintv_cache_t (memchain.c:57-59,68) */
typedef struct { bwtintv_v mem, tmp[2]; bwtintv_v *tmpv[2]; } ref_smem_ctx_t;
API void *ref_smem_ctx_new(void) { ref_smem_ctx_t *c = (ref_smem_ctx_t*)calloc(1, sizeof(*c)); c->tmpv[0] = &c->tmp[0]; c->tmpv[1] = &c->tmp[1]; return c; }
API void ref_smem_ctx_free(void *c_) { ref_smem_ctx_t *c = (ref_smem_ctx_t*)c_; free(c->mem.a); free(c->tmp[0].a); free(c->tmp[1].a); free(c); }
API int ref_bwt_smem1a_ctx(void *c_, void *hb, void *hc, int len, const uint8_t *q, int x, int min_intv, const uint64_t **out, int *n)
{
	ref_smem_ctx_t *c = (ref_smem_ctx_t*)c_;
	int r = bwt_smem1a((bwt_t*)hb, (bwt_t*)hc, len, q, x, min_intv, 0, &c->mem, c->tmpv);
	*out = (const uint64_t*)c->mem.a; *n = (int)c->mem.n;   /* bwtintv_t = 4 x u64 */
	return r;
}

/* ---------------- BWT construction primitive (is.c) ---------------- */
int is_bwt(ubyte_t *T, int n); /* is.c:208 */
/* in-place: T (n bytes of 0..3, plus one spare byte) becomes the BWT; returns primary */
API int ref_is_bwt(uint8_t *T, int n) { return is_bwt(T, n); }

/* ---------------- DP kernels (ksw.c) ---------------- */

/* ksw_extend2 (ksw.c:380); out = {score,qle,tle,gtle,gscore,max_off} */
API void ref_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
	int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0, int out[6])
{
	int qle, tle, gtle, gscore, max_off;
	out[0] = ksw_extend2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, &qle, &tle, &gtle, &gscore, &max_off);
	out[1] = qle; out[2] = tle; out[3] = gtle; out[4] = gscore; out[5] = max_off;
}
/* ksw_align2 (ksw.c:343); out = {score,te,qe,score2,te2,tb,qb}. query/target are modified and restored by the reference. */
API void ref_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, const int8_t *mat,
	int o_del, int e_del, int o_ins, int e_ins, int xtra, int out[7])
{
	kswr_t r = ksw_align2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, xtra, 0);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}
/* ksw_global2 (ksw.c:504); want_cigar=0 -> score only. returns score; *n_cigar, cigar[] (cap entries) */
API int ref_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
	int o_del, int e_del, int o_ins, int e_ins, int w, int want_cigar, int *n_cigar, uint32_t *cigar, int cap)
{
	int score, n = 0, i;
	uint32_t *cg = 0;
	if (want_cigar) score = ksw_global2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, w, &n, &cg);
	else score = ksw_global2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, w, 0, 0);
	if (n_cigar) *n_cigar = n;
	for (i = 0; i < n && i < cap; ++i) cigar[i] = cg[i];
	free(cg);
	return score;
}

/* ---------------- scoring / options / MAPQ (bwa.c, bwamem.c) ---------------- */

/* bwa_fill_scmat{,_ct,_ga} (bwa.c:146-182): which = 0 plain, 1 ct, 2 ga */
API void ref_fill_scmat(int which, int a, int b, int8_t mat[25])
{
	if (which == 0) bwa_fill_scmat(a, b, mat);
	else if (which == 1) bwa_fill_scmat_ct(a, b, mat);
	else bwa_fill_scmat_ga(a, b, mat);
}
/* mem_opt_init (bwamem.c:77): dump the defaults as a text record */
API int ref_opt_defaults(char *buf, int cap)
{
	mem_opt_t *o = mem_opt_init();
	int n = snprintf(buf, cap,
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
	free(o);
	return n;
}
/* mem_approx_mapq_se (bwamem.c:134) on a region given by the fields it reads */
API int ref_approx_mapq_se(int a, int b, int min_seed_len, float mapQ_coef_len, int mapQ_coef_fac,
	int score, int sub, int csub, int sub_n, int qb, int qe, int64_t rb, int64_t re, int seedcov, float frac_rep)
{
	mem_opt_t *o = mem_opt_init();
	mem_alnreg_t r;
	int q;
	o->a = a; o->b = b; o->min_seed_len = min_seed_len; o->mapQ_coef_len = mapQ_coef_len; o->mapQ_coef_fac = mapQ_coef_fac;
	memset(&r, 0, sizeof(r));
	r.score = score; r.sub = sub; r.csub = csub; r.sub_n = sub_n; r.qb = qb; r.qe = qe; r.rb = rb; r.re = re;
	r.seedcov = seedcov; r.frac_rep = frac_rep;
	q = mem_approx_mapq_se(o, &r);
	free(o);
	return q;
}
/* bseq_bsconvert (bwamem.c:161) */
API void ref_bsconvert(int l, const uint8_t *seq, int parent, uint8_t *out)
{
	bseq1_t s;
	memset(&s, 0, sizeof(s));
	s.l_seq = l; s.seq = (uint8_t*)seq;
	bseq_bsconvert(&s, (uint8_t)parent);
	memcpy(out, s.bisseq[parent], l);
	free(s.bisseq[parent]);
}
/* infer_bw (bwamem.h:192) */
API int ref_infer_bw(int l1, int l2, int score, int a, int q, int r) { return infer_bw(l1, l2, score, a, q, r); }
/* ---- the header-inline functions of the pairing / formatting code (mem_alnreg.h, bwamem.h, bntseq.h): the .c files around them do
 * not build here, these do -- each forward builds the reference's own structs from plain arguments and calls the inline function */
/* mem_infer_isize (mem_alnreg.h:75-83) */
API int ref_infer_isize(int64_t pos1, int64_t pos2, int isrev1, int isrev2, int len1, int len2, int64_t *isize)
{ return mem_infer_isize(pos1, pos2, isrev1, isrev2, len1, len2, isize); }
static void ref_mk_pair(bntseq_t *bns, mem_alnreg_t r[2], int64_t l_pac, const int64_t a[5], const int64_t b[5])   /* rid, rb, re, qb, qe */
{
	memset(bns, 0, sizeof(*bns)); memset(r, 0, 2 * sizeof(mem_alnreg_t));
	bns->l_pac = l_pac;
	r[0].rid = (int)a[0]; r[0].rb = a[1]; r[0].re = a[2]; r[0].qb = (int)a[3]; r[0].qe = (int)a[4];
	r[1].rid = (int)b[0]; r[1].rb = b[1]; r[1].re = b[2]; r[1].qb = (int)b[3]; r[1].qe = (int)b[4];
}
/* mem_alnreg_isize (mem_alnreg.h:86-93) */
API int ref_alnreg_isize(int64_t l_pac, const int64_t a[5], const int64_t b[5], int64_t *isize)
{ bntseq_t bns; mem_alnreg_t r[2]; ref_mk_pair(&bns, r, l_pac, a, b); return mem_alnreg_isize(&bns, &r[0], &r[1], isize); }
/* is_proper_pair (mem_alnreg.h:95-100) */
API int ref_is_proper_pair(int64_t l_pac, const int64_t a[5], const int64_t b[5], int low, int high)
{ bntseq_t bns; mem_alnreg_t r[2]; mem_pestat_t pes; ref_mk_pair(&bns, r, l_pac, a, b); memset(&pes, 0, sizeof(pes)); pes.low = low; pes.high = high; return is_proper_pair(&bns, &r[0], &r[1], pes); }
/* get_pri_idx (mem_alnreg.h:127-131) over n regions given by score and secondary_all */
API int ref_get_pri_idx(double XA_drop_ratio, int n, const int *score, const int *secondary_all, int i)
{
	mem_alnreg_t *a = (mem_alnreg_t*)calloc((size_t)n, sizeof(mem_alnreg_t));
	int k, r;
	for (k = 0; k < n; ++k) { a[k].score = score[k]; a[k].secondary_all = secondary_all[k]; }
	r = get_pri_idx(XA_drop_ratio, a, i);
	free(a);
	return r;
}
/* region_depos (mem_alnreg.h:139-144): position of a region on its contig, forward strand */
API int ref_region_depos(int64_t l_pac, int64_t contig_offset, int64_t rb, int64_t re, int *is_rev)
{
	bntseq_t bns; bntann1_t ann; mem_alnreg_t r;
	memset(&bns, 0, sizeof(bns)); memset(&ann, 0, sizeof(ann)); memset(&r, 0, sizeof(r));
	bns.l_pac = l_pac; bns.n_seqs = 1; bns.anns = &ann; ann.offset = contig_offset;
	r.rid = 0; r.rb = rb; r.re = re;
	return region_depos(&bns, &r, is_rev);
}
/* get_rlen (bwamem.h:200-208) */
API int ref_get_rlen(int n_cigar, const uint32_t *cigar) { return get_rlen(n_cigar, cigar); }
/* bns_depos (bntseq.h:92-94) */
API int64_t ref_bns_depos(int64_t l_pac, int64_t pos, int *is_rev) { bntseq_t bns; memset(&bns, 0, sizeof(bns)); bns.l_pac = l_pac; return bns_depos(&bns, pos, is_rev); }
/* hash_64 (utils.h:107) */
API uint64_t ref_hash_64(uint64_t k) { return hash_64(k); }

/* ---------------- sorting templates (ksort.h) ---------------- */
/* ks_introsort_64 / _64s / _128 / _192 are the reference's own instantiations (utils.c:46-50). */
API void ref_introsort_64(int64_t n, uint64_t *a) { ks_introsort_64((size_t)n, a); }
API void ref_introsort_64s(int64_t n, int64_t *a) { ks_introsort_64s((size_t)n, a); }
API void ref_introsort_128(int64_t n, uint64_t *a) { ks_introsort_128((size_t)n, (pair64_t*)a); }
API void ref_introsort_192(int64_t n, uint64_t *a) { ks_introsort_192((size_t)n, (trio64_t*)a); }
/* The reference instantiates ks_introsort with record types + "less-than on one field" comparators
 * (memchain.c:47,402; mem_alnreg.c:43-49,242-247).  The permutation produced depends only on the
 * sequence of comparison outcomes, so a (key, payload) record with the same template pins it. */
typedef struct { int64_t key; int64_t id; } ref_kv_t;
#define ref_kv_lt(a, b) ((a).key < (b).key)
KSORT_INIT(refkv, ref_kv_t, ref_kv_lt)
API void ref_introsort_kv(int64_t n, int64_t *kv) { ks_introsort_refkv((size_t)n, (ref_kv_t*)kv); }
#define ref_kv_gt(a, b) ((a).key > (b).key)
KSORT_INIT(refkvd, ref_kv_t, ref_kv_gt)
API void ref_introsort_kv_desc(int64_t n, int64_t *kv) { ks_introsort_refkvd((size_t)n, (ref_kv_t*)kv); }

/* ---------------- B-tree template (kbtree.h) ---------------- */
/* Same key size as mem_chain_t (72 bytes, memchain.h:84-94) so that the node fan-out
 * t = ((512-4-8)/(8+72)+1)>>1 = 3 equals the reference's (kbtree.h:75, KB_DEFAULT_SIZE=512). */
typedef struct { int64_t pos; int64_t id; char pad[56]; } ref_bk_t;
#define ref_bk_cmp(a, b) (((b).pos < (a).pos) - ((a).pos < (b).pos))
KBTREE_INIT(refbk, ref_bk_t, ref_bk_cmp)

API void *ref_bt_new(void) { return kb_init(refbk, KB_DEFAULT_SIZE); }
API int ref_bt_t(void *t) { return ((kbtree_t(refbk)*)t)->t; }
API void ref_bt_free(void *t) { kbtree_t(refbk) *b = (kbtree_t(refbk)*)t; kb_destroy(refbk, b); }
API void ref_bt_put(void *t, int64_t pos, int64_t id)
{
	ref_bk_t k; memset(&k, 0, sizeof(k)); k.pos = pos; k.id = id;
	kb_putp(refbk, (kbtree_t(refbk)*)t, &k);
}
/* kb_intervalp: returns id of lower (or -1) and of upper (or -1) */
API void ref_bt_interval(void *t, int64_t pos, int64_t out[2])
{
	ref_bk_t k, *lo = 0, *up = 0; memset(&k, 0, sizeof(k)); k.pos = pos;
	kb_intervalp(refbk, (kbtree_t(refbk)*)t, &k, &lo, &up);
	out[0] = lo? lo->id : -1; out[1] = up? up->id : -1;
}
/* in-order traversal with the iterator interface used by mem_chain (memchain.c:375-379) */
API int64_t ref_bt_traverse(void *t, int64_t *ids, int64_t cap)
{
	kbtree_t(refbk) *b = (kbtree_t(refbk)*)t;
	kbitr_t itr; int64_t n = 0;
	kb_itr_first(refbk, b, &itr);
	for (; kb_itr_valid(&itr); kb_itr_next(refbk, b, &itr)) { if (n < cap) ids[n] = kb_itr_key(ref_bk_t, &itr).id; ++n; }
	return n;
}
/* change the payload of the key found by interval lookup (mem_chain mutates chains in place through
 * the pointer kb_intervalp returns; positions never change) -- not needed for structure tests. */

/* ---------------- FASTQ chunk reader and SAM header (bwa.c) ---------------- */
#include <unistd.h>
#include "kseq.h"
KSEQ_DECLARE(gzFile)   /* instantiated by the reference itself in utils.c:53 */

API void *ref_kseq_open(const char *fn) { gzFile fp = gzopen(fn, "r"); return fp ? kseq_init(fp) : 0; }
API void ref_kseq_close(void *ks_) { kseq_t *ks = (kseq_t*)ks_; gzFile fp = ks->f->f; kseq_destroy(ks); gzclose(fp); }

/* kseq_read (kseq.h:182-222, the grammar under bis_bseq_read, bwa.c:817-850): next record flattened as
 *   name \t comment|* \t sequence \t qual|* \n        returns kseq_read's value (>= 0 length, -1 end of file, -2 truncated quality)
 * bis_bseq_read itself cannot be linked: bseq1_code_nt4 needs nst_nt4_table, defined in bntseq.c, which does not build
 * here (encode.h is not vendored).  The table is pinned separately as data (tests/golden/make_vectors.py reads it from the
 * reference source); the chunk rule and trim_readno around kseq_read stay restated. */
API int ref_kseq_next(void *ks_, char *out, size_t cap)
{
	kseq_t *ks = (kseq_t*)ks_;
	int l = kseq_read(ks);
	out[0] = 0;
	if (l < 0) return l;
	if (ks->name.l + ks->comment.l + ks->seq.l + ks->qual.l + 16 > cap) return -3;
	sprintf(out, "%s\t%s\t%s\t%s\n", ks->name.s, ks->comment.l ? ks->comment.s : "*", ks->seq.s, ks->qual.l ? ks->qual.s : "*");
	return l;
}

/* bwa_print_sam_hdr (bwa.c:654-684) prints to stdout: captured through a temporary file */
extern char *bwa_pg;
API int ref_sam_hdr(int n_seqs, const char **names, const int *lens, const char *hdr_line, const char *pg, char *out, size_t cap)
{
	bntseq_t bns;
	FILE *tmp = tmpfile();
	int saved, i;
	long sz;
	if (!tmp) return -1;
	memset(&bns, 0, sizeof(bns));
	bns.n_seqs = n_seqs;
	bns.anns = (bntann1_t*)calloc((size_t)n_seqs + 1, sizeof(bntann1_t));
	for (i = 0; i < n_seqs; ++i) { bns.anns[i].name = (char*)names[i]; bns.anns[i].len = lens[i]; }
	bwa_pg = (char*)pg;
	fflush(stdout);
	saved = dup(1);
	dup2(fileno(tmp), 1);
	bwa_print_sam_hdr(&bns, hdr_line && hdr_line[0] ? hdr_line : 0);
	fflush(stdout);
	dup2(saved, 1); close(saved);
	bwa_pg = 0;
	free(bns.anns);
	sz = ftell(tmp);
	if (sz < 0) { fseek(tmp, 0, SEEK_END); sz = ftell(tmp); }
	fseek(tmp, 0, SEEK_END); sz = ftell(tmp); rewind(tmp);
	if ((size_t)sz + 1 > cap) { fclose(tmp); return -1; }
	if (fread(out, 1, (size_t)sz, tmp) != (size_t)sz) { fclose(tmp); return -1; }
	out[sz] = 0;
	fclose(tmp);
	return (int)sz;
}
