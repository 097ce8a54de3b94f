/* pipeline.c -- D1: the chunk driver, mem_process_seqs (lib/aln/bwamem.c:432-476) re-shaped for a
 * batch device.
 *
 * The reference maps reads one at a time inside kt_for workers.  Here a chunk moves through the
 * stages as whole batches so that every kernel launch carries the work of all reads:
 *
 *   clip -> [K1+K2 seed -> K3 SA + chaining + chain filter + extension, fused on the device (k_regions)]
 *        -> for the strand searches the device declined, and on backends without the fused pass:
 *             [K1+K2 seed] -> [K3 SA] -> chain/filter (host) -> [K5 seed SW, long reads only]
 *             -> rounds of [K4 extend] driven by the per-task state machines (extend.c)
 *        -> merge/dedup (host, [K6 score-only] for concatenation tests)
 *        -> insert-size statistics (host reduction over the chunk)
 *        -> [K5 mate rescue] -> primary marking, pairing, MAPQ (host)
 *        -> [K6 CIGAR] -> MD/NM + SAM text (host)
 *
 * Device stages go through bsx_backend_t; the product binds it to the HIP kernels only.
 * Host stages are parallelised over reads with bsx_parallel_for; no result depends on scheduling.
 */
#include <math.h>
#include <pthread.h>
#include <sys/time.h>
#include "align_types.h"
#include "tune.h"
#include "pipeline.h"

static double now_s(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + tv.tv_usec * 1e-6; }

static bsx_phase_stats_t g_last_stats;   /* of the last chunk that completed */
BSX_API void bsx_last_phase_stats(bsx_phase_stats_t *out) { *out = g_last_stats; }


/* ------------------------------------------------------------------ read clipping (bwamem.c:218-303) */
static const uint8_t *find_bytes(const uint8_t *hay, size_t hlen, const uint8_t *needle, size_t nlen)
{
	size_t i;
	if (!nlen || hlen < nlen) return 0;
	for (i = 0; i + nlen <= hlen; ++i)
		if (hay[i] == needle[0] && memcmp(hay + i, needle, nlen) == 0) return hay + i;
	return 0;
}

static void clip_read(bsx_read_t *seq, const uint8_t *adaptor, int l_adaptor, const bsx_opt_t *opt)
{
	if (adaptor == 0) seq->l_adaptor = 0;
	else { /* full adaptor anywhere, else its longest prefix that is a suffix of the read */
		const uint8_t *hit = find_bytes(seq->seq, (size_t)seq->l_seq, adaptor, (size_t)l_adaptor);
		if (hit) seq->l_adaptor = seq->l_seq - (int)(hit - seq->seq);
		else {
			int i;
			for (i = l_adaptor - 1; i; --i)
				if (i <= seq->l_seq && memcmp(seq->seq + seq->l_seq - i, adaptor, (size_t)i) == 0) break;
			seq->l_adaptor = i;
		}
	}
	seq->clip5 = opt->clip5;
	seq->clip3 = opt->clip3 + seq->l_adaptor;
	if (seq->qual) { /* clip_read_by_quality, bwamem.c:275-283 */
		for (; seq->clip5 < seq->l_seq - seq->clip3; seq->clip5++)
			if (seq->qual[seq->clip5] >= opt->min_base_qual + 33) break;
		for (; seq->l_seq - seq->clip3 >= seq->clip5; seq->clip3++)
			if (seq->l_seq - seq->clip3 - 1 < 0 || seq->qual[seq->l_seq - seq->clip3 - 1] >= opt->min_base_qual + 33) break;
	}
	seq->seq0 = seq->seq;
	seq->l_seq0 = seq->l_seq;
	seq->seq += seq->clip5;
	seq->l_seq = seq->l_seq - seq->clip3 - seq->clip5;
	if (seq->l_seq < 0) seq->l_seq = 0;
}

static int pair_names_ok(const char *n1, const char *n2)   /* check_paired_read_names, bwamem.c:210-216 */
{
	size_t l;
	if (strcmp(n1, n2) == 0) return 1;
	l = strlen(n1);
	if (l > 0 && n1[l - 1] == '1' && strlen(n2) >= l && n2[l - 1] == '2' && strncmp(n1, n2, l - 1) == 0) return 1;
	return 0;
}

/* test hooks: read_clipping (bwamem.c:286-303) and check_paired_read_names (bwamem.c:210-216) as this file has them */
BSX_API void bsx_hook_clip_read(const bsx_opt_t *opt, int l_seq, uint8_t *seq, char *qual, const uint8_t *adaptor, int l_adaptor, int out[5])
{
	bsx_read_t s;
	memset(&s, 0, sizeof(s));
	s.l_seq = l_seq; s.seq = seq; s.qual = qual;
	clip_read(&s, adaptor, l_adaptor, opt);
	out[0] = s.l_adaptor; out[1] = s.clip5; out[2] = s.clip3; out[3] = s.l_seq; out[4] = (int)(s.seq - s.seq0);
}
BSX_API int bsx_hook_pair_names_ok(const char *n1, const char *n2) { return pair_names_ok(n1, n2); }

/* ------------------------------------------------------------------ chunk state */
typedef struct {
	const bsx_backend_t *be;
	const bsx_opt_t *opt;
	const bsx_index_t *idx;
	int n, is_pe, nt;
	int64_t n_processed;
	int64_t seq;         /* order of the chunks pushed through streams ($BSX_STREAM_WHOLE_CHUNK: back halves start in this order) */
	int ordered;         /* decided once, when the chunk is pushed: its back half runs on its own thread, in its turn (seq is its ticket) */
	int64_t local0;      /* these reads are a slice of a chunk (several GPUs sharing it): the index of the first one within the chunk */
	bsx_read_t *reads;
	uint32_t *roff;
	/* tasks */
	int n_tasks;
	c2r_t *tasks;
	/* strand searches the host chains itself (all of them without a device regions pass): h -> task */
	int n_host, *hmap, n_pending;
	bsx_region_t *dregs; int64_t dregs_cap, *dreg_off; int32_t *dreg_n;
	int32_t *dd_n; uint8_t *dd_idx; int dd_cap;   /* C5 done by the backend (regions_dedup): per read, the regions kept and their order; dd_n < 0: here */
	int64_t *dd_loff; uint16_t *dd_lidx; int64_t dd_lcap;   /* ... the reads with more than dd_cap regions (regions_dedup2): where a read's 16-bit indices start in dd_lidx, or -1 */
	int *read_task0;             /* first task of each read; read_task0[n] = n_tasks */
	bsx_intv_t *intv; int64_t intv_cap; int64_t *intv_off;
	uint64_t *pos; int64_t *ipos_off;   /* per interval: its occurrences' positions */
	int *need_more;              /* per task: 0 or 1+interval */
	uint64_t **xpos; int64_t **xpos_off;  /* private re-lookups for tasks that needed more */
	bsx_btree_t **trees;         /* per thread */
	reg_v *regs;                 /* per read */
	bsx_pestat_t pes;
	uint8_t *buf; bsx_seed_task_t *stasks; bsx_sa_job_t *sa_jobs;
	/* life cycle: front half (seeding .. regions per strand search), back half (merge .. SAM) */
	bsx_backend_t be_copy;
	const bsx_pestat_t *pes0; bsx_pestat_t pes0_copy;
	int arena_set, rc;
	pthread_t th; int th_live;   /* the thread running the front half (stream mode) */
	int back_done;               /* ... which ran the back half too */
	int merged;                  /* the per-read merge (C5) already ran at the end of the front half, on its thread */
	int64_t token; int max_len;  /* a number of its own (the backend keeps per-chunk uploads by it); the longest clipped read */
	double t_begin, t_front_end;
	bsx_phase_stats_t st;
} chunk_t;

/* ------------------------------------------------------------------ chaining */
static void chain_worker(void *data, long t, int tid)   /* t indexes the host-side task list */
{
	chunk_t *C = (chunk_t*)data;
	c2r_t *T = &C->tasks[C->hmap[t]];
	const bsx_intv_t *iv = C->intv + C->intv_off[t];
	int n_iv = (int)(C->intv_off[t + 1] - C->intv_off[t]);
	int rc;
	if (C->need_more[t] < 0) return; /* already chained */
	bsx_chain_free(&T->chains);
	if (C->xpos[t]) rc = bsx_chain_build(C->opt, &C->idx->ref, T->l_query, T->parent, iv, n_iv, C->xpos[t], C->xpos_off[t], C->trees[tid], &T->chains);
	else {
		/* this task's intervals are consecutive in the global interval list: rebase the offsets */
		int64_t base = C->ipos_off[C->intv_off[t]];
		int64_t *off = (int64_t*)alloca(sizeof(int64_t) * ((size_t)n_iv + 1));
		int i;
		for (i = 0; i <= n_iv; ++i) off[i] = C->ipos_off[C->intv_off[t] + i] - base;
		rc = bsx_chain_build(C->opt, &C->idx->ref, T->l_query, T->parent, iv, n_iv, C->pos + base, off, C->trees[tid], &T->chains);
	}
	if (rc > 0) { C->need_more[t] = rc; return; }
	C->need_more[t] = -1;
	bsx_chain_filter(C->opt, &T->chains);
}

/* ------------------------------------------------------------------ long-read seed filter (memchain.c:501-568) */
#define MEM_SHORT_EXT 50
#define MEM_SHORT_LEN 200
#define MEM_HSP_COEF 1.1f
#define MEM_MINSC_COEF 5.5f
#define MEM_SEEDSW_COEF 0.05f

static int seed_sw_active(const bsx_opt_t *opt, int l_query, int *min_HSP_score)
{
	double min_l = opt->min_chain_weight ? MEM_HSP_COEF * opt->min_chain_weight : MEM_MINSC_COEF * log(l_query);
	if (min_l > MEM_SEEDSW_COEF * l_query) return 0;
	*min_HSP_score = (int)(opt->a * min_l + .499);
	return 1;
}

/* mem_seed_sw window; returns 0 if the reference would return -1 without running SW */
static int seed_sw_job(const chunk_t *C, const c2r_t *T, const seed_t *s, bsx_sw_job_t *j)
{
	int qb, qe;
	int64_t rb, re, mid, l_pac = C->idx->ref.l_pac;
	if (s->len >= MEM_SHORT_LEN) return 0;
	qb = s->qbeg; qe = s->qbeg + s->len;
	rb = s->rbeg; re = s->rbeg + s->len;
	mid = (rb + re) >> 1;
	qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
	qe += MEM_SHORT_EXT; qe = qe < T->l_query ? qe : T->l_query;
	rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
	re += MEM_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
	if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return 0;
	bsx_fetch_span(&C->idx->ref, &rb, mid, &re);
	memset(j, 0, sizeof(*j));
	j->qoff = T->qoff + (uint32_t)qb; j->qlen = qe - qb; j->qdir = 1;
	j->tpos = rb; j->tlen = (int32_t)(re - rb); j->tdir = 1;
	/* the reference asks for KSW_XSTART but only uses the score, which the forward pass fixes */
	j->xtra = 0;
	j->use_ct = (uint8_t)T->parent;
	return j->tlen > 0 && j->qlen > 0;
}

static int filter_chained_seeds(chunk_t *C)
{
	BSX_VEC(bsx_sw_job_t) jobs;
	bsx_sw_res_t *res = 0;
	int t, rc = BSX_OK, minsc;
	size_t u, j, k, cur = 0;
	bsx_vec_init(jobs);
	for (t = 0; t < C->n_tasks; ++t) {
		c2r_t *T = &C->tasks[t];
		if (!seed_sw_active(C->opt, T->l_query, &minsc)) continue;
		for (u = 0; u < T->chains.n; ++u)
			for (j = 0; j < T->chains.a[u].seeds.n; ++j) {
				bsx_sw_job_t jb;
				if (seed_sw_job(C, T, &T->chains.a[u].seeds.a[j], &jb)) bsx_vec_push(jobs, jb);
			}
	}
	if (jobs.n) {
		res = (bsx_sw_res_t*)malloc(sizeof(*res) * jobs.n);
		rc = C->be->sw_batch(C->be->ctx, (int64_t)jobs.n, jobs.a, res);
		C->st.n_sw_jobs += (int64_t)jobs.n;
	}
	if (rc == BSX_OK) {
		for (t = 0; t < C->n_tasks; ++t) {
			c2r_t *T = &C->tasks[t];
			if (!seed_sw_active(C->opt, T->l_query, &minsc)) continue;
			for (u = 0; u < T->chains.n; ++u) {
				chain_t *c = &T->chains.a[u];
				for (j = k = 0; j < c->seeds.n; ++j) {
					seed_t *s = &c->seeds.a[j];
					bsx_sw_job_t jb;
					s->score = seed_sw_job(C, T, s, &jb) ? res[cur++].score : -1;
					if (s->score < 0 || s->score >= minsc) {
						s->score = s->score < 0 ? s->len * C->opt->a : s->score;
						c->seeds.a[k++] = *s;
					}
				}
				c->seeds.n = k;
			}
		}
	}
	free(res); bsx_vec_free(jobs);
	return rc;
}

/* test hook: mem_flt_chained_seeds (memchain.c:501-568) of one chain as this pipeline runs it (window, K5 batch on the given backend, filter):
 * seeds = (rbeg, qbeg, len) triples; on return keep[k] = index of the k-th surviving seed and score[k] its score; returns their number */
BSX_API int bsx_hook_flt_chained_seeds(const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx, int l_query, const uint8_t *query, int parent,
                                       int n_seeds, const int64_t *seeds, int *keep, int *score)
{
	chunk_t *C = (chunk_t*)calloc(1, sizeof(chunk_t));
	c2r_t *T = (c2r_t*)calloc(1, sizeof(c2r_t));
	chain_t ch;
	int k, n = -1;
	C->be_copy = *be; C->be = &C->be_copy; C->opt = opt; C->idx = idx; C->nt = 1; C->arena_set = -1; C->n_tasks = 1; C->tasks = T;
	T->l_query = l_query; T->qoff = 0; T->parent = parent; T->query = query;
	memset(&ch, 0, sizeof(ch));
	ch.seeds.a = (seed_t*)bsx_crealloc(0, 0, sizeof(seed_t) * (size_t)(n_seeds ? n_seeds : 1)); ch.seeds.n = ch.seeds.m = (size_t)n_seeds;
	for (k = 0; k < n_seeds; ++k) { ch.seeds.a[k].rbeg = seeds[3 * k]; ch.seeds.a[k].qbeg = (int32_t)seeds[3 * k + 1]; ch.seeds.a[k].len = (int32_t)seeds[3 * k + 2]; ch.seeds.a[k].score = k; /* carries the index */ }
	T->chains.a = &ch; T->chains.n = T->chains.m = 1;
	if (be->set_opt(be->ctx, opt) == BSX_OK && be->set_reads(be->ctx, query, (size_t)l_query) == BSX_OK) {
		/* the filter overwrites score: find the survivors by their coordinates (unique in the tests) */
		seed_t *orig = (seed_t*)malloc(sizeof(seed_t) * (size_t)(n_seeds ? n_seeds : 1));
		memcpy(orig, ch.seeds.a, sizeof(seed_t) * (size_t)n_seeds);
		if (filter_chained_seeds(C) == BSX_OK) {
			size_t u; int j;
			n = (int)ch.seeds.n;
			for (u = 0; u < ch.seeds.n; ++u) {
				keep[u] = -1;
				for (j = 0; j < n_seeds; ++j) if (orig[j].rbeg == ch.seeds.a[u].rbeg && orig[j].qbeg == ch.seeds.a[u].qbeg && orig[j].len == ch.seeds.a[u].len) { keep[u] = j; break; }
				score[u] = ch.seeds.a[u].score;
			}
		}
		free(orig);
	}
	bsx_cfree(ch.seeds.a);
	free(T); free(C);
	return n;
}

/* ------------------------------------------------------------------ extension rounds */
typedef struct { chunk_t *C; const int *active; const int *owner; const bsx_ext_res_t *res; } round_par_t;

static void advance_worker(void *data, long i, int tid)
{
	round_par_t *P = (round_par_t*)data;
	c2r_t *T = &P->C->tasks[P->active[i]];
	(void)tid;
	if (!T->done && !T->has_job) bsx_c2r_advance(P->C->opt, P->C->idx, T);
}
static void consume_worker(void *data, long i, int tid)
{
	round_par_t *P = (round_par_t*)data;
	(void)tid;
	bsx_c2r_consume(P->C->opt, P->C->idx, &P->C->tasks[P->owner[i]], &P->res[i]);
}

/* All strand searches advance in lock-step: one K4 batch per round.  Only tasks that still have
 * work stay on the active list, so the host cost of a round is proportional to its batch. */
static int extension_rounds(chunk_t *C)
{
	bsx_ext_job_t *jobs = (bsx_ext_job_t*)malloc(sizeof(bsx_ext_job_t) * ((size_t)C->n_tasks + 1));
	bsx_ext_res_t *res = (bsx_ext_res_t*)malloc(sizeof(bsx_ext_res_t) * ((size_t)C->n_tasks + 1));
	int *owner = (int*)malloc(sizeof(int) * ((size_t)C->n_tasks + 1));
	int *active = (int*)malloc(sizeof(int) * ((size_t)C->n_tasks + 1));
	int rc = BSX_OK, n_active = 0, t;
	round_par_t P;
	for (t = 0; t < C->n_tasks; ++t) if (C->tasks[t].chains.n) active[n_active++] = t; else C->tasks[t].done = 1;
	P.C = C; P.active = active; P.owner = owner; P.res = res;
	double ta = 0, tg = 0, tb = 0, tc = 0, tx;
	while (n_active > 0) {
		int i, nj = 0, na = 0;
		tx = now_s();
		/* a handful of strand searches: not worth the worker pool (which the other chunk's half may be using) */
		if (n_active < 64) for (i = 0; i < n_active; ++i) advance_worker(&P, i, 0);
		else bsx_parallel_for(C->nt, advance_worker, &P, n_active);
		ta += now_s() - tx; tx = now_s();
		for (i = 0; i < n_active; ++i) {
			c2r_t *T = &C->tasks[active[i]];
			if (T->has_job) { jobs[nj] = T->job; owner[nj++] = active[i]; active[na++] = active[i]; }
			else if (!T->done) active[na++] = active[i];
		}
		n_active = na;
		tg += now_s() - tx; tx = now_s();
		if (nj == 0) break;
		if ((rc = C->be->extend_batch(C->be->ctx, nj, jobs, res)) != BSX_OK) break;
		tb += now_s() - tx; tx = now_s();
		C->st.n_ext_jobs += nj; ++C->st.n_ext_rounds;
		if (nj < 64) for (i = 0; i < nj; ++i) consume_worker(&P, i, 0);
		else bsx_parallel_for(C->nt, consume_worker, &P, nj);
		tc += now_s() - tx;
	}
	if (bsx_phases()) fprintf(stderr, "[M::extend] advance %.3f gather %.3f batch %.3f consume %.3f (%ld jobs, %ld rounds)\n", ta, tg, tb, tc, (long)C->st.n_ext_jobs, (long)C->st.n_ext_rounds);
	free(jobs); free(res); free(owner); free(active);
	return rc;
}

/* ------------------------------------------------------------------ merge / dedup with batched concatenation tests */
typedef struct { int64_t rb, re; int qb, qe, w, parent, score; } gcache_t;
typedef struct {
	chunk_t *C; int read;
	BSX_VEC(gcache_t) cache;     /* known scores */
	BSX_VEC(gcache_t) wanted;    /* requests raised by the last attempt */
} merge_ud_t;

static int merge_score_fn(void *ud_, const reg_t *a, const reg_t *b, int w, int *score)
{
	merge_ud_t *U = (merge_ud_t*)ud_;
	gcache_t key;
	size_t i;
	key.rb = a->rb; key.re = b->re; key.qb = a->qb; key.qe = b->qe; key.w = w; key.parent = a->parent; key.score = 0;
	for (i = 0; i < U->cache.n; ++i) {
		const gcache_t *c = &U->cache.a[i];
		if (c->rb == key.rb && c->re == key.re && c->qb == key.qb && c->qe == key.qe && c->w == key.w && c->parent == key.parent) { *score = c->score; return 0; }
	}
	bsx_vec_push(U->wanted, key);
	return 1;
}

typedef struct { chunk_t *C; merge_ud_t *ud; int *pending; } merge_par_t;

/* a region as the device left it (bsx_region_t, 56 bytes) into the host's record */
static inline void reg_from_device(reg_t *r, const bsx_region_t *d)
{
	memset(r, 0, sizeof(*r));
	r->rb = d->rb; r->re = d->re; r->qb = d->qb; r->qe = d->qe; r->rid = d->rid; r->score = d->score; r->truesc = d->truesc;
	r->w = d->w; r->seedcov = d->seedcov; r->seedlen0 = d->seedlen0; r->frac_rep = d->frac_rep; r->bss = d->bss; r->parent = d->parent;
}

static void regs_copy(reg_v *dst, const reg_v *src)
{
	dst->n = src->n; dst->n_pri = src->n_pri;
	if (dst->m < src->n) { dst->a = (reg_t*)bsx_crealloc(dst->a, 0, sizeof(reg_t) * (src->n + 4)); dst->m = src->n + 4; }
	if (src->n) memcpy(dst->a, src->a, sizeof(reg_t) * src->n);
}

static void merge_worker(void *data, long i, int tid)
{
	merge_par_t *P = (merge_par_t*)data;
	chunk_t *C = P->C;
	reg_v *regs = &C->regs[i];
	int missing = 0;
	size_t k;
	(void)tid;
	if (!P->pending[i]) return;
	if (C->dd_n && C->dd_n[i] >= 0) { /* sorted and de-duplicated on the device: what is left of the concatenation, in order */
		const uint8_t *ix = C->dd_idx + (size_t)i * (size_t)C->dd_cap;
		const uint16_t *lx = C->dd_loff && C->dd_loff[i] >= 0 ? C->dd_lidx + C->dd_loff[i] : 0;   /* a long list (k_dedup_long): 16-bit indices */
		int t, kk, m = C->dd_n[i]; size_t tot = 0;   /* (every strand search of such a read finished on the device: dreg_n >= 0) */
		for (t = C->read_task0[i]; t < C->read_task0[i + 1]; ++t) tot += (size_t)C->dreg_n[t];
		if (regs->m < (size_t)m) { regs->a = (reg_t*)bsx_crealloc(regs->a, 0, sizeof(reg_t) * ((size_t)m + 2)); regs->m = (size_t)m + 2; }
		regs->n = (size_t)m; regs->n_pri = 0;
		for (kk = 0; kk < m; ++kk) {
			size_t li = lx ? lx[kk] : ix[kk];
			for (t = C->read_task0[i]; li >= (size_t)C->dreg_n[t]; ++t) li -= (size_t)C->dreg_n[t];
			reg_from_device(&regs->a[kk], &C->dregs[C->dreg_off[t] + (int64_t)li]);
			regs->a[kk].n_comp = tot > 1 ? 1 : 0;   /* mem_alnreg.c:114,118 */
		}
	} else { /* here: (re)start from the regions of the read's strand searches, concatenated in call order */
		int t; size_t tot = 0;
		for (t = C->read_task0[i]; t < C->read_task0[i + 1]; ++t) tot += C->tasks[t].regs.n;
		if (regs->m < tot) { regs->a = (reg_t*)bsx_crealloc(regs->a, 0, sizeof(reg_t) * (tot + 2)); regs->m = tot + 2; }
		regs->n = 0; regs->n_pri = 0;
		for (t = C->read_task0[i]; t < C->read_task0[i + 1]; ++t) {
			const c2r_t *T = &C->tasks[t];
			if (T->regs.n) { memcpy(regs->a + regs->n, T->regs.a, sizeof(reg_t) * T->regs.n); regs->n += T->regs.n; }
		}
		P->ud[i].wanted.n = 0;
		bsx_regs_sort_dedup(C->opt, &C->idx->ref, 1, regs, merge_score_fn, &P->ud[i], &missing);
		if (missing) return; /* retried once the scores have been computed */
	}
	P->pending[i] = 0;
	/* mem_test_and_remove_exact (mem_alnreg.c:205-211) */
	if ((C->opt->flag & BSX_F_SELF_OVLP) && regs->n > 0 && regs->a[0].truesc == C->reads[i].l_seq * C->opt->a) {
		memmove(regs->a, regs->a + 1, (regs->n - 1) * sizeof(reg_t));
		regs->n--;
	}
	for (k = 0; k < regs->n; ++k) {
		reg_t *p = &regs->a[k];
		if (p->rid >= 0 && C->idx->ref.anns[p->rid].is_alt) p->is_alt = 1;
	}
}

/* exclusive prefix sum of per-item counts: off[0..n], returns the total */
static int64_t prefix_counts(long n, const int *cnt, int64_t *off)
{
	long i; int64_t tot = 0;
	for (i = 0; i < n; ++i) { off[i] = tot; tot += cnt[i]; }
	off[n] = tot;
	return tot;
}

typedef struct { merge_par_t *P; int *cnt; int64_t *off; bsx_glb_job_t *jobs; const bsx_glb_res_t *res; } merge_aux_t;

static void merge_init_worker(void *data, long i, int tid)
{
	merge_par_t *P = (merge_par_t*)data;
	(void)tid;
	P->ud[i].C = P->C; P->ud[i].read = (int)i;
	P->pending[i] = 1;
}
static void merge_count_worker(void *data, long i, int tid)
{
	merge_aux_t *A = (merge_aux_t*)data;
	(void)tid;
	A->cnt[i] = A->P->pending[i] ? (int)A->P->ud[i].wanted.n : 0;
}
static void merge_jobs_worker(void *data, long i, int tid)
{
	merge_aux_t *A = (merge_aux_t*)data;
	chunk_t *C = A->P->C;
	int k;
	(void)tid;
	for (k = 0; k < A->cnt[i]; ++k) {
		const gcache_t *g = &A->P->ud[i].wanted.a[k];
		bsx_glb_job_t j;
		int rev = g->rb >= C->idx->ref.l_pac;
		memset(&j, 0, sizeof(j));
		j.qlen = g->qe - g->qb; j.tlen = (int32_t)(g->re - g->rb);
		j.qoff = C->roff[i] + (uint32_t)(rev ? g->qe - 1 : g->qb); j.qdir = rev ? -1 : 1;
		j.tpos = rev ? g->re - 1 : g->rb; j.tdir = rev ? -1 : 1;
		j.w0 = g->w; j.w_max = g->w > C->opt->w << 2 ? g->w : C->opt->w << 2; j.n_try = 1; j.use_ct = (uint8_t)g->parent; j.want_cigar = 0;
		A->jobs[A->off[i] + k] = j;
	}
}
static void merge_scores_worker(void *data, long i, int tid)
{
	merge_aux_t *A = (merge_aux_t*)data;
	int k;
	(void)tid;
	for (k = 0; k < A->cnt[i]; ++k) { gcache_t g = A->P->ud[i].wanted.a[k]; g.score = A->res[A->off[i] + k].score; bsx_vec_push(A->P->ud[i].cache, g); }
}
static void merge_free_worker(void *data, long i, int tid)
{
	merge_par_t *P = (merge_par_t*)data;
	(void)tid;
	bsx_vec_free(P->ud[i].cache); bsx_vec_free(P->ud[i].wanted);
}

static int merge_regions(chunk_t *C)
{
	int n = C->n, rc = BSX_OK, round;
	merge_par_t P;
	merge_aux_t A;
	P.C = C;
	P.ud = (merge_ud_t*)bsx_par_calloc(C->nt, (size_t)n, sizeof(merge_ud_t));
	P.pending = (int*)malloc(sizeof(int) * (n ? n : 1));
	A.P = &P; A.cnt = (int*)malloc(sizeof(int) * ((size_t)n + 1)); A.off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n + 1));
	bsx_parallel_for(C->nt, merge_init_worker, &P, n);
	for (round = 0; round < 64; ++round) {
		bsx_glb_res_t *res;
		int64_t nj;
		bsx_parallel_for(C->nt, merge_worker, &P, n);
		bsx_parallel_for(C->nt, merge_count_worker, &A, n);
		nj = prefix_counts(n, A.cnt, A.off);
		if (nj == 0) break;
		A.jobs = (bsx_glb_job_t*)malloc(sizeof(bsx_glb_job_t) * (size_t)nj);
		res = (bsx_glb_res_t*)malloc(sizeof(*res) * (size_t)nj);
		bsx_parallel_for(C->nt, merge_jobs_worker, &A, n);
		rc = C->be->global_batch(C->be->ctx, nj, A.jobs, res, 0, 0);
		C->st.n_glb_jobs += nj;
		A.res = res;
		if (rc == BSX_OK) bsx_parallel_for(C->nt, merge_scores_worker, &A, n);
		free(res); free(A.jobs);
		if (rc != BSX_OK) break;
	}
	bsx_parallel_for(C->nt, merge_free_worker, &P, n);
	free(P.ud); free(P.pending); free(A.cnt); free(A.off);
	return rc;
}

/* ------------------------------------------------------------------ mate rescue (mem_alnreg.c:395-513) */
typedef struct { int i, j; bsx_sw_job_t job; bsx_sw_res_t res; int have; } msw_slot_t;
typedef struct {
	reg_v saved[2]; int have_saved;
	BSX_VEC(msw_slot_t) slots;   /* in the order the candidates (i, j) are visited: every pass visits them in that order */
	int pending, cur;            /* cur: where the pass's next candidate is expected in slots */
} msw_pair_t;

static int g_msw_prof = 0;                       /* $BSX_PHASES: where mate rescue's host time goes */
static int64_t g_msw_ns = 0, g_msw_calls = 0, g_msw_elems = 0, g_msw_replay_ns = 0;
static int no_glb(void *ud, const reg_t *a, const reg_t *b, int w, int *score) { (void)ud; (void)a; (void)b; (void)w; *score = 0; return 0; }

/* one mem_alnreg_matesw_core; returns 1 if its SW result was needed but is not available yet */
static int matesw_core(chunk_t *C, msw_pair_t *M, int pi, int i, int j, const reg_t *reg, int mate_read, reg_v *mregs, int apply, bsx_regs_inc_t *inc)
{
	const bsx_opt_t *opt = C->opt;
	const bsx_refmeta_t *ref = &C->idx->ref;
	int64_t l_pac = ref->l_pac, rb, re, is;
	int l_ms = C->reads[mate_read].l_seq, rid = -1, parent, xtra;
	size_t k;
	msw_slot_t *slot = 0;
	(void)pi;
	for (k = 0; k < mregs->n; ++k)
		if (bsx_reg_isize(ref, reg, &mregs->a[k], &is) && is >= C->pes.low && is <= C->pes.high) return 0;
	rb = reg->rb + C->pes.low - l_ms; rb = rb > 0 ? rb : 0;
	re = reg->rb + C->pes.high; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < re) rid = bsx_fetch_span(ref, &rb, (rb + re) >> 1, &re);
	if (reg->rid != rid || re - rb < opt->min_seed_len) return 0;
	if (l_ms <= 0) return 0;   /* a mate clipped away entirely: ksw_align2 of an empty query scores 0 and reports no start (ksw.c:343-365), so nothing is added */
	parent = reg->bss ^ (reg->rb < l_pac);
	xtra = BSX_KSW_XSUBO | BSX_KSW_XSTART | (l_ms * opt->a < 250 ? BSX_KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
	/* (a pass asks for its candidates' slots in the order an earlier pass made them: the next one is almost always the one wanted -- the search
	 * over all slots of the pair was 5 000 comparisons a pair for a read inside a repeat family) */
	while ((size_t)M->cur < M->slots.n && (M->slots.a[M->cur].i < i || (M->slots.a[M->cur].i == i && M->slots.a[M->cur].j < j))) ++M->cur;   /* (slots of candidates this pass skipped) */
	if ((size_t)M->cur < M->slots.n && M->slots.a[M->cur].i == i && M->slots.a[M->cur].j == j) slot = &M->slots.a[M->cur++];
	else for (k = 0; k < M->slots.n; ++k) if (M->slots.a[k].i == i && M->slots.a[k].j == j) { slot = &M->slots.a[k]; M->cur = (int)k + 1; break; }
	if (!slot) {
		msw_slot_t s;
		memset(&s, 0, sizeof(s));
		s.i = i; s.j = j;
		s.job.qoff = C->roff[mate_read] + (uint32_t)l_ms - 1; s.job.qlen = l_ms; s.job.qdir = -1; s.job.qcomp = 1;  /* reverse complement of the mate */
		s.job.tpos = rb; s.job.tlen = (int32_t)(re - rb); s.job.tdir = 1;
		s.job.xtra = xtra; s.job.use_ct = (uint8_t)(parent ? 0 : 1);   /* the mate is on the other converted strand */
		bsx_vec_push(M->slots, s);
		return 1;
	}
	if (!slot->have) return 1;
	if (!apply) return 0;
	if (slot->res.score >= opt->min_seed_len && slot->res.qb >= 0) {
		reg_t b;
		int64_t d1, d2;
		int ins, pos;
		int missing = 0;
		memset(&b, 0, sizeof(b));
		b.rid = reg->rid; b.is_alt = reg->is_alt;
		b.qb = l_ms - (slot->res.qe + 1); b.qe = l_ms - slot->res.qb;
		b.rb = (l_pac << 1) - (rb + slot->res.te + 1); b.re = (l_pac << 1) - (rb + slot->res.tb);
		b.score = slot->res.score; b.csub = slot->res.score2; b.secondary = -1;
		d1 = b.re - b.rb; d2 = b.qe - b.qb;
		b.seedcov = (int)((d1 < d2 ? d1 : d2) >> 1);
		b.bss = reg->bss; b.parent = (uint8_t)(1 - parent);
		/* keep the mate list ordered by score, then de-duplicate without merging (mem_alnreg.c:478-488) */
		(void)missing; (void)ins; (void)pos;
		if (g_msw_prof) {
			double t0 = now_s();
			size_t n0 = mregs->n;
			bsx_regs_insert_dedup(opt, ref, mregs, &b, inc, no_glb);
			__atomic_fetch_add(&g_msw_ns, (int64_t)((now_s() - t0) * 1e9), __ATOMIC_RELAXED);
			__atomic_fetch_add(&g_msw_calls, 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_msw_elems, (int64_t)n0, __ATOMIC_RELAXED);
		} else bsx_regs_insert_dedup(opt, ref, mregs, &b, inc, no_glb);
	}
	return 0;
}

/* mem_alnreg_matesw for pair pi, replayed on the saved lists; 0 when complete */
static int matesw_replay(chunk_t *C, msw_pair_t *M, int pi)
{
	const bsx_opt_t *opt = C->opt;
	reg_v *pair = &C->regs[pi << 1];
	reg_v good[2];
	reg_t small[2][8];
	bsx_regs_inc_t inc[2];   /* the order by end of each mate list between the hits added to it */
	int i, missing = 0;
	size_t j;
	memset(inc, 0, sizeof(inc));
	/* The first pass only collects SW requests: a request returns before anything is applied, so the lists are still
	 * the originals.  They are saved when the first results are about to be applied, and every later pass (a rescued
	 * hit can change what the following candidates see) restarts from that copy. */
	if (M->slots.n) {
		if (!M->have_saved) { regs_copy(&M->saved[0], &pair[0]); regs_copy(&M->saved[1], &pair[1]); M->have_saved = 1; }
		else { regs_copy(&pair[0], &M->saved[0]); regs_copy(&pair[1], &M->saved[1]); }
	}
	memset(good, 0, sizeof(good));
	M->cur = 0;
	for (i = 0; i < 2; ++i) {
		good[i].a = small[i]; good[i].m = 8;
		/* (the reference copies every region within pen_unpaired of the best and then looks at the first max_matesw of the copies: only those
		 * are copied here -- a read inside a repeat family has two hundred) */
		for (j = 0; j < pair[i].n && (int)good[i].n < opt->max_matesw; ++j)
			if (pair[i].a[j].score >= pair[i].a[0].score - opt->pen_unpaired) {
				if (good[i].n == good[i].m) {
					reg_t *na = (reg_t*)malloc(sizeof(reg_t) * (good[i].m << 1));
					memcpy(na, good[i].a, sizeof(reg_t) * good[i].n);
					if (good[i].a != small[i]) free(good[i].a);
					good[i].a = na; good[i].m <<= 1;
				}
				good[i].a[good[i].n++] = pair[i].a[j];
			}
	}
	for (i = 0; i < 2; ++i)
		for (j = 0; j < good[i].n && (int)j < opt->max_matesw; ++j)
			/* once a result is missing, keep walking (without applying) only to collect further requests */
			missing |= matesw_core(C, M, pi, i, (int)j, &good[i].a[j], (pi << 1) | !i, &pair[!i], !missing, &inc[!i]);
	bsx_regs_inc_free(&inc[0]); bsx_regs_inc_free(&inc[1]);
	if (good[0].a != small[0]) free(good[0].a);
	if (good[1].a != small[1]) free(good[1].a);
	return missing;
}

/* (the pairs [p0, p0 + np) of the chunk: M, cnt and off are indexed from p0) */
typedef struct { chunk_t *C; msw_pair_t *M; int *cnt; int64_t *off; bsx_sw_job_t *jobs; const bsx_sw_res_t *res; int p0; } msw_par_t;
static void msw_worker(void *data, long pi, int tid)
{
	msw_par_t *P = (msw_par_t*)data;
	(void)tid;
	if (P->M[pi].pending) {
		double t0 = g_msw_prof ? now_s() : 0;
		P->M[pi].pending = matesw_replay(P->C, &P->M[pi], P->p0 + (int)pi);
		if (g_msw_prof) __atomic_fetch_add(&g_msw_replay_ns, (int64_t)((now_s() - t0) * 1e9), __ATOMIC_RELAXED);
	}
}
static void msw_init_worker(void *data, long pi, int tid)
{
	msw_par_t *P = (msw_par_t*)data;
	(void)tid;
	P->M[pi].pending = 1;
}
static void msw_count_worker(void *data, long pi, int tid)
{
	msw_par_t *P = (msw_par_t*)data;
	size_t k; int c = 0;
	(void)tid;
	if (P->M[pi].pending) for (k = 0; k < P->M[pi].slots.n; ++k) c += !P->M[pi].slots.a[k].have;
	P->cnt[pi] = c;
}
static void msw_jobs_worker(void *data, long pi, int tid)
{
	msw_par_t *P = (msw_par_t*)data;
	size_t k; int64_t at = P->off[pi];
	(void)tid;
	if (P->cnt[pi]) for (k = 0; k < P->M[pi].slots.n; ++k) if (!P->M[pi].slots.a[k].have) P->jobs[at++] = P->M[pi].slots.a[k].job;
}
static void msw_results_worker(void *data, long pi, int tid)
{
	msw_par_t *P = (msw_par_t*)data;
	size_t k; int64_t at = P->off[pi];
	(void)tid;
	if (P->cnt[pi]) for (k = 0; k < P->M[pi].slots.n; ++k) if (!P->M[pi].slots.a[k].have) { P->M[pi].slots.a[k].res = P->res[at++]; P->M[pi].slots.a[k].have = 1; }
}
static void msw_free_worker(void *data, long pi, int tid)
{
	msw_par_t *P = (msw_par_t*)data;
	(void)tid;
	bsx_cfree(P->M[pi].saved[0].a); bsx_cfree(P->M[pi].saved[1].a); bsx_vec_free(P->M[pi].slots);
}

/* the device's plan (k_msw.hip) into the pairs' slots: every alignment the first pass would have asked for is there with its result, in the
 * order the replay visits its candidates */
typedef struct { msw_pair_t *M; const bsx_msw_pair_t *tab; const bsx_sw_res_t *res; } msw_fill_t;
static void msw_fill_worker(void *data, long k, int tid)
{
	msw_fill_t *F = (msw_fill_t*)data;
	const bsx_msw_pair_t *T = &F->tab[k];
	int i, at = T->base;
	(void)tid;
	if (at < 0) return;   /* (left to the host's own plan: the first pass collects as before) */
	for (i = 0; i < 2; ++i) {
		uint64_t m = T->mask[i];
		while (m) {
			msw_slot_t s;
			memset(&s, 0, sizeof(s));
			s.i = i; s.j = __builtin_ctzll(m); s.have = 1; s.res = F->res[at++];
			bsx_vec_push(F->M[k].slots, s);
			m &= m - 1;
		}
	}
}

static int mate_rescue(chunk_t *C, int p0, int p1)   /* the pairs [p0, p1) */
{
	int np = p1 - p0, rc = BSX_OK, round;
	double t_batch = 0, t_all = now_s();
	int64_t n_planned = -1, n_later = 0; long n_pairs_host = 0;   /* ($BSX_PHASES) */
	msw_pair_t *M = (msw_pair_t*)bsx_par_calloc(C->nt, (size_t)np, sizeof(msw_pair_t));
	msw_par_t P;
	g_msw_prof = bsx_phases() != 0;
	P.C = C; P.M = M; P.p0 = p0;
	P.cnt = (int*)malloc(sizeof(int) * ((size_t)np + 1)); P.off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)np + 1));
	bsx_parallel_for(C->nt, msw_init_worker, &P, np);
	if (C->be->msw_plan && C->dd_n && C->n > 0 && np > 0) { /* the plan pass and its K5 batch on the device, where the lists are */
		bsx_msw_pair_t *tab = (bsx_msw_pair_t*)malloc(sizeof(bsx_msw_pair_t) * (size_t)np);
		bsx_sw_res_t *pres = 0; int64_t pcap = 0, nj = -1;
		double tb = now_s();
		rc = C->be->msw_plan(C->be->ctx, C->opt, &C->pes, C->token, C->n, C->n_tasks / C->n, C->roff, C->max_len, p0, p1, tab, &pres, &pcap, &nj);
		t_batch += now_s() - tb;
		if (rc == BSX_OK && nj >= 0) {
			msw_fill_t F; F.M = M; F.tab = tab; F.res = pres;
			bsx_parallel_for(C->nt, msw_fill_worker, &F, np);
			__atomic_fetch_add(&C->st.n_sw_jobs, nj, __ATOMIC_RELAXED);
			n_planned = nj;
			if (g_msw_prof) { long k; for (k = 0; k < np; ++k) n_pairs_host += tab[k].base < 0; }
		}
		free(tab); free(pres);
		if (rc != BSX_OK) { free(M); free(P.cnt); free(P.off); return rc; }
	}
	for (round = 0; round < 256; ++round) {
		bsx_sw_res_t *res;
		int64_t nj;
		bsx_parallel_for(C->nt, msw_worker, &P, np);
		bsx_parallel_for(C->nt, msw_count_worker, &P, np);
		nj = prefix_counts(np, P.cnt, P.off);
		if (nj == 0) break;
		n_later += nj;
		P.jobs = (bsx_sw_job_t*)malloc(sizeof(bsx_sw_job_t) * (size_t)nj);
		res = (bsx_sw_res_t*)malloc(sizeof(*res) * (size_t)nj);
		bsx_parallel_for(C->nt, msw_jobs_worker, &P, np);
		{ double tb = now_s(); rc = C->be->sw_batch(C->be->ctx, nj, P.jobs, res); t_batch += now_s() - tb; }
		__atomic_fetch_add(&C->st.n_sw_jobs, nj, __ATOMIC_RELAXED);
		P.res = res;
		if (rc == BSX_OK) bsx_parallel_for(C->nt, msw_results_worker, &P, np);
		free(res); free(P.jobs);
		if (rc != BSX_OK) break;
	}
	bsx_parallel_for(C->nt, msw_free_worker, &P, np);
	if (bsx_phases()) {
		fprintf(stderr, "[M::matesw] %d rounds, %.3f s in the K5 batches, %.3f s on the host | thread-seconds: %.2f in the per-pair passes, %.2f of them in %ld list sorts (%.1f regions each) | planned on the device: %lld alignments (%ld of %d pairs left to the host's plan), asked for by the replay afterwards: %lld\n",
		        round, t_batch, now_s() - t_all - t_batch, g_msw_replay_ns * 1e-9, g_msw_ns * 1e-9, (long)g_msw_calls, g_msw_calls ? (double)g_msw_elems / g_msw_calls : 0.0,
		        (long long)n_planned, n_pairs_host, np, (long long)n_later);
		g_msw_replay_ns = g_msw_ns = g_msw_calls = g_msw_elems = 0;
	}
	free(M); free(P.cnt); free(P.off);
	return rc;
}

/* test hook: mem_alnreg_matesw of n_reads / 2 pairs as this pipeline runs it (plan, K5 batch on the given backend, replay), on region
 * lists given as flat records: read i's regions are a[off[i] .. off[i+1]); the lists after rescue come back the same way */
#include "hook_types.h"
BSX_API int bsx_hook_mate_rescue(const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_pestat_t *pes, int n_reads,
                                 bsx_read_t *reads, const bsx_hook_reg_t *a, const int64_t *off, bsx_hook_reg_t *out, int64_t *out_off, int64_t out_cap)
{
	chunk_t *C = (chunk_t*)calloc(1, sizeof(chunk_t));
	int i, rc;
	size_t tot = 0, k;
	int64_t at = 0;
	C->be_copy = *be; C->be = &C->be_copy; C->opt = opt; C->idx = idx; C->n = n_reads; C->reads = reads; C->nt = 1; C->is_pe = 1; C->arena_set = -1;
	C->pes = *pes;
	C->roff = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n_reads + 1));
	for (i = 0; i < n_reads; ++i) { C->roff[i] = (uint32_t)tot; tot += (size_t)reads[i].l_seq; }
	C->roff[n_reads] = (uint32_t)tot;
	C->buf = (uint8_t*)malloc(tot + 16);
	for (i = 0; i < n_reads; ++i) if (reads[i].l_seq) memcpy(C->buf + C->roff[i], reads[i].seq, (size_t)reads[i].l_seq);
	C->regs = (reg_v*)calloc(n_reads ? n_reads : 1, sizeof(reg_v));
	for (i = 0; i < n_reads; ++i) {
		reg_v *v = &C->regs[i];
		v->n = (size_t)(off[i + 1] - off[i]); v->m = v->n + 4;
		v->a = (reg_t*)bsx_crealloc(0, 0, sizeof(reg_t) * v->m);
		for (k = 0; k < v->n; ++k) bsx_hook_to_reg(&a[off[i] + (int64_t)k], &v->a[k]);
	}
	rc = be->set_opt(be->ctx, opt);
	if (rc == BSX_OK) rc = be->set_reads(be->ctx, C->buf, tot);
	if (rc == BSX_OK) rc = mate_rescue(C, 0, n_reads >> 1);
	for (i = 0; i < n_reads && rc == BSX_OK; ++i) {
		out_off[i] = at;
		if (at + (int64_t)C->regs[i].n > out_cap) { rc = BSX_E_ARG; break; }
		for (k = 0; k < C->regs[i].n; ++k) bsx_hook_from_reg(&C->regs[i].a[k], &out[at++]);
	}
	out_off[n_reads] = at;
	for (i = 0; i < n_reads; ++i) bsx_cfree(C->regs[i].a);
	free(C->regs); free(C->roff); free(C->buf); free(C);
	return rc;
}

/* ------------------------------------------------------------------ output: plan -> K6 -> final */
typedef struct {
	chunk_t *C;
	samctx_t *ctx;        /* per unit (pair or single read) of the slice: unit u0 + k at ctx[k] */
	int final_pass, u0;
} out_par_t;

static void reset_flags(reg_v *r) { size_t k; for (k = 0; k < r->n; ++k) r->a[k].flag = 0; }

static void out_worker(void *data, long k_, int tid)
{
	out_par_t *P = (out_par_t*)data;
	chunk_t *C = P->C;
	samctx_t *ctx = &P->ctx[k_];
	const long u = P->u0 + k_;
	(void)tid;
	if (C->is_pe) {
		reg_v *pair = &C->regs[u << 1];
		if (!P->final_pass) { /* plan on a scratch copy */
			reg_v tmp[2];
			memset(tmp, 0, sizeof(tmp));
			bsx_mark_primary(C->opt, &pair[0], (C->local0 + (u << 1)) | 0);   /* ids are chunk-local for PE (bwamem.c:408,413): local0 = where a slice starts in its chunk */
			bsx_mark_primary(C->opt, &pair[1], (C->local0 + (u << 1)) | 1);
			reset_flags(&pair[0]); reset_flags(&pair[1]);
			regs_copy(&tmp[0], &pair[0]); regs_copy(&tmp[1], &pair[1]);
			ctx->plan = 1;
			bsx_reg2sam_pe(C->opt, C->idx, (uint64_t)((C->n_processed >> 1) + u), &C->reads[u << 1], tmp, &C->pes, ctx, bsx_rg_id);
			bsx_cfree(tmp[0].a); bsx_cfree(tmp[1].a);
		} else {
			ctx->plan = 0;
			bsx_reg2sam_pe(C->opt, C->idx, (uint64_t)((C->n_processed >> 1) + u), &C->reads[u << 1], pair, &C->pes, ctx, bsx_rg_id);
		}
	} else {
		reg_v *regs = &C->regs[u];
		if (!P->final_pass) {
			reg_v tmp;
			memset(&tmp, 0, sizeof(tmp));
			bsx_mark_primary(C->opt, regs, C->n_processed + u);
			reset_flags(regs);
			regs_copy(&tmp, regs);
			ctx->plan = 1;
			bsx_reg2sam_se(C->opt, C->idx, &C->reads[u], &tmp, ctx, bsx_rg_id);
			bsx_cfree(tmp.a);
		} else {
			ctx->plan = 0;
			bsx_reg2sam_se(C->opt, C->idx, &C->reads[u], regs, ctx, bsx_rg_id);
		}
	}
}

typedef struct {
	chunk_t *C; samctx_t *ctx; int per, u0; const int *todo, *jread, *jreg; const bsx_glb_job_t *sub; const bsx_glb_res_t *sres; const uint32_t *pool;
	const bsx_glb_tag_t *tags; const char *md;   /* NM / MD / ZC / ZR from the backend, or NULL */
} finish_par_t;
static void finish_worker(void *data, long k, int tid)
{
	finish_par_t *F = (finish_par_t*)data;
	int jj = F->todo[k], ri = F->jread[jj];
	(void)tid;
	if (F->sres[k].n_cigar < 0) return;   /* did not fit: redone with more room */
	bsx_setsam_finish(F->C->opt, F->C->idx, &F->C->reads[ri], &F->C->regs[ri].a[F->jreg[jj]], F->pool + F->sub[k].cigar_off, F->sres[k].n_cigar,
	                  &F->ctx[ri / F->per - F->u0].table[ri % F->per][F->jreg[jj]], F->tags ? &F->tags[k] : 0, F->tags ? F->md + F->tags[k].md_off : 0);
}

typedef struct { chunk_t *C; samctx_t *ctx; int per, u0; int *cnt; int64_t *off; bsx_glb_job_t *jobs; int *jread, *jreg, *todo; } plan_par_t;
static void plan_count_worker(void *data, long u, int tid)
{
	plan_par_t *Q = (plan_par_t*)data;
	int w, c = 0;
	(void)tid;
	for (w = 0; w < Q->per; ++w) c += (int)Q->ctx[u].want[w].n;
	Q->cnt[u] = c;
}
static void plan_jobs_worker(void *data, long u, int tid)
{
	plan_par_t *Q = (plan_par_t*)data;
	chunk_t *C = Q->C;
	int w; size_t k; int64_t at = Q->off[u];
	(void)tid;
	for (w = 0; w < Q->per; ++w) {
		int ri = (Q->u0 + (int)u) * Q->per + w;
		reg_v *regs = &C->regs[ri];
		Q->ctx[u].table[w] = (samrec_t*)bsx_crealloc(0, 0, sizeof(samrec_t) * (regs->n ? regs->n : 1));
		memset(Q->ctx[u].table[w], 0, sizeof(samrec_t) * (regs->n ? regs->n : 1));
		for (k = 0; k < Q->ctx[u].want[w].n; ++k, ++at) {
			int gi = Q->ctx[u].want[w].a[k];
			bsx_setsam_job(C->opt, C->idx, &C->reads[ri], C->roff[ri], &regs->a[gi], &Q->jobs[at]);
			Q->jobs[at].cigar_cap = 8;   /* most CIGARs are 1-3 operations; one that does not fit is redone with the room it asks for */
			Q->jobs[at].cigar_off = (uint32_t)(at * 8);   /* (the first round's pool layout: every job's eight words one behind the other) */
			Q->jread[at] = ri; Q->jreg[at] = gi;
			if (Q->todo) Q->todo[at] = (int)at;
		}
	}
}
static void plan_free_worker(void *data, long u, int tid)
{
	plan_par_t *Q = (plan_par_t*)data;
	int w; size_t k;
	(void)tid;
	for (w = 0; w < Q->per; ++w) {
		reg_v *regs = &Q->C->regs[(Q->u0 + u) * Q->per + w];
		if (Q->ctx[u].table[w]) for (k = 0; k < regs->n; ++k) bsx_cfree(Q->ctx[u].table[w][k].cigar);
		bsx_cfree(Q->ctx[u].table[w]); bsx_cvec_free(Q->ctx[u].want[w]);
	}
}

static pthread_mutex_t g_stat_mu = PTHREAD_MUTEX_INITIALIZER;   /* a chunk's phase times, added to by the slices of its back half */
static void stat_add(double *dst, double v) { pthread_mutex_lock(&g_stat_mu); *dst += v; pthread_mutex_unlock(&g_stat_mu); }

static int emit_sam(chunk_t *C, int u0, int u1)   /* the units (pairs, or single reads) [u0, u1) */
{
	int n_units = u1 - u0, per = C->is_pe ? 2 : 1, rc = BSX_OK, round;
	samctx_t *ctx = (samctx_t*)bsx_par_calloc(C->nt, (size_t)n_units, sizeof(samctx_t));
	out_par_t P;
	plan_par_t Q;
	BSX_VEC(int) todo;
	uint32_t *pool = 0;
	char *md = 0;
	int64_t md_cap = 0;
	size_t k, pool_len = 0;
	int64_t n_jobs;
	double t0 = now_s(), t_batch = 0;
	bsx_vec_init(todo);
	P.C = C; P.ctx = ctx; P.final_pass = 0; P.u0 = u0;
	bsx_parallel_for(C->nt, out_worker, &P, n_units);
	stat_add(&C->st.t_primary, now_s() - t0); t0 = now_s();
	Q.C = C; Q.ctx = ctx; Q.per = per; Q.u0 = u0;
	Q.cnt = (int*)malloc(sizeof(int) * ((size_t)n_units + 1)); Q.off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_units + 1));
	bsx_parallel_for(C->nt, plan_count_worker, &Q, n_units);
	n_jobs = prefix_counts(n_units, Q.cnt, Q.off);
	Q.jobs = (bsx_glb_job_t*)malloc(sizeof(bsx_glb_job_t) * (size_t)(n_jobs ? n_jobs : 1));
	Q.jread = (int*)malloc(sizeof(int) * (size_t)(n_jobs ? n_jobs : 1)); Q.jreg = (int*)malloc(sizeof(int) * (size_t)(n_jobs ? n_jobs : 1));
	bsx_vec_reserve(todo, (size_t)n_jobs + 1);
	Q.todo = todo.a;   /* filled by the workers, each job its own index: the first round is every job */
	bsx_parallel_for(C->nt, plan_jobs_worker, &Q, n_units);
	todo.n = (size_t)n_jobs;
	for (round = 0; round < 8 && todo.n && rc == BSX_OK; ++round) { /* a CIGAR that does not fit is redone with the room it asked for */
		/* the first round is every job, in place (a million 48-byte records are not copied); a later one the few whose CIGAR did not fit */
		bsx_glb_job_t *sub = round == 0 ? Q.jobs : (bsx_glb_job_t*)malloc(sizeof(*sub) * todo.n);
		bsx_glb_res_t *sres = (bsx_glb_res_t*)malloc(sizeof(*sres) * todo.n);
		bsx_glb_tag_t *tags = C->be->global_batch_tags ? (bsx_glb_tag_t*)malloc(sizeof(*tags) * todo.n) : 0;
		size_t off = 0, nt = 0;
		if (round == 0) off = todo.n * 8;   /* (cigar_off set with the jobs) */
		else for (k = 0; k < todo.n; ++k) { sub[k] = Q.jobs[todo.a[k]]; sub[k].cigar_off = (uint32_t)off; off += sub[k].cigar_cap; }
		if (off > pool_len) { pool_len = off; pool = (uint32_t*)realloc(pool, pool_len * 4 + 4); }
		{
			double tb = now_s();
			rc = tags ? C->be->global_batch_tags(C->be->ctx, (int64_t)todo.n, sub, sres, pool, off, tags, &md, &md_cap)
			          : C->be->global_batch(C->be->ctx, (int64_t)todo.n, sub, sres, pool, off);
			t_batch += now_s() - tb;
		}
		__atomic_fetch_add(&C->st.n_glb_jobs, (int64_t)todo.n, __ATOMIC_RELAXED);
		if (rc == BSX_OK) {
			finish_par_t F;
			F.C = C; F.ctx = ctx; F.per = per; F.u0 = u0; F.todo = todo.a; F.jread = Q.jread; F.jreg = Q.jreg; F.sub = sub; F.sres = sres; F.pool = pool;
			F.tags = tags; F.md = md;
			bsx_parallel_for(C->nt, finish_worker, &F, (long)todo.n);
			for (k = 0; k < todo.n; ++k) {
				int jj = todo.a[k];
				if (sres[k].n_cigar < 0) { Q.jobs[jj].cigar_cap = (uint32_t)(-sres[k].n_cigar) + 2; todo.a[nt++] = jj; }
			}
		}
		todo.n = nt;
		if (round) free(sub);
		free(sres); free(tags);
	}
	if (rc == BSX_OK && todo.n) rc = BSX_E_INTERNAL;
	if (bsx_phases()) fprintf(stderr, "[M::cigar] %d rounds, %.3f s in the K6 batches, %.3f s on the host\n", round, t_batch, now_s() - t0 - t_batch);
	stat_add(&C->st.t_cigar, now_s() - t0); t0 = now_s();
	if (rc == BSX_OK) { P.final_pass = 1; bsx_parallel_for(C->nt, out_worker, &P, n_units); }
	/* the records own the CIGAR buffers now; release what the reference frees in mem_alnreg_freeSAM */
	if (C->arena_set < 0) bsx_parallel_for(C->nt, plan_free_worker, &Q, n_units);   /* arena memory is rewound with the chunk */
	stat_add(&C->st.t_sam, now_s() - t0);
	free(ctx); free(pool); free(md); free(Q.cnt); free(Q.off); free(Q.jobs); free(Q.jread); free(Q.jreg);
	bsx_vec_free(todo);
	return rc;
}

static void clip_worker(void *data, long i, int tid)
{
	chunk_t *C = (chunk_t*)data;
	const bsx_opt_t *opt = C->opt;
	(void)tid;
	if (C->is_pe) clip_read(&C->reads[i], (i & 1) ? opt->adaptor2 : opt->adaptor1, (i & 1) ? opt->l_adaptor2 : opt->l_adaptor1, opt);
	else clip_read(&C->reads[i], opt->adaptor1, opt->l_adaptor1, opt);
	C->reads[i].sam = 0;
}

/* which converted index a read is searched against, in the reference's call order (bis_worker1, bwamem.c:311-376): 1 = parent (C>T read),
 * 0 = daughter (G>A read) */
static int strand_order(const bsx_opt_t *opt, int is_pe, int second_of_pair, int order[2])
{
	int no = 0;
	if (!is_pe) {
		if (!(opt->parent & 1) || opt->parent >> 1) order[no++] = 0;
		if (!(opt->parent & 1) || !(opt->parent >> 1)) order[no++] = 1;
	} else if (!second_of_pair) { order[no++] = 1; if (!opt->parent) order[no++] = 0; }
	else { order[no++] = 0; if (!opt->parent) order[no++] = 1; }
	return no;
}
BSX_API int bsx_hook_strand_order(const bsx_opt_t *opt, int is_pe, int second_of_pair, int order[2]) { return strand_order(opt, is_pe, second_of_pair, order); }

static void setup_worker(void *data, long i, int tid)
{
	chunk_t *C = (chunk_t*)data;
	const bsx_opt_t *opt = C->opt;
	int order[2], no, k;
	(void)tid;
	if (C->reads[i].l_seq) memcpy(C->buf + C->roff[i], C->reads[i].seq, (size_t)C->reads[i].l_seq);
	no = strand_order(opt, C->is_pe, (int)(i & 1), order);
	for (k = 0; k < no; ++k) {
		int t = C->read_task0[i] + k;
		c2r_t *T = &C->tasks[t];
		memset(T, 0, C2R_HEADER_BYTES);
		T->read_idx = (int)i; T->parent = order[k]; T->qoff = C->roff[i]; T->l_query = C->reads[i].l_seq; T->query = C->reads[i].seq;
		C->stasks[t].qoff = T->qoff; C->stasks[t].len = T->l_query; C->stasks[t].parent = T->parent;
	}
}

static void sa_jobs_worker(void *data, long t, int tid)
{
	chunk_t *C = (chunk_t*)data;
	int64_t k, c;
	(void)tid;
	for (k = C->intv_off[t]; k < C->intv_off[t + 1]; ++k)
		for (c = 0; c < C->ipos_off[k + 1] - C->ipos_off[k]; ++c) {
			bsx_sa_job_t *j = &C->sa_jobs[C->ipos_off[k] + c];
			j->k = C->intv[k].x[0] + (uint64_t)c; j->parent = C->tasks[C->hmap[t]].parent; j->pad = 0;
		}
}

static void release_worker(void *data, long t, int tid)
{
	chunk_t *C = (chunk_t*)data;
	(void)tid;
	bsx_cvec_free(C->tasks[t].regs);
}
static void release_host_worker(void *data, long h, int tid) { chunk_t *C = (chunk_t*)data; (void)tid; bsx_c2r_release(&C->tasks[C->hmap[h]]); }
static void init_host_worker(void *data, long h, int tid) { chunk_t *C = (chunk_t*)data; c2r_t *T = &C->tasks[C->hmap[h]]; (void)tid; bsx_c2r_init(T); T->done = T->has_job = 0; }

/* regions the device produced -> the task's region list (every other mem_alnreg_t field is still zero here) */
static void adopt_worker(void *data, long t, int tid)
{
	chunk_t *C = (chunk_t*)data;
	c2r_t *T = &C->tasks[t];
	int k, n = C->dreg_n[t];
	(void)tid;
	if (n < 0) return;
	T->done = 1;
	/* a read whose regions were sorted and de-duplicated on the device takes the survivors straight from the downloaded block
	 * (merge_worker): nine million regions a chunk come down, a third of them are left, and a list per strand search was two
	 * million allocations and 1.4 GB of writes */
	if (C->dd_n && C->n > 0 && C->dd_n[t / (C->n_tasks / C->n)] >= 0) return;
	for (k = 0; k < n; ++k) {
		reg_t r;
		reg_from_device(&r, &C->dregs[C->dreg_off[t] + k]);
		bsx_cvec_push(T->regs, r);
	}
}
static void release_regs_worker(void *data, long i, int tid) { (void)tid; bsx_cfree(((chunk_t*)data)->regs[i].a); }

/* ------------------------------------------------------------------ the chunk */
#define FCHECK(x) do { rc = (x); if (rc != BSX_OK) return rc; } while (0)

/* The next chunk handed to bsx_process_seqs / bsx_stream_push by this thread is a slice of a larger chunk, starting at read `first` of
 * it (an even number for pairs): the reference hashes a pair's regions by the pair's index WITHIN the chunk (bwamem.c:408,413), the one
 * place where a read's position in its chunk enters the result besides the insert-size statistics. */
static __thread int64_t g_next_local0 = 0;
BSX_API void bsx_chunk_slice_offset(int64_t first) { g_next_local0 = first; }

static chunk_t *chunk_new(const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx, int64_t n_processed, int n,
                          bsx_read_t *reads, const bsx_pestat_t *pes0)
{
	chunk_t *C = (chunk_t*)calloc(1, sizeof(chunk_t));
	C->be_copy = *be; C->be = &C->be_copy;
	C->opt = opt; C->idx = idx; C->n = n; C->reads = reads; C->n_processed = n_processed; C->nt = bsx_host_threads(opt);
	C->local0 = g_next_local0; g_next_local0 = 0;
	C->is_pe = (opt->flag & BSX_F_PE) ? 1 : 0;
	if (pes0) { C->pes0_copy = *pes0; C->pes0 = &C->pes0_copy; }
	C->arena_set = -1;
	C->t_begin = now_s();
	{ static int64_t g_token = 0; C->token = __atomic_add_fetch(&g_token, 1, __ATOMIC_RELAXED); }
	return C;
}

/* Strand searches the host chains itself: C->hmap[0..n_host), the first n_reseed of them seeded here (K1+K2 batch), the
 * others with the interval lists the device handed back.  K3, chaining, chain filter, extension rounds; afterwards their
 * regions are in place like the adopted ones and everything else is released. */
static int host_path(chunk_t *C, int n_reseed, const bsx_intv_t *decl_intv, const int64_t *decl_off)
{
	const bsx_backend_t *be = C->be;
	const bsx_opt_t *opt = C->opt;
	int rc = BSX_OK, i, t, nt = C->nt;
	double t0;
	if (C->n_host == 0) return BSX_OK;
	bsx_parallel_for(nt, init_host_worker, C, C->n_host);

	/* K1+K2 */
	t0 = now_s();
	C->intv_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)C->n_host + 1));
	C->intv_off[0] = 0;
	if (n_reseed == C->n_tasks) rc = be->seed_batch(be->ctx, opt, n_reseed, C->stasks, &C->intv, &C->intv_cap, C->intv_off);   /* hmap is the identity */
	else {
		bsx_seed_task_t *sub = (bsx_seed_task_t*)malloc(sizeof(*sub) * ((size_t)n_reseed + 1));
		for (t = 0; t < n_reseed; ++t) sub[t] = C->stasks[C->hmap[t]];
		rc = be->seed_batch(be->ctx, opt, n_reseed, sub, &C->intv, &C->intv_cap, C->intv_off);
		free(sub);
	}
	if (rc != BSX_OK) goto out;
	if (C->n_host > n_reseed) { /* append the interval lists the device handed back */
		int64_t base = C->intv_off[n_reseed], add = decl_off[C->n_host - n_reseed];
		if (C->intv_cap < base + add) { C->intv_cap = base + add + 16; C->intv = (bsx_intv_t*)realloc(C->intv, sizeof(bsx_intv_t) * (size_t)C->intv_cap); }
		if (add) memcpy(C->intv + base, decl_intv, sizeof(bsx_intv_t) * (size_t)add);
		for (t = n_reseed; t <= C->n_host; ++t) C->intv_off[t] = base + decl_off[t - n_reseed];
	}
	C->st.t_seed += now_s() - t0; C->st.n_intv += C->intv_off[C->n_host];
	if (bsx_phases() && be->regions_batch)
		for (t = 0; t < C->n_host && t < 80; ++t) {
			int64_t k, occ = 0, big = 0;
			for (k = C->intv_off[t]; k < C->intv_off[t + 1]; ++k) { occ += (int64_t)(C->intv[k].x[2] < (uint64_t)opt->max_occ ? C->intv[k].x[2] : (uint64_t)opt->max_occ); big += C->intv[k].x[2] > (uint64_t)opt->max_occ; }
			fprintf(stderr, "[M::declined] task %d status %d len %d: %ld intervals, %ld occurrences (capped), %ld intervals beyond max_occ\n", C->hmap[t], C->dreg_n[C->hmap[t]],
			        C->tasks[C->hmap[t]].l_query, (long)(C->intv_off[t + 1] - C->intv_off[t]), (long)occ, (long)big);
		}

	/* K3: the first min(occ, max_occ) occurrences of every interval */
	t0 = now_s();
	{
		int64_t n_iv = C->intv_off[C->n_host], k, nj = 0;
		bsx_sa_job_t *sj;
		C->ipos_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_iv + 1));
		for (k = 0; k < n_iv; ++k) { C->ipos_off[k] = nj; nj += (int64_t)(C->intv[k].x[2] < opt->max_occ ? C->intv[k].x[2] : opt->max_occ); }
		C->ipos_off[n_iv] = nj;
		sj = (bsx_sa_job_t*)malloc(sizeof(*sj) * ((size_t)nj + 1));
		C->sa_jobs = sj;
		bsx_parallel_for(nt, sa_jobs_worker, C, C->n_host);
		C->pos = (uint64_t*)malloc(8 * ((size_t)nj + 1));
		rc = be->sa_batch(be->ctx, nj, sj, C->pos);
		free(sj);
		C->st.n_sa += nj;
		if (rc != BSX_OK) goto out;
	}
	C->st.t_sa += now_s() - t0;

	/* chaining (host); intervals that must be walked past max_occ get their remaining occurrences looked up */
	t0 = now_s();
	C->need_more = (int*)calloc((size_t)C->n_host + 1, sizeof(int));
	C->xpos = (uint64_t**)calloc((size_t)C->n_host + 1, sizeof(uint64_t*));
	C->xpos_off = (int64_t**)calloc((size_t)C->n_host + 1, sizeof(int64_t*));
	C->trees = (bsx_btree_t**)malloc(sizeof(bsx_btree_t*) * nt);
	for (i = 0; i < nt; ++i) C->trees[i] = bsx_bt_new();
	for (;;) {
		int any = 0;
		bsx_parallel_for(nt, chain_worker, C, C->n_host);
		for (t = 0; t < C->n_host; ++t) {
			int n_iv, k, want;
			int64_t nj, c;
			bsx_sa_job_t *sj;
			if (C->need_more[t] <= 0) continue;
			any = 1;
			n_iv = (int)(C->intv_off[t + 1] - C->intv_off[t]);
			want = C->need_more[t] - 1;
			if (!C->xpos_off[t]) {
				C->xpos_off[t] = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_iv + 1));
				for (k = 0; k <= n_iv; ++k) C->xpos_off[t][k] = C->ipos_off[C->intv_off[t] + k] - C->ipos_off[C->intv_off[t]];
			}
			{ /* give interval `want` all of its occurrences, keep the others */
				int64_t *cnt = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_iv);
				for (k = 0; k < n_iv; ++k) cnt[k] = C->xpos_off[t][k + 1] - C->xpos_off[t][k];
				cnt[want] = (int64_t)C->intv[C->intv_off[t] + want].x[2];
				for (k = 0, nj = 0; k < n_iv; ++k) { C->xpos_off[t][k] = nj; nj += cnt[k]; }
				C->xpos_off[t][n_iv] = nj;
				free(cnt);
			}
			sj = (bsx_sa_job_t*)malloc(sizeof(*sj) * ((size_t)nj + 1));
			for (k = 0; k < n_iv; ++k)
				for (c = 0; c < C->xpos_off[t][k + 1] - C->xpos_off[t][k]; ++c) {
					bsx_sa_job_t *j = &sj[C->xpos_off[t][k] + c];
					j->k = C->intv[C->intv_off[t] + k].x[0] + (uint64_t)c; j->parent = C->tasks[C->hmap[t]].parent; j->pad = 0;
				}
			free(C->xpos[t]);
			C->xpos[t] = (uint64_t*)malloc(8 * ((size_t)nj + 1));
			rc = be->sa_batch(be->ctx, nj, sj, C->xpos[t]);
			free(sj);
			C->st.n_sa += nj;
			if (rc != BSX_OK) goto out;
			C->need_more[t] = 0;
		}
		if (!any) break;
	}
	rc = filter_chained_seeds(C);
	if (rc != BSX_OK) goto out;
	C->st.t_chain += now_s() - t0;

	/* K4 rounds */
	t0 = now_s();
	rc = extension_rounds(C);
	C->st.t_extend += now_s() - t0;
out:
	bsx_parallel_for(nt, release_host_worker, C, C->n_host);
	if (C->trees) { for (i = 0; i < nt; ++i) bsx_bt_free(C->trees[i]); free(C->trees); C->trees = 0; }
	free(C->intv); free(C->intv_off); free(C->pos); free(C->ipos_off);
	for (t = 0; t < C->n_host; ++t) { if (C->xpos) free(C->xpos[t]); if (C->xpos_off) free(C->xpos_off[t]); }
	free(C->need_more); free(C->xpos); free(C->xpos_off);
	C->intv = 0; C->intv_cap = 0; C->intv_off = 0; C->pos = 0; C->ipos_off = 0; C->need_more = 0; C->xpos = 0; C->xpos_off = 0;
	C->n_host = 0;
	return rc;
}

static int chunk_merge(chunk_t *C);
/* front half: clipping, strand searches, seeding .. regions of every strand search (mem_align1_core's first part,
 * lib/aln/bwamem.c:183-208, for the whole chunk) */
static int chunk_front(chunk_t *C)
{
	const bsx_backend_t *be = C->be;
	const bsx_opt_t *opt = C->opt;
	bsx_read_t *reads = C->reads;
	int rc = BSX_OK, i, t, n = C->n, nt = C->nt, n_reseed = 0;
	bsx_intv_t *decl_intv = 0; int64_t decl_cap = 0, *decl_off = 0;
	size_t tot = 0;
	double t0, t_batch_end = 0;

	C->arena_set = bsx_arenas_begin(nt);
	/* clipping + chunk read buffer + strand searches in the reference's call order (bwamem.c:325-333,352-372) */
	t0 = now_s();
	if (C->is_pe)
		for (i = 0; i < n; i += 2)
			if (!pair_names_ok(reads[i].name, reads[i + 1].name)) {
				fprintf(stderr, "[bsx] paired reads have different names: \"%s\", \"%s\"\n", reads[i].name, reads[i + 1].name);
				return BSX_E_FORMAT;
			}
	bsx_parallel_for(nt, clip_worker, C, n);
	C->roff = (uint32_t*)bsx_big_get(C->arena_set, 0, sizeof(uint32_t) * ((size_t)n + 1));
	for (i = 0; i < n; ++i) { C->roff[i] = (uint32_t)tot; tot += (size_t)reads[i].l_seq; if (reads[i].l_seq > C->max_len) C->max_len = reads[i].l_seq; }
	C->roff[n] = (uint32_t)tot;
	if (tot >= 0xffff0000ull) return BSX_E_ARG;
	C->buf = (uint8_t*)bsx_big_get(C->arena_set, 1, tot + 16);
	{
		int per_read = C->is_pe ? (opt->parent ? 1 : 2) : ((opt->parent & 1) ? 1 : 2);
		C->read_task0 = (int*)bsx_big_get(C->arena_set, 2, sizeof(int) * ((size_t)n + 1));
		for (i = 0; i <= n; ++i) C->read_task0[i] = i * per_read;
		C->n_tasks = n * per_read;
	}
	C->tasks = (c2r_t*)bsx_big_get(C->arena_set, 3, sizeof(c2r_t) * ((size_t)C->n_tasks + 1));
	C->stasks = (bsx_seed_task_t*)bsx_big_get(C->arena_set, 4, sizeof(bsx_seed_task_t) * ((size_t)C->n_tasks + 1));
	bsx_parallel_for(nt, setup_worker, C, n);
	C->st.n_tasks = C->n_tasks;
	FCHECK(be->set_opt(be->ctx, opt));
	FCHECK(be->set_reads(be->ctx, C->buf, tot));
	C->st.t_prep = now_s() - t0;

	/* seeding through regions in one device pass where the backend has it; what it declines (and everything,
	 * on a backend without it) goes through the batch kernels and the host chaining below */
	t0 = now_s();
	C->hmap = (int*)bsx_big_get(C->arena_set, 5, sizeof(int) * ((size_t)C->n_tasks + 1));
	if (be->regions_batch) {
		C->dreg_off = (int64_t*)bsx_big_get(C->arena_set, 6, sizeof(int64_t) * ((size_t)C->n_tasks + 1));
		C->dreg_n = (int32_t*)bsx_big_get(C->arena_set, 7, sizeof(int32_t) * ((size_t)C->n_tasks + 1));
		/* the regions array is grown by the backend: start from the block kept from the last chunk of this set */
		C->dregs_cap = (int64_t)(2 * (size_t)C->n_tasks + 4096);
		C->dregs = (bsx_region_t*)bsx_big_get(C->arena_set, 8, sizeof(bsx_region_t) * (size_t)C->dregs_cap);
		decl_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)C->n_tasks + 1));
		rc = be->regions_batch(be->ctx, opt, C->n_tasks, C->stasks, &C->dregs, &C->dregs_cap, C->dreg_off, C->dreg_n, &decl_intv, &decl_cap, decl_off);
		t_batch_end = now_s();
		bsx_big_update(C->arena_set, 8, C->dregs, sizeof(bsx_region_t) * (size_t)C->dregs_cap);
		if (rc != BSX_OK) goto out;
		if (be->regions_dedup && C->n_tasks == n * (C->n_tasks / (n ? n : 1))) { /* C5 of every read whose strand searches all finished on the device */
			C->dd_cap = be->dedup_cap;
			C->dd_n = (int32_t*)bsx_big_get(C->arena_set, 10, sizeof(int32_t) * ((size_t)n + 1));
			C->dd_idx = (uint8_t*)bsx_big_get(C->arena_set, 11, (size_t)C->dd_cap * ((size_t)n + 1));
			if (be->regions_dedup2) {
				C->dd_loff = (int64_t*)bsx_big_get(C->arena_set, 12, sizeof(int64_t) * ((size_t)n + 1));
				rc = be->regions_dedup2(be->ctx, opt, n, C->n_tasks / (n ? n : 1), C->dd_n, C->dd_idx, C->dd_loff, &C->dd_lidx, &C->dd_lcap);
			} else rc = be->regions_dedup(be->ctx, opt, n, C->n_tasks / (n ? n : 1), C->dd_n, C->dd_idx);
			if (rc != BSX_OK) goto out;
			if (bsx_phases()) { /* what the device's sort + de-duplication leaves of the regions that came down */
				int64_t kept = 0, held = 0, left_all = 0, left_reads = 0; int per = C->n_tasks / (n ? n : 1), i, k;
				for (i = 0; i < n; ++i) {
					int64_t all = 0;
					for (k = 0; k < per; ++k) if (C->dreg_n[i * per + k] > 0) all += C->dreg_n[i * per + k];
					if (C->dd_n[i] >= 0) { kept += C->dd_n[i]; held += all; } else { left_all += all; ++left_reads; }
				}
				fprintf(stderr, "[M::regions] device de-duplication: %lld of %lld regions kept; %lld regions of %lld reads left to the host's\n", (long long)kept, (long long)held, (long long)left_all, (long long)left_reads);
			}
		}
		{ double ta = now_s(); bsx_parallel_for(nt, adopt_worker, C, C->n_tasks); if (bsx_phases()) fprintf(stderr, "[M::regions] regions_batch %.3f s, device de-duplication %.3f s, adopting the regions %.3f s\n", t_batch_end - t0, ta - t_batch_end, now_s() - ta); }
		/* host list: first the strand searches that must be seeded again, then the ones whose intervals came back */
		for (t = 0; t < C->n_tasks; ++t) if (C->dreg_n[t] == -1) C->hmap[C->n_host++] = t;
		n_reseed = C->n_host;
		for (t = 0; t < C->n_tasks; ++t) if (C->dreg_n[t] < -1 && C->dreg_n[t] != BSX_REGIONS_PENDING) C->hmap[C->n_host++] = t;
		for (t = 0; t < C->n_tasks; ++t) if (C->dreg_n[t] == BSX_REGIONS_PENDING) ++C->n_pending;
		if (bsx_phases()) {
			long h[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
			for (t = 0; t < C->n_tasks; ++t) ++h[C->dreg_n[t] >= 0 ? 0 : C->dreg_n[t] == BSX_REGIONS_PENDING ? 1 : (-C->dreg_n[t] < 9 ? -C->dreg_n[t] : 9)];
			fprintf(stderr, "[M::regions] on device %ld | declined: seeding overflow %ld, read length %ld, intervals %ld, occurrences %ld, chains %ld, tied starts %ld, band %ld, regions %ld, output %ld\n",
			        h[0], h[1], h[9], h[8], h[2], h[3], h[4], h[5], h[6], h[7]);
		}
	} else {
		for (t = 0; t < C->n_tasks; ++t) C->hmap[t] = t;
		C->n_host = n_reseed = C->n_tasks;
	}
	C->st.t_regions = now_s() - t0; C->st.n_host_tasks = C->n_host;
	rc = host_path(C, n_reseed, decl_intv, decl_off);
	if (rc == BSX_OK && !C->n_pending) rc = chunk_merge(C);
out:
	free(decl_intv); free(decl_off);
	return rc;
}

/* The strand searches the device was still seeding again when the front half returned (reads inside tandem repeats):
 * collect their regions now; what even that pass could not hold goes through the host path. */
static int finish_pending(chunk_t *C)
{
	const bsx_backend_t *be = C->be;
	int rc, t, n_list = 0, *list = (int*)malloc(sizeof(int) * (size_t)C->n_pending);
	for (t = 0; t < C->n_tasks; ++t) if (C->dreg_n[t] == BSX_REGIONS_PENDING) list[n_list++] = t;
	rc = be->regions_finish ? be->regions_finish(be->ctx, &C->dregs, &C->dregs_cap, C->dreg_off, C->dreg_n) : BSX_E_ARG;
	bsx_big_update(C->arena_set, 8, C->dregs, sizeof(bsx_region_t) * (size_t)C->dregs_cap);
	C->st.n_redo_tasks += n_list;
	C->n_pending = 0; C->n_host = 0;
	if (rc == BSX_OK) {
		int i;
		for (i = 0; i < n_list; ++i) {
			t = list[i];
			if (C->dreg_n[t] >= 0) adopt_worker(C, t, 0);
			else if (C->dreg_n[t] == -1) C->hmap[C->n_host++] = t;
			else rc = BSX_E_ARG;
		}
		if (bsx_phases()) fprintf(stderr, "[M::regions] seeded again on the device: %d strand searches, %d of them left to the host\n", n_list, C->n_host);
		C->st.n_host_tasks += C->n_host;
		if (rc == BSX_OK) rc = host_path(C, C->n_host, 0, 0);
	}
	free(list);
	return rc;
}

/* What follows the insert-size statistics -- mate rescue, primary marking / pairing, CIGARs, SAM text -- is per pair (per read, single-end):
 * nothing of it looks beyond its pair.  The pushing thread used to run those stages one after the other over the whole chunk, waiting for the
 * chunk's K5 batch (150-300 ms with other chunks' kernels on the device) and then for its K6 batch with nothing else to do; on a repeat-rich
 * genome that thread's back half is what bounds the stream (1.1-1.2 s of a 1.3 s step, round 5).  So the units are cut into slices, and two
 * threads take the slices alternately, each slice going through all four stages: while one thread waits for its slice's K5 / K6 batch (the
 * lane's back-half batches are serialised by a lock in the shim: one stream, one set of staging buffers), the other runs host stages of the
 * next slice.  Every pair is still processed by the same code in the same order of stages: the SAM cannot change ("back_slices": 1 = the
 * stages over the whole chunk, as before). */
typedef struct { chunk_t *C; int first, step, n_sl, n_units, helper, rc; } slice_run_t;
static int one_slice(chunk_t *C, int u0, int u1)
{
	int rc = BSX_OK;
	if (u1 <= u0) return BSX_OK;
	if (C->is_pe && !(C->opt->flag & BSX_F_NO_RESCUE)) {
		double t0 = now_s();
		rc = mate_rescue(C, u0, u1);
		stat_add(&C->st.t_matesw, now_s() - t0);
		if (rc != BSX_OK) return rc;
	}
	return emit_sam(C, u0, u1);
}
static void *slice_thread(void *arg)
{
	slice_run_t *R = (slice_run_t*)arg;
	chunk_t *C = R->C;
	int k;
	if (R->helper) bsx_arenas_bind_extra(C->arena_set, R->helper - 1);   /* its own arena beside the chunk's: the caller of a parallel loop allocates too */
	for (k = R->first; k < R->n_sl && R->rc == BSX_OK; k += R->step) {
		const int u0 = (int)((int64_t)R->n_units * k / R->n_sl), u1 = (int)((int64_t)R->n_units * (k + 1) / R->n_sl);
		R->rc = one_slice(C, u0, u1);
	}
	if (R->helper) bsx_arenas_bind(-1);
	return 0;
}
static int back_slices(chunk_t *C)
{
	const int n_units = C->is_pe ? C->n >> 1 : C->n;
	int n_sl = (int)bsx_tune_long("back_slices", 4), n_thr = (int)bsx_tune_long("back_threads", 2), j, rc = BSX_OK;
	slice_run_t R[4];
	pthread_t th[4];
	int live[4] = {0, 0, 0, 0};
	if (n_sl > n_units / 1024) n_sl = n_units / 1024;   /* (small chunks: nothing to overlap) */
	if (n_sl < 1) n_sl = 1;
	if (n_thr > 4) n_thr = 4;
	if (n_thr > n_sl) n_thr = n_sl;
	if (n_thr < 1 || C->arena_set < 0) n_thr = 1;
	for (j = 0; j < n_thr; ++j) { R[j].C = C; R[j].first = j; R[j].step = n_thr; R[j].n_sl = n_sl; R[j].n_units = n_units; R[j].helper = j; R[j].rc = BSX_OK; }
	for (j = 1; j < n_thr; ++j) live[j] = pthread_create(&th[j], 0, slice_thread, &R[j]) == 0;
	for (j = 1; j < n_thr; ++j) if (!live[j]) { R[j].helper = 0; }   /* (no thread: its slices run here, below) */
	(void)slice_thread(&R[0]);
	for (j = 1; j < n_thr; ++j) { if (live[j]) pthread_join(th[j], 0); else (void)slice_thread(&R[j]); }
	for (j = 0; j < n_thr; ++j) if (rc == BSX_OK) rc = R[j].rc;
	return rc;
}

/* back half: per-read merge, insert-size statistics, mate rescue, pairing, CIGARs, SAM text (the rest of
 * mem_process_seqs, lib/aln/bwamem.c:374-412) */
/* C5 of every read (mem_sort_deduplicate: what the device left to the host, the concatenation tests' score-only K6 batches, the reads'
 * region lists built).  It needs nothing but the front half's products, so it runs at the end of the front half, on that thread -- chunks
 * further back in the stream have time there, the pushing thread (whose back halves bound the stream) does not -- unless strand searches
 * are still being seeded again on the side stream: then it waits for the back half to collect them. */
static int chunk_merge(chunk_t *C)
{
	int rc;
	double t0 = now_s();
	C->regs = (reg_v*)bsx_big_get(C->arena_set, 9, sizeof(reg_v) * ((size_t)C->n + 1));
	memset(C->regs, 0, sizeof(reg_v) * ((size_t)C->n + 1));
	rc = merge_regions(C);
	C->st.t_merge = now_s() - t0;
	C->merged = 1;
	return rc;
}

static int chunk_back(chunk_t *C)
{
	const bsx_opt_t *opt = C->opt;
	int rc = BSX_OK;
	double t0;
	bsx_arenas_bind(C->arena_set);
	if (C->n_pending) FCHECK(finish_pending(C));
	if (!C->merged) FCHECK(chunk_merge(C));
	if (C->is_pe) {
		t0 = now_s();
		if (C->pes0) C->pes = *C->pes0;
		else C->pes = bsx_pestat(opt, &C->idx->ref, C->n, C->regs);
		C->st.t_pestat = now_s() - t0;
	}
	return back_slices(C);
}

static void chunk_free(chunk_t *C)
{
	int nt = C->nt;
	double t0 = now_s();
	bsx_arenas_bind(C->arena_set);
	if (C->tasks) {
		if (C->arena_set < 0) bsx_parallel_for(nt, release_worker, C, C->n_tasks);   /* arena memory is rewound, not freed */
		bsx_big_put(C->arena_set, 3, C->tasks);
	}
	if (C->regs) { if (C->arena_set < 0) bsx_parallel_for(nt, release_regs_worker, C, C->n); bsx_big_put(C->arena_set, 9, C->regs); }
	bsx_big_put(C->arena_set, 0, C->roff); bsx_big_put(C->arena_set, 2, C->read_task0);
	bsx_big_put(C->arena_set, 4, C->stasks); bsx_big_put(C->arena_set, 1, C->buf);
	bsx_big_put(C->arena_set, 5, C->hmap); bsx_big_put(C->arena_set, 8, C->dregs); bsx_big_put(C->arena_set, 6, C->dreg_off); bsx_big_put(C->arena_set, 7, C->dreg_n);
	if (C->dd_n) { bsx_big_put(C->arena_set, 10, C->dd_n); bsx_big_put(C->arena_set, 11, C->dd_idx); }
	if (C->dd_loff) bsx_big_put(C->arena_set, 12, C->dd_loff);
	free(C->dd_lidx);
	bsx_arenas_end(C->arena_set);
	C->st.t_cleanup = now_s() - t0;
	C->st.t_total = now_s() - C->t_begin;
	g_last_stats = C->st;
	if (bsx_verbose >= 3) {
		const bsx_phase_stats_t *S = &C->st;
		fprintf(stderr, "[M::%s] Processed %d reads in %.3f real sec (%s)\n", "bsx_process_seqs", C->n, S->t_total, C->be->name ? C->be->name : "?");
		if (bsx_phases())
			fprintf(stderr, "[M::phases] regions %.3f (host tasks %ld) seed %.3f sa %.3f chain %.3f extend %.3f (%ld jobs, %ld rounds) merge %.3f pestat %.3f matesw %.3f primary %.3f cigar %.3f sam %.3f | tasks %ld intv %ld sa %ld sw %ld glb %ld\n",
				S->t_regions, (long)S->n_host_tasks, S->t_seed, S->t_sa, S->t_chain, S->t_extend, (long)S->n_ext_jobs, (long)S->n_ext_rounds, S->t_merge, S->t_pestat,
				S->t_matesw, S->t_primary, S->t_cigar, S->t_sam, (long)S->n_tasks, (long)S->n_intv, (long)S->n_sa, (long)S->n_sw_jobs, (long)S->n_glb_jobs);
	}
	free(C);
}

/* ------------------------------------------------------------------ the chunk, synchronously (mem_process_seqs) */
BSX_API int bsx_process_seqs_backend(const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx,
                                     int64_t n_processed, int n, bsx_read_t *reads, const bsx_pestat_t *pes0)
{
	chunk_t *C;
	int rc;
	if (!be || !opt || !idx || n < 0) return BSX_E_ARG;
	if (n == 0) return BSX_OK;
	if ((opt->flag & BSX_F_PE) && (n & 1)) return BSX_E_ARG;
	C = chunk_new(be, opt, idx, n_processed, n, reads, pes0);
	rc = chunk_front(C);
	if (rc == BSX_OK) rc = chunk_back(C);
	chunk_free(C);
	return rc;
}

BSX_API int bsx_process_seqs(bsx_device_t *dev, const bsx_opt_t *opt, const bsx_index_t *idx,
                             int64_t n_processed, int n, bsx_read_t *reads, const bsx_pestat_t *pes0)
{
	bsx_backend_t be;
	int rc = bsx_hip_backend(dev, &be);   /* the HIP kernels are the only product backend */
	if (rc != BSX_OK) return rc;
	return bsx_process_seqs_backend(&be, opt, idx, n_processed, n, reads, pes0);
}

/* ------------------------------------------------------------------ chunks as a two-deep pipeline
 * The reference overlaps reading, aligning and writing of consecutive chunks (kt_pipeline, lib/aln/align.c:100-170).
 * Here the overlap that matters is inside the aligning step: the front half of a chunk is device-bound, the back
 * half host-bound, so chunk k+1's front half runs (on its own device lane, from its own thread) while chunk k's
 * back half runs on the caller's thread.  Chunks stay independent: each has its own insert-size statistics. */
#define STREAM_MAX_DEPTH 6
struct bsx_stream {
	bsx_backend_t be[STREAM_MAX_DEPTH];
	int depth;                /* chunks in flight: depth-1 front halves ahead of the back half being run */
	const bsx_opt_t *opt; const bsx_index_t *idx;
	bsx_pestat_t pes0; int has_pes0;
	chunk_t *q[STREAM_MAX_DEPTH]; int n_q;   /* in flight, oldest first */
	int64_t n_pushed;
};

static int g_whole_chunk_threads = -1;   /* $BSX_STREAM_WHOLE_CHUNK=N: the chunk's own thread runs its back half too, at most N back halves at a time (0: the pushing thread runs them, one by one) */
static pthread_mutex_t g_back_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_back_cv = PTHREAD_COND_INITIALIZER;
static int g_back_running = 0;
static int64_t g_back_next = 0, g_back_seq = 0;   /* chunk sequence numbers of the streams of this process */
extern void (*bsx_pes_hist_hook)(void *ud, int64_t *hist, int n_bins);

static void *front_thread(void *arg)
{
	chunk_t *C = (chunk_t*)arg;
	C->rc = chunk_front(C);
	C->t_front_end = now_s();
	/* With the back half on the chunk's own thread as well, back halves of consecutive chunks overlap each other: one
	 * chunk's serial stretches and waits for its K5/K6 batches are filled with the other's parallel loops (the worker
	 * pool serves several loops at once).  Chunks still complete in order: the stream joins the threads in order. */
	if (C->ordered && C->rc == BSX_OK) {   /* (ranks sharing a chunk exchange histograms in chunk order: back halves stay on the pushing thread -- decided at push time) */
		/* at most g_whole_chunk_threads back halves at a time, started in chunk order */
		pthread_mutex_lock(&g_back_mu);
		while (g_back_running >= g_whole_chunk_threads || g_back_next != C->seq) pthread_cond_wait(&g_back_cv, &g_back_mu);
		++g_back_running; ++g_back_next;
		pthread_cond_broadcast(&g_back_cv);
		pthread_mutex_unlock(&g_back_mu);
		C->rc = chunk_back(C); C->back_done = 1;
		pthread_mutex_lock(&g_back_mu);
		--g_back_running;
		pthread_cond_broadcast(&g_back_cv);
		pthread_mutex_unlock(&g_back_mu);
	} else if (C->ordered) { /* a failed front half: its turn passes */
		pthread_mutex_lock(&g_back_mu);
		while (g_back_next != C->seq) pthread_cond_wait(&g_back_cv, &g_back_mu);
		++g_back_next;
		pthread_cond_broadcast(&g_back_cv);
		pthread_mutex_unlock(&g_back_mu);
	}
	bsx_arenas_bind(-1);
	return 0;
}

BSX_API int bsx_stream_open_backends(int depth, const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx,
                                     const bsx_pestat_t *pes0, bsx_stream_t **out)
{
	bsx_stream_t *s;
	int i;
	if (!be || !opt || !idx || !out || depth < 1 || depth > STREAM_MAX_DEPTH) return BSX_E_ARG;
	g_whole_chunk_threads = (int)bsx_tune_long("stream_whole_chunk", 0);
	s = (bsx_stream_t*)calloc(1, sizeof(*s));
	for (i = 0; i < depth; ++i) s->be[i] = be[i];
	s->depth = depth; s->opt = opt; s->idx = idx;
	if (pes0) { s->pes0 = *pes0; s->has_pes0 = 1; }
	*out = s;
	return BSX_OK;
}

BSX_API int bsx_stream_open(bsx_device_t *dev, const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_pestat_t *pes0, bsx_stream_t **out)
{
	bsx_backend_t b[STREAM_MAX_DEPTH];
	const char *e = getenv("BSX_STREAM_DEPTH");
	/* front halves in flight + 1: against a genome of hg38's size a front half is about twice as long as a back half and its kernels
	 * leave gaps (table-bound and gather-bound launches share the device well), so three of them keep the device busier than two
	 * (measured +5 %); against small genomes two are enough (a third was -4 % at 128 Mbp) */
	int rc, i, depth = e ? atoi(e) : (idx && idx->ref.l_pac >= 1000000000LL ? 4 : 3);
	if (depth < 1) depth = 1;
	if (depth > STREAM_MAX_DEPTH) depth = STREAM_MAX_DEPTH;
	for (i = 0; i < depth; ++i) if ((rc = bsx_hip_backend_lane(dev, i, &b[i])) != BSX_OK) return rc;
	return bsx_stream_open_backends(depth, b, opt, idx, pes0, out);
}
BSX_API int bsx_stream_depth(const bsx_stream_t *s) { return s ? s->depth : 0; }

/* wait for the chunk's front half, run its back half (its reads get their SAM text), release it */
static int chunk_finish(chunk_t *C)
{
	int rc;
	double t0 = now_s(), t1, t2;
	if (C->th_live) { pthread_join(C->th, 0); C->th_live = 0; }
	t1 = now_s();
	rc = C->rc;
	if (rc == BSX_OK && !C->back_done) rc = chunk_back(C);
	t2 = now_s();
	if (bsx_phases()) fprintf(stderr, "[M::stream] chunk begun at %.3f: front half %.3f s (prep %.3f, device pass %.3f), waited %.3f s for it from %.3f, back half %.3f s until %.3f\n",
	                                  C->t_begin, C->t_front_end - C->t_begin, C->st.t_prep, C->st.t_regions, t1 - t0, t0, t2 - t1, t2);
	chunk_free(C);
	return rc;
}

BSX_API int bsx_stream_push(bsx_stream_t *s, int64_t n_processed, int n, bsx_read_t *reads)
{
	int rc = BSX_OK, i;
	if (!s || n < 0) return BSX_E_ARG;
	if ((s->opt->flag & BSX_F_PE) && (n & 1)) return BSX_E_ARG;
	/* the new chunk's front half starts right away on a free device lane: the tail of a front half (a few straggling
	 * strand searches being seeded again) leaves the device almost idle, and the host is busy with an older chunk */
	if (n > 0) {
		chunk_t *C = chunk_new(&s->be[s->n_pushed % s->depth], s->opt, s->idx, n_processed, n, reads, s->has_pes0 ? &s->pes0 : 0);
		++s->n_pushed;
		/* a ticket only for the chunks that take a turn: a chunk pushed while the histogram hook is set (or without $BSX_STREAM_WHOLE_CHUNK)
		 * never advances g_back_next, so handing it a number would leave every later ordered chunk waiting for a turn that never comes */
		C->ordered = g_whole_chunk_threads > 0 && !bsx_pes_hist_hook;
		if (C->ordered) { pthread_mutex_lock(&g_back_mu); C->seq = g_back_seq++; pthread_mutex_unlock(&g_back_mu); }
		if (pthread_create(&C->th, 0, front_thread, C) == 0) C->th_live = 1;
		else { C->rc = chunk_front(C); bsx_arenas_bind(-1); if (C->ordered) { pthread_mutex_lock(&g_back_mu); while (g_back_next != C->seq) pthread_cond_wait(&g_back_cv, &g_back_mu); ++g_back_next; pthread_cond_broadcast(&g_back_cv); pthread_mutex_unlock(&g_back_mu); } }
		s->q[s->n_q++] = C;
	}
	while (s->n_q > (n > 0 ? s->depth - 1 : 0) && rc == BSX_OK) { /* complete the oldest: its lane is the next push's */
		rc = chunk_finish(s->q[0]);
		for (i = 1; i < s->n_q; ++i) s->q[i - 1] = s->q[i];
		--s->n_q;
	}
	if (rc != BSX_OK) { while (s->n_q) (void)chunk_finish(s->q[--s->n_q]); }
	return rc;
}

static int stream_drain(bsx_stream_t *s)
{
	int rc = BSX_OK, rc2, i;
	while (s->n_q) {
		rc2 = chunk_finish(s->q[0]);
		if (rc == BSX_OK) rc = rc2;
		for (i = 1; i < s->n_q; ++i) s->q[i - 1] = s->q[i];
		--s->n_q;
	}
	return rc;
}

BSX_API int bsx_stream_flush(bsx_stream_t *s) { return s ? stream_drain(s) : BSX_E_ARG; }

BSX_API void bsx_stream_close(bsx_stream_t *s)
{
	if (!s) return;
	(void)stream_drain(s);
	free(s);
}
