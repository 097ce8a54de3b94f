/* extend.c -- C4: chains -> alignment regions, as a resumable per-task state machine.
 *
 * mem_chain2region / mem_chain2region1 (lib/aln/memchain.c:742-904) walk the seeds of every chain
 * best-first; whether a seed is extended depends on the regions produced so far by the same strand
 * search, so the work of one task is a strictly ordered sequence of ksw_extend2 calls (left then
 * right, each with up to MAX_BAND_TRY=2 band widths).  Across tasks there is no dependency.  The
 * pipeline therefore advances all tasks in lock-step rounds: every task that needs a DP call posts
 * ONE job, the whole round runs as one device batch (K4), and the results are fed back.
 */
#include "align_types.h"
#include "pipeline.h"

#define MAX_BAND_TRY 2   /* memchain.c:611 */
#define getbss(parent, l_pac, rb) ((((rb) > (l_pac)) == (parent)) ? 1 : 0)

/* asymmetric_flt_seed, memchain.c:138-149: reference T under a read C, or reference A under a read G */
static int seed_violates_conversion(const bsx_index_t *idx, const uint8_t *query, const seed_t *s)
{
	int i;
	for (i = 0; i < s->len; ++i) {
		int r = bsx_ref_base(idx->ref.l_pac, idx->pac, s->rbeg + i);
		int q = query[s->qbeg + i];
		if ((r == 3 && q == 1) || (r == 0 && q == 2)) return 1;
	}
	return 0;
}

static void c2r_open_list(c2r_t *t, const seed_v *seeds)
{
	size_t i;
	if (t->m_srt < (int)seeds->n) { bsx_cfree(t->srt); t->m_srt = (int)seeds->n + 8; t->srt = (uint64_t*)bsx_crealloc(0, 0, sizeof(uint64_t) * t->m_srt); }
	for (i = 0; i < seeds->n; ++i) t->srt[i] = (uint64_t)seeds->a[i].score << 32 | i;
	t->n_srt = (int)seeds->n;
	bsx_introsort_u64(seeds->n, t->srt);   /* ks_introsort_64, memchain.c:752 */
	t->k = t->n_srt - 1;
}

static void c2r_left_job(const bsx_opt_t *opt, c2r_t *t, const seed_t *s)
{
	bsx_ext_job_t *j = &t->job;
	memset(j, 0, sizeof(*j));
	j->qoff = t->qoff + (uint32_t)s->qbeg - 1; j->qdir = -1; j->qlen = s->qbeg;
	j->tpos = s->rbeg - 1; j->tdir = -1; j->tlen = (int32_t)(s->rbeg - t->rmax[0]);
	j->h0 = s->len * opt->a; j->w = opt->w << t->tryi; j->end_bonus = opt->pen_clip5; j->parent = (uint8_t)t->parent;
	t->has_job = 1;
}

static void c2r_right_job(const bsx_opt_t *opt, c2r_t *t, const seed_t *s)
{
	bsx_ext_job_t *j = &t->job;
	int qe = s->qbeg + s->len;
	memset(j, 0, sizeof(*j));
	j->qoff = t->qoff + (uint32_t)qe; j->qdir = 1; j->qlen = t->l_query - qe;
	j->tpos = s->rbeg + s->len; j->tdir = 1; j->tlen = (int32_t)(t->rmax[1] - (s->rbeg + s->len));
	j->h0 = t->sc0; j->w = opt->w << t->tryi; j->end_bonus = opt->pen_clip3; j->parent = (uint8_t)t->parent;
	t->has_job = 1;
}

/* region complete: strand-boundary check, seed coverage, book-keeping (memchain.c:839-869) */
static void c2r_finish_region(const bsx_index_t *idx, c2r_t *t, const chain_t *c, const seed_v *seeds, const seed_t *s)
{
	reg_t *r = &t->cur;
	int64_t l_pac = idx->ref.l_pac;
	size_t i;
	r->bss = (uint8_t)getbss(t->parent, l_pac, r->rb);
	r->parent = (uint8_t)t->parent;
	if (getbss(t->parent, l_pac, r->re) == r->bss) {
		for (i = 0, r->seedcov = 0; i < seeds->n; ++i) {
			const seed_t *q = &seeds->a[i];
			if (q->qbeg >= r->qb && q->qbeg + q->len <= r->qe && q->rbeg >= r->rb && q->rbeg + q->len <= r->re) r->seedcov += q->len;
		}
		r->w = t->aw[0] > t->aw[1] ? t->aw[0] : t->aw[1];
		r->seedlen0 = s->len;
		r->frac_rep = c->frac_rep;
		bsx_cvec_push(t->regs, *r);
	}
	--t->k;
	t->stage = 0;
}

/* after the left side is settled: skip or post the right extension (memchain.c:689-693) */
static void c2r_begin_right(const bsx_opt_t *opt, const bsx_index_t *idx, c2r_t *t, const chain_t *c, const seed_v *seeds, const seed_t *s)
{
	if (s->qbeg + s->len == t->l_query) {
		t->cur.qe = t->l_query; t->cur.re = s->rbeg + s->len;
		c2r_finish_region(idx, t, c, seeds, s);
		return;
	}
	t->sc0 = t->cur.score;
	t->tryi = 0;
	t->stage = 2;
	c2r_right_job(opt, t, s);
}

/* Run the task until it posts a job (returns 1) or is finished (returns 0). */
int bsx_c2r_advance(const bsx_opt_t *opt, const bsx_index_t *idx, c2r_t *t)
{
	int64_t l_pac = idx->ref.l_pac;
	t->has_job = 0;
	if (t->done) return 0;
	for (;;) {
		chain_t *c;
		const seed_v *seeds;
		if (t->ci >= (int)t->chains.n) { t->done = 1; return 0; }
		c = &t->chains.a[t->ci];
		if (!t->chain_open) {
			if (c->seeds.n == 0) { ++t->ci; continue; }
			bsx_chain_ref_span(opt, t->l_query, l_pac, c, t->rmax);
			t->rid = bsx_fetch_span(&idx->ref, &t->rmax[0], c->seeds.a[0].rbeg, &t->rmax[1]);
			t->n0 = (int)t->regs.n;
			t->pass = 0;
			c2r_open_list(t, &c->seeds);
			t->chain_open = 1;
			t->stage = 0;
		}
		seeds = t->pass ? &c->seeds_extra : &c->seeds;
		if (t->stage != 0) return 0; /* a job is outstanding; nothing to do until its result arrives */
		while (t->k >= 0) {
			const seed_t *s = &seeds->a[(uint32_t)t->srt[t->k]];
			size_t u;
			if (seed_violates_conversion(idx, t->query, s)) { --t->k; continue; }
			/* is the seed inside a region this strand search already produced? (memchain.c:761-790) */
			for (u = 0; u < t->regs.n; ++u) {
				const reg_t *reg = &t->regs.a[u];
				int64_t rd;
				int qd, w, max_gap;
				if (s->rbeg < reg->rb || s->rbeg + s->len > reg->re || s->qbeg < reg->qb || s->qbeg + s->len > reg->qe) continue;
				if (s->len - reg->seedlen0 > .1 * t->l_query) continue;
				qd = s->qbeg - reg->qb; rd = s->rbeg - reg->rb;
				max_gap = bsx_cal_max_gap(opt, (int)(qd < rd ? qd : rd));
				w = max_gap < reg->w ? max_gap : reg->w;
				if (qd - rd < w && rd - qd < w) break;
				qd = reg->qe - (s->qbeg + s->len); rd = reg->re - (s->rbeg + s->len);
				max_gap = bsx_cal_max_gap(opt, (int)(qd < rd ? qd : rd));
				w = max_gap < reg->w ? max_gap : reg->w;
				if (qd - rd < w && rd - qd < w) break;
			}
			if (u < t->regs.n) { /* contained: extend anyway only if an overlapping long seed disagrees (memchain.c:794-819) */
				int i;
				for (i = t->k + 1; i < t->n_srt; ++i) {
					const seed_t *o;
					if (t->srt[i] == 0) continue;
					o = &seeds->a[(uint32_t)t->srt[i]];
					if (o->len < s->len * .95) continue;
					if (s->qbeg <= o->qbeg && s->qbeg + s->len - o->qbeg >= s->len >> 2 && o->qbeg - s->qbeg != o->rbeg - s->rbeg) break;
					if (o->qbeg <= s->qbeg && o->qbeg + o->len - s->qbeg >= s->len >> 2 && s->qbeg - o->qbeg != s->rbeg - o->rbeg) break;
				}
				if (i == t->n_srt) { t->srt[t->k] = 0; --t->k; continue; }
			}
			/* extend this seed (memchain.c:822-836) */
			memset(&t->cur, 0, sizeof(t->cur));
			t->cur.w = t->aw[0] = t->aw[1] = opt->w;
			t->cur.score = t->cur.truesc = -1;
			t->cur.rid = t->rid;
			if (s->qbeg == 0) { /* nothing to the left (memchain.c:623-626) */
				t->cur.score = t->cur.truesc = s->len * opt->a; t->cur.qb = 0; t->cur.rb = s->rbeg;
				c2r_begin_right(opt, idx, t, c, seeds, s);
				if (t->has_job) return 1;
				continue;
			}
			t->tryi = 0;
			t->stage = 1;
			c2r_left_job(opt, t, s);
			return 1;
		}
		/* list exhausted: fall back to the contained seeds if the chain produced nothing (memchain.c:898-901) */
		if (t->pass == 0 && (int)t->regs.n == t->n0 && c->seeds_extra.n > 0) {
			t->pass = 1;
			c2r_open_list(t, &c->seeds_extra);
			continue;
		}
		++t->ci; t->chain_open = 0;
	}
}

/* Feed the result of the posted job back (left: memchain.c:641-671, right: memchain.c:700-729). */
void bsx_c2r_consume(const bsx_opt_t *opt, const bsx_index_t *idx, c2r_t *t, const bsx_ext_res_t *res)
{
	chain_t *c = &t->chains.a[t->ci];
	const seed_v *seeds = t->pass ? &c->seeds_extra : &c->seeds;
	const seed_t *s = &seeds->a[(uint32_t)t->srt[t->k]];
	reg_t *r = &t->cur;
	int prev = r->score, aw = opt->w << t->tryi;
	t->has_job = 0;
	r->score = res->score;
	if (t->stage == 1) {
		t->aw[0] = aw;
		if (!(r->score == prev || res->max_off < (aw >> 1) + (aw >> 2)) && t->tryi + 1 < MAX_BAND_TRY) {
			++t->tryi; c2r_left_job(opt, t, s); return;
		}
		if (res->gscore <= 0 || res->gscore <= r->score - opt->pen_clip5) { /* local */
			r->qb = s->qbeg - res->qle; r->rb = s->rbeg - res->tle; r->truesc = r->score;
		} else { /* to the read end */
			r->qb = 0; r->rb = s->rbeg - res->gtle; r->truesc = res->gscore;
		}
		c2r_begin_right(opt, idx, t, c, seeds, s);
	} else {
		int qe = s->qbeg + s->len;
		t->aw[1] = aw;
		if (!(r->score == prev || res->max_off < (aw >> 1) + (aw >> 2)) && t->tryi + 1 < MAX_BAND_TRY) {
			++t->tryi; c2r_right_job(opt, t, s); return;
		}
		if (res->gscore <= 0 || res->gscore <= r->score - opt->pen_clip3) {
			r->qe = qe + res->qle; r->re = s->rbeg + s->len + res->tle; r->truesc += r->score - t->sc0;
		} else {
			r->qe = t->l_query; r->re = s->rbeg + s->len + res->gtle; r->truesc += res->gscore - t->sc0;
		}
		c2r_finish_region(idx, t, c, seeds, s);
	}
}

void bsx_c2r_release(c2r_t *t)
{
	bsx_chain_free(&t->chains);
	bsx_cvec_free(t->chains);
	bsx_cfree(t->srt); t->srt = 0; t->m_srt = 0;
}
