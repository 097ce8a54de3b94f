/* chain.c -- C1/C2: seed chaining and chain filtering on the host.
 *
 * The device has already produced, per strand search, the sorted SA intervals (K1+K2) and the
 * reference positions of their occurrences (K3).  What remains of mem_chain
 * (lib/aln/memchain.c:268-393) is the branchy, order-dependent clustering of seeds into chains
 * through a B-tree keyed by reference position, and mem_chain_flt (memchain.c:406-488).
 * Float arithmetic keeps the reference's operand types (float options, int operands).
 */
#include <math.h>
#include "align_types.h"

#define getbss(parent, l_pac, rb) ((((rb) > (l_pac)) == (parent)) ? 1 : 0)   /* mem_getbss, memchain.c:265 */

/* merge_seed_to_chain, memchain.c:227-256: 1 if the seed was absorbed by chain c */
static int chain_absorb(const bsx_opt_t *opt, int64_t l_pac, chain_t *c, const seed_t *s, int seed_rid)
{
	const seed_t *first = &c->seeds.a[0], *last = &c->seeds.a[c->seeds.n - 1];
	int64_t qdist, rdist;
	if (seed_rid != c->rid) return 0;
	if (s->qbeg >= first->qbeg && s->qbeg + s->len <= last->qbeg + last->len &&
	    s->rbeg >= first->rbeg && s->rbeg + s->len <= last->rbeg + last->len) {
		bsx_cvec_push(c->seeds_extra, *s);   /* contained on both axes: kept as a back-up seed */
		return 1;
	}
	if ((last->rbeg < l_pac || first->rbeg < l_pac) && s->rbeg >= l_pac) return 0;  /* other strand */
	qdist = s->qbeg - last->qbeg;
	rdist = s->rbeg - last->rbeg;
	if (rdist >= 0 && qdist - rdist <= opt->w && rdist - qdist <= opt->w &&
	    qdist - last->len < opt->max_chain_gap && rdist - last->len < opt->max_chain_gap) {
		bsx_cvec_push(c->seeds, *s);
		return 1;
	}
	return 0;
}

void bsx_chain_free(chain_v *chains)
{
	size_t i;
	for (i = 0; i < chains->n; ++i) { bsx_cvec_free(chains->a[i].seeds); bsx_cvec_free(chains->a[i].seeds_extra); }
	chains->n = 0;
}

int bsx_chain_build(const bsx_opt_t *opt, const bsx_refmeta_t *ref, int l_seq, int parent,
                    const bsx_intv_t *intv, int n_intv, const uint64_t *pos, const int64_t *pos_off,
                    bsx_btree_t *tree, chain_v *out)
{
	int i, b, e, l_rep;
	int64_t l_pac = ref->l_pac;
	chain_v pool;
	int32_t *ids;
	int n_ids;
	float frac_rep;

	out->n = 0;
	if (l_seq < opt->min_seed_len) return 0;
	/* read length covered by over-represented seeds (memchain.c:294-301) */
	for (i = 0, b = e = l_rep = 0; i < n_intv; ++i) {
		int sb, se;
		if (intv[i].x[2] <= opt->max_occ) continue;
		sb = (int)(intv[i].info >> 32); se = (int)(uint32_t)intv[i].info;
		if (sb > e) { l_rep += e - b; b = sb; e = se; }
		else e = e > se ? e : se;
	}
	l_rep += e - b;

	bsx_vec_init(pool);
	bsx_bt_clear(tree);
	for (i = 0; i < n_intv; ++i) {
		const bsx_intv_t *p = &intv[i];
		int slen = (int)((uint32_t)p->info - (uint32_t)(p->info >> 32));
		int64_t avail = pos_off[i + 1] - pos_off[i];
		uint32_t count; uint64_t k;
		/* visit every occurrence while few chains came out of this interval, else cap at max_occ
		 * (memchain.c:325-326) */
		for (k = count = 0; k < p->x[2] && count < opt->max_occ && ((count > 5 && k < opt->max_occ) || count <= 5); ++k) {
			seed_t s;
			int rid, to_add = 0;
			if ((int64_t)k >= avail) { /* the caller must look up more occurrences of interval i */
				size_t u;
				for (u = 0; u < pool.n; ++u) { bsx_cvec_free(pool.a[u].seeds); bsx_cvec_free(pool.a[u].seeds_extra); }
				bsx_cvec_free(pool);
				return 1 + i;
			}
			s.rbeg = (int64_t)pos[pos_off[i] + (int64_t)k];
			s.qbeg = (int32_t)(p->info >> 32);
			s.score = s.len = slen;
			rid = bsx_intv2rid(ref, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;   /* spans two contigs or the strand boundary */
			if ((opt->bsstrand & 1) && getbss(parent, l_pac, s.rbeg) != opt->bsstrand >> 1) continue;
			if (bsx_bt_size(tree)) {
				int32_t lower = bsx_bt_lower(tree, s.rbeg);
				if (lower < 0 || !chain_absorb(opt, l_pac, &pool.a[lower], &s, rid)) to_add = 1;
			} else to_add = 1;
			if (to_add) {
				chain_t c;
				memset(&c, 0, sizeof(c));
				++count;
				bsx_cvec_push(c.seeds, s);
				c.rid = rid;
				c.is_alt = !!ref->anns[rid].is_alt;
				c.pos = s.rbeg;
				bsx_cvec_push(pool, c);
				bsx_bt_put(tree, c.pos, (int32_t)(pool.n - 1));
			}
		}
	}
	/* in-order traversal of the tree gives the chain order (memchain.c:372-379) */
	ids = (int32_t*)bsx_crealloc(0, 0, sizeof(int32_t) * (pool.n + 1));
	n_ids = bsx_bt_traverse(tree, ids);
	frac_rep = (float)l_rep / l_seq;
	bsx_cvec_reserve(*out, (size_t)n_ids + 1);
	for (i = 0; i < n_ids; ++i) { out->a[i] = pool.a[ids[i]]; out->a[i].frac_rep = frac_rep; }
	out->n = (size_t)n_ids;
	bsx_cfree(ids);
	bsx_cvec_free(pool);
	return 0;
}

/* mem_chain_weight, memchain.c:158-180 */
static int chain_weight(const chain_t *c)
{
	int64_t end;
	int w = 0, tmp;
	size_t j;
	for (j = 0, end = 0; j < c->seeds.n; ++j) {
		const seed_t *s = &c->seeds.a[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0; j < c->seeds.n; ++j) {
		const seed_t *s = &c->seeds.a[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
		end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

static int chain_w_desc(const void *a, const void *b) { return ((const chain_t*)a)->w > ((const chain_t*)b)->w; }

#define CH_BEG(c) ((c).seeds.a[0].qbeg)
#define CH_END(c) ((c).seeds.a[(c).seeds.n - 1].qbeg + (c).seeds.a[(c).seeds.n - 1].len)

void bsx_chain_filter(const bsx_opt_t *opt, chain_v *chns)
{
	uint32_t i, k;
	BSX_VEC(int) keep;
	if (chns->n == 0) return;
	bsx_vec_init(keep);
	for (i = k = 0; i < chns->n; ++i) {
		chain_t *c = &chns->a[i];
		c->first = -1; c->kept = 0;
		c->w = (uint32_t)chain_weight(c) & 0x1fffffffu;
		if ((int)c->w < opt->min_chain_weight) { bsx_cvec_free(c->seeds); bsx_cvec_free(c->seeds_extra); }
		else chns->a[k++] = *c;
	}
	chns->n = k;
	if (chns->n == 0) { bsx_cvec_free(keep); return; }
	bsx_introsort(chns->a, chns->n, sizeof(chain_t), chain_w_desc);
	chns->a[0].kept = 3;
	bsx_cvec_push(keep, 0);
	for (i = 1; i < chns->n; ++i) {
		int large_overlap = 0;
		for (k = 0; k < keep.n; ++k) {
			chain_t *ci = &chns->a[i], *ck = &chns->a[keep.a[k]];
			int b_max = CH_BEG(*ck) > CH_BEG(*ci) ? CH_BEG(*ck) : CH_BEG(*ci);
			int e_min = CH_END(*ck) < CH_END(*ci) ? CH_END(*ck) : CH_END(*ci);
			if (e_min > b_max && (!ck->is_alt || ci->is_alt)) {
				int li = CH_END(*ci) - CH_BEG(*ci), lj = CH_END(*ck) - CH_BEG(*ck);
				int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
					large_overlap = 1;
					if (ck->first < 0) ck->first = (int)i;
					if ((int)ci->w < (int)ck->w * opt->drop_ratio && (int)ck->w - (int)ci->w >= opt->min_seed_len << 1) break;
				}
			}
		}
		if (k == keep.n) {
			bsx_cvec_push(keep, (int)i);
			chns->a[i].kept = large_overlap ? 2 : 3;
		}
	}
	for (i = 0; i < keep.n; ++i) {
		chain_t *c = &chns->a[keep.a[i]];
		if (c->first >= 0) chns->a[c->first].kept = 1;
	}
	bsx_cvec_free(keep);
	for (i = k = 0; i < chns->n; ++i) { /* at most max_chain_extend shadowed chains survive */
		if (chns->a[i].kept == 0 || chns->a[i].kept == 3) continue;
		if (++k >= opt->max_chain_extend) break;
	}
	for (; i < chns->n; ++i)
		if (chns->a[i].kept < 3) chns->a[i].kept = 0;
	for (i = k = 0; i < chns->n; ++i) {
		chain_t *c = &chns->a[i];
		if (c->kept == 0) { bsx_cvec_free(c->seeds); bsx_cvec_free(c->seeds_extra); }
		else chns->a[k++] = *c;
	}
	chns->n = k;
}

/* cal_max_gap, memchain.c:576-582 */
int bsx_cal_max_gap(const bsx_opt_t *opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

/* mem_chain_reference_span, memchain.c:585-605 */
void bsx_chain_ref_span(const bsx_opt_t *opt, int l_query, int64_t l_pac, const chain_t *c, int64_t rmax[2])
{
	size_t i;
	rmax[0] = l_pac << 1; rmax[1] = 0;
	for (i = 0; i < c->seeds.n; ++i) {
		const seed_t *s = &c->seeds.a[i];
		int64_t b = s->rbeg - (s->qbeg + bsx_cal_max_gap(opt, s->qbeg));
		int64_t e = s->rbeg + s->len + ((l_query - s->qbeg - s->len) + bsx_cal_max_gap(opt, l_query - s->qbeg - s->len));
		rmax[0] = rmax[0] < b ? rmax[0] : b;
		rmax[1] = rmax[1] > e ? rmax[1] : e;
	}
	rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) {
		if (c->seeds.a[0].rbeg < l_pac) rmax[1] = l_pac;
		else rmax[0] = l_pac;
	}
}
