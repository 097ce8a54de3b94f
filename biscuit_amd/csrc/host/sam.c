/* sam.c -- K6 host side + F1: CIGAR post-processing (MD/NM/ZC/ZR), SAM record text, and the
 * SE/PE output drivers with MAPQ and pairing decisions.
 *
 * Follows mem_alnreg_setSAM / mem_alnreg_formatSAM / mem_reg2sam_{se,pe,pe_nopairing}
 * (lib/aln/mem_alnreg_format.c) and the MD/NM part of bis_bwa_gen_cigar2 (lib/aln/bwa.c:342-418).
 * The banded global DP itself (ksw_global2) runs on the device; to batch it, the output drivers are
 * executed twice per read (pair): a *plan* pass on a scratch copy that only records which regions
 * get a CIGAR, then -- after one K6 batch for the whole chunk -- the *final* pass, in which
 * "setSAM" copies the finished record into the region at exactly the moment the reference would
 * compute it (the SA/XA tag logic looks at which regions already carry a CIGAR).
 */
#include <math.h>
#include <limits.h>
#include "align_types.h"
#include "pipeline.h"

char bsx_rg_id[256];

/* ------------------------------------------------------------------ tiny string builder */
typedef struct { char *s; size_t l, m; } sbuf_t;
static inline void sb_need(sbuf_t *b, size_t extra)
{
	if (b->l + extra + 1 > b->m) { b->m = (b->l + extra + 1) * 2 + 64; b->s = (char*)realloc(b->s, b->m); }
}
/* room for the usual record (two copies of the read, the tags) in one allocation instead of a doubling series */
static inline void sb_reserve(sbuf_t *b, size_t m) { if (m > b->m) { b->m = m; b->s = (char*)realloc(b->s, b->m); if (b->l == 0) b->s[0] = 0; } }
static inline void sb_putc(sbuf_t *b, int c) { sb_need(b, 1); b->s[b->l++] = (char)c; b->s[b->l] = 0; }
static inline void sb_putsn(sbuf_t *b, const char *p, size_t n) { sb_need(b, n); memcpy(b->s + b->l, p, n); b->l += n; b->s[b->l] = 0; }
static inline void sb_puts(sbuf_t *b, const char *p) { sb_putsn(b, p, strlen(p)); }
static inline void sb_putl(sbuf_t *b, long c)   /* kputl/kputw: decimal, '-' for negatives */
{
	char buf[32];
	int l = 0;
	unsigned long x = c < 0 ? -(unsigned long)c : (unsigned long)c;
	do { buf[l++] = (char)(x % 10 + '0'); x /= 10; } while (x > 0);
	if (c < 0) buf[l++] = '-';
	sb_need(b, l);
	while (l > 0) b->s[b->l++] = buf[--l];
	b->s[b->l] = 0;
}
#define sb_putw(b, c) sb_putl(b, (long)(int)(c))

/* a builder that starts in a caller's stack buffer: moved to the heap the first time it has to grow */
static inline void sb_room_from_stack(sbuf_t *b, const char *stack, size_t extra)
{
	if (b->l + extra + 1 <= b->m) return;
	b->m = (b->l + extra + 1) * 2 + 64;
	if (b->s == stack) { char *h = (char*)malloc(b->m); memcpy(h, stack, b->l + 1); b->s = h; }
	else b->s = (char*)realloc(b->s, b->m);
}

/* ------------------------------------------------------------------ K6 job + finish */
static int infer_bw(int l1, int l2, int score, int a, int q, int r)   /* bwamem.h:192-198 */
{
	int w, d;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	d = l1 - l2; d = d < 0 ? -d : d;
	if (w < d) w = d;
	return w;
}

void bsx_setsam_job(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_read_t *s, uint32_t qoff, const reg_t *reg, bsx_glb_job_t *job)
{
	int w1 = infer_bw(reg->qe - reg->qb, (int)(reg->re - reg->rb), reg->truesc, opt->a, opt->o_del, opt->e_del);
	int w2 = infer_bw(reg->qe - reg->qb, (int)(reg->re - reg->rb), reg->truesc, opt->a, opt->o_ins, opt->e_ins);
	int w = w1 > w2 ? w1 : w2, rev = reg->rb >= idx->ref.l_pac;
	(void)s;
	if (w > opt->w) w = w < reg->w ? w : reg->w;
	memset(job, 0, sizeof(*job));
	job->qlen = reg->qe - reg->qb; job->tlen = (int32_t)(reg->re - reg->rb);
	/* reverse-strand hits are aligned on reversed sequences so that indels left-align (bwa.c:307-312) */
	job->qoff = qoff + (uint32_t)(rev ? reg->qe - 1 : reg->qb); job->qdir = rev ? -1 : 1;
	job->tpos = rev ? reg->re - 1 : reg->rb; job->tdir = rev ? -1 : 1;
	job->w0 = w; job->w_max = opt->w << 2; job->truesc = reg->truesc; job->n_try = 3;
	job->use_ct = reg->parent; job->want_cigar = 1;
}

/* tag != NULL: NM / MD / ZC / ZR came with the CIGAR (bsx_global_batch_tags: computed by the device over the same job) */
void bsx_setsam_finish(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_read_t *s, const reg_t *reg,
                       const uint32_t *cg, int n_cigar, samrec_t *out, const bsx_glb_tag_t *tag, const char *tag_md)
{
	int64_t l_pac = idx->ref.l_pac, rpos;
	int rev = reg->rb >= l_pac, k, x, y, u, i, is_rev;
	int n_mm = 0, n_gap = 0, n_conv_ct = 0, n_ret_c = 0, n_conv_ga = 0, n_ret_g = 0;
	const char *int2base = reg->rb < l_pac ? "ACGTN" : "TGCAN";
	const uint8_t *query = s->seq;
	int parent = reg->parent, l_MD;
	/* MD (at most two characters per reference base, plus the last count) in a stack buffer unless the alignment is unusually
	 * long (one heap block per record otherwise: a million per chunk); the reference bases of the alignment, decoded once in
	 * alignment order: a forward walk over pac on either strand */
	char md_stack[1024];
	uint8_t rb_stack[480], *rbase = rb_stack;
	sbuf_t md = {md_stack, 0, sizeof(md_stack)};   /* grows through sb_room_from_stack only */
	uint32_t *cigar;
	(void)opt;
	md_stack[0] = 0;
	if (tag && tag->l_md >= 0) {
		l_MD = tag->l_md + 1;
		memset(out, 0, sizeof(*out));
		out->NM = tag->NM; out->ZC = tag->ZC; out->ZR = tag->ZR; out->bss_u = tag->bss_u;
	} else { /* a backend without them (the batch entry points one by one, the CPU checker's): here */
		const int64_t rl = reg->re - reg->rb, f0 = rev ? (l_pac << 1) - reg->re : reg->rb;   /* forward coordinate of the first base */
		int64_t j;
		if (rl > (int64_t)sizeof(rb_stack)) {
			rbase = (uint8_t*)malloc((size_t)rl);
		}
		if (rev) for (j = 0; j < rl; ++j) rbase[j] = (uint8_t)(3 - bsx_pac_get(idx->pac, f0 + j));
		else for (j = 0; j < rl; ++j) rbase[j] = (uint8_t)bsx_pac_get(idx->pac, f0 + j);
	/* MD / NM / ZC / ZR over the alignment (bwa.c:342-418); conversions are MD mismatches but not NM */
	for (k = 0, x = y = u = 0; k < n_cigar; ++k) {
		int op = cg[k] & 0xf, len = (int)(cg[k] >> 4);
		if (op == 0) {
			for (i = 0; i < len; ++i) {
				int q = rev ? query[reg->qe - 1 - (x + i)] : query[reg->qb + x + i];
				int r = rbase[y + i];
				if (q == r) {
					if (q == 1) ++n_ret_c;
					if (q == 2) ++n_ret_g;
					++u;
				} else {
					sb_room_from_stack(&md, md_stack, 16);
					sb_putw(&md, u); sb_putc(&md, int2base[r]); u = 0;
					if (parent && q == 3 && r == 1) ++n_conv_ct;
					else if (!parent && q == 0 && r == 2) ++n_conv_ga;
					else ++n_mm;
				}
			}
			x += len; y += len;
		} else if (op == 2) {
			if (k > 0 && k < n_cigar - 1) {
				sb_room_from_stack(&md, md_stack, 16 + (size_t)len);
				sb_putw(&md, u); sb_putc(&md, '^');
				for (i = 0; i < len; ++i) sb_putc(&md, int2base[rbase[y + i]]);
				u = 0; n_gap += len;
			}
			y += len;
		} else if (op == 1) { x += len; n_gap += len; }
	}
	sb_room_from_stack(&md, md_stack, 16);
	sb_putw(&md, u);
	l_MD = (int)md.l + 1;
	memset(out, 0, sizeof(*out));
	out->NM = n_mm + n_gap;
	out->ZC = parent ? n_conv_ct : n_conv_ga;
	out->ZR = parent ? n_ret_c : n_ret_g;
	out->bss_u = (n_conv_ct == 0 && n_conv_ga == 0) ? 1 : 0;
	}
	/* position, strand, D squeezing and clipping (mem_alnreg_format.c:79-120) */
	cigar = (uint32_t*)bsx_crealloc(0, 0, 4 * ((size_t)n_cigar + 2) + l_MD + 4);   /* chunk lifetime: from the worker's arena */
	memcpy(cigar, cg, 4 * (size_t)n_cigar);
	rpos = bsx_depos(l_pac, reg->rb < l_pac ? reg->rb : reg->re - 1, &is_rev);
	out->is_rev = (uint32_t)is_rev;
	if (n_cigar > 0) {
		if ((cigar[0] & 0xf) == 2) { rpos += cigar[0] >> 4; --n_cigar; memmove(cigar, cigar + 1, (size_t)n_cigar * 4); }
		else if ((cigar[n_cigar - 1] & 0xf) == 2) --n_cigar;
	}
	if (reg->qb != 0 || reg->qe != s->l_seq || s->clip5 || s->clip3) {
		int clip5 = is_rev ? s->l_seq - reg->qe + s->clip3 : reg->qb + s->clip5;
		int clip3 = is_rev ? reg->qb + s->clip5 : s->l_seq - reg->qe + s->clip3;
		if (clip5) { memmove(cigar + 1, cigar, (size_t)n_cigar * 4); cigar[0] = (uint32_t)clip5 << 4 | 3; ++n_cigar; }
		if (clip3) cigar[n_cigar++] = (uint32_t)clip3 << 4 | 3;
	}
	if (tag && tag->l_md >= 0) memcpy(cigar + n_cigar, tag_md, (size_t)tag->l_md + 1);
	else memcpy(cigar + n_cigar, md.s, md.l + 1);
	if (md.s != md_stack) free(md.s);
	if (rbase != rb_stack) free(rbase);
	out->n_cigar = n_cigar;
	out->cigar = cigar;
	out->pos = (int)(rpos - idx->ref.anns[reg->rid].offset);
	out->valid = 1;
}

/* ------------------------------------------------------------------ "setSAM" inside the drivers */
typedef struct {
	const bsx_opt_t *opt; const bsx_index_t *idx; samctx_t *ctx; const char *rg_id;
} drv_t;

/* which = read of the pair (0/1); regs = that read's region vector */
static void set_sam(drv_t *D, int which, reg_v *regs, reg_t *reg)
{
	int ri = (int)(reg - regs->a);
	if (reg->n_cigar > 0) return;   /* mem_alnreg_format.c:42 */
	if (D->ctx->plan) {
		bsx_cvec_push(D->ctx->want[which], ri);
		reg->n_cigar = 1;  /* stands for "will have a CIGAR" during planning */
		return;
	}
	{
		samrec_t *t = &D->ctx->table[which][ri];
		if (!t->valid) { fprintf(stderr, "[bsx] internal: CIGAR of region %d was not planned\n", ri); abort(); }
		reg->is_rev = t->is_rev;
		reg->flag |= reg->is_rev ? 0x10 : 0;
		reg->NM = t->NM; reg->ZC = t->ZC; reg->ZR = t->ZR; reg->bss_u = t->bss_u;
		reg->n_cigar = t->n_cigar;
		if (t->n_cigar > 0) reg->cigar = t->cigar;
		reg->pos = t->pos;
	}
}

static int get_pri_idx(double XA_drop_ratio, const reg_t *a, int i)   /* mem_alnreg.h:122-126 */
{
	int k = a[i].secondary_all;
	if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
	return -1;
}

static int get_rlen(int n_cigar, const uint32_t *cigar)   /* bwamem.h:200-208 */
{
	int k, l;
	for (k = l = 0; k < n_cigar; ++k) { int op = cigar[k] & 0xf; if (op == 0 || op == 2) l += (int)(cigar[k] >> 4); }
	return l;
}

/* mem_alnreg_tagXAXB, mem_alnreg_format.c:126-191 */
static void tag_XAXB(drv_t *D, int which, bsx_read_t *s, const reg_t *p0, reg_v *regs0, sbuf_t *out)
{
	const bsx_opt_t *opt = D->opt;
	int cnt_pri = 0, cnt_alt = 0;
	size_t i;
	(void)s;
	if (!regs0 || (opt->flag & BSX_F_ALL)) return;
	for (i = 0; i < regs0->n; ++i) {
		int r = get_pri_idx(opt->XA_drop_ratio, regs0->a, (int)i);
		if (r >= 0 && regs0->a + r == p0) { if (regs0->a[i].is_alt) ++cnt_alt; else ++cnt_pri; }
	}
	if (cnt_pri <= opt->max_XA_hits && cnt_alt <= opt->max_XA_hits_alt) {
		sbuf_t str = {0, 0, 0};
		int n = 0;
		for (i = 0; i < regs0->n; ++i) {
			reg_t *q = &regs0->a[i];
			int r = get_pri_idx(opt->XA_drop_ratio, regs0->a, (int)i), k;
			if (r < 0 || regs0->a + r != p0) continue;
			if (q->n_cigar == 0) { set_sam(D, which, regs0, q); if (q->n_cigar == 0) continue; }
			if (D->ctx->plan) { ++n; continue; }
			if (n) sb_putc(&str, ';');
			sb_puts(&str, D->idx->ref.anns[q->rid].name);
			sb_putc(&str, ','); sb_putc(&str, "+-"[q->is_rev]); sb_putl(&str, q->pos + 1); sb_putc(&str, ',');
			for (k = 0; k < q->n_cigar; ++k) { sb_putw(&str, q->cigar[k] >> 4); sb_putc(&str, "MIDSHN"[q->cigar[k] & 0xf]); }
			sb_putc(&str, ','); sb_putw(&str, q->NM);
			++n;
		}
		if (str.l) { sb_putsn(out, "\tXA:Z:", 6); sb_puts(out, str.s); }
		free(str.s);
	}
	if (cnt_pri > 0 || cnt_alt > 0) { sb_putsn(out, "\tXB:Z:", 6); sb_putw(out, cnt_pri); sb_putc(out, ','); sb_putw(out, cnt_alt); }
}

/* mem_alnreg_tagSA, mem_alnreg_format.c:194-228: every other region that already carries a CIGAR */
static void tag_SA(drv_t *D, const reg_t *p0, const reg_v *regs0, sbuf_t *out)
{
	sbuf_t str = {0, 0, 0};
	size_t i;
	if (!regs0 || (p0->flag & 0x100)) return;
	for (i = 0; i < regs0->n; ++i) {
		const reg_t *q = &regs0->a[i];
		int k;
		if (q == p0 || q->n_cigar == 0 || (q->flag & 0x100)) continue;
		sb_puts(&str, D->idx->ref.anns[q->rid].name); sb_putc(&str, ',');
		sb_putl(&str, q->pos + 1); sb_putc(&str, ',');
		sb_putc(&str, "+-"[q->is_rev]); sb_putc(&str, ',');
		for (k = 0; k < q->n_cigar; ++k) { sb_putw(&str, q->cigar[k] >> 4); sb_putc(&str, "MIDSH"[q->cigar[k] & 0xf]); }
		sb_putc(&str, ','); sb_putw(&str, q->mapq);
		sb_putc(&str, ','); sb_putw(&str, q->NM);
		sb_putc(&str, ';');
	}
	if (str.l) { sb_putsn(out, "\tSA:Z:", 6); sb_puts(out, str.s); }
	free(str.s);
}

static int is_proper_pair(const bsx_refmeta_t *ref, const reg_t *r1, const reg_t *r2, const bsx_pestat_t *pes)
{
	int64_t isize;
	if (!bsx_reg_isize(ref, r1, r2, &isize)) return 0;
	return isize >= pes->low && isize <= pes->high;
}

static void put_cigar(sbuf_t *str, const bsx_opt_t *opt, const reg_t *p, int is_primary)
{
	int i;
	for (i = 0; i < p->n_cigar; ++i) {
		int c = p->cigar[i] & 0xf;
		if (!(opt->flag & BSX_F_SOFTCLIP) && !p->is_alt && (c == 3 || c == 4)) c = is_primary ? 3 : 4;
		sb_putw(str, p->cigar[i] >> 4); sb_putc(str, "MIDSH"[c]);
	}
}

/* mem_alnreg_formatSAM, mem_alnreg_format.c:237-436 */
static void format_sam(drv_t *D, int which, sbuf_t *str, bsx_read_t *s, const reg_t *p0, const reg_t *m0,
                       reg_v *regs0, int is_primary, const bsx_pestat_t *pes)
{
	const bsx_opt_t *opt = D->opt;
	const bsx_refmeta_t *ref = &D->idx->ref;
	reg_t p = *p0, m;
	int i;
	if (D->ctx->plan) { /* planning only needs the side effects on CIGAR availability */
		if (D->ctx->trace && D->ctx->n_trace < D->ctx->m_trace) {
			int *t = D->ctx->trace[D->ctx->n_trace++];
			t[0] = which;
			t[1] = regs0 && p0 >= regs0->a && p0 < regs0->a + regs0->n ? (int)(p0 - regs0->a) : -1;
			t[2] = !m0 ? -2 : m0->rid < 0 ? -1 : (int)m0->hash;   /* the hook numbers the mate's regions in `hash` */
			t[3] = p0->flag; t[4] = (int)p0->mapq; t[5] = is_primary;
		}
		if (regs0) tag_XAXB(D, which, s, p0, regs0, str);
		return;
	}
	memset(&m, 0, sizeof(m));
	if (m0) m = *m0;
	p.flag |= m0 ? 0x1 : 0;
	p.flag |= m0 && m.rid < 0 ? 0x8 : 0;
	if (m0 && m0->bss_u == 0) p.bss_u = 0;
	if (p.rid >= 0 && m0 && m.rid >= 0 && pes && is_proper_pair(ref, &p, &m, pes)) { p.flag |= 2; m.flag |= 2; }
	if (p.rid < 0 && m0 && m.rid >= 0) { p.rid = m.rid; p.pos = m.pos; p.is_rev = m.is_rev; p.n_cigar = 0; }
	if (m0 && m.rid < 0 && p.rid >= 0) { m.rid = p.rid; m.pos = p.pos; m.is_rev = p.is_rev; m.n_cigar = 0; }
	p.flag |= m0 && m.is_rev ? 0x20 : 0;

	sb_puts(str, s->name);
	if (s->comment) { sb_putc(str, '_'); sb_puts(str, s->comment); }
	sb_putc(str, '\t');
	sb_putw(str, (p.flag & 0xffff) | (p.flag & 0x10000 ? 0x100 : 0)); sb_putc(str, '\t');
	if (p.rid >= 0) {
		sb_puts(str, ref->anns[p.rid].name); sb_putc(str, '\t');
		sb_putl(str, p.pos + 1); sb_putc(str, '\t');
		sb_putw(str, p.mapq); sb_putc(str, '\t');
		if (p.n_cigar) put_cigar(str, opt, &p, is_primary);
		else sb_putc(str, '*');
	} else sb_putsn(str, "*\t0\t0\t*", 7);
	sb_putc(str, '\t');
	if (m0 && m.rid >= 0) {
		if (p.rid == m.rid) sb_putc(str, '=');
		else sb_puts(str, ref->anns[m.rid].name);
		sb_putc(str, '\t');
		sb_putl(str, m.pos + 1); sb_putc(str, '\t');
		if (p.rid == m.rid) { /* TLEN: leftmost forward start to rightmost reverse end (differs from BWA) */
			int64_t q0 = -1, q1 = -1;
			if (p.is_rev) q1 = p.pos + get_rlen(p.n_cigar, p.cigar) - 1; else q0 = p.pos;
			if (m.is_rev) q1 = m.pos + get_rlen(m.n_cigar, m.cigar) - 1; else q0 = m.pos;
			if (p.n_cigar > 0 && m.n_cigar > 0 && q0 >= 0 && q1 >= 0) sb_putl(str, q1 - q0 + 1);
			else sb_putc(str, '0');
		} else sb_putc(str, '0');
	} else sb_putsn(str, "*\t0\t0", 5);
	sb_putc(str, '\t');
	if (p.flag & 0x100) sb_putsn(str, "*\t*", 3);
	else {
		int qb = 0, qe = s->l_seq0;
		if (p.n_cigar && !is_primary && !(opt->flag & BSX_F_SOFTCLIP) && !p.is_alt) { /* hard clip */
			int c0 = p.cigar[0] & 0xf, c1 = p.cigar[p.n_cigar - 1] & 0xf;
			if (p.is_rev) {
				if (c0 == 4 || c0 == 3) qe -= (int)(p.cigar[0] >> 4);
				if (c1 == 4 || c1 == 3) qb += (int)(p.cigar[p.n_cigar - 1] >> 4);
			} else {
				if (c0 == 4 || c0 == 3) qb += (int)(p.cigar[0] >> 4);
				if (c1 == 4 || c1 == 3) qe -= (int)(p.cigar[p.n_cigar - 1] >> 4);
			}
		}
		sb_need(str, (size_t)(qe - qb > 0 ? qe - qb : 0) * 2 + 4);
		if (p.is_rev) {
			for (i = qe - 1; i >= qb; --i) str->s[str->l++] = "TGCAN"[(int)s->seq0[i]];
			str->s[str->l] = 0;
			sb_putc(str, '\t');
			if (s->qual) { sb_need(str, (size_t)(qe - qb > 0 ? qe - qb : 0) + 2); for (i = qe - 1; i >= qb; --i) str->s[str->l++] = s->qual[i]; str->s[str->l] = 0; }
			else sb_putc(str, '*');
		} else {
			for (i = qb; i < qe; ++i) str->s[str->l++] = "ACGTN"[(int)s->seq0[i]];
			str->s[str->l] = 0;
			sb_putc(str, '\t');
			if (s->qual) { sb_need(str, (size_t)(qe - qb > 0 ? qe - qb : 0) + 2); for (i = qb; i < qe; ++i) str->s[str->l++] = s->qual[i]; str->s[str->l] = 0; }
			else sb_putc(str, '*');
		}
	}
	if (p.n_cigar) {
		sb_putsn(str, "\tNM:i:", 6); sb_putw(str, p.NM);
		sb_putsn(str, "\tMD:Z:", 6); sb_puts(str, (char*)(p.cigar + p.n_cigar));
		sb_putsn(str, "\tZC:i:", 6); sb_putw(str, p.ZC);
		sb_putsn(str, "\tZR:i:", 6); sb_putw(str, p.ZR);
	}
	if (p.score >= 0) { sb_putsn(str, "\tAS:i:", 6); sb_putw(str, p.score); }
	if (p.sub >= 0) { sb_putsn(str, "\tXS:i:", 6); sb_putw(str, p.sub > p.csub ? p.sub : p.csub); }
	if (D->rg_id && D->rg_id[0]) { sb_putsn(str, "\tRG:Z:", 6); sb_puts(str, D->rg_id); }
	if (regs0) tag_SA(D, p0, regs0, str);
	if (is_primary && p.alt_sc > 0) { char buf[64]; snprintf(buf, sizeof(buf), "\tPA:f:%.3f", (double)p.score / p.alt_sc); sb_puts(str, buf); }
	sb_putsn(str, "\tXL:i:", 6); sb_putw(str, s->l_seq);
	if (regs0) tag_XAXB(D, which, s, p0, regs0, str);
	if ((opt->flag & BSX_F_REF_HDR) && p.rid >= 0 && ref->anns[p.rid].anno != 0 && ref->anns[p.rid].anno[0] != 0) {
		size_t tmp, k;
		sb_putsn(str, "\tXR:Z:", 6);
		tmp = str->l;
		sb_puts(str, ref->anns[p.rid].anno);
		for (k = tmp; k < str->l; ++k) if (str->s[k] == '\t') str->s[k] = ' ';
	}
	if (s->barcode) { sb_putsn(str, "\tCB:Z:", 6); sb_puts(str, s->barcode); }
	if (s->umi) { sb_putsn(str, "\tRX:Z:", 6); sb_puts(str, s->umi); }
	sb_putsn(str, "\tMC:Z:", 6);
	if (m.n_cigar) put_cigar(str, opt, &m, is_primary);
	else sb_putc(str, '*');
	sb_putsn(str, "\tMQ:i:", 6); sb_putw(str, m.mapq);
	sb_putsn(str, "\tYD:A:", 6);
	if (p.bss_u) sb_putc(str, 'u');
	else sb_putc(str, "fr"[p.bss]);
	sb_putc(str, '\n');
}

typedef BSX_VEC(int) int_v;

/* mem_alnreg_select_format, mem_alnreg_format.c:445-488 */
static int_v select_format(drv_t *D, int which, reg_v *regs)
{
	const bsx_opt_t *opt = D->opt;
	int_v out;
	int l; size_t k;
	bsx_vec_init(out);
	for (k = 0, l = 0; k < regs->n; ++k) {
		reg_t *p = &regs->a[k];
		if (p->rb < 0 || p->re < 0) continue;
		if (p->score < opt->T) continue;
		if (p->secondary >= 0 && (p->is_alt || !(opt->flag & BSX_F_ALL))) continue;
		if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < regs->a[p->secondary].score * opt->drop_ratio) continue;
		if (l && p->secondary < 0) p->flag |= (opt->flag & BSX_F_NO_MULTI) ? 0x10000 : 0x800;
		if (p->secondary >= 0) p->flag |= 0x100;
		p->mapq = p->secondary < 0 ? (unsigned)bsx_approx_mapq_se(opt, p) : 0;
		if (!(opt->flag & BSX_F_KEEP_SUPP_MAPQ) && l && !p->is_alt) p->mapq = p->mapq < regs->a[0].mapq ? p->mapq : regs->a[0].mapq;
		set_sam(D, which, regs, p);
		bsx_vec_push(out, (int)k);
		++l;
	}
	return out;
}

static void init_unmapped(reg_t *r, int flag) { memset(r, 0, sizeof(*r)); r->rid = -1; r->flag = flag; }

/* mem_reg2sam_se, mem_alnreg_format.c:492-515 */
void bsx_reg2sam_se(const bsx_opt_t *opt, const bsx_index_t *idx, bsx_read_t *s, reg_v *regs, samctx_t *ctx, const char *rg_id)
{
	drv_t D = { opt, idx, ctx, rg_id };
	sbuf_t str = {0, 0, 0};
	if (!ctx->plan) sb_reserve(&str, (size_t)s->l_seq0 * 3 + 200);
	int_v sel = select_format(&D, 0, regs);
	if (sel.n > 0) {
		size_t i;
		for (i = 0; i < sel.n; ++i) format_sam(&D, 0, &str, s, &regs->a[sel.a[i]], NULL, regs, !i, NULL);
	} else {
		reg_t reg; init_unmapped(&reg, 0x4);
		format_sam(&D, 0, &str, s, &reg, NULL, regs, 1, NULL);
	}
	if (!ctx->plan) s->sam = str.s; else free(str.s);
	bsx_vec_free(sel);
}

/* mem_reg2sam_pe_nopairing, mem_alnreg_format.c:519-559 */
static void pe_nopairing(drv_t *D, bsx_read_t s[2], reg_v regs[2], const bsx_pestat_t *pes)
{
	reg_t *best[2] = {0, 0}, unmapped[2];
	int_v sel[2];
	int i;
	for (i = 0; i < 2; ++i) {
		sel[i] = select_format(D, i, &regs[i]);
		if (sel[i].n > 0) best[i] = &regs[i].a[sel[i].a[0]];
		else { init_unmapped(&unmapped[i], 0x40 << i | 0x1 | 0x4); best[i] = &unmapped[i]; }
	}
	for (i = 0; i < 2; ++i) {
		sbuf_t str = {0, 0, 0};
		if (!D->ctx->plan) sb_reserve(&str, (size_t)s[i].l_seq0 * 3 + 200);
		if (sel[i].n) {
			size_t j;
			for (j = 0; j < sel[i].n; ++j) {
				reg_t *p = &regs[i].a[sel[i].a[j]];
				if (!best[!i]) p->flag |= 0x8;
				format_sam(D, i, &str, &s[i], p, best[!i], &regs[i], !j, pes);
			}
		} else format_sam(D, i, &str, &s[i], best[i], best[!i], NULL, 1, pes);
		if (!D->ctx->plan) s[i].sam = str.s; else free(str.s);
	}
	for (i = 0; i < 2; ++i) bsx_vec_free(sel[i]);
}

#define raw_mapq(diff, a) ((int)(6.02 * (diff) / (a) + .499))
#define imin(a, b) ((a) < (b) ? (a) : (b))
#define imax(a, b) ((a) > (b) ? (a) : (b))

/* mem_reg2sam_pe, mem_alnreg_format.c:562-696 */
void bsx_reg2sam_pe(const bsx_opt_t *opt, const bsx_index_t *idx, uint64_t id, bsx_read_t s[2], reg_v regs[2],
                    const bsx_pestat_t *pes, samctx_t *ctx, const char *rg_id)
{
	drv_t D = { opt, idx, ctx, rg_id };
	int i, is_multi[2], pscore, sub_pscore, n_sub, z[2] = {0, 0}, score_unpaired;
	size_t k, j;
	for (i = 0; i < 2; ++i)
		for (k = 0; k < regs[i].n; ++k) regs[i].a[k].flag |= (0x40 << i) | 1;
	if (opt->flag & BSX_F_NOPAIRING) { pe_nopairing(&D, s, regs, pes); return; }
	if (regs[0].n_pri == 0 || regs[1].n_pri == 0) { pe_nopairing(&D, s, regs, pes); return; }
	for (i = 0; i < 2; ++i) { /* a second good primary hit on either end disables pairing */
		for (j = 1; j < regs[i].n_pri; ++j)
			if (regs[i].a[j].secondary < 0 && regs[i].a[j].score >= opt->T) break;
		is_multi[i] = j < regs[i].n_pri ? 1 : 0;
	}
	if (is_multi[0] || is_multi[1]) { pe_nopairing(&D, s, regs, pes); return; }
	bsx_pair(opt, &idx->ref, pes, regs, (int)id, &pscore, &sub_pscore, &n_sub, z);
	if (pscore <= 0) { pe_nopairing(&D, s, regs, pes); return; }
	score_unpaired = regs[0].a[0].score + regs[1].a[0].score - opt->pen_unpaired;
	if (pscore > score_unpaired) {
		int q_pe, q_se[2];
		reg_t *c[2];
		sub_pscore = imax(sub_pscore, score_unpaired);
		q_pe = raw_mapq(pscore - sub_pscore, opt->a);
		if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
		q_pe = imax(0, imin(60, q_pe));
		q_pe = (int)(q_pe * (1. - .5 * (regs[0].a[0].frac_rep + regs[1].a[0].frac_rep)) + .499);
		c[0] = &regs[0].a[z[0]]; c[1] = &regs[1].a[z[1]];
		for (i = 0; i < 2; ++i) {
			if (c[i]->secondary >= 0) { c[i]->sub = regs[i].a[c[i]->secondary].score; c[i]->secondary = -2; }
			q_se[i] = bsx_approx_mapq_se(opt, c[i]);
		}
		q_se[0] = imax(q_se[0], imin(q_pe, q_se[0] + 40));
		q_se[1] = imax(q_se[1], imin(q_pe, q_se[1] + 40));
		c[0]->mapq = (unsigned)imin(q_se[0], raw_mapq(c[0]->score - c[0]->csub, opt->a));
		c[1]->mapq = (unsigned)imin(q_se[1], raw_mapq(c[1]->score - c[1]->csub, opt->a));
	} else {
		z[0] = z[1] = 0;
		regs[0].a[0].mapq = (unsigned)bsx_approx_mapq_se(opt, &regs[0].a[0]);
		regs[1].a[0].mapq = (unsigned)bsx_approx_mapq_se(opt, &regs[1].a[0]);
	}
	for (i = 0; i < 2; ++i) { /* a chosen secondary trades places with its primary */
		reg_v *r = &regs[i];
		int kk = r->a[z[i]].secondary_all;
		if (kk >= 0 && (size_t)kk < r->n_pri) {
			for (j = 0; j < r->n; ++j)
				if (r->a[j].secondary_all == kk || j == (size_t)kk) r->a[j].secondary_all = z[i];
			r->a[z[i]].secondary_all = -1;
		}
	}
	for (i = 0; i < 2; ++i) set_sam(&D, i, &regs[i], &regs[i].a[z[i]]);
	for (i = 0; i < 2; ++i) {
		sbuf_t str = {0, 0, 0};
		if (!ctx->plan) sb_reserve(&str, (size_t)s[i].l_seq0 * 3 + 200);
		reg_v *r = &regs[i];
		format_sam(&D, i, &str, &s[i], &r->a[z[i]], &regs[!i].a[z[!i]], r, 1, pes);
		if (r->n_pri < r->n) { /* best ALT hit as an extra supplementary record */
			reg_t *p = &r->a[r->n_pri];
			if (p->score >= opt->T && p->secondary < 0) {
				p->flag |= 0x800;
				set_sam(&D, i, r, p);
				format_sam(&D, i, &str, &s[i], p, NULL, r, 0, pes);
			}
		}
		if (!ctx->plan) s[i].sam = str.s; else free(str.s);
	}
}

/* test hook: mem_alnreg_formatSAM of one record (regions given with their SAM side filled in); p_idx names the record's region in
 * regs0 when regs0 is given (the XA/XB/SA tags look at the list), m may be NULL; returns the length written to buf (cap bytes) */
#include "hook_types.h"
BSX_API int bsx_hook_format_sam(const bsx_opt_t *opt, const bsx_index_t *idx, bsx_read_t *s, const bsx_hook_reg_t *p, const bsx_hook_reg_t *m,
                                const bsx_hook_reg_t *regs0, int n_regs0, int p_idx, int is_primary, const bsx_pestat_t *pes, const char *rg_id,
                                char *buf, int cap)
{
	samctx_t ctx;
	drv_t D;
	reg_t P, M;
	reg_v v;
	sbuf_t str = {0, 0, 0};
	int k, l;
	memset(&ctx, 0, sizeof(ctx)); memset(&v, 0, sizeof(v));
	D.opt = opt; D.idx = idx; D.ctx = &ctx; D.rg_id = rg_id;
	if (regs0) {
		v.n = v.m = (size_t)n_regs0;
		v.a = (reg_t*)calloc(n_regs0 ? n_regs0 : 1, sizeof(reg_t));
		for (k = 0; k < n_regs0; ++k) bsx_hook_to_reg(&regs0[k], &v.a[k]);
	}
	bsx_hook_to_reg(p, &P);
	if (m) bsx_hook_to_reg(m, &M);
	format_sam(&D, 0, &str, s, regs0 && p_idx >= 0 ? &v.a[p_idx] : &P, m ? &M : 0, regs0 ? &v : 0, is_primary, pes);
	l = (int)str.l;
	if (l < cap) memcpy(buf, str.s, (size_t)l + 1); else l = -l;
	free(str.s); free(v.a);
	return l;
}

/* test hook: the part of mem_alnreg_setSAM after the alignment (mem_alnreg_format.c:79-120: position and strand, leading / trailing deletion
 * squeezed out, clipping added) given a CIGAR and its NM/MD/ZC/ZR; out_cigar receives the final operations (cap entries), returns their number */
BSX_API int bsx_hook_setsam_finish(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_read_t *s, const bsx_hook_reg_t *reg, const uint32_t *cg, int n_cigar,
                                   const bsx_glb_tag_t *tag, const char *md, int out[3], uint32_t *out_cigar, int cap, char *out_md, int md_cap)
{
	reg_t r;
	samrec_t rec;
	int k;
	bsx_hook_to_reg(reg, &r);
	bsx_setsam_finish(opt, idx, s, &r, cg, n_cigar, &rec, tag, md);
	out[0] = rec.pos; out[1] = (int)rec.is_rev; out[2] = rec.NM;
	if (rec.n_cigar > cap) return -1;
	for (k = 0; k < rec.n_cigar; ++k) out_cigar[k] = rec.cigar[k];
	snprintf(out_md, (size_t)md_cap, "%s", (const char*)(rec.cigar + rec.n_cigar));
	k = rec.n_cigar;
	bsx_cfree(rec.cigar);
	return k;
}

/* ---- test hooks (tests/test_oracle_golden.py): restated header-inline functions against the reference's own (ref_vectors.npz) */
BSX_API int bsx_hook_get_rlen(int n_cigar, const uint32_t *cigar) { return get_rlen(n_cigar, cigar); }
BSX_API int bsx_hook_get_pri_idx(double XA_drop_ratio, int n, const int *score, const int *secondary_all, int i)
{
	reg_t *a = (reg_t*)calloc((size_t)n, sizeof(reg_t));
	int k, r;
	for (k = 0; k < n; ++k) { a[k].score = score[k]; a[k].secondary_all = secondary_all[k]; }
	r = get_pri_idx(XA_drop_ratio, a, i);
	free(a);
	return r;
}
BSX_API int bsx_hook_is_proper_pair(int64_t l_pac, const int64_t a[5], const int64_t b[5], int low, int high)
{
	bsx_refmeta_t ref; reg_t r[2]; bsx_pestat_t pes;
	memset(&ref, 0, sizeof(ref)); memset(r, 0, sizeof(r)); memset(&pes, 0, sizeof(pes));
	ref.l_pac = l_pac; pes.low = low; pes.high = high;
	r[0].rid = (int)a[0]; r[0].rb = a[1]; r[0].re = a[2]; r[0].qb = (int)a[3]; r[0].qe = (int)a[4];
	r[1].rid = (int)b[0]; r[1].rb = b[1]; r[1].re = b[2]; r[1].qb = (int)b[3]; r[1].qe = (int)b[4];
	return is_proper_pair(&ref, &r[0], &r[1], &pes);
}
