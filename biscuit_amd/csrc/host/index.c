/* index.c -- the BISCUIT index: loader (what `align` needs) and an own builder.
 *
 * File formats (little endian, no magic):
 *   <b>.{par,dau}.bwt : u64 primary; u64 L2[1..4]; u32 words: per 128 symbols 4 x u64 running
 *                       counts then 8 x u32 2-bit symbols (first symbol in the top bits); one
 *                       trailing count block              (lib/aln/bwt.c:402-410,476-491;
 *                                                          lib/aln/bwtindex.c:130-154)
 *   <b>.{par,dau}.sa  : u64 primary; u64 L2[1..4]; u64 sa_intv; u64 seq_len; u64 sa[1..n_sa-1]
 *                                                         (lib/aln/bwt.c:412-454)
 *   <b>.bis.pac       : forward genome, 2 bits/base, N -> lrand48()&3 after srand48(11)
 *                                                         (lib/aln/bntseq.c:635-685)
 *   <b>.bis.ann/.amb  : text                              (lib/aln/bntseq.c:514-539,108-161)
 * The converted texts are  parent  = C>T of [fwd ; revcomp(fwd)],
 *                          daughter = G>A of [fwd ; revcomp(fwd)]   (lib/aln/bntseq.c:585-600).
 * BWT and SA of a text are unique, so the builder may use any suffix sorter (here: SA-IS, Nong,
 * Zhang & Chan 2009, written from the paper) and still emit byte-identical files.
 */
#include <zlib.h>
#include <errno.h>
#include <ctype.h>
#include "bsx_core.h"

/* ------------------------------------------------------------------------------------------
 * coordinate helpers
 * ------------------------------------------------------------------------------------------ */
int bsx_pos2rid(const bsx_refmeta_t *r, int64_t pos_f) /* bns_pos2rid, bntseq.c:356-369 */
{
	int left, mid, right;
	if (pos_f >= r->l_pac) return -1;
	left = 0; mid = 0; right = r->n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= r->anns[mid].offset) {
			if (mid == r->n_seqs - 1) break;
			if (pos_f < r->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

int bsx_intv2rid(const bsx_refmeta_t *r, int64_t rb, int64_t re) /* bns_intv2rid, bntseq.c:371-379 */
{
	int is_rev, rid_b, rid_e;
	if (rb < r->l_pac && re > r->l_pac) return -2;
	rid_b = bsx_pos2rid(r, bsx_depos(r->l_pac, rb, &is_rev));
	rid_e = rb < re ? bsx_pos2rid(r, bsx_depos(r->l_pac, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

int bsx_fetch_span(const bsx_refmeta_t *r, int64_t *beg, int64_t mid, int64_t *end) /* bns_fetch_seq, bntseq.c:428-452 */
{
	int64_t far_beg, far_end;
	int is_rev, rid;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	rid = bsx_pos2rid(r, bsx_depos(r->l_pac, mid, &is_rev));
	far_beg = r->anns[rid].offset;
	far_end = far_beg + r->anns[rid].len;
	if (is_rev) {
		int64_t tmp = far_beg;
		far_beg = (r->l_pac << 1) - far_end;
		far_end = (r->l_pac << 1) - tmp;
	}
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	return rid;
}

/* ------------------------------------------------------------------------------------------
 * loader
 * ------------------------------------------------------------------------------------------ */
static int read_all(FILE *fp, void *dst, uint64_t n)
{
	char *p = (char*)dst;
	while (n) {
		size_t k = n > (1u << 30) ? (1u << 30) : (size_t)n;
		size_t got = fread(p, 1, k, fp);
		if (got == 0) return -1;
		p += got; n -= got;
	}
	return 0;
}

static int load_fmi(const char *base, const char *tag, bsx_fmi_t *f)
{
	char fn[4096];
	FILE *fp;
	long sz;
	uint64_t hdr[5], primary, skipped[4], sa_intv, seq_len;

	memset(f, 0, sizeof(*f));
	snprintf(fn, sizeof(fn), "%s.%s.bwt", base, tag);
	if ((fp = fopen(fn, "rb")) == 0) { fprintf(stderr, "[bsx] cannot open %s\n", fn); return BSX_E_IO; }
	fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
	if (sz < 40 || fread(hdr, 8, 5, fp) != 5) { fclose(fp); return BSX_E_FORMAT; }
	f->primary = hdr[0];
	f->L2[0] = 0; f->L2[1] = hdr[1]; f->L2[2] = hdr[2]; f->L2[3] = hdr[3]; f->L2[4] = hdr[4];
	f->seq_len = f->L2[4];
	f->bwt_size = (uint64_t)(sz - 40) >> 2;
	f->bwt = (uint32_t*)calloc(f->bwt_size + 16, 4);
	if (!f->bwt || read_all(fp, f->bwt, f->bwt_size << 2) < 0) { fclose(fp); return BSX_E_FORMAT; }
	fclose(fp);

	snprintf(fn, sizeof(fn), "%s.%s.sa", base, tag);
	if ((fp = fopen(fn, "rb")) == 0) { fprintf(stderr, "[bsx] cannot open %s\n", fn); return BSX_E_IO; }
	if (fread(&primary, 8, 1, fp) != 1 || fread(skipped, 8, 4, fp) != 4 ||
	    fread(&sa_intv, 8, 1, fp) != 1 || fread(&seq_len, 8, 1, fp) != 1) { fclose(fp); return BSX_E_FORMAT; }
	if (primary != f->primary || seq_len != f->seq_len) {
		fprintf(stderr, "[bsx] SA-BWT inconsistency in %s\n", fn); fclose(fp); return BSX_E_FORMAT;
	}
	f->sa_intv = (int)sa_intv;
	f->n_sa = (f->seq_len + f->sa_intv) / f->sa_intv;
	f->sa = (uint64_t*)calloc(f->n_sa, 8);
	f->sa[0] = (uint64_t)-1;
	if (read_all(fp, f->sa + 1, (f->n_sa - 1) * 8) < 0) { fclose(fp); return BSX_E_FORMAT; }
	fclose(fp);
	return BSX_OK;
}

static int load_refmeta(const char *base, bsx_refmeta_t *r)
{
	char fn[4096], str[8192];
	FILE *fp;
	long long xx;
	int i, c;

	memset(r, 0, sizeof(*r));
	snprintf(fn, sizeof(fn), "%s.bis.ann", base);
	if ((fp = fopen(fn, "r")) == 0) { fprintf(stderr, "[bsx] cannot open %s\n", fn); return BSX_E_IO; }
	if (fscanf(fp, "%lld%d%u", &xx, &r->n_seqs, &r->seed) != 3) { fclose(fp); return BSX_E_FORMAT; }
	r->l_pac = xx;
	r->anns = (bsx_ann_t*)calloc(r->n_seqs ? r->n_seqs : 1, sizeof(bsx_ann_t));
	for (i = 0; i < r->n_seqs; ++i) {
		bsx_ann_t *p = r->anns + i;
		char *q = str;
		if (fscanf(fp, "%u%8191s", &p->gi, str) != 2) { fclose(fp); return BSX_E_FORMAT; }
		p->name = strdup(str);
		while ((size_t)(q - str) < sizeof(str) - 1 && (c = fgetc(fp)) != '\n' && c != EOF) *q++ = c;
		while (c != '\n' && c != EOF) c = fgetc(fp);
		if (c == EOF) { fclose(fp); return BSX_E_FORMAT; }
		*q = 0;
		if (q - str > 1 && strcmp(str, " (null)") != 0) p->anno = strdup(str + 1);
		else p->anno = strdup("");
		if (fscanf(fp, "%lld%d%d", &xx, &p->len, &p->n_ambs) != 3) { fclose(fp); return BSX_E_FORMAT; }
		p->offset = xx;
	}
	fclose(fp);

	snprintf(fn, sizeof(fn), "%s.bis.amb", base);
	if ((fp = fopen(fn, "r")) == 0) { fprintf(stderr, "[bsx] cannot open %s\n", fn); return BSX_E_IO; }
	{
		int n_seqs;
		if (fscanf(fp, "%lld%d%d", &xx, &n_seqs, &r->n_holes) != 3) { fclose(fp); return BSX_E_FORMAT; }
		if (xx != r->l_pac || n_seqs != r->n_seqs) { fprintf(stderr, "[bsx] inconsistent .ann and .amb files\n"); fclose(fp); return BSX_E_FORMAT; }
		r->ambs = r->n_holes ? (bsx_amb_t*)calloc(r->n_holes, sizeof(bsx_amb_t)) : 0;
		for (i = 0; i < r->n_holes; ++i) {
			bsx_amb_t *p = r->ambs + i;
			if (fscanf(fp, "%lld%d%8191s", &xx, &p->len, str) != 3) { fclose(fp); return BSX_E_FORMAT; }
			p->offset = xx; p->amb = str[0];
		}
	}
	fclose(fp);

	/* optional <base>.alt: first column = ALT contig names, '@' lines skipped (bntseq.c:183-214) */
	snprintf(fn, sizeof(fn), "%s.alt", base);
	if ((fp = fopen(fn, "r")) != 0) {
		char line[8192];
		while (fgets(line, sizeof(line), fp)) {
			size_t l = strcspn(line, "\t\r\n");
			line[l] = 0;
			if (line[0] == '@' || l == 0) continue;
			for (i = 0; i < r->n_seqs; ++i)
				if (strcmp(r->anns[i].name, line) == 0) { r->anns[i].is_alt = 1; break; }
		}
		fclose(fp);
	}
	return BSX_OK;
}

BSX_API int bsx_index_load(const char *base, bsx_index_t **out)
{
	bsx_index_t *idx;
	char fn[4096];
	FILE *fp;
	int rc;
	*out = 0;
	idx = (bsx_index_t*)calloc(1, sizeof(*idx));
	if ((rc = load_fmi(base, "par", &idx->fmi[1])) != BSX_OK) { bsx_index_free(idx); return rc; }
	if ((rc = load_fmi(base, "dau", &idx->fmi[0])) != BSX_OK) { bsx_index_free(idx); return rc; }
	if ((rc = load_refmeta(base, &idx->ref)) != BSX_OK) { bsx_index_free(idx); return rc; }
	snprintf(fn, sizeof(fn), "%s.bis.pac", base);
	if ((fp = fopen(fn, "rb")) == 0) { bsx_index_free(idx); return BSX_E_IO; }
	idx->pac = (uint8_t*)calloc(idx->ref.l_pac / 4 + 1 + 16, 1);
	if (read_all(fp, idx->pac, idx->ref.l_pac / 4 + 1) < 0) { fclose(fp); bsx_index_free(idx); return BSX_E_FORMAT; }
	fclose(fp);
	if (idx->fmi[1].seq_len != (uint64_t)idx->ref.l_pac * 2 || idx->fmi[0].seq_len != (uint64_t)idx->ref.l_pac * 2) {
		fprintf(stderr, "[bsx] index length mismatch: l_pac=%lld, par=%llu, dau=%llu\n", (long long)idx->ref.l_pac,
			(unsigned long long)idx->fmi[1].seq_len, (unsigned long long)idx->fmi[0].seq_len);
		bsx_index_free(idx); return BSX_E_FORMAT;
	}
	*out = idx;
	return BSX_OK;
}

BSX_API void bsx_index_free(bsx_index_t *idx)
{
	int i;
	if (!idx) return;
	for (i = 0; i < 2; ++i) { free(idx->fmi[i].bwt); free(idx->fmi[i].sa); }
	for (i = 0; i < idx->ref.n_seqs; ++i) { free(idx->ref.anns[i].name); free(idx->ref.anns[i].anno); }
	free(idx->ref.anns); free(idx->ref.ambs); free(idx->pac);
	free(idx);
}

BSX_API int64_t bsx_index_l_pac(const bsx_index_t *idx) { return idx->ref.l_pac; }
BSX_API int bsx_index_n_seqs(const bsx_index_t *idx) { return idx->ref.n_seqs; }
BSX_API const uint8_t *bsx_index_pac(const bsx_index_t *idx) { return idx->pac; }
BSX_API int bsx_index_contig(const bsx_index_t *idx, int i, const char **name, int64_t *offset, int64_t *len)
{
	if (!idx || i < 0 || i >= idx->ref.n_seqs) return BSX_E_ARG;
	*name = idx->ref.anns[i].name; *offset = idx->ref.anns[i].offset; *len = idx->ref.anns[i].len;
	return BSX_OK;
}

/* ------------------------------------------------------------------------------------------
 * SA-IS (induced sorting).  s has n symbols, s[n-1] is a unique smallest sentinel.
 * cs = bytes per symbol (1 at the top level, 4 in recursive levels).
 * ------------------------------------------------------------------------------------------ */
#define SCH(i) (cs == 4 ? ((const int32_t*)s)[i] : (int32_t)((const uint8_t*)s)[i])
#define TGET(i) ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(i, b) (t[(i) >> 3] = (uint8_t)((b) ? (t[(i) >> 3] | (1u << ((i) & 7))) : (t[(i) >> 3] & ~(1u << ((i) & 7)))))
#define IS_LMS(i) ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void sa_buckets(const void *s, int32_t *bkt, int32_t n, int32_t K, int cs, int end)
{
	int32_t i, sum = 0;
	for (i = 0; i <= K; ++i) bkt[i] = 0;
	for (i = 0; i < n; ++i) ++bkt[SCH(i)];
	for (i = 0; i <= K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}

static void sa_induce(const uint8_t *t, int32_t *SA, const void *s, int32_t *bkt, int32_t n, int32_t K, int cs)
{
	int32_t i, j;
	sa_buckets(s, bkt, n, K, cs, 0);
	for (i = 0; i < n; ++i) { j = SA[i] - 1; if (j >= 0 && !TGET(j)) SA[bkt[SCH(j)]++] = j; }
	sa_buckets(s, bkt, n, K, cs, 1);
	for (i = n - 1; i >= 0; --i) { j = SA[i] - 1; if (j >= 0 && TGET(j)) SA[--bkt[SCH(j)]] = j; }
}

static void sa_is(const void *s, int32_t *SA, int32_t n, int32_t K, int cs)
{
	uint8_t *t = (uint8_t*)calloc(n / 8 + 1, 1);
	int32_t *bkt, *SA1, *s1;
	int32_t i, j, n1, name, prev;

	TSET(n - 2, 0); TSET(n - 1, 1);
	for (i = n - 3; i >= 0; --i)
		TSET(i, (SCH(i) < SCH(i + 1) || (SCH(i) == SCH(i + 1) && TGET(i + 1))) ? 1 : 0);
	bkt = (int32_t*)malloc(sizeof(int32_t) * ((size_t)K + 1));
	sa_buckets(s, bkt, n, K, cs, 1);
	for (i = 0; i < n; ++i) SA[i] = -1;
	for (i = 1; i < n; ++i) if (IS_LMS(i)) SA[--bkt[SCH(i)]] = i;
	sa_induce(t, SA, s, bkt, n, K, cs);
	free(bkt);
	/* compact sorted LMS substrings and name them */
	for (i = 0, n1 = 0; i < n; ++i) if (IS_LMS(SA[i])) SA[n1++] = SA[i];
	for (i = n1; i < n; ++i) SA[i] = -1;
	name = 0; prev = -1;
	for (i = 0; i < n1; ++i) {
		int32_t pos = SA[i], d, diff = 0;
		for (d = 0; d < n; ++d) {
			if (prev == -1 || SCH(pos + d) != SCH(prev + d) || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
			else if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) break;
		}
		if (diff) { ++name; prev = pos; }
		SA[n1 + (pos >> 1)] = name - 1;
	}
	for (i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
	SA1 = SA; s1 = SA + n - n1;
	if (name < n1) sa_is(s1, SA1, n1, name - 1, 4);
	else for (i = 0; i < n1; ++i) SA1[s1[i]] = i;
	/* induce the final order from the sorted LMS suffixes */
	bkt = (int32_t*)malloc(sizeof(int32_t) * ((size_t)K + 1));
	sa_buckets(s, bkt, n, K, cs, 1);
	for (i = 1, j = 0; i < n; ++i) if (IS_LMS(i)) s1[j++] = i;
	for (i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
	for (i = n1; i < n; ++i) SA[i] = -1;
	for (i = n1 - 1; i >= 0; --i) { j = SA[i]; SA[i] = -1; SA[--bkt[SCH(j)]] = j; }
	sa_induce(t, SA, s, bkt, n, K, cs);
	free(bkt); free(t);
}

/* ------------------------------------------------------------------------------------------
 * builder
 * ------------------------------------------------------------------------------------------ */
static const uint8_t nt4_of_char[256] = {
#define R16(v) v,v,v,v,v,v,v,v,v,v,v,v,v,v,v,v
	R16(4), R16(4),
	4,4,4,4,4,4,4,4,4,4,4,4,4,5,4,4, R16(4),
	4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
	4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
	R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4)
#undef R16
};
const uint8_t *bsx_nt4_table(void) { return nt4_of_char; }

/* ------------------------------------------------------------------------------------------
 * genome sink: contigs and bases streamed in, <base>.bis.{pac,ann,amb} content out (bis_bns_fasta2bntseq,
 * lib/aln/bntseq.c:542-633).  N -> random base with the reference's generator and seed (bntseq.c:558-559,495).
 * Both the FASTA reader and the synthetic-genome generator (sim.c) feed it, so a genome never has to exist as
 * text or as one byte per base.
 * ------------------------------------------------------------------------------------------ */
struct bsx_gsink {
	bsx_index_t *idx;
	int64_t cap;             /* bases the pac array has room for */
	BSX_VEC(bsx_amb_t) holes;
	int n_anns, m_anns;
	int lasts;               /* last character of the current contig (runs of one ambiguity code form one hole) */
	unsigned short xs[3];    /* the 48-bit state srand48(11) sets (bntseq.c:558-559), kept per sink: nrand48 over it is lrand48's sequence */
};

bsx_gsink_t *bsx_gsink_new(void)
{
	bsx_gsink_t *g = (bsx_gsink_t*)calloc(1, sizeof(*g));
	g->idx = (bsx_index_t*)calloc(1, sizeof(bsx_index_t));
	g->idx->ref.seed = 11;
	bsx_vec_init(g->holes);
	g->xs[0] = 0x330E; g->xs[1] = 11; g->xs[2] = 0;
	return g;
}

void bsx_gsink_contig(bsx_gsink_t *g, const char *name, const char *comment)
{
	bsx_refmeta_t *r = &g->idx->ref;
	bsx_ann_t *p;
	if (g->n_anns == g->m_anns) { g->m_anns = g->m_anns ? g->m_anns << 1 : 16; r->anns = (bsx_ann_t*)realloc(r->anns, sizeof(bsx_ann_t) * (size_t)g->m_anns); }
	p = &r->anns[g->n_anns++];
	memset(p, 0, sizeof(*p));
	p->name = strdup(name); p->anno = strdup(comment ? comment : "");
	p->offset = r->l_pac;
	r->n_seqs = g->n_anns;
	g->lasts = 0;
}

int bsx_gsink_bases(bsx_gsink_t *g, const char *chars, int64_t n)
{
	bsx_refmeta_t *r = &g->idx->ref;
	bsx_ann_t *p;
	int64_t k, l = r->l_pac;
	uint8_t *pac;
	if (g->n_anns == 0) return BSX_E_FORMAT;
	p = &r->anns[g->n_anns - 1];
	if ((int64_t)p->len + n > 0x7fffffffLL) { fprintf(stderr, "[bsx] contig %s is longer than 2^31 bases\n", p->name); return BSX_E_FORMAT; }
	if (l + n + 64 > g->cap) {
		int64_t nc = g->cap ? g->cap : (1 << 20);
		while (nc < l + n + 64) nc <<= 1;
		g->idx->pac = (uint8_t*)realloc(g->idx->pac, (size_t)(nc / 4 + 32));
		if (!g->idx->pac) return BSX_E_NOMEM;
		memset(g->idx->pac + g->cap / 4, 0, (size_t)(nc / 4 + 32 - g->cap / 4));
		g->cap = nc;
	}
	pac = g->idx->pac;
	for (k = 0; k < n; ++k, ++l) {
		int ch = (unsigned char)chars[k], c = nt4_of_char[ch];
		if (c >= 4) {
			if (g->lasts == ch) ++g->holes.a[g->holes.n - 1].len;
			else {
				bsx_amb_t h; h.offset = l; h.len = 1; h.amb = (char)ch;
				bsx_vec_push(g->holes, h);
				++p->n_ambs;
			}
			c = (int)(nrand48(g->xs) & 3);
		}
		g->lasts = ch;
		pac[l >> 2] |= (uint8_t)(c << ((~l & 3) << 1));
	}
	p->len += (int32_t)n;
	r->l_pac = l;
	return BSX_OK;
}

bsx_index_t *bsx_gsink_finish(bsx_gsink_t *g)
{
	bsx_index_t *idx = g->idx;
	idx->ref.n_holes = (int32_t)g->holes.n;
	idx->ref.ambs = g->holes.a;
	if (!idx->pac) idx->pac = (uint8_t*)calloc(32, 1);
	free(g);
	return idx;
}

/* minimal FASTA reader (plain or gzip), streamed into the sink */
BSX_API int bsx_index_from_fasta(const char *fn, bsx_index_t **out)
{
	gzFile fp = gzopen(fn, "r");
	char *line;
	size_t cap = 1 << 16;
	bsx_gsink_t *g;
	int rc = BSX_OK, n = 0;
	*out = 0;
	if (!fp) return BSX_E_IO;
	gzbuffer(fp, 1 << 20);
	line = (char*)malloc(cap);
	g = bsx_gsink_new();
	while (rc == BSX_OK && gzgets(fp, line, (int)cap)) {
		size_t l = strlen(line), i, m;
		while (l == cap - 1 && line[l - 1] != '\n') { /* long line */
			cap <<= 1; line = (char*)realloc(line, cap);
			if (!gzgets(fp, line + l, (int)(cap - l))) break;
			l += strlen(line + l);
		}
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (line[0] == '>') {
			char *p = line + 1, *q, *comment = 0;
			for (q = p; *q && !isspace((unsigned char)*q); ++q);
			if (*q) { *q++ = 0; while (*q && isspace((unsigned char)*q)) ++q; if (*q) comment = q; }
			bsx_gsink_contig(g, p, comment);
			++n;
		} else if (n > 0) {
			for (i = 0, m = 0; i < l; ++i) if (!isspace((unsigned char)line[i])) line[m++] = line[i];
			rc = bsx_gsink_bases(g, line, (int64_t)m);
		}
	}
	free(line);
	gzclose(fp);
	*out = bsx_gsink_finish(g);
	if (rc == BSX_OK && (n == 0 || (*out)->ref.l_pac <= 0)) rc = BSX_E_FORMAT;
	if (rc != BSX_OK) { bsx_index_free(*out); *out = 0; }
	return rc;
}

/* FM index of one converted text on the host: suffix-sort text+sentinel (SA-IS, 32-bit), derive BWT, occ blocks, SA samples */
static int host_build_fmi(const uint8_t *text, int64_t n, bsx_fmi_t *f)
{
	uint8_t *s = (uint8_t*)malloc((size_t)n + 1);
	int32_t *SA = (int32_t*)malloc(sizeof(int32_t) * ((size_t)n + 1));
	uint64_t L2[5] = {0, 0, 0, 0, 0}, primary = 0, c[4] = {0, 0, 0, 0};
	uint64_t n_occ, bwt_words, k, *sa_s, n_sa, sa_intv = 32, seq_len = (uint64_t)n;
	uint32_t *out;
	uint8_t *bw;
	int64_t i;

	if (!s || !SA) { free(s); free(SA); return BSX_E_NOMEM; }
	for (i = 0; i < n; ++i) { s[i] = text[i] + 1; ++L2[1 + text[i]]; }
	s[n] = 0;
	for (i = 2; i <= 4; ++i) L2[i] += L2[i - 1];
	sa_is(s, SA, (int32_t)(n + 1), 4, 1);
	free(s);
	/* rank r (0..n) <-> suffix SA[r]; row 0 is the sentinel suffix */
	bw = (uint8_t*)malloc((size_t)n);
	n_sa = (seq_len + sa_intv) / sa_intv;
	sa_s = (uint64_t*)calloc(n_sa, 8);
	{
		uint64_t r, w = 0;
		for (r = 0; r <= (uint64_t)n; ++r) {
			int32_t p = SA[r];
			if ((r & (sa_intv - 1)) == 0) sa_s[r / sa_intv] = (uint64_t)p;
			if (p == 0) { primary = r; continue; }
			bw[w++] = text[p - 1];
		}
	}
	free(SA);
	/* interleave: every 128 symbols 4 x u64 counts then 8 words; one trailing count block */
	n_occ = (seq_len + 127) / 128 + 1;
	bwt_words = ((seq_len + 15) >> 4) + n_occ * 8;
	out = (uint32_t*)calloc(bwt_words + 16, 4);
	for (i = 0, k = 0; i < n; ++i) {
		if ((i & 127) == 0) { memcpy(out + k, c, 32); k += 8; }
		if ((i & 15) == 0) ++k;
		out[k - 1] |= (uint32_t)bw[i] << ((15 - (i & 15)) << 1);
		++c[bw[i]];
	}
	memcpy(out + k, c, 32); k += 8;
	free(bw);
	if (k != bwt_words) { free(out); free(sa_s); return BSX_E_INTERNAL; }
	memset(f, 0, sizeof(*f));
	f->primary = primary; memcpy(f->L2, L2, sizeof(L2)); f->seq_len = seq_len;
	f->bwt_size = bwt_words; f->bwt = out;
	f->sa_intv = (int)sa_intv; f->n_sa = n_sa; f->sa = sa_s;
	f->sa[0] = (uint64_t)-1;   /* as the loader leaves it (lib/aln/bwt.c:448-452) */
	return BSX_OK;
}

static int save_fmi(const char *base, const char *tag, const bsx_fmi_t *f)
{
	char fn[4096];
	FILE *fp;
	uint64_t sa_intv = (uint64_t)f->sa_intv;
	int ok;
	if (!f->bwt || !f->sa) return BSX_E_ARG;
	snprintf(fn, sizeof(fn), "%s.%s.bwt", base, tag);
	if ((fp = fopen(fn, "wb")) == 0) return BSX_E_IO;
	ok = fwrite(&f->primary, 8, 1, fp) == 1 && fwrite(f->L2 + 1, 8, 4, fp) == 4 && fwrite(f->bwt, 4, f->bwt_size, fp) == f->bwt_size;
	if (fclose(fp) != 0 || !ok) return BSX_E_IO;
	snprintf(fn, sizeof(fn), "%s.%s.sa", base, tag);
	if ((fp = fopen(fn, "wb")) == 0) return BSX_E_IO;
	ok = fwrite(&f->primary, 8, 1, fp) == 1 && fwrite(f->L2 + 1, 8, 4, fp) == 4 && fwrite(&sa_intv, 8, 1, fp) == 1 &&
	     fwrite(&f->seq_len, 8, 1, fp) == 1 && fwrite(f->sa + 1, 8, f->n_sa - 1, fp) == f->n_sa - 1;
	if (fclose(fp) != 0 || !ok) return BSX_E_IO;
	return BSX_OK;
}

struct build_par { bsx_index_t *idx; int rc[2]; };
static void build_strand(void *data, long i, int tid)   /* i: 1 = parent (C>T), 0 = daughter (G>A) */
{
	struct build_par *P = (struct build_par*)data;
	const uint8_t *pac = P->idx->pac;
	int64_t k, l_pac = P->idx->ref.l_pac;
	uint8_t *text = (uint8_t*)malloc((size_t)l_pac * 2);
	(void)tid;
	if (!text) { P->rc[i] = BSX_E_NOMEM; return; }
	for (k = 0; k < l_pac; ++k) {
		uint8_t c = (uint8_t)bsx_pac_get(pac, k), r = (uint8_t)(3 - bsx_pac_get(pac, l_pac - 1 - k));
		if (i) { if (c == 1) c = 3; if (r == 1) r = 3; }
		else   { if (c == 2) c = 0; if (r == 2) r = 0; }
		text[k] = c; text[l_pac + k] = r;
	}
	P->rc[i] = host_build_fmi(text, l_pac * 2, &P->idx->fmi[i]);
	free(text);
}

/* both FM indices of idx (pac + annotation present) on the host; texts of at most 2^31 - 2 symbols.  Larger genomes
 * are indexed on the device (bsx_device_build_index). */
BSX_API int bsx_index_build_host(bsx_index_t *idx)
{
	struct build_par bp;
	int i;
	if (!idx || !idx->pac) return BSX_E_ARG;
	if (idx->ref.l_pac <= 0 || idx->ref.l_pac * 2 + 1 >= 0x7fffffffLL) { fprintf(stderr, "[bsx] genome too large for the host's 32-bit suffix sorter: build the index on the device\n"); return BSX_E_ARG; }
	for (i = 0; i < 2; ++i) { free(idx->fmi[i].bwt); free(idx->fmi[i].sa); memset(&idx->fmi[i], 0, sizeof(bsx_fmi_t)); }
	bp.idx = idx; bp.rc[0] = bp.rc[1] = BSX_OK;
	bsx_parallel_for(2, build_strand, &bp, 2);   /* the two strands are independent: side by side */
	return bp.rc[1] != BSX_OK ? bp.rc[1] : bp.rc[0];
}

/* the seven files of `biscuit index` (lib/aln/bwtindex.c:206-347): <base>.bis.{pac,ann,amb}, <base>.{par,dau}.{bwt,sa} */
BSX_API int bsx_index_save(const bsx_index_t *idx, const char *base)
{
	char fn[4096];
	FILE *fp;
	int64_t l_pac, k;
	int i, rc;
	if (!idx || !idx->pac) return BSX_E_ARG;
	l_pac = idx->ref.l_pac;
	snprintf(fn, sizeof(fn), "%s.bis.pac", base);
	if ((fp = fopen(fn, "wb")) == 0) return BSX_E_IO;
	fwrite(idx->pac, 1, (size_t)((l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1)), fp);
	{ uint8_t ct = 0; if (l_pac % 4 == 0) fwrite(&ct, 1, 1, fp); ct = (uint8_t)(l_pac % 4); fwrite(&ct, 1, 1, fp); }
	if (fclose(fp) != 0) return BSX_E_IO;
	snprintf(fn, sizeof(fn), "%s.bis.ann", base);
	if ((fp = fopen(fn, "w")) == 0) return BSX_E_IO;
	fprintf(fp, "%lld %d %u\n", (long long)l_pac, idx->ref.n_seqs, idx->ref.seed);
	for (i = 0; i < idx->ref.n_seqs; ++i) {
		const bsx_ann_t *p = &idx->ref.anns[i];
		fprintf(fp, "%d %s", (int)p->gi, p->name);
		fprintf(fp, " %s\n", p->anno && p->anno[0] ? p->anno : "(null)");
		fprintf(fp, "%lld %d %d\n", (long long)p->offset, p->len, p->n_ambs);
	}
	if (fclose(fp) != 0) return BSX_E_IO;
	snprintf(fn, sizeof(fn), "%s.bis.amb", base);
	if ((fp = fopen(fn, "w")) == 0) return BSX_E_IO;
	fprintf(fp, "%lld %d %u\n", (long long)l_pac, idx->ref.n_seqs, (unsigned)idx->ref.n_holes);
	for (k = 0; k < idx->ref.n_holes; ++k) fprintf(fp, "%lld %d %c\n", (long long)idx->ref.ambs[k].offset, idx->ref.ambs[k].len, idx->ref.ambs[k].amb);
	if (fclose(fp) != 0) return BSX_E_IO;
	if ((rc = save_fmi(base, "par", &idx->fmi[1])) != BSX_OK) return rc;
	return save_fmi(base, "dau", &idx->fmi[0]);
}

BSX_API int bsx_index_build(const char *fasta, const char *base)
{
	bsx_index_t *idx = 0;
	int rc = bsx_index_from_fasta(fasta, &idx);
	if (rc != BSX_OK) return rc;
	if ((rc = bsx_index_build_host(idx)) == BSX_OK) rc = bsx_index_save(idx, base);
	bsx_index_free(idx);
	return rc;
}
