/* sim.c -- seeded synthetic genome / bisulfite read generator for bench.py and the large-size tests.
 * (hg38 and real WGBS reads are not available offline; everything generated here is labelled
 * synthetic wherever it is reported.)  xorshift64* PRNG, so every box regenerates identical data.
 *   genome : i.i.d. bases + planted repeat families (300-3000 bp, 2-6 copies, 0-5 % divergence,
 *            some reverse-complemented) + short tandem repeats + one N run per contig; profile 1 adds
 *            interspersed repeat families with up to a million copies (gen_hard)
 *   pairs  : directional protocol -- R1 = bisulfite-converted strand, R2 = reverse complement of
 *            the fragment; C->T with retention 0.70 at CpG and 0.01 elsewhere; substitutions.
 */
#include "bsx_core.h"

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r) { uint64_t x = r->s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; r->s = x; return x * 0x2545F4914F6CDD1DULL; }
static inline uint64_t rng_below(rng_t *r, uint64_t n) { return rng_next(r) % n; }
static inline double rng_unit(rng_t *r) { return (rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* profile 1 ("hg38-like"): what makes a mammalian genome hard for a seed-and-extend aligner and what the clean profile lacks -- interspersed
 * repeat families with 10^5..10^6 copies.  Fractions follow hg38's RepeatMasker table in round numbers:
 *   SINE-like   one 300 bp consensus,  10 % of the genome (1.0 M copies at 3.1 Gbp), each copy 5-15 % diverged from the consensus
 *   LINE-like   one 6 kb consensus,    17 %, copies truncated at the 5' end (a suffix of 300..6000 bp, short ones most common), 5-20 % diverged
 *   satellite   a 171 bp unit in arrays of 50 kb..2 Mb, 3 %, units 1-3 % diverged from each other
 *   LTR-like    one 7 kb element with 400 bp terminal repeats, 8 %, 5-18 % diverged
 * on top of the clean profile's segmental duplications (repeat_frac) and short tandem repeats (ten times as many here).  ~43 % of the
 * genome in all; half of the copies reverse-complemented.  A 19-mer of a young copy then has thousands of occurrences: SA intervals
 * beyond max_occ = 500 (memchain.c:325-326), strand searches with thousands of seeds, reads that align equally well in many places. */
static void plant_family(uint8_t *g, int64_t n, rng_t *R, const uint8_t *cons, int cl, double frac, double div_lo, double div_hi, int truncate5)
{
	int64_t planted = 0, want = (int64_t)((double)n * frac);
	while (planted < want) {
		int l = cl, off = 0, rc = (int)(rng_next(R) & 1), j;
		double div = div_lo + rng_unit(R) * (div_hi - div_lo);
		int64_t d;
		if (truncate5) { // most copies are short 3' ends: length ~ 300 + a squared uniform of the rest
			const double u = rng_unit(R);
			l = 300 + (int)((cl - 300) * u * u);
			off = cl - l;
		}
		if ((int64_t)l * 4 > n) break;
		d = (int64_t)rng_below(R, (uint64_t)(n - l));
		for (j = 0; j < l; ++j) {
			uint8_t b = rc ? (uint8_t)(3 - cons[off + l - 1 - j]) : cons[off + j];
			if (rng_unit(R) < div) b = (uint8_t)((b + 1 + rng_below(R, 3)) & 3);
			g[d + j] = b;
		}
		planted += l;
	}
}
static void gen_hard(uint8_t *g, int64_t n, rng_t *R)
{
	uint8_t *cons = (uint8_t*)malloc(8192);
	int64_t planted, want;
	int j;
	for (j = 0; j < 8192; ++j) cons[j] = (uint8_t)(rng_next(R) & 3);
	plant_family(g, n, R, cons, 6000, 0.17, 0.05, 0.20, 1);          /* LINE-like first: the younger families land on top of it */
	for (j = 0; j < 400; ++j) cons[6600 + j] = cons[j];               /* LTR-like: 7 kb with the same 400 bp at both ends */
	plant_family(g, n, R, cons, 7000, 0.08, 0.05, 0.18, 0);
	for (j = 0; j < 300; ++j) cons[j] = (uint8_t)(rng_next(R) & 3);
	for (j = 280; j < 300; ++j) cons[j] = 0;                          /* a poly-A tail */
	plant_family(g, n, R, cons, 300, 0.10, 0.05, 0.15, 0);            /* SINE-like */
	for (j = 0; j < 171; ++j) cons[j] = (uint8_t)(rng_next(R) & 3);
	for (planted = 0, want = (int64_t)((double)n * 0.03); planted < want; ) { /* satellite arrays */
		int64_t al = 50000 + (int64_t)rng_below(R, 1950000), d, k;
		const double div = 0.01 + rng_unit(R) * 0.02;
		if (al > want - planted) al = want - planted;
		if (al * 4 > n) al = n / 8;
		if (al < 171) break;
		d = (int64_t)rng_below(R, (uint64_t)(n - al));
		for (k = 0; k < al; ++k) {
			uint8_t b = cons[k % 171];
			if (rng_unit(R) < div) b = (uint8_t)((b + 1 + rng_below(R, 3)) & 3);
			g[d + k] = b;
		}
		planted += al;
	}
	free(cons);
}

/* the genome as one byte per base (0..3); N runs are decided by the writers below */
static uint8_t *gen_genome(int64_t n, uint64_t seed, double repeat_frac, int profile)
{
	rng_t R; uint8_t *g;
	int64_t i, planted = 0;
	R.s = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
	g = (uint8_t*)malloc((size_t)n);
	if (!g) return 0;
	for (i = 0; i < n; i += 32) { uint64_t x = rng_next(&R); int k; for (k = 0; k < 32 && i + k < n; ++k) g[i + k] = (uint8_t)((x >> (2 * k)) & 3); }
	if (profile == 1) gen_hard(g, n, &R);
	while (planted < (int64_t)(n * repeat_frac)) {
		int64_t l = 300 + (int64_t)rng_below(&R, 2700), s, d;
		int copies = 1 + (int)rng_below(&R, 5), k;
		if (l * 4 > n) l = n / 4;
		s = (int64_t)rng_below(&R, (uint64_t)(n - l));
		for (k = 0; k < copies; ++k) {
			double div = rng_unit(&R) * 0.05;
			int rc = rng_unit(&R) < 0.3;
			int64_t j;
			d = (int64_t)rng_below(&R, (uint64_t)(n - l));
			if ((d > s ? d - s : s - d) < l) continue;
			for (j = 0; j < l; ++j) {
				uint8_t b = rc ? (uint8_t)(3 - g[s + l - 1 - j]) : g[s + j];
				if (rng_unit(&R) < div) b = (uint8_t)((b + 1 + rng_below(&R, 3)) & 3);
				g[d + j] = b;
			}
			planted += l;
		}
	}
	for (i = 0; i < (profile == 1 ? n / 20000 : n / 200000) + 1; ++i) { /* tandem repeats */
		int ul = 2 + (int)rng_below(&R, 28), reps = 5 + (int)rng_below(&R, 35), k;
		int64_t d = (int64_t)rng_below(&R, (uint64_t)(n - (int64_t)ul * reps - 1));
		for (k = ul; k < ul * reps; ++k) g[d + k] = g[d + k % ul];
	}
	return g;
}
/* contig c of n_contigs: [*b, *e), with one N run [*nb, *nb + *nl) */
static void contig_span(int64_t n, int n_contigs, int c, int64_t *b, int64_t *e, int64_t *nb, int64_t *nl)
{
	*b = n / n_contigs * c; *e = c == n_contigs - 1 ? n : n / n_contigs * (c + 1);
	*nb = *b + (*e - *b) / 3; *nl = (*e - *b) / 200 < 1000 ? (*e - *b) / 200 : 1000;
}

BSX_API int bsx_sim_genome2(const char *fasta, int64_t n, uint64_t seed, int n_contigs, double repeat_frac, int profile);
BSX_API int bsx_sim_genome(const char *fasta, int64_t n, uint64_t seed, int n_contigs, double repeat_frac) { return bsx_sim_genome2(fasta, n, seed, n_contigs, repeat_frac, 0); }
/* profile 0: the clean genome of rounds 1-2; 1: with high-copy interspersed repeat families (gen_hard) */
BSX_API int bsx_sim_genome2(const char *fasta, int64_t n, uint64_t seed, int n_contigs, double repeat_frac, int profile)
{
	uint8_t *g;
	int64_t i;
	int c;
	FILE *fp;
	if (n < 1000 || n_contigs < 1 || (profile == 1 && n < 100000)) return BSX_E_ARG;
	if ((g = gen_genome(n, seed, repeat_frac, profile)) == 0) return BSX_E_NOMEM;
	if ((fp = fopen(fasta, "wb")) == 0) { free(g); return BSX_E_IO; }
	for (c = 0; c < n_contigs; ++c) {
		int64_t b, e, nb, nl;
		char line[64];
		contig_span(n, n_contigs, c, &b, &e, &nb, &nl);
		fprintf(fp, ">chr%d\n", c + 1);
		for (i = b; i < e; i += 60) {
			int k, m = (int)(e - i < 60 ? e - i : 60);
			for (k = 0; k < m; ++k) line[k] = (i + k >= nb && i + k < nb + nl) ? 'N' : "ACGT"[g[i + k]];
			line[m] = '\n';
			fwrite(line, 1, (size_t)m + 1, fp);
		}
	}
	fclose(fp);
	free(g);
	return BSX_OK;
}

/* the same genome handed straight to the index builder's sink: an index with pac + annotation and no FM indices yet
 * (bsx_index_build_host / bsx_device_build_index make those).  What bsx_sim_genome + bsx_index_from_fasta give, without
 * the FASTA text: a 3.1 Gbp genome is 0.78 GB this way. */
BSX_API int bsx_sim_genome_index2(int64_t n, uint64_t seed, int n_contigs, double repeat_frac, int profile, bsx_index_t **out);
BSX_API int bsx_sim_genome_index(int64_t n, uint64_t seed, int n_contigs, double repeat_frac, bsx_index_t **out) { return bsx_sim_genome_index2(n, seed, n_contigs, repeat_frac, 0, out); }
BSX_API int bsx_sim_genome_index2(int64_t n, uint64_t seed, int n_contigs, double repeat_frac, int profile, bsx_index_t **out)
{
	uint8_t *g;
	bsx_gsink_t *S;
	int64_t i;
	int c, rc = BSX_OK;
	char *buf;
	*out = 0;
	if (n < 1000 || n_contigs < 1 || (profile == 1 && n < 100000)) return BSX_E_ARG;
	if ((g = gen_genome(n, seed, repeat_frac, profile)) == 0) return BSX_E_NOMEM;
	buf = (char*)malloc(1 << 20);
	S = bsx_gsink_new();
	for (c = 0; c < n_contigs && rc == BSX_OK; ++c) {
		int64_t b, e, nb, nl;
		char nm[32];
		contig_span(n, n_contigs, c, &b, &e, &nb, &nl);
		snprintf(nm, sizeof(nm), "chr%d", c + 1);
		bsx_gsink_contig(S, nm, 0);
		for (i = b; i < e && rc == BSX_OK; i += 1 << 20) {
			int64_t k, m = e - i < (1 << 20) ? e - i : (1 << 20);
			for (k = 0; k < m; ++k) buf[k] = (i + k >= nb && i + k < nb + nl) ? 'N' : "ACGT"[g[i + k]];
			rc = bsx_gsink_bases(S, buf, m);
		}
	}
	free(buf); free(g);
	*out = bsx_gsink_finish(S);
	if (rc != BSX_OK) { bsx_index_free(*out); *out = 0; }
	return rc;
}

/* n_pairs read pairs as an interleaved bsx_read_t array (caller frees with bsx_sim_free_reads) */
BSX_API int bsx_sim_pairs_truth(const bsx_index_t *idx, int64_t n_pairs, int read_len, uint64_t seed, int frag_lo, int frag_hi,
                                double sub_rate, double pbat_frac, bsx_read_t **out, int64_t *truth);
BSX_API int bsx_sim_pairs(const bsx_index_t *idx, int64_t n_pairs, int read_len, uint64_t seed, int frag_lo, int frag_hi,
                          double sub_rate, double pbat_frac, bsx_read_t **out)
{
	return bsx_sim_pairs_truth(idx, n_pairs, read_len, seed, frag_lo, frag_hi, sub_rate, pbat_frac, out, 0);
}
/* truth (optional, 2 per pair): forward-strand start of the fragment, fragment length << 1 | taken from the reverse strand */
BSX_API int bsx_sim_pairs_truth(const bsx_index_t *idx, int64_t n_pairs, int read_len, uint64_t seed, int frag_lo, int frag_hi,
                                double sub_rate, double pbat_frac, bsx_read_t **out, int64_t *truth)
{
	rng_t R;
	int64_t l_pac = idx->ref.l_pac, p;
	bsx_read_t *reads;
	uint8_t *frag;
	if (frag_lo < read_len) frag_lo = read_len;
	if (frag_hi < frag_lo) frag_hi = frag_lo;
	if (l_pac <= frag_hi + 2) return BSX_E_ARG;
	R.s = seed * 0x9E3779B97F4A7C15ULL + 0xABCDEFULL;
	reads = (bsx_read_t*)calloc((size_t)n_pairs * 2, sizeof(bsx_read_t));
	frag = (uint8_t*)malloc((size_t)frag_hi + 8);
	if (!reads || !frag) { free(reads); free(frag); return BSX_E_NOMEM; }
	for (p = 0; p < n_pairs; ++p) {
		int fl = frag_lo + (int)rng_below(&R, (uint64_t)(frag_hi - frag_lo + 1)), i, rev, e, swap;
		int64_t s = (int64_t)rng_below(&R, (uint64_t)(l_pac - fl));
		int rid = bsx_pos2rid(&idx->ref, s);
		if (bsx_pos2rid(&idx->ref, s + fl - 1) != rid) { --p; continue; }   /* fragment inside one contig */
		rev = (int)(rng_next(&R) & 1);
		if (truth) { truth[p * 2] = s; truth[p * 2 + 1] = (int64_t)fl << 1 | rev; }
		for (i = 0; i < fl; ++i) frag[i] = rev ? (uint8_t)(3 - bsx_pac_get(idx->pac, s + fl - 1 - i)) : (uint8_t)bsx_pac_get(idx->pac, s + i);
		for (i = 0; i < fl; ++i) { /* bisulfite conversion of the fragment's top strand */
			if (frag[i] == 1) {
				double keep = (i + 1 < fl && frag[i + 1] == 2) ? 0.70 : 0.01;
				if (rng_unit(&R) >= keep) frag[i] = 3;
			}
		}
		swap = rng_unit(&R) < pbat_frac;   /* PBAT-like pair: the two reads trade roles */
		for (e = 0; e < 2; ++e) {
			bsx_read_t *r = &reads[p * 2 + e];
			char nm[32];
			int from_end = (e == 1) ^ swap;
			snprintf(nm, sizeof(nm), "p%09lld", (long long)p);
			r->name = strdup(nm);
			r->seq = r->seq0 = (uint8_t*)malloc((size_t)read_len + 1);
			r->qual = (char*)malloc((size_t)read_len + 1);
			for (i = 0; i < read_len; ++i) {
				uint8_t b = from_end ? (uint8_t)(3 - frag[fl - 1 - i]) : frag[i];
				if (rng_unit(&R) < sub_rate) b = (uint8_t)((b + 1 + rng_below(&R, 3)) & 3);
				r->seq[i] = b; r->qual[i] = 'I';
			}
			r->qual[read_len] = 0;
			r->l_seq = r->l_seq0 = read_len; r->id = (int)(p * 2 + e);
		}
	}
	free(frag);
	*out = reads;
	return BSX_OK;
}

/* undo clipping and drop the SAM text so that the same array can be processed again */
BSX_API void bsx_sim_reset_reads(bsx_read_t *reads, int64_t n)
{
	int64_t i;
	for (i = 0; i < n; ++i) {
		if (reads[i].seq0) { reads[i].seq = reads[i].seq0; reads[i].l_seq = reads[i].l_seq0; }
		free(reads[i].sam); reads[i].sam = 0;
		reads[i].clip5 = reads[i].clip3 = reads[i].l_adaptor = 0;
	}
}

BSX_API int64_t bsx_sim_sam_bytes(const bsx_read_t *reads, int64_t n)
{
	int64_t i, tot = 0;
	for (i = 0; i < n; ++i) if (reads[i].sam) tot += (int64_t)strlen(reads[i].sam);
	return tot;
}

BSX_API void bsx_sim_free_reads(bsx_read_t *reads, int64_t n)
{
	int64_t i;
	if (!reads) return;
	for (i = 0; i < n; ++i) { free(reads[i].name); free(reads[i].comment); free(reads[i].barcode); free(reads[i].umi); free(reads[i].seq0); free(reads[i].qual); free(reads[i].sam); }
	free(reads);
}

/* the pairs of bsx_sim_pairs as two FASTQ files (end-to-end runs of the command line) */
BSX_API int bsx_sim_write_fastq(const bsx_read_t *reads, int64_t n, const char *fq1, const char *fq2, int append)
{
	FILE *f[2];
	int64_t i;
	int k;
	f[0] = fopen(fq1, append ? "a" : "w"); f[1] = fopen(fq2, append ? "a" : "w");
	if (!f[0] || !f[1]) { if (f[0]) fclose(f[0]); if (f[1]) fclose(f[1]); return BSX_E_IO; }
	for (i = 0; i < n; ++i) {
		const bsx_read_t *r = &reads[i];
		FILE *o = f[i & 1];
		fputc('@', o); fputs(r->name, o); fputc('\n', o);
		for (k = 0; k < r->l_seq; ++k) fputc("ACGTN"[r->seq[k] < 4 ? r->seq[k] : 4], o);
		fputs("\n+\n", o); fputs(r->qual, o); fputc('\n', o);
	}
	fclose(f[0]); fclose(f[1]);
	return BSX_OK;
}
