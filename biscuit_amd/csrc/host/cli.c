/* cli.c -- L1: the `biscuit align` command line, main_align (lib/aln/align.c:319-598):
 * same getopt string, option -> field mapping, -A rescaling, -x presets, header, chunk size
 * (chunk_size x n_threads bases, align.c:576) and chunk order.  SAM goes to stdout. */
#include <unistd.h>
#include <ctype.h>
#include <math.h>
#include <getopt.h>
#include <pthread.h>
#include "bsx_core.h"
#include "tune.h"
#include "fastq.h"
#include "pipeline.h"

const uint8_t *bsx_nt4_table(void);
BSX_API char *bsx_pg_line = 0;
/* Chunk-level sharding for multi-GPU runs (one process per GPU): chunks are independent in the
 * reference (insert-size statistics are per chunk, bwamem.c:464-467), so rank r of `world` processes
 * chunks r, r+world, ... of the same input stream with the same n_processed offsets, and the SAM is
 * identical to the single-process run once the chunks are written back in order.  The launcher
 * (biscuit_amd/multi_gpu.py) installs an emit hook and gathers the text over RCCL. */
BSX_API int bsx_shard_rank = 0, bsx_shard_world = 1;
/* 0: the chunks of the input are dealt to the ranks (chunk k to rank k % world).  1: every rank takes every chunk and aligns its own
 * slice of the chunk's pairs (for inputs with fewer chunks than GPUs); the insert-size statistics are those of the whole
 * chunk (bsx_pes_hist_hook, region.c), the slices leave in rank order as chunks k * world + rank of the gather */
BSX_API int bsx_shard_mode = 0;
void bsx_pestat_sync_empty(const bsx_opt_t *opt);
void bsx_chunk_slice_offset(int64_t first);
extern void (*bsx_pes_hist_hook)(void *ud, int64_t *hist, int n_bins);
#define CHUNK_WORLD (bsx_shard_mode ? 1 : bsx_shard_world)
#define CHUNK_RANK (bsx_shard_mode ? 0 : bsx_shard_rank)
/* this rank's slice of a chunk (within-chunk sharding): reads [*a, *a + return) of n, whole pairs */
static int shard_slice(int n, int is_pe, int *a)
{
	const int unit = is_pe ? 2 : 1, U = n / unit;
	const int ua = (int)((int64_t)U * bsx_shard_rank / bsx_shard_world), ub = (int)((int64_t)U * (bsx_shard_rank + 1) / bsx_shard_world);
	*a = ua * unit;
	return (ub - ua) * unit;
}
/* keep reads [a, a + m) of the chunk, release the others */
static void shard_trim(bsx_read_t *seqs, int n, int a, int m)
{
	int i;
	for (i = 0; i < a; ++i) bsx_read_free(&seqs[i]);
	for (i = a + m; i < n; ++i) bsx_read_free(&seqs[i]);
	if (a > 0 && m > 0) memmove(seqs, seqs + a, (size_t)m * sizeof(bsx_read_t));
}
BSX_API void (*bsx_emit_hook)(void *ud, int64_t chunk, const char *text, size_t len) = 0;
BSX_API void *bsx_emit_ud = 0;   /* "@PG\t..." set by the program entry, printed after the header lines */

static int usage(void)
{
	fprintf(stderr,
		"\nUsage: biscuit_align [options] <fai-index base> <in1.fq> [in2.fq]\n\n"
		"MI355X implementation of `biscuit align`; options, defaults and SAM output follow the reference:\n"
		"  algorithm : -@ INT threads  -b INT parent/daughter restriction  -f INT BSW/BSC restriction\n"
		"              -k INT min seed  -w INT band  -d INT z-drop  -r FLOAT re-seed factor  -y INT 3rd-round occ\n"
		"              -J/-K STR adaptors  -z INT min base quality  -5/-3 INT extra clipping  -c INT max occ\n"
		"              -D FLOAT chain drop ratio  -W INT min chain weight  -m INT mate-rescue rounds  -S -P -e -9\n"
		"  scoring   : -A -B INT  -O -E -L INT[,INT]  -U INT\n"
		"  I/O       : -1/-2 STR reads on the command line  -i -p -R STR -F -H STR/FILE -j -q -T INT -g INT[,INT]\n"
		"              -a -C -V -Y -M -I FLOAT[,FLOAT[,INT[,INT]]] -v INT -h\n"
		"  device    : $BSX_DEVICE selects the HIP device ordinal (default 0)\n\n");
	return 1;
}

static void update_a(bsx_opt_t *opt, const bsx_opt_t *opt0)   /* align.c:169-182 */
{
	if (opt0->a) {
		if (!opt0->b) opt->b *= opt->a;
		if (!opt0->T) opt->T *= opt->a;
		if (!opt0->o_del) opt->o_del *= opt->a;
		if (!opt0->e_del) opt->e_del *= opt->a;
		if (!opt0->o_ins) opt->o_ins *= opt->a;
		if (!opt0->e_ins) opt->e_ins *= opt->a;
		if (!opt0->zdrop) opt->zdrop *= opt->a;
		if (!opt0->pen_clip5) opt->pen_clip5 *= opt->a;
		if (!opt0->pen_clip3) opt->pen_clip3 *= opt->a;
		if (!opt0->pen_unpaired) opt->pen_unpaired *= opt->a;
	}
}

static void infer_alt(bsx_refmeta_t *r)   /* infer_alt_chromosomes, align.c:184-224 */
{
	int i, n, found[25];
	for (i = 0; i < r->n_seqs; ++i) if (r->anns[i].is_alt) return;
	memset(found, 0, sizeof(found));
	for (i = 0; i < r->n_seqs; ++i) {
		const char *nm = r->anns[i].name;
		if (strncmp(nm, "chr", 3) != 0) continue;
		if (strlen(nm) == 4) {
			if (toupper(nm[3]) == 'X') found[22] = 1;
			else if (toupper(nm[3]) == 'Y') found[23] = 1;
			else if (toupper(nm[3]) == 'M') found[24] = 1;
			else if (isdigit((unsigned char)nm[3])) { int k = nm[3] - '0'; if (k > 0 && k <= 22) found[k - 1] = 1; }
		} else if (strlen(nm) == 5 && isdigit((unsigned char)nm[3]) && isdigit((unsigned char)nm[4])) {
			int k = atoi(nm + 3); if (k > 0 && k <= 22) found[k - 1] = 1;
		}
	}
	for (i = n = 0; i < 25; ++i) if (found[i]) ++n;
	if (n < 20) return;
	for (i = 0; i < r->n_seqs; ++i) {
		const char *nm = r->anns[i].name;
		if (strncmp(nm, "chrUn", 5) == 0 || strstr(nm, "_random") || strstr(nm, "_hap") || strstr(nm, "_alt")) r->anns[i].is_alt = 1;
	}
}

static char *escape(char *s)   /* bwa_escape, bwa.c:686-701 */
{
	char *p, *q;
	for (p = q = s; *p; ++p) {
		if (*p == '\\') {
			++p;
			if (*p == 't') *q++ = '\t'; else if (*p == 'n') *q++ = '\n'; else if (*p == 'r') *q++ = '\r'; else if (*p == '\\') *q++ = '\\';
		} else *q++ = *p;
	}
	*q = 0;
	return s;
}
static char *insert_header(const char *s, char *hdr)   /* bwa_insert_header, bwa.c:736-748 */
{
	size_t len = 0;
	if (s == 0 || s[0] != '@') return hdr;
	if (hdr) { len = strlen(hdr); hdr = (char*)realloc(hdr, len + strlen(s) + 2); hdr[len++] = '\n'; strcpy(hdr + len, s); }
	else hdr = strdup(s);
	escape(hdr + len);
	return hdr;
}
static char *set_rg(const char *s)   /* bwa_set_rg, bwa.c:703-734 */
{
	char *p, *q, *r, *rg_line;
	memset(bsx_rg_id, 0, 256);
	if (strstr(s, "@RG") != s) { fprintf(stderr, "[E::%s] the read group line is not started with @RG\n", __func__); return 0; }
	rg_line = strdup(s);
	escape(rg_line);
	if ((p = strstr(rg_line, "\tID:")) == 0) { fprintf(stderr, "[E::%s] no ID at the read group line\n", __func__); free(rg_line); return 0; }
	p += 4;
	for (q = p; *q && *q != '\t' && *q != '\n'; ++q);
	if (q - p + 1 > 256) { fprintf(stderr, "[E::%s] @RG:ID is longer than 255 characters\n", __func__); free(rg_line); return 0; }
	for (q = p, r = bsx_rg_id; *q && *q != '\t' && *q != '\n'; ++q) *r++ = *q;
	return rg_line;
}

static int ann_name_lt(const void *a, const void *b) { return strcmp((*(bsx_ann_t* const*)a)->name, (*(bsx_ann_t* const*)b)->name) < 0; }

/* bwa_print_sam_hdr, bwa.c:654-684: @SQ lines sorted by name with the reference's introsort */
BSX_API char *bsx_sam_header(const bsx_index_t *idx, const char *hdr_line, const char *pg_line)
{
	BSX_VEC(char) out;
	int i, n_SQ = 0;
	char buf[4096];
	bsx_vec_init(out);
#define OUTS(s_) do { const char *z_ = (s_); size_t l_ = strlen(z_); bsx_vec_reserve(out, out.n + l_ + 1); memcpy(out.a + out.n, z_, l_); out.n += l_; } while (0)
	if (hdr_line) {
		const char *p = hdr_line;
		while ((p = strstr(p, "@SQ\t")) != 0) { if (p == hdr_line || *(p - 1) == '\n') ++n_SQ; p += 4; }
	}
	if (n_SQ == 0) {
		const bsx_ann_t **ap = (const bsx_ann_t**)malloc(sizeof(*ap) * (idx->ref.n_seqs + 1));
		for (i = 0; i < idx->ref.n_seqs; ++i) ap[i] = &idx->ref.anns[i];
		bsx_introsort(ap, idx->ref.n_seqs, sizeof(*ap), ann_name_lt);
		for (i = 0; i < idx->ref.n_seqs; ++i) { snprintf(buf, sizeof(buf), "@SQ\tSN:%s\tLN:%d\n", ap[i]->name, ap[i]->len); OUTS(buf); }
		free(ap);
	} else if (n_SQ != idx->ref.n_seqs && bsx_verbose >= 2)
		fprintf(stderr, "[W::%s] %d @SQ lines provided with -H; %d sequences in the index. Continue anyway.\n", __func__, n_SQ, idx->ref.n_seqs);
	if (hdr_line) { OUTS(hdr_line); OUTS("\n"); }
	if (pg_line) { OUTS(pg_line); OUTS("\n"); }
	bsx_vec_reserve(out, out.n + 1);
	out.a[out.n] = 0;
	return out.a;
}

/* bseq_classify (bwa.c:118-138) for -p: consecutive equal names form a pair */
static void classify(int n, bsx_read_t *seqs, int m[2], bsx_read_t *sep[2])
{
	int i, has_last;
	BSX_VEC(bsx_read_t) a[2];
	bsx_vec_init(a[0]); bsx_vec_init(a[1]);
	for (i = 1, has_last = 1; i < n; ++i) {
		if (has_last) {
			if (strcmp(seqs[i].name, seqs[i - 1].name) == 0) { bsx_vec_push(a[1], seqs[i - 1]); bsx_vec_push(a[1], seqs[i]); has_last = 0; }
			else bsx_vec_push(a[0], seqs[i - 1]);
		} else has_last = 1;
	}
	if (has_last && n > 0) bsx_vec_push(a[0], seqs[i - 1]);
	sep[0] = a[0].a; m[0] = (int)a[0].n; sep[1] = a[1].a; m[1] = (int)a[1].n;
	if (bsx_verbose >= 3) fprintf(stderr, "[%s] %d SE sequences; %d PE sequences\n", "bseq_classify", m[0], m[1]);
}

/* set by the product entry: chunks go through bsx_stream_* on the device `ud` names instead of one at a time */
static int g_use_stream = 0;
/* set by the product entry: how to release what its open_device callback made */
static void (*g_close_device)(void *ud) = 0;

typedef int (*process_fn)(void *ud, const bsx_opt_t *opt, const bsx_index_t *idx, int64_t n_processed, int n, bsx_read_t *reads, const bsx_pestat_t *pes0);

/* shared by the product entry (HIP) and the test-only entry that injects another backend */
/* ---- step 0 and step 2 of the reference's kt_pipeline (align.c:100-170) as two helper threads: one parses the next
 * chunks of FASTQ ahead of the aligner, one writes finished chunks out, so that neither sits on the thread that runs
 * the back half of the alignment.  Bounded queues of chunk records between them. */
typedef struct { bsx_read_t *seqs; int n; int64_t idx; int ok; int64_t n_before; } chunk_rec_t;   /* n_before: reads of the input ahead of the chunk (-1: count along) */
typedef struct {
	pthread_mutex_t mu; pthread_cond_t cv;
	chunk_rec_t q[4]; int head, count, closed;
} chunk_q_t;
static void cq_init(chunk_q_t *Q) { memset(Q, 0, sizeof(*Q)); pthread_mutex_init(&Q->mu, 0); pthread_cond_init(&Q->cv, 0); }
static void cq_put(chunk_q_t *Q, chunk_rec_t r)
{
	pthread_mutex_lock(&Q->mu);
	while (Q->count == 4) pthread_cond_wait(&Q->cv, &Q->mu);
	Q->q[(Q->head + Q->count++) & 3] = r;
	pthread_cond_broadcast(&Q->cv);
	pthread_mutex_unlock(&Q->mu);
}
static void cq_close(chunk_q_t *Q) { pthread_mutex_lock(&Q->mu); Q->closed = 1; pthread_cond_broadcast(&Q->cv); pthread_mutex_unlock(&Q->mu); }
static int cq_get(chunk_q_t *Q, chunk_rec_t *r)   /* 0 when the queue is closed and empty */
{
	int got = 0;
	pthread_mutex_lock(&Q->mu);
	while (Q->count == 0 && !Q->closed) pthread_cond_wait(&Q->cv, &Q->mu);
	if (Q->count) { *r = Q->q[Q->head]; Q->head = (Q->head + 1) & 3; --Q->count; got = 1; pthread_cond_broadcast(&Q->cv); }
	pthread_mutex_unlock(&Q->mu);
	return got;
}
static void emit_chunk(bsx_read_t *seqs, int n, int64_t chunk_idx, int ok);
static volatile int g_write_error = 0;   /* a write to stdout failed (the reference aborts there: err_fputs, utils.c:214) */
/* Several ranks over plain files: a scanner thread lists the chunk boundaries without building records (fastq.c) and this rank
 * parses only its own chunks r, r+N, ..., seeking to each.  Compressed or piped input: every rank inflates everything (threads of
 * fastq.c, ahead of the parser), parses its own chunks and walks the others' by the record grammar without building records (the chunk
 * rule is cumulative). */
static bsx_fq_scan_t *shard_scan_start(const char *fn1, const char *fn2, int chunk)
{
	bsx_fq_scan_t *s;
	if (CHUNK_WORLD <= 1 || bsx_tune_long("no_chunk_scan", 0)) return 0;
	s = bsx_fq_scan_start(fn1, fn2, chunk);
	if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] rank %d of %d: %s\n", "main_align", bsx_shard_rank, bsx_shard_world,
	                              s ? "chunk boundaries by a scan of the input, this rank parses its own chunks only" : "compressed or piped input: this rank inflates all of it, parses its own chunks and walks the others without building records");
	return s;
}
typedef struct { chunk_q_t *Q; bsx_fq_t *f1, *f2; const char *fn1, *fn2; int chunk, has_bc, copy_comment; volatile int stop, failed; } reader_t;
static void *reader_main(void *arg)
{
	reader_t *R = (reader_t*)arg;
	int64_t idx = 0;
	bsx_fq_scan_t *scan = shard_scan_start(R->fn1, R->fn2, R->chunk);
	/* several ranks over input that cannot be sought in (compressed, piped): the chunks of the other ranks are walked without building
	 * their records (bsx_fq_skip_chunk); this rank's own are parsed in place, the inflate threads of fastq.c running ahead of both */
	const int skip_mode = !scan && CHUNK_WORLD > 1 && !bsx_tune_long("no_chunk_skip", 0);
	int64_t n_before = 0;
	bsx_fq_pair_t *P = (scan || skip_mode) ? 0 : bsx_fq_pair_open(R->f1, R->f2, R->has_bc);
	if (scan) idx = bsx_shard_rank;
	while (!R->stop) {
		chunk_rec_t r;
		int i;
		r.n_before = -1;
		if (skip_mode) {
			if (idx % bsx_shard_world != bsx_shard_rank) {
				const int n = bsx_fq_skip_chunk(R->f1, R->f2, R->chunk);
				if (n == 0) break;
				n_before += n; ++idx;
				continue;
			}
			r.seqs = bsx_fq_read_chunk(R->f1, R->f2, R->chunk, R->has_bc, &r.n);
			r.n_before = n_before;
			n_before += r.n;
		} else
		if (scan) { /* this rank's next chunk: where it starts is known, the parser threads start there */
			bsx_fq_chunkpos_t cp;
			if (!bsx_fq_scan_get(scan, idx, &cp)) break;
			if (bsx_fq_seek(R->f1, cp.off1) != 0 || (R->f2 && bsx_fq_seek(R->f2, cp.off2) != 0)) { fprintf(stderr, "[E::%s] cannot seek in the input\n", "reader"); R->failed = 1; break; }
			P = bsx_fq_pair_open_n(R->f1, R->f2, R->has_bc, R->f2 ? cp.n / 2 : cp.n);   /* (no parsing ahead into the chunks of other ranks) */
			r.seqs = bsx_fq_pair_read_chunk(P, R->chunk, &r.n);
			bsx_fq_pair_close(P); P = 0;
			if (r.seqs == 0 || r.n != cp.n) { fprintf(stderr, "[E::%s] chunk %ld: %d reads where the scan counted %d\n", "reader", (long)idx, r.n, cp.n); if (r.seqs) { for (i = 0; i < r.n; ++i) bsx_read_free(&r.seqs[i]); free(r.seqs); } R->failed = 1; break; }
			r.n_before = cp.n_before;
		} else r.seqs = bsx_fq_pair_read_chunk(P, R->chunk, &r.n);
		if (r.seqs == 0 || r.n == 0) { free(r.seqs); break; }
		if (!R->copy_comment) for (i = 0; i < r.n; ++i) { free(r.seqs[i].comment); r.seqs[i].comment = 0; }
		r.idx = idx; r.ok = 1;
		idx += scan ? bsx_shard_world : 1;
		cq_put(R->Q, r);
	}
	cq_close(R->Q);
	if (P) bsx_fq_pair_close(P);
	bsx_fq_scan_close(scan);
	return 0;
}
static void *writer_main(void *arg)
{
	chunk_q_t *Q = (chunk_q_t*)arg;
	chunk_rec_t r;
	while (cq_get(Q, &r)) emit_chunk(r.seqs, r.n, r.idx, r.ok);
	return 0;
}

/* write out (or hand to the hook) the SAM text of a finished chunk and release its reads */
static void emit_chunk(bsx_read_t *seqs, int n, int64_t chunk_idx, int ok)
{
	int i;
	if (ok && bsx_emit_hook) {
		size_t tot = 0, at = 0; char *all;
		for (i = 0; i < n; ++i) if (seqs[i].sam) tot += strlen(seqs[i].sam);
		all = (char*)malloc(tot + 1);
		for (i = 0; i < n; ++i) if (seqs[i].sam) { size_t l = strlen(seqs[i].sam); memcpy(all + at, seqs[i].sam, l); at += l; }
		bsx_emit_hook(bsx_emit_ud, chunk_idx, all, tot);
		free(all);
	}
	for (i = 0; i < n; ++i) { if (ok && !bsx_emit_hook && !g_write_error && seqs[i].sam && fputs(seqs[i].sam, stdout) == EOF) g_write_error = 1; bsx_read_free(&seqs[i]); }
	free(seqs);
}

BSX_API int bsx_align_main_with(int argc, char **argv, process_fn process, void *ud, int (*open_device)(int ordinal, const bsx_index_t *idx, void **ud))
{
	bsx_opt_t opt_, opt0, *opt = &opt_;
	int c, i, ignore_alt = 0, auto_alt = 1, copy_comment = 0, device = 0, rc = 0;
	char *p, *rg_line = 0, *hdr_line = 0, *seq1 = 0, *seq2 = 0;
	const char *mode = 0;
	bsx_pestat_t *pes0 = 0;
	bsx_index_t *idx = 0;
	bsx_fq_t *f1 = 0, *f2 = 0;
	int64_t n_processed = 0;
	const uint8_t *nt4 = bsx_nt4_table();

	g_write_error = 0;   /* per call: a failed write of an earlier call in this process must not fail this one */
	if (getenv("BSX_DEVICE")) device = atoi(getenv("BSX_DEVICE"));
	bsx_opt_init(opt);
	opt->flag |= BSX_F_NO_MULTI;   /* align.c:335 */
	memset(&opt0, 0, sizeof(opt0));
	if (argc < 2) return usage();
	optind = 1;
	while ((c = getopt(argc, argv, ":@:1:2:3:5:9ab:c:d:ef:g:hijk:m:pqr:s:v:w:x:y:z:A:B:CD:E:FG:H:I:J:K:L:MN:O:PQ:R:ST:U:VW:X:Y")) >= 0) {
		if (c == 'k') opt->min_seed_len = atoi(optarg), opt0.min_seed_len = 1;
		else if (c == '1') seq1 = strdup(optarg);
		else if (c == '2') seq2 = strdup(optarg);
		else if (c == 'x') mode = optarg;
		else if (c == 'b') opt->parent = (uint8_t)atoi(optarg);
		else if (c == 'f') opt->bsstrand = (uint8_t)atoi(optarg);
		else if (c == 'i') auto_alt = 0;
		else if (c == 'w') opt->w = atoi(optarg), opt0.w = 1;
		else if (c == 'A') opt->a = atoi(optarg), opt0.a = 1;
		else if (c == 'B') opt->b = atoi(optarg), opt0.b = 1;
		else if (c == 'T') opt->T = atoi(optarg), opt0.T = 1;
		else if (c == 'U') opt->pen_unpaired = atoi(optarg), opt0.pen_unpaired = 1;
		else if (c == '@') opt->n_threads = atoi(optarg), opt->n_threads = opt->n_threads > 1 ? opt->n_threads : 1;
		else if (c == 'P') opt->flag |= BSX_F_NOPAIRING;
		else if (c == 'a') opt->flag |= BSX_F_ALL;
		else if (c == 'p') opt->flag |= BSX_F_PE | BSX_F_SMARTPE;
		else if (c == 'q') opt->flag |= BSX_F_KEEP_SUPP_MAPQ;
		else if (c == 'M') opt->flag |= BSX_F_NO_MULTI;
		else if (c == 'S') opt->flag |= BSX_F_NO_RESCUE;
		else if (c == 'e') opt->flag |= BSX_F_SELF_OVLP;
		else if (c == 'F') opt->flag |= BSX_F_ALN_REG;
		else if (c == 'Y') opt->flag |= BSX_F_SOFTCLIP;
		else if (c == 'V') opt->flag |= BSX_F_REF_HDR;
		else if (c == 'c') opt->max_occ = (uint32_t)atoi(optarg), opt0.max_occ = 1;
		else if (c == 'd') opt->zdrop = atoi(optarg), opt0.zdrop = 1;
		else if (c == 'v') bsx_verbose = atoi(optarg);
		else if (c == 'j') ignore_alt = 1;
		else if (c == 'r') opt->split_factor = (float)atof(optarg), opt0.split_factor = 1.;
		else if (c == 'D') opt->drop_ratio = (float)atof(optarg), opt0.drop_ratio = 1.;
		else if (c == 'm') opt->max_matesw = atoi(optarg), opt0.max_matesw = 1;
		else if (c == 's') opt->split_width = atoi(optarg), opt0.split_width = 1;
		else if (c == 'G') opt->max_chain_gap = atoi(optarg), opt0.max_chain_gap = 1;
		else if (c == 'N') opt->max_chain_extend = (uint32_t)atoi(optarg), opt0.max_chain_extend = 1;
		else if (c == 'W') opt->min_chain_weight = atoi(optarg), opt0.min_chain_weight = 1;
		else if (c == 'y') opt->max_mem_intv = (uint64_t)atol(optarg), opt0.max_mem_intv = 1;
		else if (c == 'C') copy_comment = 1;
		else if (c == 'J') {
			opt->l_adaptor1 = (int)strlen(optarg); opt->adaptor1 = (uint8_t*)calloc(opt->l_adaptor1 + 1, 1);
			for (i = 0; i < opt->l_adaptor1; ++i) opt->adaptor1[i] = nt4[(unsigned char)optarg[i]];
		} else if (c == 'K') {
			opt->l_adaptor2 = (int)strlen(optarg); opt->adaptor2 = (uint8_t*)calloc(opt->l_adaptor2 + 1, 1);
			for (i = 0; i < opt->l_adaptor2; ++i) opt->adaptor2[i] = nt4[(unsigned char)optarg[i]];
		} else if (c == 'z') opt->min_base_qual = atoi(optarg);
		else if (c == '5') opt->clip5 = atoi(optarg);
		else if (c == '3') opt->clip3 = atoi(optarg);
		else if (c == '9') opt->has_bc = 1;
		else if (c == 'X') opt->mask_level = (float)atof(optarg);
		else if (c == 'g') {
			opt0.max_XA_hits = opt0.max_XA_hits_alt = 1;
			opt->max_XA_hits = opt->max_XA_hits_alt = (int)strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt->max_XA_hits_alt = (int)strtol(p + 1, &p, 10);
		} else if (c == 'Q') {
			opt0.mapQ_coef_len = 1;
			opt->mapQ_coef_len = (float)atoi(optarg);
			opt->mapQ_coef_fac = opt->mapQ_coef_len > 0 ? log(opt->mapQ_coef_len) : 0;
		} else if (c == 'O') {
			opt0.o_del = opt0.o_ins = 1;
			opt->o_del = opt->o_ins = (int)strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt->o_ins = (int)strtol(p + 1, &p, 10);
		} else if (c == 'E') {
			opt0.e_del = opt0.e_ins = 1;
			opt->e_del = opt->e_ins = (int)strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt->e_ins = (int)strtol(p + 1, &p, 10);
		} else if (c == 'L') {
			opt0.pen_clip5 = opt0.pen_clip3 = 1;
			opt->pen_clip5 = opt->pen_clip3 = (int)strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt->pen_clip3 = (int)strtol(p + 1, &p, 10);
		} else if (c == 'R') {
			if ((rg_line = set_rg(optarg)) == 0) return 1;
		} else if (c == 'H') {
			if (optarg[0] != '@') {
				FILE *fp;
				if ((fp = fopen(optarg, "r")) != 0) {
					char *buf = (char*)calloc(1, 0x10000);
					while (fgets(buf, 0xffff, fp)) { size_t l = strlen(buf); if (l && buf[l - 1] == '\n') buf[l - 1] = 0; hdr_line = insert_header(buf, hdr_line); }
					free(buf); fclose(fp);
				}
			} else hdr_line = insert_header(optarg, hdr_line);
		} else if (c == 'I') { /* align.c:434-453 */
			pes0 = (bsx_pestat_t*)calloc(1, sizeof(bsx_pestat_t));
			pes0->avg = strtod(optarg, &p);
			pes0->std = pes0->avg * .1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0->std = strtod(p + 1, &p);
			pes0->high = (int)(pes0->avg + 4. * pes0->std + .499);
			pes0->low  = (int)(pes0->avg - 4. * pes0->std + .499);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0->high = (int)(strtod(p + 1, &p) + .499);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes0->low = (int)(strtod(p + 1, &p) + .499);
			if (bsx_verbose >= 3)
				fprintf(stderr, "[M::%s] mean insert size: %.3f, stddev: %.3f, max: %d, min: %d\n", "main_align", pes0->avg, pes0->std, pes0->high, pes0->low);
		} else if (c == 'h') return usage();
		else if (c == ':') { usage(); fprintf(stderr, "Option needs an argument: -%c\n", optopt); return 1; }
		else if (c == '?') { usage(); fprintf(stderr, "Unrecognized option: -%c\n", optopt); return 1; }
		else return usage();
	}
	if (rg_line) { hdr_line = insert_header(rg_line, hdr_line); free(rg_line); }
	if (opt->n_threads < 1) opt->n_threads = 1;
	if ((optind + 1 >= argc || optind + 3 < argc) && !seq1) { usage(); fprintf(stderr, "Missing fai-index base or FASTQ file\n"); return 1; }
	if (mode) { /* -x presets, align.c:476-512 */
		if (strcmp(mode, "intractg") == 0) {
			if (!opt0.o_del) opt->o_del = 16;
			if (!opt0.o_ins) opt->o_ins = 16;
			if (!opt0.b) opt->b = 9;
			if (!opt0.pen_clip5) opt->pen_clip5 = 5;
			if (!opt0.pen_clip3) opt->pen_clip3 = 5;
		} else if (strcmp(mode, "pacbio") == 0 || strcmp(mode, "pbref") == 0 || strcmp(mode, "pbread") == 0 || strcmp(mode, "ont2d") == 0) {
			if (!opt0.o_del) opt->o_del = 1;
			if (!opt0.e_del) opt->e_del = 1;
			if (!opt0.o_ins) opt->o_ins = 1;
			if (!opt0.e_ins) opt->e_ins = 1;
			if (!opt0.b) opt->b = 1;
			if (opt0.split_factor == 0.) opt->split_factor = 10.;
			if (strcmp(mode, "pbread") == 0) {
				opt->flag |= BSX_F_ALL | BSX_F_SELF_OVLP | BSX_F_ALN_REG;
				if (!opt0.min_chain_weight) opt->min_chain_weight = 40;
				if (!opt0.max_occ) opt->max_occ = 1000;
				if (!opt0.min_seed_len) opt->min_seed_len = 13;
				if (!opt0.max_chain_extend) opt->max_chain_extend = 25;
				if (opt0.drop_ratio == 0.) opt->drop_ratio = .001;
			} else if (strcmp(mode, "ont2d") == 0) {
				if (!opt0.min_chain_weight) opt->min_chain_weight = 20;
				if (!opt0.min_seed_len) opt->min_seed_len = 14;
				if (!opt0.pen_clip5) opt->pen_clip5 = 0;
				if (!opt0.pen_clip3) opt->pen_clip3 = 0;
			} else {
				if (!opt0.min_chain_weight) opt->min_chain_weight = 40;
				if (!opt0.min_seed_len) opt->min_seed_len = 17;
				if (!opt0.pen_clip5) opt->pen_clip5 = 0;
				if (!opt0.pen_clip3) opt->pen_clip3 = 0;
			}
		} else { fprintf(stderr, "[E::%s] unknown read type '%s'\n", "main_align", mode); return 1; }
	} else update_a(opt, &opt0);
	bsx_opt_fill_matrices(opt);
	if (optind >= argc) { usage(); fprintf(stderr, "Missing fai-index base\n"); return 1; }
	if ((rc = bsx_index_load(argv[optind], &idx)) != BSX_OK) { fprintf(stderr, "[E::%s] fail to locate the index files (%s)\n", "main_align", bsx_strerror(rc)); return 1; }
	if (auto_alt) infer_alt(&idx->ref);
	if (ignore_alt) for (i = 0; i < idx->ref.n_seqs; ++i) idx->ref.anns[i].is_alt = 0;
	if (open_device) ud = 0;
	if (open_device && (rc = open_device(device, idx, &ud)) != BSX_OK) { fprintf(stderr, "[E::%s] %s\n", "main_align", bsx_strerror(rc)); ud = 0; rc = 1; goto cleanup; }
	if (!seq1) {
		if ((f1 = bsx_fq_open(argv[optind + 1])) == 0) { fprintf(stderr, "[E::%s] fail to open file `%s'.\n", "main_align", argv[optind + 1]); rc = 1; goto cleanup; }
		if (optind + 2 < argc) {
			if (opt->flag & BSX_F_PE) { if (bsx_verbose >= 2) fprintf(stderr, "[W::%s] when '-p' is in use, the second query file is ignored.\n", "main_align"); }
			else {
				if ((f2 = bsx_fq_open(argv[optind + 2])) == 0) { fprintf(stderr, "[E::%s] fail to open file `%s'.\n", "main_align", argv[optind + 2]); rc = 1; goto cleanup; }
				opt->flag |= BSX_F_PE;
			}
		}
	}
	if (!(opt->flag & BSX_F_ALN_REG) && bsx_shard_rank == 0) {
		char *h = bsx_sam_header(idx, hdr_line, bsx_pg_line);
		if (bsx_emit_hook) bsx_emit_hook(bsx_emit_ud, -1, h, strlen(h)); else if (fputs(h, stdout) == EOF) g_write_error = 1;
		free(h);
	}
	{
		/* $BSX_CHUNK_SIZE overrides the per-thread chunk size (tests only; the reference's is fixed at 10 Mbp) */
		int chunk = (getenv("BSX_CHUNK_SIZE") ? atoi(getenv("BSX_CHUNK_SIZE")) : opt->chunk_size) * opt->n_threads, done_cmdline = 0;
		int64_t chunk_idx = -1;
		bsx_fq_scan_t *scan = 0;
		/* pipelined mode: chunks in flight, oldest first (the reference's kt_pipeline keeps reading/aligning/writing
		 * of consecutive chunks in flight the same way, align.c:165-170) */
		struct { bsx_read_t *seqs; int n; int64_t idx; } pend[8];
		int n_pend = 0, depth = 1;
		bsx_stream_t *stream = 0;
		if (g_use_stream && !(opt->flag & BSX_F_SMARTPE)) {
			if ((rc = bsx_stream_open((bsx_device_t*)ud, opt, idx, pes0, &stream)) != BSX_OK) { fprintf(stderr, "[E::%s] %s\n", "main_align", bsx_strerror(rc)); rc = 1; goto cleanup; }
			depth = bsx_stream_depth(stream);
		}
		if (stream && !seq1) { /* reader thread -> this thread (stream push / back halves) -> writer thread */
			chunk_q_t in_q, out_q;
			reader_t R;
			pthread_t th_r, th_w;
			chunk_rec_t r;
			cq_init(&in_q); cq_init(&out_q);
			R.Q = &in_q; R.f1 = f1; R.f2 = f2; R.fn1 = argv[optind + 1]; R.fn2 = f2 ? argv[optind + 2] : 0; R.chunk = chunk; R.has_bc = opt->has_bc; R.copy_comment = copy_comment; R.stop = 0; R.failed = 0;
			pthread_create(&th_r, 0, reader_main, &R);
			pthread_create(&th_w, 0, writer_main, &out_q);
			while (cq_get(&in_q, &r)) {
				int64_t size = 0;
				for (i = 0; i < r.n; ++i) size += r.seqs[i].l_seq;
				if (r.n_before >= 0) n_processed = r.n_before;   /* the reader skipped the chunks of the other ranks */
				if (CHUNK_WORLD > 1 && r.idx % bsx_shard_world != bsx_shard_rank) { /* another rank's chunk */
					n_processed += r.n;
					for (i = 0; i < r.n; ++i) bsx_read_free(&r.seqs[i]);
					free(r.seqs);
					continue;
				}
				if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] read %d sequences (%ld bp)...\n", "process", r.n, (long)size);
				if (bsx_shard_mode && bsx_shard_world > 1) { /* this rank's slice of the chunk */
					int a = 0, m = shard_slice(r.n, (opt->flag & BSX_F_PE) != 0, &a);
					const int n_all = r.n, short_chunk = r.n / ((opt->flag & BSX_F_PE) ? 2 : 1) < bsx_shard_world;
					shard_trim(r.seqs, r.n, a, m);
					r.n = m; r.idx = r.idx * bsx_shard_world + bsx_shard_rank;
					/* a chunk with fewer pairs than ranks leaves some slices empty: every rank brings its pipeline to this chunk first, so that
					 * the exchange of the insert-size histograms is everybody's next one */
					if (short_chunk) { rc = bsx_stream_flush(stream); while (rc == BSX_OK && n_pend) { chunk_rec_t d; d.seqs = pend[0].seqs; d.n = pend[0].n; d.idx = pend[0].idx; d.ok = 1; d.n_before = -1; cq_put(&out_q, d); for (i = 1; i < n_pend; ++i) pend[i - 1] = pend[i]; --n_pend; } }
					if (rc == BSX_OK && m > 0) { bsx_chunk_slice_offset(a); rc = bsx_stream_push(stream, n_processed + a, m, r.seqs); }
					if (rc == BSX_OK && short_chunk) { rc = bsx_stream_flush(stream); if (m == 0 && (opt->flag & BSX_F_PE) && !pes0) bsx_pestat_sync_empty(opt); }
					if (m > 0 && !short_chunk) { pend[n_pend].seqs = r.seqs; pend[n_pend].n = m; pend[n_pend].idx = r.idx; ++n_pend; }
					else { chunk_rec_t d; d.seqs = r.seqs; d.n = m; d.idx = r.idx; d.ok = rc == BSX_OK; d.n_before = -1; cq_put(&out_q, d); }   /* (done, or empty: straight to the writer) */
					n_processed += n_all;
				} else {
				rc = bsx_stream_push(stream, n_processed, r.n, r.seqs);
				pend[n_pend].seqs = r.seqs; pend[n_pend].n = r.n; pend[n_pend].idx = r.idx; ++n_pend;
				n_processed += r.n;
				}
				if (rc != BSX_OK) { fprintf(stderr, "[E::%s] alignment failed: %s\n", "main_align", bsx_strerror(rc)); rc = 1; R.stop = 1; break; }
				if (g_write_error) { rc = 1; R.stop = 1; break; }
				while (n_pend > depth - 1) { /* the push completed the oldest chunk in flight */
					chunk_rec_t d; d.seqs = pend[0].seqs; d.n = pend[0].n; d.idx = pend[0].idx; d.ok = 1; d.n_before = -1;
					cq_put(&out_q, d);
					for (i = 1; i < n_pend; ++i) pend[i - 1] = pend[i];
					--n_pend;
				}
			}
			if (rc) { chunk_rec_t d; while (cq_get(&in_q, &d)) { for (i = 0; i < d.n; ++i) bsx_read_free(&d.seqs[i]); free(d.seqs); } }   /* let the reader finish */
			if (rc == 0 && (rc = bsx_stream_flush(stream)) != BSX_OK) { fprintf(stderr, "[E::%s] alignment failed: %s\n", "main_align", bsx_strerror(rc)); rc = 1; }
			/* after an error chunks are still in flight: closing the stream joins their front halves and runs their back halves, which
			 * read the pending chunks' reads -- so close it before those reads go to the writer (which frees them) */
			if (rc) { bsx_stream_close(stream); stream = 0; }
			for (i = 0; i < n_pend; ++i) { chunk_rec_t d; d.seqs = pend[i].seqs; d.n = pend[i].n; d.idx = pend[i].idx; d.ok = rc == 0; d.n_before = -1; cq_put(&out_q, d); }
			n_pend = 0;
			cq_close(&out_q);
			pthread_join(th_r, 0); pthread_join(th_w, 0);
			if (R.failed) rc = 1;   /* the reader gave up on the input: the SAM is incomplete */
			if (stream) bsx_stream_close(stream);
			stream = 0;
			goto loop_done;
		}
		for (;;) {
			int n = 0;
			bsx_read_t *seqs = 0;
			int64_t size = 0;
			if (seq1) { /* reads given with -1/-2 (align.c:77-81, bwa.c:749-764) */
				if (done_cmdline) break;
				done_cmdline = 1;
				n = seq2 ? 2 : 1;
				seqs = (bsx_read_t*)calloc(2, sizeof(bsx_read_t));
				for (i = 0; i < n; ++i) {
					const char *sq = i ? seq2 : seq1;
					size_t l = strlen(sq), k;
					seqs[i].name = strdup("inputread");
					seqs[i].seq = seqs[i].seq0 = (uint8_t*)malloc(l + 1);
					for (k = 0; k < l; ++k) seqs[i].seq[k] = nt4[(unsigned char)sq[k]];
					seqs[i].l_seq = seqs[i].l_seq0 = (int)l;
				}
				if (seq2) opt->flag |= BSX_F_PE;
			} else {
				if (chunk_idx < 0) scan = shard_scan_start(argv[optind + 1], f2 ? argv[optind + 2] : 0, chunk);
				if (scan) { /* straight to this rank's next chunk (see reader_main) */
					bsx_fq_chunkpos_t cp;
					const int64_t k = chunk_idx < 0 ? CHUNK_RANK : chunk_idx + CHUNK_WORLD;
					if (!bsx_fq_scan_get(scan, k, &cp)) break;
					if (bsx_fq_seek(f1, cp.off1) != 0 || (f2 && bsx_fq_seek(f2, cp.off2) != 0)) { fprintf(stderr, "[E::%s] cannot seek in the input\n", "main_align"); rc = 1; break; }
					chunk_idx = k - 1;
					n_processed = cp.n_before;
				}
				if (!scan && CHUNK_WORLD > 1 && (chunk_idx + 1) % bsx_shard_world != bsx_shard_rank && !bsx_tune_long("no_chunk_skip", 0)) {
					/* another rank's chunk of input that cannot be sought in: walked, not parsed (bsx_fq_skip_chunk) */
					const int ns = bsx_fq_skip_chunk(f1, f2, chunk);
					if (ns == 0) break;
					n_processed += ns; ++chunk_idx;
					continue;
				}
				seqs = bsx_fq_read_chunk(f1, f2, chunk, opt->has_bc, &n);
				if (seqs == 0 || n == 0) { free(seqs); break; }
				if (!copy_comment) for (i = 0; i < n; ++i) { free(seqs[i].comment); seqs[i].comment = 0; }
			}
			for (i = 0; i < n; ++i) size += seqs[i].l_seq;
			++chunk_idx;
			if (CHUNK_WORLD > 1 && chunk_idx % bsx_shard_world != bsx_shard_rank) { /* another rank's chunk */
				n_processed += n;
				for (i = 0; i < n; ++i) bsx_read_free(&seqs[i]);
				free(seqs);
				continue;
			}
			if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] read %d sequences (%ld bp)...\n", "process", n, (long)size);
			if (bsx_shard_mode && bsx_shard_world > 1) { /* within-chunk sharding without the stream (a backend without one): slice, align, emit */
				int a = 0, m = shard_slice(n, (opt->flag & BSX_F_PE) != 0, &a);
				const int n_all = n;
				if (opt->flag & BSX_F_SMARTPE) { fprintf(stderr, "[E::%s] -p cannot be combined with ranks sharing a chunk\n", "main_align"); rc = 1; break; }
				if (stream) { rc = bsx_stream_flush(stream); for (i = 0; i < n_pend; ++i) emit_chunk(pend[i].seqs, pend[i].n, pend[i].idx, rc == 0); n_pend = 0; }
				shard_trim(seqs, n, a, m);
				if (rc == 0 && m > 0) { bsx_chunk_slice_offset(a); rc = process(ud, opt, idx, n_processed + a, m, seqs, pes0); }
				else if (rc == 0 && (opt->flag & BSX_F_PE) && !pes0) bsx_pestat_sync_empty(opt);
				if (rc != BSX_OK) { fprintf(stderr, "[E::%s] alignment failed: %s\n", "main_align", bsx_strerror(rc)); rc = 1; }
				n_processed += n_all;
				emit_chunk(seqs, m, chunk_idx * bsx_shard_world + bsx_shard_rank, rc == 0);
				if (g_write_error) rc = 1;
				if (rc) break;
				continue;
			}
			if (opt->flag & BSX_F_SMARTPE) { /* -p: split into single and paired reads (align.c:108-146) */
				bsx_read_t *sep[2]; int m[2];
				bsx_opt_t tmp = *opt;
				for (i = 0; i < n; ++i) seqs[i].id = i;
				classify(n, seqs, m, sep);
				if (m[0]) { tmp.flag &= ~BSX_F_PE; rc = process(ud, &tmp, idx, n_processed, m[0], sep[0], 0); for (i = 0; i < m[0] && rc == 0; ++i) seqs[sep[0][i].id] = sep[0][i]; }
				if (m[1] && rc == 0) { tmp.flag |= BSX_F_PE; rc = process(ud, &tmp, idx, n_processed + m[0], m[1], sep[1], pes0); for (i = 0; i < m[1] && rc == 0; ++i) seqs[sep[1][i].id] = sep[1][i]; }
				free(sep[0]); free(sep[1]);
			} else if (stream) {
				rc = bsx_stream_push(stream, n_processed, n, seqs);
				pend[n_pend].seqs = seqs; pend[n_pend].n = n; pend[n_pend].idx = chunk_idx; ++n_pend;
				if (rc != BSX_OK) { fprintf(stderr, "[E::%s] alignment failed: %s\n", "main_align", bsx_strerror(rc)); rc = 1; }
				n_processed += n;
				while (rc == 0 && n_pend > depth - 1) { /* the push completed the oldest chunk in flight */
					emit_chunk(pend[0].seqs, pend[0].n, pend[0].idx, 1);
					for (i = 1; i < n_pend; ++i) pend[i - 1] = pend[i];
					--n_pend;
					if (g_write_error) rc = 1;
				}
				if (rc) break;
				continue;
			} else rc = process(ud, opt, idx, n_processed, n, seqs, pes0);
			if (rc != BSX_OK) { fprintf(stderr, "[E::%s] alignment failed: %s\n", "main_align", bsx_strerror(rc)); rc = 1; }
			n_processed += n;
			emit_chunk(seqs, n, chunk_idx, rc == 0);
			if (g_write_error) rc = 1;
			if (rc) break;
		}
		if (stream) {
			if (rc == 0 && (rc = bsx_stream_flush(stream)) != BSX_OK) { fprintf(stderr, "[E::%s] alignment failed: %s\n", "main_align", bsx_strerror(rc)); rc = 1; }
			if (rc) { bsx_stream_close(stream); stream = 0; }   /* as above: before the pending reads are freed */
			for (i = 0; i < n_pend; ++i) emit_chunk(pend[i].seqs, pend[i].n, pend[i].idx, rc == 0);
			if (stream) bsx_stream_close(stream);
		}
loop_done:
		bsx_fq_scan_close(scan);
	}
	if (bsx_fq_error(f1) || bsx_fq_error(f2)) { fprintf(stderr, "[E::%s] damaged or truncated compressed input: the SAM is incomplete\n", "main_align"); rc = 1; }
	if (fflush(stdout) != 0 || ferror(stdout)) g_write_error = 1;
	if (g_write_error) { fprintf(stderr, "[E::%s] failed to write the output: the SAM is incomplete\n", "main_align"); rc = 1; }
cleanup:
	if (open_device && ud && g_close_device) g_close_device(ud);   /* the device this call opened: index replica, lanes, streams */
	free(hdr_line); free(opt->adaptor1); free(opt->adaptor2); free(pes0); free(seq1); free(seq2);
	bsx_fq_close(f1); bsx_fq_close(f2);
	bsx_index_free(idx);
	return rc;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * Several GPUs without Python in the data path (north_star: "host code stays C ... a trivial RCCL gather"): the same command line started
 * once per GPU by any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun, mpirun with a wrapper, a shell loop).  Every process
 * aligns its share of the chunks (cli.c above: bsx_shard_rank / _world / _mode) on GPU $LOCAL_RANK and hands each chunk's SAM text to the
 * gather (gather.c): over RCCL in the product, over Unix sockets for the CPU checker.  The output: $BSX_OUT when set -- every rank writes its
 * own chunks into that file at their offsets when all ranks are on one node and it is a regular file, else the text goes through rank 0 --
 * or rank 0's stdout.  $BSX_GATHER_ID: the path through which the ranks find each other (the unique ids' file / the sockets' name); by
 * default next to the output, or in /tmp under the launcher's MASTER_PORT.  Settings: shard_pairs=1 (every rank takes its slice of EVERY
 * chunk: inputs with fewer chunks than GPUs), gather_via_rank0=1, gather_transport=rccl|socket.
 * biscuit_amd/multi_gpu.py (torch.distributed) drives the same sharding through the emit hook and remains available. */
typedef struct {
	bsx_gather_t *G; FILE *out; int direct, failed;
	bsx_transport_t tg, tr;
	int argc; char **argv; bsx_process_fn process; void *ud; int (*open_device)(int, const bsx_index_t*, void**); int rc;
} ranks_t;
static ranks_t *g_ranks = 0;
static void ranks_emit(void *ud, int64_t chunk, const char *text, size_t len)   /* the aligner's writer thread, chunk by chunk; -1: the header (rank 0) */
{
	ranks_t *R = (ranks_t*)ud;
	if (chunk < 0) {
		if (R->direct) bsx_gather_set_header(R->G, text, len);
		else if (R->out && len && fwrite(text, 1, len, R->out) != len) R->failed = 1;
		return;
	}
	{
		char *copy = (char*)malloc(len ? len : 1);
		memcpy(copy, text, len);
		if (bsx_gather_submit(R->G, chunk, copy, len) != BSX_OK) R->failed = 1;   /* blocks while a few chunks wait for their round */
	}
}
static void ranks_sink(void *ud, int64_t chunk, const void *buf, size_t n)      /* rank 0, the gather's thread: chunks in input order */
{
	ranks_t *R = (ranks_t*)ud;
	(void)chunk;
	if (R->out && !R->failed && n && fwrite(buf, 1, n, R->out) != n) { R->failed = 1; fprintf(stderr, "[E::%s] failed to write the output: the SAM is incomplete\n", "main_align"); }
}
static void ranks_hist(void *ud, int64_t *hist, int nb)
{
	ranks_t *R = (ranks_t*)ud;
	if (R->tr.all_reduce_sum && R->tr.all_reduce_sum(R->tr.ctx, hist, nb) != BSX_OK) { R->failed = 1; fprintf(stderr, "[E::%s] adding the insert-size histograms over the ranks failed\n", "main_align"); }
}
static void *ranks_aligner(void *arg)
{
	ranks_t *R = (ranks_t*)arg;
	R->rc = bsx_align_main_with(R->argc, R->argv, R->process, R->ud, R->open_device);
	bsx_gather_close_input(R->G);
	return 0;
}
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
extern void *bsx_pes_hist_ud;

BSX_API int bsx_align_main_ranks_with(int argc, char **argv, bsx_process_fn process, void *ud, int (*open_device)(int ordinal, const bsx_index_t *idx, void **ud), int use_rccl)
{
	const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local_rank = env_int("LOCAL_RANK", rank), local_world = env_int("LOCAL_WORLD_SIZE", world);
	const char *out_path = getenv("BSX_OUT"), *id_env = getenv("BSX_GATHER_ID"), *tk = bsx_tune_str("gather_transport");
	const int pairs = bsx_tune_long("shard_pairs", 0) != 0, want_rccl = tk ? strcmp(tk, "socket") != 0 : use_rccl;
	char id_path[4096], dev_env[32];
	ranks_t R;
	pthread_t th;
	int rc, direct;
	int64_t n_chunks = 0, mine, *all;
	if (world <= 1 || bsx_emit_hook) return bsx_align_main_with(argc, argv, process, ud, open_device);   /* (a launcher with its own hook: multi_gpu.py) */
	if (rank < 0 || rank >= world) { fprintf(stderr, "[E::%s] RANK %d of WORLD_SIZE %d\n", "main_align", rank, world); return 1; }
	memset(&R, 0, sizeof(R));
	if (id_env && *id_env) snprintf(id_path, sizeof(id_path), "%s", id_env);
	else if (out_path && *out_path) snprintf(id_path, sizeof(id_path), "%s.ranks", out_path);
	else snprintf(id_path, sizeof(id_path), "/tmp/bsx_ranks_%s_%d", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", (int)getppid());
	if (!getenv("BSX_DEVICE")) { snprintf(dev_env, sizeof(dev_env), "%d", local_rank); setenv("BSX_DEVICE", dev_env, 1); }
	rc = want_rccl ? bsx_transport_rccl(rank, world, local_rank, id_path, &R.tg, pairs ? &R.tr : 0) : bsx_transport_socket(rank, world, id_path, &R.tg, pairs ? &R.tr : 0);
	if (rc != BSX_OK) { fprintf(stderr, "[E::%s] rank %d: no connection to the other ranks (%s)\n", "main_align", rank, bsx_strerror(rc)); return 1; }
	direct = out_path && *out_path && !bsx_tune_long("gather_via_rank0", 0) && bsx_gather_direct_ok(out_path, world, local_world);
	R.direct = direct;
	if (!direct && rank == 0) {
		R.out = out_path && *out_path ? fopen(out_path, "wb") : stdout;
		if (!R.out) { fprintf(stderr, "[E::%s] cannot open %s\n", "main_align", out_path); R.failed = 1; }
	}
	if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] rank %d of %d on device %d: %s over %s, %s\n", "main_align", rank, world, local_rank, pairs ? "a slice of every chunk" : "every world-th chunk",
	                              want_rccl ? "RCCL" : "sockets", direct ? "every rank writes its own chunks" : "records through rank 0");
	if ((rc = bsx_gather_open(&R.tg, direct ? out_path : 0, ranks_sink, &R, 3, &R.G)) != BSX_OK) { R.tg.close(R.tg.ctx); if (R.tr.close) R.tr.close(R.tr.ctx); return 1; }
	R.argc = argc; R.argv = argv; R.process = process; R.ud = ud; R.open_device = open_device; R.rc = 1;
	bsx_shard_rank = rank; bsx_shard_world = world; bsx_shard_mode = pairs;
	bsx_emit_hook = ranks_emit; bsx_emit_ud = &R;
	if (pairs) { bsx_pes_hist_hook = ranks_hist; bsx_pes_hist_ud = &R; }
	g_ranks = &R;
	if (pthread_create(&th, 0, ranks_aligner, &R) != 0) { (void)ranks_aligner(&R); }
	rc = bsx_gather_run(R.G, &n_chunks);
	pthread_join(th, 0);
	bsx_emit_hook = 0; bsx_emit_ud = 0; bsx_pes_hist_hook = 0; bsx_pes_hist_ud = 0; bsx_shard_rank = 0; bsx_shard_world = 1; bsx_shard_mode = 0; g_ranks = 0;
	if (R.out && R.out != stdout && fclose(R.out) != 0) R.failed = 1;
	else if (R.out == stdout && fflush(stdout) != 0) R.failed = 1;
	/* every rank leaves with the worst status of all */
	mine = (R.rc != 0 || rc != BSX_OK || R.failed) ? 1 : 0;
	all = (int64_t*)calloc((size_t)world, sizeof(int64_t));
	if (R.tg.all_gather(R.tg.ctx, &mine, 1, all) == BSX_OK) { int k; for (k = 0; k < world; ++k) if (all[k]) mine = 1; }
	else mine = 1;
	free(all);
	bsx_gather_free(R.G);
	R.tg.close(R.tg.ctx);
	if (R.tr.close) R.tr.close(R.tr.ctx);
	return (int)mine;
}

/* ---- product entry: HIP device only ---- */
static int hip_open(int ordinal, const bsx_index_t *idx, void **ud)
{
	bsx_device_t *dev = 0;
	int rc = bsx_device_open(ordinal, &dev);
	if (rc != BSX_OK) return rc;
	if ((rc = bsx_device_upload_index(dev, idx)) != BSX_OK) { bsx_device_close(dev); return rc; }
	if (bsx_verbose >= 3) fprintf(stderr, "[M::%s] index resident on %s\n", "main_align", bsx_device_name(dev));
	*ud = dev;
	return BSX_OK;
}
static int hip_process(void *ud, const bsx_opt_t *opt, const bsx_index_t *idx, int64_t np, int n, bsx_read_t *reads, const bsx_pestat_t *pes0)
{
	return bsx_process_seqs((bsx_device_t*)ud, opt, idx, np, n, reads, pes0);
}

static void hip_close(void *ud) { bsx_device_close((bsx_device_t*)ud); }

BSX_API int bsx_align_main(int argc, char **argv)
{
	int rc;
	g_use_stream = getenv("BSX_NO_STREAM") ? 0 : 1;
	g_close_device = hip_close;
	rc = bsx_align_main_ranks_with(argc, argv, hip_process, 0, hip_open, 1);   /* (one process: straight to bsx_align_main_with) */
	g_close_device = 0; g_use_stream = 0;
	return rc;
}
