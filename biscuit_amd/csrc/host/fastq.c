/* fastq.c -- I1: FASTA/FASTQ reader and chunker.
 * Record grammar follows klib's kseq_read (lib/aln/kseq.h:182-222): header '>' or '@', name up to
 * the first white space, optional comment = rest of the line, multi-line sequence until a line that
 * starts with '>', '@' or '+', then (FASTQ) quality lines until as many characters as bases.
 * Chunking follows bis_bseq_read (lib/aln/bwa.c:817-850): stop at the first even read count once
 * the chunk holds >= chunk_size bases.
 */
#include <zlib.h>
#include <ctype.h>
#include <pthread.h>
#include <sys/stat.h>
#include "bsx_core.h"
#include "fastq.h"

const uint8_t *bsx_nt4_table(void);

/* ---- where a reader gets its bytes.  A plain regular file is read in place (offsets into it mean something: the chunk scan and
 * bsx_fq_seek).  Compressed or piped input is inflated AHEAD of the parser by threads of its own (with the aligner on
 * the GPU, inflating and parsing the FASTQ is what the command line waits for):
 *   - a BGZF file (gzip members of <= 64 KB with a 'BC' extra field, what bgzip and most sequencers' tools write) is inflated block by
 *     block by several workers, each block on its own (raw deflate, the member's size is in its header);
 *   - any other gzip stream (one member, or members of unknown size) has one thread that runs zlib's gzread into a ring of buffers,
 *     so that inflating overlaps parsing.
 * $BSX_INFLATE_THREADS: workers per BGZF file (default 3); 0 = no threads at all, every read a plain gzread as before. */
#define SRC_SLOT (1 << 20)
#define SRC_RING 24
typedef struct { unsigned char *data; int n, cap; unsigned char *comp; int ncomp; int state; } src_slot_t;   /* state: 0 free, 1 compressed block loaded, 2 being inflated, 3 ready */
typedef struct fq_src {
	int mode;                 /* 1: gzread thread, 2: BGZF workers */
	gzFile gz; FILE *fp;
	src_slot_t ring[SRC_RING];
	int64_t head, tail, next_job;   /* consumer's slot, producer's next slot, next compressed slot a worker takes */
	int eof, stop, failed, reported, head_pos;
	pthread_mutex_t mu; pthread_cond_t cv;
	pthread_t th_read, th_work[8]; int n_work;
} fq_src_t;

struct bsx_fq {
	gzFile fp;
	fq_src_t *src;
	unsigned char *buf;
	int begin, end, is_eof, last_char, failed;
	int64_t base;        /* file offset of buf[0] (plain files: the chunk scan and bsx_fq_seek) */
	int64_t last_off;    /* where last_char was read */
	BSX_VEC(char) name, comment, seq, qual;
};

#define FQ_BUFSZ (1 << 18)

static void *src_gz_main(void *arg)   /* mode 1: the one thread that inflates */
{
	fq_src_t *S = (fq_src_t*)arg;
	for (;;) {
		src_slot_t *sl;
		int n;
		pthread_mutex_lock(&S->mu);
		while (S->tail - S->head >= SRC_RING && !S->stop) pthread_cond_wait(&S->cv, &S->mu);
		if (S->stop) { pthread_mutex_unlock(&S->mu); return 0; }
		sl = &S->ring[S->tail % SRC_RING];
		pthread_mutex_unlock(&S->mu);
		if (!sl->data) { sl->data = (unsigned char*)malloc(SRC_SLOT); sl->cap = SRC_SLOT; }
		n = gzread(S->gz, sl->data, SRC_SLOT);
		pthread_mutex_lock(&S->mu);
		if (n <= 0) { /* the end, or damage: a truncated stream reads as 0 bytes with Z_BUF_ERROR set */
			int errnum = Z_OK;
			(void)gzerror(S->gz, &errnum);
			S->eof = 1; if (n < 0 || errnum < 0) S->failed = 1;
			pthread_cond_broadcast(&S->cv); pthread_mutex_unlock(&S->mu); return 0;
		}
		sl->n = n; sl->state = 3; ++S->tail;
		pthread_cond_broadcast(&S->cv);
		pthread_mutex_unlock(&S->mu);
	}
}
/* a BGZF block header at the file's current position: its total size, 0 at a clean end of file, -1 if it is not one */
static int bgzf_block_size(FILE *fp, unsigned char hdr[18])
{
	size_t k = fread(hdr, 1, 18, fp);
	if (k == 0) return 0;
	if (k < 18 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4) || hdr[10] != 6 || hdr[11] != 0 || hdr[12] != 'B' || hdr[13] != 'C' || hdr[14] != 2 || hdr[15] != 0) return -1;
	return (hdr[16] | hdr[17] << 8) + 1;
}
static void *src_bgzf_read_main(void *arg)   /* mode 2: loads the compressed blocks, in order */
{
	fq_src_t *S = (fq_src_t*)arg;
	for (;;) {
		src_slot_t *sl;
		unsigned char hdr[18];
		int bs;
		pthread_mutex_lock(&S->mu);
		while (S->tail - S->head >= SRC_RING && !S->stop) pthread_cond_wait(&S->cv, &S->mu);
		if (S->stop) { pthread_mutex_unlock(&S->mu); return 0; }
		sl = &S->ring[S->tail % SRC_RING];
		pthread_mutex_unlock(&S->mu);
		bs = bgzf_block_size(S->fp, hdr);
		if (bs > 0) {
			if (!sl->comp) sl->comp = (unsigned char*)malloc(65536 + 64);
			if (bs < 26 || bs > 65536 || fread(sl->comp, 1, (size_t)bs - 18, S->fp) != (size_t)bs - 18) bs = -1;
			else sl->ncomp = bs - 18 - 8;   /* the deflate stream; behind it CRC32 and ISIZE */
		}
		pthread_mutex_lock(&S->mu);
		if (bs <= 0) { S->eof = 1; if (bs < 0) S->failed = 1; pthread_cond_broadcast(&S->cv); pthread_mutex_unlock(&S->mu); return 0; }
		sl->state = 1; ++S->tail;
		pthread_cond_broadcast(&S->cv);
		pthread_mutex_unlock(&S->mu);
	}
}
static void *src_bgzf_work_main(void *arg)   /* mode 2: a worker inflates the next loaded block */
{
	fq_src_t *S = (fq_src_t*)arg;
	z_stream z;
	memset(&z, 0, sizeof(z));
	if (inflateInit2(&z, -15) != Z_OK) return 0;
	for (;;) {
		src_slot_t *sl;
		int ok;
		pthread_mutex_lock(&S->mu);
		while (!(S->next_job < S->tail) && !S->stop && !S->eof) pthread_cond_wait(&S->cv, &S->mu);
		if (S->stop || !(S->next_job < S->tail)) { pthread_mutex_unlock(&S->mu); break; }
		sl = &S->ring[S->next_job % SRC_RING];
		++S->next_job;
		sl->state = 2;
		pthread_mutex_unlock(&S->mu);
		if (!sl->data) { sl->data = (unsigned char*)malloc(65536); sl->cap = 65536; }
		inflateReset(&z);
		z.next_in = sl->comp; z.avail_in = (unsigned)sl->ncomp; z.next_out = sl->data; z.avail_out = 65536;
		ok = inflate(&z, Z_FINISH) == Z_STREAM_END;
		if (ok) { /* the member's trailer: CRC32 and ISIZE of what it inflates to (RFC 1952) */
			const unsigned char *t = sl->comp + sl->ncomp;
			const unsigned long crc = (unsigned long)t[0] | (unsigned long)t[1] << 8 | (unsigned long)t[2] << 16 | (unsigned long)t[3] << 24;
			const unsigned long isz = (unsigned long)t[4] | (unsigned long)t[5] << 8 | (unsigned long)t[6] << 16 | (unsigned long)t[7] << 24;
			ok = isz == z.total_out && crc == crc32(crc32(0L, Z_NULL, 0), sl->data, (uInt)z.total_out);
		}
		pthread_mutex_lock(&S->mu);
		sl->n = ok ? (int)z.total_out : -1;   /* -1: the stream ends here (src_read), whatever lies behind the damage */
		if (!ok) S->failed = 1;
		sl->state = 3;
		pthread_cond_broadcast(&S->cv);
		pthread_mutex_unlock(&S->mu);
	}
	inflateEnd(&z);
	return 0;
}
static void src_report(fq_src_t *S)
{
	if (S->failed && !S->reported) { S->reported = 1; fprintf(stderr, "[E::%s] the compressed input is damaged or truncated: the reads behind the damage are not aligned\n", "fastq"); }
}
/* the next bytes of the stream, up to cap of them; 0 at its end (or at the first damaged block: bsx_fq_error tells the two apart) */
static int src_read(fq_src_t *S, unsigned char *dst, int cap)
{
	int got = 0;
	while (got == 0) {
		src_slot_t *sl;
		int k;
		pthread_mutex_lock(&S->mu);
		for (;;) {
			sl = &S->ring[S->head % SRC_RING];
			if (S->head < S->tail && sl->state == 3) break;
			if (S->head >= S->tail && S->eof) { pthread_mutex_unlock(&S->mu); src_report(S); return 0; }
			pthread_cond_wait(&S->cv, &S->mu);
		}
		pthread_mutex_unlock(&S->mu);
		if (sl->n < 0) { src_report(S); return 0; }   /* a damaged block: nothing behind it is handed on (the slot is never released: every later call ends here) */
		k = sl->n - S->head_pos; k = k < cap ? k : cap;
		if (k > 0) { memcpy(dst, sl->data + S->head_pos, (size_t)k); S->head_pos += k; got = k; }
		if (S->head_pos >= sl->n) {
			pthread_mutex_lock(&S->mu);
			sl->state = 0; S->head_pos = 0; ++S->head;
			pthread_cond_broadcast(&S->cv);
			pthread_mutex_unlock(&S->mu);
		}
	}
	return got;
}
/* What kind of input a name is, decided WITHOUT reading from it unless it is a regular file: a FIFO or a process substitution
 * (`align ref <(zcat r1.gz) <(zcat r2.gz)`) gives every byte once, so it is opened exactly once, by the reader that keeps it (zlib's gzopen
 * tells gzip from plain text itself).  0: plain regular file, 1: regular gzip file, 2: regular BGZF file, 3: anything else (pipes, devices,
 * standard input), -1: cannot be opened. */
static int src_kind(const char *fn)
{
	struct stat st;
	FILE *fp;
	unsigned char hdr[18];
	int kind = 0;
	if (strcmp(fn, "-") == 0) return 3;
	if (stat(fn, &st) != 0) return -1;
	if (!S_ISREG(st.st_mode)) return 3;
	if ((fp = fopen(fn, "rb")) == 0) return -1;
	if (fread(hdr, 1, 2, fp) == 2 && hdr[0] == 0x1f && hdr[1] == 0x8b) { rewind(fp); kind = bgzf_block_size(fp, hdr) > 0 ? 2 : 1; }
	fclose(fp);
	return kind;
}
static fq_src_t *src_open(const char *fn)
{
	const char *e = getenv("BSX_INFLATE_THREADS");
	int nw = e ? atoi(e) : 3, i;
	const int kind = src_kind(fn);
	fq_src_t *S;
	if (nw <= 0 || kind <= 0) return 0;   /* a plain regular file is read in place (and a name that cannot be opened fails in bsx_fq_open) */
	S = (fq_src_t*)calloc(1, sizeof(*S));
	pthread_mutex_init(&S->mu, 0); pthread_cond_init(&S->cv, 0);
	if (kind == 2) {
		S->mode = 2;
		if ((S->fp = fopen(fn, "rb")) == 0) { free(S); return 0; }
		S->n_work = nw < 8 ? nw : 8;
		pthread_create(&S->th_read, 0, src_bgzf_read_main, S);
		for (i = 0; i < S->n_work; ++i) pthread_create(&S->th_work[i], 0, src_bgzf_work_main, S);
	} else {
		S->mode = 1;
		S->gz = strcmp(fn, "-") == 0 ? gzdopen(0, "r") : gzopen(fn, "r");
		if (!S->gz) { free(S); return 0; }
		gzbuffer(S->gz, 1 << 20);
		pthread_create(&S->th_read, 0, src_gz_main, S);
	}
	return S;
}
static void src_close(fq_src_t *S)
{
	int i;
	if (!S) return;
	pthread_mutex_lock(&S->mu); S->stop = 1; pthread_cond_broadcast(&S->cv); pthread_mutex_unlock(&S->mu);
	pthread_join(S->th_read, 0);
	for (i = 0; i < S->n_work; ++i) pthread_join(S->th_work[i], 0);
	for (i = 0; i < SRC_RING; ++i) { free(S->ring[i].data); free(S->ring[i].comp); }
	if (S->gz) gzclose(S->gz);
	if (S->fp) fclose(S->fp);
	free(S);
}

bsx_fq_t *bsx_fq_open(const char *fn)
{
	bsx_fq_t *f;
	fq_src_t *src = src_open(fn);
	gzFile fp = 0;
	if (!src) {
		fp = strcmp(fn, "-") == 0 ? gzdopen(0, "r") : gzopen(fn, "r");
		if (!fp) return 0;
		gzbuffer(fp, 1 << 20);
	}
	f = (bsx_fq_t*)calloc(1, sizeof(*f));
	f->fp = fp; f->src = src;
	f->buf = (unsigned char*)malloc(FQ_BUFSZ);
	return f;
}

void bsx_fq_close(bsx_fq_t *f)
{
	if (!f) return;
	if (f->fp) gzclose(f->fp);
	src_close(f->src);
	free(f->buf); bsx_vec_free(f->name); bsx_vec_free(f->comment); bsx_vec_free(f->seq); bsx_vec_free(f->qual);
	free(f);
}

static inline int fq_more(bsx_fq_t *f)
{
	int n, errnum = Z_OK;
	if (f->src) return src_read(f->src, f->buf, FQ_BUFSZ);
	n = gzread(f->fp, f->buf, FQ_BUFSZ);
	if (n <= 0) { (void)gzerror(f->fp, &errnum); if ((n < 0 || errnum < 0) && !f->failed) { f->failed = 1; fprintf(stderr, "[E::%s] the input is damaged or truncated: the reads behind the damage are not aligned\n", "fastq"); } }
	return n;
}
/* non-zero once the reader has met damaged or truncated compressed input: what it returned as "end of file" was not one */
int bsx_fq_error(const bsx_fq_t *f) { return f ? (f->failed || (f->src && f->src->failed)) : 0; }
static inline int fq_getc(bsx_fq_t *f)
{
	if (f->is_eof && f->begin >= f->end) return -1;
	if (f->begin >= f->end) {
		f->base += f->end;
		f->begin = 0;
		f->end = fq_more(f);
		if (f->end <= 0) { f->is_eof = 1; f->end = 0; return -1; }
	}
	return (int)f->buf[f->begin++];
}

#define vec_putc(v, c) do { if ((v).n + 1 >= (v).m) bsx_vec_reserve(v, (v).n + 2); (v).a[(v).n++] = (char)(c); } while (0)
/* refill the buffer if it is exhausted; 0 at end of file */
static inline int fq_fill(bsx_fq_t *f)
{
	if (f->begin < f->end) return 1;
	if (f->is_eof) return 0;
	f->base += f->end;
	f->begin = 0;
	f->end = fq_more(f);
	if (f->end <= 0) { f->is_eof = 1; f->end = 0; return 0; }
	return 1;
}

/* read up to a delimiter: 0 = any white space, 1 = end of line; returns the delimiter or -1.  Whole runs are taken
 * from the buffer at once (memchr / a tight scan) instead of a call per character. */
static int fq_until(bsx_fq_t *f, int line_only, void *v_, int append)
{
	BSX_VEC(char) *v = v_;
	int c = -1, got = 0;
	if (!append) v->n = 0;
	for (;;) {
		const unsigned char *p, *q, *e;
		size_t len;
		if (!fq_fill(f)) { c = -1; break; }
		got = 1;
		p = f->buf + f->begin; e = f->buf + f->end;
		if (line_only) q = (const unsigned char*)memchr(p, '\n', (size_t)(e - p));
		else { for (q = p; q < e && !isspace(*q); ++q); if (q == e) q = 0; }
		len = (size_t)((q ? q : e) - p);
		if (len) { bsx_vec_reserve(*v, v->n + len + 2); memcpy(v->a + v->n, p, len); v->n += len; }
		f->begin += (int)len + (q ? 1 : 0);
		if (q) { c = *q; break; }
	}
	if (!got && c < 0) return -1;
	if (line_only && v->n > 0 && v->a[v->n - 1] == '\r') --v->n;
	bsx_vec_reserve(*v, v->n + 1);
	v->a[v->n] = 0;
	return c; /* -1 at EOF */
}

/* skip the rest of the current line; -1 if the file ends first */
static int fq_skip_line(bsx_fq_t *f)
{
	for (;;) {
		const unsigned char *p, *q;
		if (!fq_fill(f)) return -1;
		p = f->buf + f->begin;
		q = (const unsigned char*)memchr(p, '\n', (size_t)(f->end - f->begin));
		if (q) { f->begin += (int)(q - p) + 1; return '\n'; }
		f->begin = f->end;
	}
}

/* returns sequence length, -1 at end of file, -2 on a truncated quality string */
static int fq_read(bsx_fq_t *f)
{
	int c;
	if (f->last_char == 0) {
		while ((c = fq_getc(f)) != -1 && c != '>' && c != '@');
		if (c == -1) return -1;
		f->last_char = c; f->last_off = f->base + f->begin - 1;
	}
	f->comment.n = f->seq.n = f->qual.n = 0;
	if ((c = fq_until(f, 0, &f->name, 0)) < 0 && f->name.n == 0) return -1;
	if (c != '\n' && c >= 0) fq_until(f, 1, &f->comment, 0);
	bsx_vec_reserve(f->seq, 256);
	while ((c = fq_getc(f)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		vec_putc(f->seq, c);
		fq_until(f, 1, &f->seq, 1);
	}
	if (c == '>' || c == '@') { f->last_char = c; f->last_off = f->base + f->begin - 1; }
	bsx_vec_reserve(f->seq, f->seq.n + 1);
	f->seq.a[f->seq.n] = 0;
	if (c != '+') { if (c == -1) f->last_char = 0; return (int)f->seq.n; }
	bsx_vec_reserve(f->qual, f->seq.n + 2);
	if (fq_skip_line(f) < 0) return -2;   /* rest of the '+' line */
	while (f->qual.n < f->seq.n) { if (fq_until(f, 1, &f->qual, 1) < 0) break; }
	f->last_char = 0;
	if (f->seq.n != f->qual.n) return -2;
	return (int)f->seq.n;
}

static void to_read(const bsx_fq_t *f, bsx_read_t *s, int has_bc)
{
	const uint8_t *nt4 = bsx_nt4_table();
	size_t i, l = f->name.n;
	memset(s, 0, sizeof(*s));
	/* trim_readno, bwa.c:58-63 */
	s->name = (char*)malloc(l + 1); memcpy(s->name, f->name.a, l); s->name[l] = 0;
	if (l > 2 && s->name[l - 2] == '/' && isdigit((unsigned char)s->name[l - 1])) s->name[l - 2] = 0;
	s->comment = f->comment.n ? strdup(f->comment.a) : 0;
	if (has_bc) { /* name_..._BARCODE_UMI (bis_kseq2bseq1, bwa.c:766-815): the last two '_' fields */
		char *tmp = strdup(s->name), *tok, *bc = 0, *umi = 0, *sv = 0;   /* (strtok_r: the two input files are parsed by two threads) */
		tok = strtok_r(tmp, "_", &sv);
		bc = strtok_r(NULL, "_", &sv); umi = strtok_r(NULL, "_", &sv);
		while ((tok = strtok_r(NULL, "_", &sv)) != NULL) { bc = umi; umi = tok; }
		s->barcode = bc ? strdup(bc) : 0; s->umi = umi ? strdup(umi) : 0;
		free(tmp);
	}
	s->l_seq = s->l_seq0 = (int)f->seq.n;
	s->seq = s->seq0 = (uint8_t*)malloc(f->seq.n + 1);
	for (i = 0; i < f->seq.n; ++i) s->seq[i] = nt4[(unsigned char)f->seq.a[i]];
	s->qual = f->qual.n ? strdup(f->qual.a) : 0;
}

bsx_read_t *bsx_fq_read_chunk(bsx_fq_t *f1, bsx_fq_t *f2, int chunk_size, int has_bc, int *n_)
{
	int size = 0, m = 0, n = 0;
	bsx_read_t *seqs = 0;
	while (fq_read(f1) >= 0) {
		if (f2 && fq_read(f2) < 0) { fprintf(stderr, "[W::%s] the 2nd file has fewer sequences.\n", __func__); break; }
		if (n + 2 > m) { m = m ? m << 1 : 256; seqs = (bsx_read_t*)realloc(seqs, (size_t)m * sizeof(bsx_read_t)); }
		to_read(f1, &seqs[n], has_bc); seqs[n].id = n; size += seqs[n++].l_seq;
		if (f2) { to_read(f2, &seqs[n], has_bc); seqs[n].id = n; size += seqs[n++].l_seq; }
		if (size >= chunk_size && (n & 1) == 0) break;
	}
	if (size == 0 && f2 && fq_read(f2) >= 0) fprintf(stderr, "[W::%s] the 1st file has fewer sequences.\n", __func__);
	*n_ = n;
	return seqs;
}

/* ---- the same chunks with one parser thread per file: each turns its file into blocks of records ahead of the
 * consumer, which pairs them up under the chunk rule above.  Inflating and parsing two gzipped files is the slowest host
 * step of the command line once the aligner is fast; this halves it. */
#define FEED_BLOCK 2048
#define FEED_RING 8
typedef struct { bsx_read_t *r; int n, eof; } feed_block_t;
typedef struct {
	bsx_fq_t *f; int has_bc;
	long limit, taken;      /* limit >= 0: the thread parses this many records and reports the end of its input (a chunk whose size is known) */
	pthread_t th;
	pthread_mutex_t mu; pthread_cond_t cv;
	feed_block_t ring[FEED_RING]; int head, count, stop;
	feed_block_t cur; int cur_i, done;
} feed_t;
struct bsx_fq_pair { feed_t a, b; int has_b; };

static void *feed_main(void *arg)
{
	feed_t *F = (feed_t*)arg;
	for (;;) {
		feed_block_t blk;
		blk.r = (bsx_read_t*)malloc(sizeof(bsx_read_t) * FEED_BLOCK); blk.n = 0; blk.eof = 0;
		while (blk.n < FEED_BLOCK) {
			if (F->limit >= 0 && F->taken >= F->limit) { blk.eof = 1; break; }
			if (fq_read(F->f) < 0) { blk.eof = 1; break; }
			to_read(F->f, &blk.r[blk.n++], F->has_bc);
			++F->taken;
		}
		pthread_mutex_lock(&F->mu);
		while (F->count == FEED_RING && !F->stop) pthread_cond_wait(&F->cv, &F->mu);
		if (F->stop) { int i; pthread_mutex_unlock(&F->mu); for (i = 0; i < blk.n; ++i) bsx_read_free(&blk.r[i]); free(blk.r); return 0; }
		F->ring[(F->head + F->count++) % FEED_RING] = blk;
		pthread_cond_broadcast(&F->cv);
		pthread_mutex_unlock(&F->mu);
		if (blk.eof) return 0;
	}
}
static void feed_start(feed_t *F, bsx_fq_t *f, int has_bc, long limit)
{
	memset(F, 0, sizeof(*F));
	F->f = f; F->has_bc = has_bc; F->limit = limit;
	pthread_mutex_init(&F->mu, 0); pthread_cond_init(&F->cv, 0);
	pthread_create(&F->th, 0, feed_main, F);
}
/* next record of the file, 0 at its end */
static int feed_next(feed_t *F, bsx_read_t *out)
{
	for (;;) {
		if (F->cur_i < F->cur.n) { *out = F->cur.r[F->cur_i++]; return 1; }
		free(F->cur.r); F->cur.r = 0; F->cur.n = F->cur_i = 0;
		if (F->done) return 0;
		pthread_mutex_lock(&F->mu);
		while (F->count == 0) pthread_cond_wait(&F->cv, &F->mu);
		F->cur = F->ring[F->head]; F->head = (F->head + 1) % FEED_RING; --F->count;
		pthread_cond_broadcast(&F->cv);
		pthread_mutex_unlock(&F->mu);
		if (F->cur.eof) F->done = 1;
	}
}
static void feed_stop(feed_t *F)
{
	int i;
	pthread_mutex_lock(&F->mu); F->stop = 1; pthread_cond_broadcast(&F->cv); pthread_mutex_unlock(&F->mu);
	pthread_join(F->th, 0);
	for (; F->cur_i < F->cur.n; ++F->cur_i) bsx_read_free(&F->cur.r[F->cur_i]);
	free(F->cur.r);
	while (F->count) { feed_block_t *b = &F->ring[F->head]; for (i = 0; i < b->n; ++i) bsx_read_free(&b->r[i]); free(b->r); F->head = (F->head + 1) % FEED_RING; --F->count; }
}

/* n_records >= 0: each file's parser stops after that many records (the rest of the file belongs to other chunks, other ranks) */
bsx_fq_pair_t *bsx_fq_pair_open_n(bsx_fq_t *f1, bsx_fq_t *f2, int has_bc, long n_records)
{
	bsx_fq_pair_t *P = (bsx_fq_pair_t*)calloc(1, sizeof(*P));
	feed_start(&P->a, f1, has_bc, n_records);
	if (f2) { feed_start(&P->b, f2, has_bc, n_records); P->has_b = 1; }
	return P;
}
bsx_fq_pair_t *bsx_fq_pair_open(bsx_fq_t *f1, bsx_fq_t *f2, int has_bc) { return bsx_fq_pair_open_n(f1, f2, has_bc, -1); }
void bsx_fq_pair_close(bsx_fq_pair_t *P)
{
	if (!P) return;
	feed_stop(&P->a);
	if (P->has_b) feed_stop(&P->b);
	free(P);
}
bsx_read_t *bsx_fq_pair_read_chunk(bsx_fq_pair_t *P, int chunk_size, int *n_)
{
	int size = 0, m = 0, n = 0;
	bsx_read_t *seqs = 0, ra, rb;
	while (feed_next(&P->a, &ra)) {
		if (P->has_b && !feed_next(&P->b, &rb)) { fprintf(stderr, "[W::%s] the 2nd file has fewer sequences.\n", "bsx_fq_read_chunk"); bsx_read_free(&ra); break; }
		if (n + 2 > m) { m = m ? m << 1 : 256; seqs = (bsx_read_t*)realloc(seqs, (size_t)m * sizeof(bsx_read_t)); }
		seqs[n] = ra; seqs[n].id = n; size += seqs[n++].l_seq;
		if (P->has_b) { seqs[n] = rb; seqs[n].id = n; size += seqs[n++].l_seq; }
		if (size >= chunk_size && (n & 1) == 0) break;
	}
	if (size == 0 && P->has_b && feed_next(&P->b, &rb)) { fprintf(stderr, "[W::%s] the 1st file has fewer sequences.\n", "bsx_fq_read_chunk"); bsx_read_free(&rb); }
	*n_ = n;
	return seqs;
}

/* ---- chunk boundaries by a light scan.  With several GPUs every rank aligns chunks r, r+N, ...: the chunk rule is cumulative, so a rank
 * has to know where its chunks start, but it does not have to build the records of the others.  Over plain (seekable, uncompressed)
 * files a scanner thread walks the record grammar above counting bases only -- no copies, no allocations: a few memchr calls per
 * record -- and lists each chunk's byte offsets in the files and its read count; the rank's parser seeks to its own chunks. */
/* length of the rest of the current line as fq_until(line_only) would append it; *last: its last character (or `first` if the rest is
 * empty); returns the delimiter or -1 at end of file (*got: something was there) */
static int scan_line(bsx_fq_t *f, int first, int64_t *len, int *got_)
{
	int64_t n = 0; int last = first, c = -1, got = 0;
	for (;;) {
		const unsigned char *p, *q;
		size_t l;
		if (!fq_fill(f)) { c = -1; break; }
		got = 1;
		p = f->buf + f->begin;
		q = (const unsigned char*)memchr(p, '\n', (size_t)(f->end - f->begin));
		l = (size_t)((q ? q : f->buf + f->end) - p);
		if (l) { last = p[l - 1]; n += (int64_t)l; }
		f->begin += (int)l + (q ? 1 : 0);
		if (q) { c = '\n'; break; }
	}
	if (got_) *got_ = got;
	if (!got && c < 0) { *len = first >= 0 ? 1 : 0; return -1; }   /* (fq_until leaves the vector as it was) */
	if (first >= 0) ++n;               /* the character the caller had already taken */
	if (n > 0 && last == '\r') --n;    /* fq_until strips one trailing CR per call */
	*len = n;
	return c;
}
/* fq_read without building the record: sequence length, -1 at end of file, -2 truncated; *off = where the record's header starts */
static int fq_scan(bsx_fq_t *f, int64_t *off)
{
	int c, got;
	int64_t slen = 0, qlen = 0, l;
	if (f->last_char == 0) {
		while ((c = fq_getc(f)) != -1 && c != '>' && c != '@');
		if (c == -1) return -1;
		f->last_char = c; f->last_off = f->base + f->begin - 1;
	}
	*off = f->last_off;
	{ /* name: up to the first white space */
		int64_t nl = 0; int gotn = 0;
		c = -1;
		for (;;) {
			const unsigned char *p, *q, *e;
			if (!fq_fill(f)) { c = -1; break; }
			gotn = 1;
			p = f->buf + f->begin; e = f->buf + f->end;
			for (q = p; q < e && !isspace(*q); ++q);
			nl += q - p;
			if (q < e) { c = *q; f->begin += (int)(q - p) + 1; break; }
			f->begin = f->end;
		}
		if (c < 0 && (nl == 0 || !gotn)) return -1;
	}
	if (c != '\n' && c >= 0) fq_skip_line(f);
	while ((c = fq_getc(f)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		scan_line(f, c, &l, 0);
		slen += l;
	}
	if (c == '>' || c == '@') { f->last_char = c; f->last_off = f->base + f->begin - 1; }
	if (c != '+') { if (c == -1) f->last_char = 0; return (int)slen; }
	if (fq_skip_line(f) < 0) return -2;
	while (qlen < slen) { if (scan_line(f, -1, &l, &got) < 0) { qlen += l; break; } qlen += l; }
	f->last_char = 0;
	if (qlen != slen) return -2;
	return (int)slen;
}
static int64_t fq_next_off(const bsx_fq_t *f) { return f->last_char ? f->last_off : f->base + f->begin; }

int bsx_fq_plain_file(const char *fn)   /* a regular, uncompressed file: offsets into it mean something */
{
	return src_kind(fn) == 0;   /* (never opens a pipe: its bytes come once, and closing a FIFO's read end can kill its writer) */
}
int bsx_fq_seek(bsx_fq_t *f, int64_t off)
{
	if (f->src) return -1;   /* (only plain files are sought in, and those have no inflate threads) */
	if (gzseek(f->fp, (z_off_t)off, SEEK_SET) < 0) return -1;
	f->begin = f->end = 0; f->is_eof = 0; f->last_char = 0; f->base = off; f->last_off = 0;
	return 0;
}

struct bsx_fq_scan {
	bsx_fq_t *f1, *f2;
	int chunk_size;
	pthread_t th; pthread_mutex_t mu; pthread_cond_t cv;
	BSX_VEC(bsx_fq_chunkpos_t) tab;
	int done, stop;
};
static void *scan_main(void *arg)
{
	bsx_fq_scan_t *S = (bsx_fq_scan_t*)arg;
	int64_t n_before = 0;
	for (;;) {
		bsx_fq_chunkpos_t cp;
		int64_t size = 0, o;
		int n = 0, l, stop;
		cp.off1 = fq_next_off(S->f1); cp.off2 = S->f2 ? fq_next_off(S->f2) : 0; cp.n_before = n_before;
		while ((l = fq_scan(S->f1, &o)) >= 0) {   /* the rule of bsx_fq_read_chunk */
			int l2 = 0;
			if (S->f2 && (l2 = fq_scan(S->f2, &o)) < 0) break;
			size += l; ++n;
			if (S->f2) { size += l2; ++n; }
			if (size >= S->chunk_size && (n & 1) == 0) break;
		}
		cp.n = n; cp.pad = 0;
		pthread_mutex_lock(&S->mu);
		if (n > 0) bsx_vec_push(S->tab, cp);
		if (n == 0) S->done = 1;
		stop = S->stop;
		pthread_cond_broadcast(&S->cv);
		pthread_mutex_unlock(&S->mu);
		if (n == 0 || stop) break;
		n_before += n;
	}
	pthread_mutex_lock(&S->mu); S->done = 1; pthread_cond_broadcast(&S->cv); pthread_mutex_unlock(&S->mu);
	return 0;
}
bsx_fq_scan_t *bsx_fq_scan_start(const char *fn1, const char *fn2, int chunk_size)
{
	bsx_fq_scan_t *S;
	if (!bsx_fq_plain_file(fn1) || (fn2 && !bsx_fq_plain_file(fn2))) return 0;
	S = (bsx_fq_scan_t*)calloc(1, sizeof(*S));
	S->chunk_size = chunk_size;
	if ((S->f1 = bsx_fq_open(fn1)) == 0 || (fn2 && (S->f2 = bsx_fq_open(fn2)) == 0)) { bsx_fq_close(S->f1); free(S); return 0; }
	bsx_vec_init(S->tab);
	pthread_mutex_init(&S->mu, 0); pthread_cond_init(&S->cv, 0);
	pthread_create(&S->th, 0, scan_main, S);
	return S;
}
int bsx_fq_scan_get(bsx_fq_scan_t *S, int64_t k, bsx_fq_chunkpos_t *out)   /* waits for chunk k; 0: the input has fewer chunks */
{
	int ok;
	pthread_mutex_lock(&S->mu);
	while ((int64_t)S->tab.n <= k && !S->done) pthread_cond_wait(&S->cv, &S->mu);
	ok = (int64_t)S->tab.n > k;
	if (ok) *out = S->tab.a[k];
	pthread_mutex_unlock(&S->mu);
	return ok;
}
void bsx_fq_scan_close(bsx_fq_scan_t *S)
{
	if (!S) return;
	pthread_mutex_lock(&S->mu); S->stop = 1; pthread_mutex_unlock(&S->mu);
	pthread_join(S->th, 0);
	bsx_fq_close(S->f1); bsx_fq_close(S->f2);
	bsx_vec_free(S->tab);
	free(S);
}
/* test hook: the whole table at once */
BSX_API int64_t bsx_fq_scan_table(const char *fn1, const char *fn2, int chunk_size, int64_t cap, bsx_fq_chunkpos_t *out)
{
	bsx_fq_scan_t *S = bsx_fq_scan_start(fn1, fn2, chunk_size);
	int64_t k = 0;
	bsx_fq_chunkpos_t cp;
	if (!S) return -1;
	while (bsx_fq_scan_get(S, k, &cp)) { if (k < cap) out[k] = cp; ++k; }
	bsx_fq_scan_close(S);
	return k;
}

/* the next chunk walked without building its records (the chunk rule of bsx_fq_read_chunk over fq_scan): how many reads it holds.  What a
 * rank does with the chunks of the other ranks when the input cannot be sought in (compressed, piped): the stream has to be inflated
 * and its grammar followed, but nothing is copied or allocated. */
int bsx_fq_skip_chunk(bsx_fq_t *f1, bsx_fq_t *f2, int chunk_size)
{
	int64_t size = 0, o;
	int n = 0, l;
	while ((l = fq_scan(f1, &o)) >= 0) {
		int l2 = 0;
		if (f2 && (l2 = fq_scan(f2, &o)) < 0) { fprintf(stderr, "[W::%s] the 2nd file has fewer sequences.\n", "bsx_fq_read_chunk"); break; }   /* (what the parsing ranks say) */
		size += l; ++n;
		if (f2) { size += l2; ++n; }
		if (size >= chunk_size && (n & 1) == 0) break;
	}
	if (size == 0 && f2 && fq_scan(f2, &o) >= 0) fprintf(stderr, "[W::%s] the 1st file has fewer sequences.\n", "bsx_fq_read_chunk");
	return n;
}

void bsx_read_free(bsx_read_t *s)
{
	free(s->name); free(s->comment); free(s->barcode); free(s->umi); free(s->seq0); free(s->qual); free(s->sam);
}
