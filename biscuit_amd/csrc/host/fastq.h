/* fastq.h -- FASTA/FASTQ chunk reader (fastq.c) */
#ifndef BSX_FASTQ_H
#define BSX_FASTQ_H
#include "bsx.h"
typedef struct bsx_fq bsx_fq_t;
bsx_fq_t *bsx_fq_open(const char *fn);
void bsx_fq_close(bsx_fq_t *f);
int bsx_fq_error(const bsx_fq_t *f);   /* damaged or truncated compressed input was met: the end of input the reader reported was not one */
/* bis_bseq_read (lib/aln/bwa.c:817-850): interleaves f1/f2 when f2 != NULL; NULL at end of input */
bsx_read_t *bsx_fq_read_chunk(bsx_fq_t *f1, bsx_fq_t *f2, int chunk_size, int has_bc, int *n);
void bsx_read_free(bsx_read_t *s);
/* the same chunks, with one parser thread per file working ahead of the caller */
typedef struct bsx_fq_pair bsx_fq_pair_t;
int bsx_fq_skip_chunk(bsx_fq_t *f1, bsx_fq_t *f2, int chunk_size);   /* reads in the next chunk, walked without building records */
bsx_fq_pair_t *bsx_fq_pair_open(bsx_fq_t *f1, bsx_fq_t *f2, int has_bc);
bsx_fq_pair_t *bsx_fq_pair_open_n(bsx_fq_t *f1, bsx_fq_t *f2, int has_bc, long n_records);   /* the parsers stop after n_records records per file */
bsx_read_t *bsx_fq_pair_read_chunk(bsx_fq_pair_t *p, int chunk_size, int *n);
void bsx_fq_pair_close(bsx_fq_pair_t *p);
/* chunk boundaries over plain files by a scan that builds no records (fastq.c): NULL from _start when a file is compressed or not seekable */
typedef struct { int64_t off1, off2, n_before; int n, pad; } bsx_fq_chunkpos_t;   /* byte offsets of the chunk in the two files, reads before it, reads in it */
typedef struct bsx_fq_scan bsx_fq_scan_t;
int bsx_fq_plain_file(const char *fn);
bsx_fq_scan_t *bsx_fq_scan_start(const char *fn1, const char *fn2, int chunk_size);
int bsx_fq_scan_get(bsx_fq_scan_t *s, int64_t k, bsx_fq_chunkpos_t *out);
void bsx_fq_scan_close(bsx_fq_scan_t *s);
int bsx_fq_seek(bsx_fq_t *f, int64_t off);   /* reposition at a record boundary the scan reported */
#endif
