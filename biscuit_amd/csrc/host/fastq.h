/* fastq.h -- FASTA/FASTQ chunk reader (fastq.c) */
#ifndef BSX_FASTQ_H
#define BSX_FASTQ_H
#include "bsx.h"
typedef struct bsx_fq bsx_fq_t;
bsx_fq_t *bsx_fq_open(const char *fn);
void bsx_fq_close(bsx_fq_t *f);
/* bis_bseq_read (lib/aln/bwa.c:817-850): interleaves f1/f2 when f2 != NULL; NULL at end of input */
bsx_read_t *bsx_fq_read_chunk(bsx_fq_t *f1, bsx_fq_t *f2, int chunk_size, int has_bc, int *n);
void bsx_read_free(bsx_read_t *s);
/* the same chunks, with one parser thread per file working ahead of the caller */
typedef struct bsx_fq_pair bsx_fq_pair_t;
bsx_fq_pair_t *bsx_fq_pair_open(bsx_fq_t *f1, bsx_fq_t *f2, int has_bc);
bsx_read_t *bsx_fq_pair_read_chunk(bsx_fq_pair_t *p, int chunk_size, int *n);
void bsx_fq_pair_close(bsx_fq_pair_t *p);
#endif
