/* bsx.h -- C ABI of the MI355X-native `biscuit align` hot path.
 *
 * The reference (zhou-lab/biscuit, lib/aln) has no plugin/FFI layer; its seams are plain C
 * functions.  Every entry point below replaces one of them (cited file:line under
 * /root/reference) with a batch form: plain pointers + sizes, int status (0 = ok, <0 = BSX_E_*)
 * instead of abort(), no globals.  The kernel-level calls run on the GPU (HIP, gfx950); there is
 * no CPU fallback inside the product library: without a usable HIP device every device call
 * returns BSX_E_NODEVICE.
 */
#ifndef BSX_H
#define BSX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSX_EXPORT __attribute__((visibility("default")))

#define BSX_OK            0
#define BSX_E_NODEVICE  (-1)   /* no HIP device / HIP runtime error */
#define BSX_E_ARG       (-2)
#define BSX_E_IO        (-3)
#define BSX_E_NOMEM     (-4)
#define BSX_E_FORMAT    (-5)
#define BSX_E_INTERNAL  (-6)

/* ---- flags: identical values to MEM_F_* (lib/aln/bwamem.h:42-52) ---- */
#define BSX_F_PE             0x2
#define BSX_F_NOPAIRING      0x4
#define BSX_F_ALL            0x8
#define BSX_F_NO_MULTI       0x10
#define BSX_F_NO_RESCUE      0x20
#define BSX_F_SELF_OVLP      0x40
#define BSX_F_ALN_REG        0x80
#define BSX_F_REF_HDR        0x100
#define BSX_F_SOFTCLIP       0x200
#define BSX_F_SMARTPE        0x400
#define BSX_F_KEEP_SUPP_MAPQ 0x1000

/* ksw xtra flags (lib/aln/ksw.h:31-35) */
#define BSX_KSW_XBYTE  0x10000
#define BSX_KSW_XSTOP  0x20000
#define BSX_KSW_XSUBO  0x40000
#define BSX_KSW_XSTART 0x80000

/* Alignment options: field-for-field mem_opt_t (lib/aln/bwamem.h:54-124), same defaults
 * (mem_opt_init, lib/aln/bwamem.c:77-128). */
typedef struct bsx_opt {
	int a, b;
	int o_del, e_del;
	int o_ins, e_ins;
	int pen_unpaired;
	int pen_clip5, pen_clip3;
	int w;
	int zdrop;
	uint64_t max_mem_intv;
	int T;
	int flag;
	int min_seed_len;
	int min_chain_weight;
	uint32_t max_chain_extend;
	float split_factor;
	int split_width;
	uint32_t max_occ;
	int max_chain_gap;
	int n_threads;
	int chunk_size;
	float mask_level;
	float drop_ratio;
	float XA_drop_ratio;
	float mask_level_redun;
	float mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins;
	int max_matesw;
	int max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
	uint8_t parent;     /* -b */
	uint8_t bsstrand;   /* -f */
	int8_t ctmat[25];
	int8_t gamat[25];
	uint8_t *adaptor1; int l_adaptor1;
	uint8_t *adaptor2; int l_adaptor2;
	int clip5, clip3, min_base_qual;
	uint8_t has_bc;
} bsx_opt_t;

/* insert-size statistics: mem_pestat_t (lib/aln/bwamem.h:126-131) */
typedef struct bsx_pestat {
	int low, high;
	int set;
	int failed;
	double avg, std;
} bsx_pestat_t;

/* one read: the fields of bseq1_t (lib/aln/bwa.h:52-61) that mem_process_seqs reads/writes */
typedef struct bsx_read {
	int l_seq, id;
	char *name, *comment, *barcode, *umi, *qual, *sam;
	uint8_t *seq;        /* nt4 codes; advanced by clip5 after clipping */
	uint8_t *seq0;       /* start of the unclipped sequence (owner of the allocation) */
	int l_seq0;
	int l_adaptor;
	int clip5, clip3;
} bsx_read_t;

/* bi-interval: bwtintv_t (lib/aln/bwt.h:80-82) */
typedef struct bsx_intv {
	uint64_t x[3];
	uint64_t info;       /* beg<<32 | end */
} bsx_intv_t;

/* ------------------------------------------------------------------------------------------
 * Kernel-level batch jobs.  Sequences are never copied into jobs: a job names a *view*
 *   query view : chunk read buffer offset qoff, length qlen, walking direction qdir (+1/-1),
 *                optional complement (3-b for b<4)
 *   target view: reference coordinate tpos in the forward-reverse space [0, 2*l_pac)
 *                (bns_get_seq semantics, lib/aln/bntseq.c:402-422), length tlen, direction tdir
 * element i of a view is buf[qoff + i*qdir] resp. ref(tpos + i*tdir).
 * ------------------------------------------------------------------------------------------ */

/* one strand search: mem_collect_intv (lib/aln/memchain.c:50-106) for read x parent */
typedef struct bsx_seed_task {
	uint32_t qoff;       /* offset of the (clipped) read in the chunk read buffer */
	int32_t  len;
	int32_t  parent;     /* 1: C>T read vs parent index, 0: G>A read vs daughter index */
} bsx_seed_task_t;

/* suffix-array lookup: bwt_sa (lib/aln/bwt.c:87-97) */
typedef struct bsx_sa_job {
	uint64_t k;
	int32_t  parent;     /* which index */
	int32_t  pad;
} bsx_sa_job_t;

/* banded extension: ksw_extend2 (lib/aln/ksw.c:380-479) */
typedef struct bsx_ext_job {
	int64_t  tpos;
	uint32_t qoff;
	int32_t  qlen, tlen;
	int32_t  h0;
	int32_t  w;
	int32_t  end_bonus;
	int8_t   qdir, tdir;
	uint8_t  parent;     /* 1: ctmat, 0: gamat (lib/aln/memchain.c:654,713) */
	uint8_t  pad;
} bsx_ext_job_t;
typedef struct bsx_ext_res {
	int32_t score, qle, tle, gtle, gscore, max_off;
} bsx_ext_res_t;

/* One alignment region as mem_chain2region leaves it (the fields of mem_alnreg_t, lib/aln/mem_alnreg.h:34-66,
 * that lib/aln/memchain.c:822-869 sets; everything else is zero at that point). */
typedef struct bsx_region {
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0;
	float   frac_rep;
	uint8_t bss, parent, pad[2];
} bsx_region_t;

/* local SW with 2nd-best + start recovery: ksw_align2 (lib/aln/ksw.c:343-365) */
typedef struct bsx_sw_job {
	int64_t  tpos;
	uint32_t qoff;
	int32_t  qlen, tlen;
	int32_t  xtra;
	int8_t   qdir, tdir;
	uint8_t  qcomp;      /* complement the query view (mate rescue, lib/aln/mem_alnreg.c:411) */
	uint8_t  use_ct;     /* 1: ctmat, 0: gamat */
} bsx_sw_job_t;
typedef struct bsx_sw_res {   /* kswr_t, lib/aln/ksw.h:37-43 */
	int32_t score, te, qe, score2, te2, tb, qb;
} bsx_sw_res_t;

/* banded global alignment + traceback with the band-doubling loop of mem_alnreg_setSAM
 * (lib/aln/mem_alnreg_format.c:63-77) and band set-up of bis_bwa_gen_cigar2
 * (lib/aln/bwa.c:314-340) folded in.  n_try==1 && !want_cigar is the score-only call made by
 * mem_test_reg_concatenation (lib/aln/mem_alnreg.c:92). */
typedef struct bsx_glb_job {
	int64_t  tpos;
	uint32_t qoff;
	int32_t  qlen, tlen;
	int32_t  w0;          /* first band passed as w_ */
	int32_t  w_max;       /* opt->w<<2 cap applied before each try */
	int32_t  truesc;      /* stop when score >= truesc - a */
	int32_t  n_try;       /* 3 for setSAM, 1 for the concatenation test */
	uint32_t cigar_off;   /* where this job's CIGAR goes in the output pool (u32 units) */
	uint32_t cigar_cap;
	int8_t   qdir, tdir;
	uint8_t  use_ct;
	uint8_t  want_cigar;
} bsx_glb_job_t;
typedef struct bsx_glb_res {
	int32_t score;
	int32_t n_cigar;      /* <0: cigar_cap too small, -n_cigar needed */
	int32_t w_used;
	int32_t pad;
} bsx_glb_res_t;

/* ------------------------------------------------------------------------------------------
 * Index (host side): <base>.{par,dau}.{bwt,sa}, <base>.bis.{ann,amb,pac}[, <base>.alt]
 * replaces bwa_idx_load_from_disk (lib/aln/bwa.c:525-554)
 * ------------------------------------------------------------------------------------------ */
typedef struct bsx_index bsx_index_t;
int  bsx_index_load(const char *base, bsx_index_t **out);
void bsx_index_free(bsx_index_t *idx);
/* builds all seven files from a FASTA: main_biscuit_index (lib/aln/bwtindex.c:206-347); host suffix sorter, genomes up
 * to 1.07 Gbp (2 x l_pac < 2^31).  In steps: bsx_index_from_fasta (bis_bns_fasta2bntseq, lib/aln/bntseq.c:542-633: pac +
 * annotation), then the two FM indices on the host (bsx_index_build_host) or, for genomes of any size, on the device
 * (bsx_device_build_index below: what bwt_bwtgen, lib/aln/bwt_gen.c:1595-1607, is for in the reference), then bsx_index_save. */
int  bsx_index_build(const char *fasta, const char *base);
int  bsx_index_from_fasta(const char *fasta, bsx_index_t **out);
int  bsx_index_build_host(bsx_index_t *idx);
int  bsx_index_save(const bsx_index_t *idx, const char *base);
int64_t bsx_index_l_pac(const bsx_index_t *idx);
int  bsx_index_n_seqs(const bsx_index_t *idx);
const uint8_t *bsx_index_pac(const bsx_index_t *idx);    /* 2 bits per base, the .bis.pac layout */
int  bsx_index_contig(const bsx_index_t *idx, int i, const char **name, int64_t *offset, int64_t *len);

/* options: mem_opt_init (lib/aln/bwamem.c:77-128) + the three matrices (lib/aln/bwa.c:146-182) */
void bsx_opt_init(bsx_opt_t *opt);
void bsx_opt_fill_matrices(bsx_opt_t *opt);

/* ------------------------------------------------------------------------------------------
 * Device (HIP) side
 * ------------------------------------------------------------------------------------------ */
typedef struct bsx_device bsx_device_t;
/* open HIP device `ordinal`, create streams; BSX_E_NODEVICE if none */
int  bsx_device_open(int ordinal, bsx_device_t **out);
void bsx_device_close(bsx_device_t *dev);
/* copy both FM indices, SA samples and pac into HBM (resident for the life of dev) */
int  bsx_device_upload_index(bsx_device_t *dev, const bsx_index_t *idx);
/* Build both FM indices (BWT with occurrence blocks + suffix-array samples) of idx's genome on the device, from its pac,
 * and leave them resident exactly as bsx_device_upload_index would: 64-bit suffix sorting in HBM, hg38-sized genomes
 * included (bwt_bwtgen + bwt_bwtupdate_core + bwt_cal_sa, lib/aln/bwtindex.c:258-340).  fill_host != 0 also copies the
 * file-format arrays into idx (for bsx_index_save and for host-side consumers); they are byte-identical to the host
 * builder's and the reference's. */
int  bsx_device_build_index(bsx_device_t *dev, bsx_index_t *idx, int fill_host);
/* copy scoring matrices / penalties used by the DP kernels */
int  bsx_device_set_opt(bsx_device_t *dev, const bsx_opt_t *opt);
/* upload the chunk read buffer (nt4 codes of all clipped reads, concatenated) */
int  bsx_device_set_reads(bsx_device_t *dev, const uint8_t *buf, size_t n);
const char *bsx_device_name(const bsx_device_t *dev);

/* K1+K2: all three seeding passes of mem_collect_intv + final ordering by info.
 * out_off has n+1 entries; *out / *out_cap is a caller-owned growable array (realloc'd). */
int bsx_seed_batch(bsx_device_t *dev, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
                   bsx_intv_t **out, int64_t *out_cap, int64_t *out_off);
/* K3 */
int bsx_sa_batch(bsx_device_t *dev, int64_t n, const bsx_sa_job_t *jobs, uint64_t *pos);
/* K4 */
int bsx_extend_batch(bsx_device_t *dev, int64_t n, const bsx_ext_job_t *jobs, bsx_ext_res_t *res);
/* K1+K2+K3 and the chaining/extension logic between them in one pass (mem_chain + mem_chain_flt +
 * mem_chain2region, lib/aln/memchain.c:268-488,742-904; called per strand search from lib/aln/bwamem.c:352-372):
 * regions of task i = (*out)[out_off[i] .. out_off[i] + out_n[i]).  out_n[i] < 0 means the device declined the task
 * (a read too long for its on-chip rows, an over-represented interval that has to be walked past max_occ, more
 * occurrences/chains/regions than its tables hold): the caller runs that task through bsx_sa_batch/bsx_extend_batch and its own chaining instead.  For those
 * tasks the sorted SA intervals are handed back so that they need not be seeded again: the j-th declined task with
 * out_n != -1 (in task order) has (*decl_intv)[decl_off[j] .. decl_off[j+1]); out_n == -1 means its interval list
 * overflowed as well and bsx_seed_batch has to redo it.  decl_off must have room for n + 1 entries. */
int bsx_regions_batch(bsx_device_t *dev, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
                      bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n,
                      bsx_intv_t **decl_intv, int64_t *decl_cap, int64_t *decl_off);
/* out_n[i] == BSX_REGIONS_PENDING: the strand search overflowed the seeding lists (a read inside a tandem repeat) and is
 * being seeded again with much longer lists while the call returns; bsx_regions_finish waits for those, appends their
 * regions to *out and fills out_off/out_n in (-1: seed and chain it on the host).  Same arrays as the batch call. */
#define BSX_REGIONS_PENDING (-100)
int bsx_regions_finish(bsx_device_t *dev, bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n);
/* C5 for the regions the last bsx_regions_batch left on the device: mem_sort_deduplicate (lib/aln/mem_alnreg.c:112-202) of every
 * read, a read's regions being those of its per_read consecutive strand searches concatenated in call order
 * (lib/aln/bwamem.c:352-372).  out_n[i] >= 0: the read keeps that many regions, out_idx[i * bsx_regions_dedup_cap() + k]
 * naming the k-th by its index in the concatenation; out_n[i] == -1: left to the caller (a strand search of the read was
 * not finished on the device, too many regions, or two regions have to be tested for concatenation, mem_alnreg.c:63-108). */
int bsx_regions_dedup_cap(void);
int bsx_regions_dedup(bsx_device_t *dev, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx);
/* ... and the reads with MORE regions than that, up to bsx_regions_dedup_long_cap() (a wavefront per read, k_dedup_long): as above, and for
 * such a read long_off[i] >= 0 says where its out_n[i] indices (16 bits each) start in *long_idx (a malloc'd array the call grows:
 * *long_cap entries); long_off[i] == -1: the read's list, if it has one, is in out_idx. */
int bsx_regions_dedup_long_cap(void);
int bsx_regions_dedup2(bsx_device_t *dev, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx,
                       int64_t *long_off, uint16_t **long_idx, int64_t *long_cap);
/* K5 */
int bsx_sw_batch(bsx_device_t *dev, int64_t n, const bsx_sw_job_t *jobs, bsx_sw_res_t *res);
/* K6 */
int bsx_global_batch(bsx_device_t *dev, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res,
                     uint32_t *cigar_pool, size_t cigar_pool_len);
/* K6 with the second half of bis_bwa_gen_cigar2 (lib/aln/bwa.c:342-418) done on the device as well: NM, the MD string
 * and BISCUIT's conversion / retention counts of every job that has a CIGAR, walked over the job's own (possibly
 * reversed) sequences.  tags[i].l_md < 0: job i has no CIGAR yet (res[i].n_cigar <= 0).  The MD strings (NUL-terminated)
 * are packed into *md, a buffer the library grows with realloc as bsx_seed_batch grows *out. */
typedef struct bsx_glb_tag {
	int32_t  NM, ZC, ZR;
	int32_t  l_md;        /* strlen of the MD string */
	uint64_t md_off;      /* where it starts in *md */
	uint8_t  bss_u;       /* no conversion seen (bwa.c:415-416) */
	uint8_t  pad[7];
} bsx_glb_tag_t;
int bsx_global_batch_tags(bsx_device_t *dev, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res,
                          uint32_t *cigar_pool, size_t cigar_pool_len, bsx_glb_tag_t *tags, char **md, int64_t *md_cap);

/* Settings of the library that never change its output (launch shapes, table sizes, which of two equivalent paths runs: what the tests and
 * the A/B tools switch).  One registry (csrc/host/tune.c has the table of names): bsx_tune_set(name, value) between calls of the library
 * (value NULL: back to the default; BSX_E_ARG for an unknown name), or "$BSX_TUNE=name=value,name=value" for a whole process.  bsx_phases():
 * the diagnostic level ($BSX_PHASES or the setting "phases"). */
int bsx_tune_set(const char *name, const char *value);
const char *bsx_tune_str(const char *name);
long bsx_tune_long(const char *name, long dflt);
int bsx_tune_is_set(const char *name);
int bsx_phases(void);
const char *bsx_tune_name(int i);
const char *bsx_tune_doc(int i);

/* device-side work counters of the last seed/sa batch (algorithmic-bytes model):
 * c[0]=bwt_occ4 calls, c[1]=same-block bwt_2occ4 calls, c[2]=bwt_occ calls (inside bwt_sa, from k_sa and
 * from the region kernels), c[3]=bwt_sa calls */
int bsx_device_counters(bsx_device_t *dev, uint64_t c[4], int reset);
/* the seeding passes of bsx_regions_batch one by one, since the last reset (read it BEFORE bsx_device_counters / bsx_device_seed_table with
 * reset: they zero the same counters): w[0], w[1] = 64-byte FM blocks and table entries read by the chunk-wide first pass (and bsx_seed_batch
 * launches), w[2], w[3] = by the second pass over the strand searches seeded again inside a chunk's launch sequence, w[4] = its launches,
 * w[5] = its strand searches; ms[0], ms[1] = the two passes' summed HIP-event times.  (Each pass's bytes over that pass's time: bench.py) */
int bsx_device_seed_passes(bsx_device_t *dev, uint64_t w[6], double ms[2], int reset);
/* Several GPUs sharing ONE chunk: each process aligns a slice of the chunk's pairs.  Two things tie a read to its chunk:
 * mem_pestat (bwamem.c:464-467), whose result is a function of the chunk's histogram of insert sizes -- bsx_pes_hist_hook, when set, is
 * called with this process's histogram (2 * max_ins + 1 counters) and must return the sum over all processes in place (an all-reduce;
 * bsx_pestat_sync_empty: the call of a process whose slice is empty) -- and the pair's index within the chunk, which seeds the hash that
 * breaks ties between equal hits (bwamem.c:408,413): bsx_chunk_slice_offset(first) says where the NEXT chunk this thread passes to
 * bsx_process_seqs / bsx_stream_push starts within its chunk.  n_processed is the global index of the slice's first read, as always. */
extern void (*bsx_pes_hist_hook)(void *ud, int64_t *hist, int n_bins);
extern void *bsx_pes_hist_ud;
void bsx_pestat_sync_empty(const bsx_opt_t *opt);
void bsx_chunk_slice_offset(int64_t first);
/* what the region kernels (bsx_regions_batch: K3 + C1 + C2 + K4 + C4) were given and made since the last reset, summed over the chunks:
 * w[0] strand searches, w[1] SA intervals read (32 B each), w[2] seed occurrences whose position was looked up (8 B each),
 * w[3] alignment regions written (56 B each), w[4] read bases of the strand searches -- the terms of the family's algorithmic bytes */
int bsx_device_region_work(bsx_device_t *dev, uint64_t w[5], int reset);
/* the seeding kernel's table of k-mer intervals: entries read since the last reset, depth K of the resident table (0: none) */
int bsx_device_seed_table(bsx_device_t *dev, uint64_t *lookups, int *depth, int reset);
/* average GPU time (ms, HIP events on the launch stream) and launch count of each kernel since
 * the last reset: k = 0 seed (the chunk-wide launch of bsx_regions_batch), 1 sa, 2 extend, 3 sw, 4 global, 5 regions (first tier),
 * 6 regions (tiers 1b, 2 and 3), 7 seed outside the chunk-wide launch (the second seeding pass inside a chunk's sequence; bsx_seed_batch launches: the strand searches the host chains) */
int bsx_device_kernel_time(bsx_device_t *dev, int k, double *total_ms, int64_t *launches, int reset);

/* ------------------------------------------------------------------------------------------
 * Several GPUs ("replicas + gather"): one process per GPU, the per-chunk alignment records brought together in input order.
 * The reference has nothing to bind here (it is one process: align.c:100-170 writes its chunks in order from kt_pipeline's third step);
 * what these replace is that ordered write, across processes.  csrc/host/gather.c has the protocol (rounds: sizes to everybody, then either
 * the payload to rank 0 or every rank's own pwrite() into the output file at its offset), over a small communication vtable:
 *   bsx_transport_rccl   RCCL over xGMI (ncclAllGather / ncclSend / ncclRecv / ncclAllReduce; the unique id travels through a file);
 *   bsx_transport_local  the ranks as threads of one process (tests of the protocol).
 * ------------------------------------------------------------------------------------------ */
typedef struct bsx_transport {
	void *ctx;
	int rank, world;
	int (*all_gather)(void *ctx, const int64_t *mine, int n, int64_t *all);   /* n values of every rank, in rank order, to every rank */
	int (*send)(void *ctx, int dst, const void *buf, size_t n_bytes);          /* host memory; returns when buf may be reused */
	int (*recv_many)(void *ctx, int n_src, const int *src, void *const *buf, const size_t *n_bytes);   /* the receives of a round, posted together */
	int (*all_reduce_sum)(void *ctx, int64_t *buf, int n);                     /* in place (the insert-size histograms of ranks sharing a chunk) */
	void (*close)(void *ctx);
} bsx_transport_t;
int bsx_transport_local(int world, bsx_transport_t *out);   /* out[0 .. world): one per thread-rank */
/* rank / world of this process, the HIP device its staging buffers live on, and a path all ranks can read: rank 0 writes the unique ids there
 * (and removes the file when the last communicator is up).  Two communicators: `gather` for the rounds, `reduce` for all_reduce_sum calls
 * made from another thread (either may be NULL).  BSX_E_NODEVICE when librccl cannot be loaded. */
int bsx_transport_rccl(int rank, int world, int device, const char *id_path, bsx_transport_t *gather, bsx_transport_t *reduce);
/* the ranks as processes of one node talking through rank 0 over Unix-domain sockets <path>.0 / <path>.1 (the CPU checker's multi-process
 * runs; a fallback where librccl cannot be loaded) */
int bsx_transport_socket(int rank, int world, const char *path, bsx_transport_t *gather, bsx_transport_t *reduce);
typedef struct bsx_gather bsx_gather_t;
/* direct_path != NULL: every rank writes its own chunks into that file (bsx_gather_direct_ok says whether it may); else rank 0's `sink`
 * is called with every chunk in input order.  max_pending: chunks a rank may hold before bsx_gather_submit blocks. */
int bsx_gather_open(const bsx_transport_t *tr, const char *direct_path, void (*sink)(void *ud, int64_t chunk, const void *buf, size_t n), void *ud,
                    int max_pending, bsx_gather_t **out);
int bsx_gather_set_header(bsx_gather_t *g, const void *hdr, size_t n);       /* rank 0, direct form: what the file starts with */
int bsx_gather_submit(bsx_gather_t *g, int64_t chunk, void *buf, size_t n);  /* buf: malloc'd, the gather's from now on */
int bsx_gather_close_input(bsx_gather_t *g);                                 /* no more chunks from this rank */
int bsx_gather_run(bsx_gather_t *g, int64_t *n_chunks);                      /* the rounds, on the calling thread, until the input has ended everywhere */
void bsx_gather_stats(const bsx_gather_t *g, int64_t out[3]);                /* rounds, payload bytes rank 0 received, bytes this rank wrote itself */
void bsx_gather_free(bsx_gather_t *g);
int bsx_gather_direct_ok(const char *path, int world, int local_world);

/* ------------------------------------------------------------------------------------------
 * The whole path: mem_process_seqs (lib/aln/bwamem.c:432-476, declared bwamem.h:184)
 *   reads[i].{name,comment,seq(nt4),qual,l_seq,barcode,umi} in; reads[i].sam (malloc'd) out;
 *   PE input interleaved, n even.  Returns BSX_OK or an error instead of aborting.
 * ------------------------------------------------------------------------------------------ */
int bsx_process_seqs(bsx_device_t *dev, const bsx_opt_t *opt, const bsx_index_t *idx,
                     int64_t n_processed, int n, bsx_read_t *reads, const bsx_pestat_t *pes0);

/* The same, for a sequence of chunks, as a pipeline: what lib/aln/align.c:100-170 (kt_pipeline over
 * read / mem_process_seqs / write) does for I/O, done here for the two halves of the aligning step itself.  The
 * device-bound front half of a chunk (seeding .. regions) runs on its own device lane and thread while the
 * host-bound back half of an older chunk (merge, pairing, CIGAR, SAM text) runs on the caller's thread; up to
 * depth-1 front halves are in flight ahead of it (depth 3, or 4 against genomes of 1 Gbp and more, unless $BSX_STREAM_DEPTH says otherwise; at most 6).
 *   bsx_stream_push(chunk k) returns once chunk k-(depth-1) is complete (its reads[i].sam are set);
 *   bsx_stream_flush completes the chunks still in flight, in order.  Every chunk is independent, exactly as with
 *   bsx_process_seqs (own insert-size statistics unless pes0 is given): the output does not depend on the depth.
 * opt and idx must outlive the stream; a chunk's reads must stay valid until it is complete. */
typedef struct bsx_stream bsx_stream_t;
int  bsx_stream_open(bsx_device_t *dev, const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_pestat_t *pes0, bsx_stream_t **out);
int  bsx_stream_depth(const bsx_stream_t *s);
int  bsx_stream_push(bsx_stream_t *s, int64_t n_processed, int n, bsx_read_t *reads);
int  bsx_stream_flush(bsx_stream_t *s);
void bsx_stream_close(bsx_stream_t *s);

/* `biscuit align` command line: main_align (lib/aln/align.c:319-598).  SAM on `out` (stdout). */
int bsx_align_main(int argc, char **argv);

/* SAM header: bwa_print_sam_hdr (lib/aln/bwa.c:654-684); returns malloc'd text */
char *bsx_sam_header(const bsx_index_t *idx, const char *hdr_line, const char *pg_line);

BSX_EXPORT const char *bsx_version(void);
BSX_EXPORT const char *bsx_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
