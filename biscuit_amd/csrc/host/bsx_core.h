/* bsx_core.h -- internal host-side types of the MI355X biscuit-align path.
 * Host code is C (the reference's language); device work goes through bsx_backend_t, whose only
 * product implementation is the HIP one (csrc/hip). */
#ifndef BSX_CORE_H
#define BSX_CORE_H

#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "bsx.h"

#define BSX_API __attribute__((visibility("default")))

/* ---------- tiny growable array ---------- */
#define BSX_VEC(type) struct { size_t n, m; type *a; }
#define bsx_vec_init(v) ((v).n = (v).m = 0, (v).a = 0)
#define bsx_vec_free(v) (free((v).a), (v).a = 0, (v).n = (v).m = 0)
#define bsx_vec_reserve(v, cap) do { \
		if ((v).m < (size_t)(cap)) { size_t m_ = (v).m ? (v).m : 4; while (m_ < (size_t)(cap)) m_ <<= 1; \
			(v).a = realloc((v).a, m_ * sizeof(*(v).a)); (v).m = m_; } } while (0)
#define bsx_vec_push(v, x) do { if ((v).n == (v).m) bsx_vec_reserve(v, (v).n + 1); (v).a[(v).n++] = (x); } while (0)
#define bsx_vec_pushp(v) (((v).n == (v).m ? (void)((v).m = (v).m ? (v).m << 1 : 4, (v).a = realloc((v).a, (v).m * sizeof(*(v).a))) : (void)0), &(v).a[(v).n++])

/* ---------- chunk-lifetime arenas ----------
 * Seeds, chains and regions live exactly as long as one chunk.  Each worker thread bump-allocates
 * them from its own arena; nothing is freed individually, the arenas are rewound when the chunk is
 * done and their pages are reused by the next chunk (no malloc lock, no page faults, no cleanup pass). */
typedef struct bsx_arena bsx_arena_t;
/* Two chunks can be in flight (front half of one, back half of the previous): there are two arena sets. */
BSX_API int  bsx_arenas_begin(int n_threads);   /* take a free set (-1: arenas off), bind it to the caller and to its parallel loops */
BSX_API void bsx_arenas_bind(int set);          /* another thread continues work on the chunk that owns `set` */
void bsx_arenas_bind_extra(int set, int k);     /* ... a further thread works on it BESIDE its owner, with arena k of the set's spares as its own */
BSX_API void bsx_arenas_end(int set);           /* rewind the set, release it, unbind the caller */
void *bsx_arena_alloc(bsx_arena_t *a, size_t n);
/* large arrays that live as long as the arena set (chunk after chunk): slot = a small fixed index per array */
void *bsx_big_get(int set, int slot, size_t bytes);
void  bsx_big_put(int set, int slot, void *p);                     /* frees only when the set is off (-1) */
void  bsx_big_update(int set, int slot, void *p, size_t cap);      /* the user realloc'd the block */
BSX_API extern __thread bsx_arena_t *bsx_tls_arena;
static inline void *bsx_crealloc(void *p, size_t old_bytes, size_t new_bytes)
{
	if (bsx_tls_arena) {
		void *q = bsx_arena_alloc(bsx_tls_arena, new_bytes);
		if (p && old_bytes) memcpy(q, p, old_bytes < new_bytes ? old_bytes : new_bytes);
		return q;
	}
	return realloc(p, new_bytes);
}
static inline void bsx_cfree(void *p) { if (!bsx_tls_arena) free(p); }
#define bsx_cvec_free(v) (bsx_cfree((v).a), (v).a = 0, (v).n = (v).m = 0)
#define bsx_cvec_reserve(v, cap) do { \
		if ((v).m < (size_t)(cap)) { size_t m_ = (v).m ? (v).m : 4; while (m_ < (size_t)(cap)) m_ <<= 1; \
			(v).a = bsx_crealloc((v).a, (v).m * sizeof(*(v).a), m_ * sizeof(*(v).a)); (v).m = m_; } } while (0)
#define bsx_cvec_push(v, x) do { if ((v).n == (v).m) bsx_cvec_reserve(v, (v).n + 1); (v).a[(v).n++] = (x); } while (0)

#define bsx_min(a, b) ((a) < (b) ? (a) : (b))
#define bsx_max(a, b) ((a) > (b) ? (a) : (b))

/* ---------- FM index of one converted text (bwt_t, lib/aln/bwt.h:54-71) ---------- */
typedef struct {
	uint64_t primary;
	uint64_t L2[5];
	uint64_t seq_len;
	uint64_t bwt_size;      /* in u32 words, occ blocks interleaved */
	uint32_t *bwt;
	int sa_intv;
	uint64_t n_sa;
	uint64_t *sa;
} bsx_fmi_t;

/* ---------- reference meta (bntseq_t, lib/aln/bntseq.h:41-62) ---------- */
typedef struct {
	int64_t offset;
	int32_t len;
	int32_t n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} bsx_ann_t;
typedef struct {
	int64_t offset;
	int32_t len;
	char amb;
} bsx_amb_t;
typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	bsx_ann_t *anns;
	int32_t n_holes;
	bsx_amb_t *ambs;
} bsx_refmeta_t;

struct bsx_index {
	bsx_fmi_t fmi[2];       /* [1] = parent (C>T), [0] = daughter (G>A); lib/aln/bwa.c:535-536 */
	bsx_refmeta_t ref;
	uint8_t *pac;           /* l_pac/4+1 bytes, forward unconverted genome */
};

#define bsx_pac_get(pac, l) ((pac)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

/* genome sink (index.c): contigs and bases streamed in, an index holding pac + annotation (no FM indices yet) out */
typedef struct bsx_gsink bsx_gsink_t;
bsx_gsink_t *bsx_gsink_new(void);
void bsx_gsink_contig(bsx_gsink_t *g, const char *name, const char *comment);
int  bsx_gsink_bases(bsx_gsink_t *g, const char *chars, int64_t n);   /* FASTA characters of the current contig */
bsx_index_t *bsx_gsink_finish(bsx_gsink_t *g);

/* coordinate helpers (bntseq.c:356-452, bntseq.h:91-93) */
static inline int64_t bsx_depos(int64_t l_pac, int64_t pos, int *is_rev)
{
	return (*is_rev = (pos >= l_pac)) ? (l_pac << 1) - 1 - pos : pos;
}
BSX_API int bsx_pos2rid(const bsx_refmeta_t *r, int64_t pos_f);
BSX_API int bsx_intv2rid(const bsx_refmeta_t *r, int64_t rb, int64_t re);
/* base at forward-reverse coordinate p in [0, 2*l_pac) (bns_get_seq, bntseq.c:402-422) */
static inline int bsx_ref_base(int64_t l_pac, const uint8_t *pac, int64_t p)
{
	if (p < l_pac) return bsx_pac_get(pac, p);
	p = (l_pac << 1) - 1 - p;
	return 3 - bsx_pac_get(pac, p);
}
/* clamp [*beg,*end) to the contig containing mid; returns rid (bns_fetch_seq, bntseq.c:428-452) */
BSX_API int bsx_fetch_span(const bsx_refmeta_t *r, int64_t *beg, int64_t mid, int64_t *end);

/* ---------- sorting: exact re-implementation of the reference's introsort permutation ---------- */
typedef int (*bsx_lt_fn)(const void *a, const void *b);   /* returns a < b */
BSX_API void bsx_introsort(void *base, size_t n, size_t width, bsx_lt_fn lt);
BSX_API void bsx_introsort_u64(size_t n, uint64_t *a);
BSX_API void bsx_introsort_i64(size_t n, int64_t *a);
BSX_API uint64_t bsx_hash64(uint64_t key);

/* ---------- B-tree with the reference's node geometry (t = 3) ---------- */
typedef struct bsx_btree bsx_btree_t;
BSX_API bsx_btree_t *bsx_bt_new(void);
BSX_API void bsx_bt_clear(bsx_btree_t *t);
BSX_API void bsx_bt_free(bsx_btree_t *t);
BSX_API int  bsx_bt_size(const bsx_btree_t *t);
BSX_API void bsx_bt_put(bsx_btree_t *t, int64_t pos, int32_t id);
/* id of the key kb_intervalp would return as `lower` (equal key or predecessor), or -1 */
BSX_API int32_t bsx_bt_lower(const bsx_btree_t *t, int64_t pos);
/* in-order ids; returns count */
BSX_API int  bsx_bt_traverse(const bsx_btree_t *t, int32_t *ids);

/* ---------- thread pool (kt_for equivalent; results never depend on scheduling) ---------- */
typedef void (*bsx_for_fn)(void *data, long i, int tid);
BSX_API void bsx_parallel_for(int n_threads, bsx_for_fn fn, void *data, long n);
void *bsx_par_calloc(int n_threads, size_t n, size_t size);   /* calloc whose zeroing is shared by the worker pool (per-chunk tables of 50-70 MB: a serial memset was 20 ms of every back half) */
/* worker threads for host stages: $BSX_HOST_THREADS if set, else opt->n_threads (-@ also fixes the chunk size) */
BSX_API int bsx_host_threads(const bsx_opt_t *opt);

/* ---------- device backend: the batch seams of include/bsx.h behind one vtable ---------- */
typedef struct bsx_backend {
	void *ctx;
	const char *name;
	int (*set_opt)(void *ctx, const bsx_opt_t *opt);
	int (*set_reads)(void *ctx, const uint8_t *buf, size_t n);
	int (*seed_batch)(void *ctx, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
	                  bsx_intv_t **out, int64_t *out_cap, int64_t *out_off);
	int (*sa_batch)(void *ctx, int64_t n, const bsx_sa_job_t *jobs, uint64_t *pos);
	int (*extend_batch)(void *ctx, int64_t n, const bsx_ext_job_t *jobs, bsx_ext_res_t *res);
	int (*sw_batch)(void *ctx, int64_t n, const bsx_sw_job_t *jobs, bsx_sw_res_t *res);
	int (*global_batch)(void *ctx, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res,
	                    uint32_t *cigar_pool, size_t cigar_pool_len);
	/* optional (may be NULL): the same with MD / NM / ZC / ZR, see bsx_global_batch_tags */
	int (*global_batch_tags)(void *ctx, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res,
	                         uint32_t *cigar_pool, size_t cigar_pool_len, bsx_glb_tag_t *tags, char **md, int64_t *md_cap);
	/* optional (may be NULL): seeding through regions in one device pass, see bsx_regions_batch */
	int (*regions_batch)(void *ctx, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
	                     bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n,
	                     bsx_intv_t **decl_intv, int64_t *decl_cap, int64_t *decl_off);
	/* optional, with regions_batch: collect the strand searches it reported as BSX_REGIONS_PENDING */
	int (*regions_finish)(void *ctx, bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n);
	/* optional, with regions_batch: mem_sort_deduplicate of every read over the regions still on the device, see bsx_regions_dedup
	 * (dedup_cap entries of out_idx per read) */
	int (*regions_dedup)(void *ctx, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx);
	int dedup_cap;
	/* optional, with regions_dedup: the same, and the reads with more than dedup_cap regions as well (bsx_regions_dedup2): long_off[i] >= 0 =
	 * read i's out_n[i] indices are 16-bit entries of *long_idx from there; -1 = they are in out_idx as above */
	int (*regions_dedup2)(void *ctx, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx,
	                      int64_t *long_off, uint16_t **long_idx, int64_t *long_cap);
	/* optional, with regions_dedup: mate rescue's plan pass (mem_alnreg.c:385-419: which candidates of the pairs [p0, p1) need an alignment, over
	 * which window) and its K5 batch, on the lists the de-duplication left on the device.  table: a bsx_msw_pair_t per pair; *n_jobs = -1: not
	 * applicable, plan on the host.  token: changes with the chunk (the reads' offsets roff[0 .. n_reads] are uploaded once per chunk) */
	int (*msw_plan)(void *ctx, const bsx_opt_t *opt, const bsx_pestat_t *pes, int64_t token, int64_t n_reads, int per_read, const uint32_t *roff, int max_len,
	                int p0, int p1, void *table, bsx_sw_res_t **res, int64_t *res_cap, int64_t *n_jobs);
} bsx_backend_t;
/* what msw_plan says about a pair: base = its first job in res (-1: the pair is left to the host's own plan), n_c[i] = candidates of read i it
 * looked at, mask[i] = which of them have a job (bit j: candidate j), in job order: read 0's, then read 1's */
typedef struct { int32_t base, n_c[2], pad; uint64_t mask[2]; } bsx_msw_pair_t;

/* mem_process_seqs equivalent over an arbitrary backend (the product passes the HIP backend;
 * tests may pass the CPU restatement that lives under oracle/) */
BSX_API int bsx_process_seqs_backend(const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx,
                                     int64_t n_processed, int n, bsx_read_t *reads, const bsx_pestat_t *pes0);

/* HIP backend constructors (csrc/hip/shim.hip): lane = one of the device's independent stream + staging sets */
int bsx_hip_backend(bsx_device_t *dev, bsx_backend_t *out);          /* lane 0 */
int bsx_hip_backend_lane(bsx_device_t *dev, int lane, bsx_backend_t *out);
/* the chunk pipeline over two arbitrary backends (tests run it over two CPU-restatement contexts) */
BSX_API int bsx_stream_open_backends(int depth, const bsx_backend_t *be, const bsx_opt_t *opt, const bsx_index_t *idx,
                                     const bsx_pestat_t *pes0, bsx_stream_t **out);   /* be[depth]: one backend context per chunk in flight */

/* per-phase wall-clock accounting of the last bsx_process_seqs* call (seconds) */
typedef struct {
	double t_seed, t_sa, t_chain, t_extend, t_merge, t_pestat, t_matesw, t_primary, t_cigar, t_sam, t_total, t_prep, t_cleanup, t_regions;
	int64_t n_tasks, n_intv, n_sa, n_ext_jobs, n_ext_rounds, n_sw_jobs, n_glb_jobs, n_host_tasks;
	int64_t n_redo_tasks;   /* strand searches the device seeded a second time with longer lists */
} bsx_phase_stats_t;
BSX_API void bsx_last_phase_stats(bsx_phase_stats_t *out);

/* `biscuit align` driver over an arbitrary chunk processor (cli.c); the product entry bsx_align_main
 * binds it to the HIP device, oracle/ binds it to the CPU restatement for tests and the CPU baseline */
typedef int (*bsx_process_fn)(void *ud, const bsx_opt_t *opt, const bsx_index_t *idx, int64_t n_processed, int n,
                              bsx_read_t *reads, const bsx_pestat_t *pes0);
BSX_API int bsx_align_main_with(int argc, char **argv, bsx_process_fn process, void *ud,
                                int (*open_device)(int ordinal, const bsx_index_t *idx, void **ud));
/* the same, as one of $WORLD_SIZE processes (rank $RANK, GPU $LOCAL_RANK) whose chunks are brought together by the gather (gather.c): over RCCL
 * (use_rccl) or Unix sockets; with WORLD_SIZE unset or 1 it is bsx_align_main_with */
BSX_API int bsx_align_main_ranks_with(int argc, char **argv, bsx_process_fn process, void *ud,
                                      int (*open_device)(int ordinal, const bsx_index_t *idx, void **ud), int use_rccl);
BSX_API extern char *bsx_pg_line;

BSX_API extern int bsx_verbose;

#endif
